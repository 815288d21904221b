// Probe: SIMD issue cost of the VALU instructions the projection is made of, on gfx950.
// Each wave runs ITER x 16 independent instructions of one kind; 4 or 8 waves per SIMD.
// Reports cycles per wave-instruction per SIMD relative to the wall clock (2.4 GHz assumed) and to v_fma_f32.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(OP)                                                                                       \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                         \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                 : "v"(m), "v"(c), "s"(sm)                                                               \
                 : "vcc", "scc", "s20", "s21", "s22", "s23", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47")
// operands: %N = the register itself, %8 = m (vgpr), %9 = c (vgpr), %10 = sm (sgpr)
#define S(x) #x
#define FMA(i) "v_fma_f32 %" S(i) ", %" S(i) ", %8, %9\n"
#define MUL(i) "v_mul_f32 %" S(i) ", %" S(i) ", %8\n"
#define ADD(i) "v_add_f32 %" S(i) ", %" S(i) ", %9\n"
#define MULS(i) "v_mul_f32 %" S(i) ", %10, %" S(i) "\n"
#define FMAC(i) "v_fmac_f32 %" S(i) ", %8, %9\n"
#define CMPV(i) "v_cmp_lt_f32 vcc, %" S(i) ", %8\n"
#define CMPS(i) "v_cmp_lt_f32 s[20:21], %" S(i) ", %8\n"
#define CMPCLS(i) "v_cmp_class_f32 s[20:21], %" S(i) ", %9\n"
#define CNDV(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %8, vcc\n"
#define CNDS(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %8, s[22:23]\n"
#define CVTI(i) "v_cvt_i32_f32 %" S(i) ", %" S(i) "\n"
#define CVTF(i) "v_cvt_f32_i32 %" S(i) ", %" S(i) "\n"
#define AND(i) "v_and_b32 %" S(i) ", %" S(i) ", %8\n"
#define LSHR(i) "v_lshrrev_b32 %" S(i) ", 5, %" S(i) "\n"
#define ADDU(i) "v_add_u32 %" S(i) ", %" S(i) ", %8\n"
#define MUL24(i) "v_mul_u32_u24 %" S(i) ", %" S(i) ", %8\n"
#define MAD24(i) "v_mad_u32_u24 %" S(i) ", %" S(i) ", %8, %9\n"
#define MULLO(i) "v_mul_lo_u32 %" S(i) ", %" S(i) ", %8\n"
#define MULHI(i) "v_mul_hi_u32 %" S(i) ", %" S(i) ", %8\n"
#define LSHLADD(i) "v_lshl_add_u32 %" S(i) ", %" S(i) ", 5, %8\n"
#define ADD3(i) "v_add3_u32 %" S(i) ", %" S(i) ", %8, %9\n"
#define BFE(i) "v_bfe_u32 %" S(i) ", %" S(i) ", 5, 5\n"
#define ANDOR(i) "v_and_or_b32 %" S(i) ", %" S(i) ", %8, %9\n"
#define RCP(i) "v_rcp_f32 %" S(i) ", %" S(i) "\n"
#define MAXF(i) "v_max_f32 %" S(i) ", %" S(i) ", %8\n"
#define MED3(i) "v_med3_f32 %" S(i) ", %" S(i) ", %8, %9\n"
#define LDEXP(i) "v_ldexp_f32 %" S(i) ", %" S(i) ", %9\n"
#define DIVFIX(i) "v_div_fixup_f32 %" S(i) ", %" S(i) ", %8, %9\n"
#define MOV(i) "v_mov_b32 %" S(i) ", %8\n"
#define TRUNC(i) "v_trunc_f32 %" S(i) ", %" S(i) "\n"
#define FLOOR(i) "v_floor_f32 %" S(i) ", %" S(i) "\n"
#define CVTU(i) "v_cvt_u32_f32 %" S(i) ", %" S(i) "\n"
#define MADU64(i) "v_mad_u64_u32 v[40:41], s[20:21], %" S(i) ", %8, v[42:43]\n"
#define LSHLADD64(i) "v_lshl_add_u64 v[40:41], v[42:43], 2, v[44:45]\n"
#define PERM(i) "v_perm_b32 %" S(i) ", %" S(i) ", %8, %9\n"
#define XAD(i) "v_xad_u32 %" S(i) ", %" S(i) ", %8, %9\n"
#define LSHLOR(i) "v_lshl_or_b32 %" S(i) ", %" S(i) ", 5, %8\n"
#define CNDE64V(i) "v_cndmask_b32_e64 %" S(i) ", %" S(i) ", %8, vcc\n"
#define CMPU(i) "v_cmp_lt_u32 s[20:21], %" S(i) ", %8\n"
#define CMPUV(i) "v_cmp_lt_u32 vcc, %" S(i) ", %8\n"
#define MINABS(i) "v_min_f32_e64 %" S(i) ", |%" S(i) "|, |%8|\n"
#define MAX3(i) "v_max3_f32 %" S(i) ", |%" S(i) "|, |%8|, %9\n"
#define BFI(i) "v_bfi_b32 %" S(i) ", %" S(i) ", %8, %9\n"
#define FRACT(i) "v_fract_f32 %" S(i) ", %" S(i) "\n"
#define SUBF(i) "v_sub_f32 %" S(i) ", %" S(i) ", %9\n"
#define MADU64X(i) "v_mad_u64_u32 v[40:41], s[20:21], %" S(i) ", %8, v[42:43]\n"
#define LSHLADD64X(i) "v_lshl_add_u64 v[40:41], v[42:43], 2, v[44:45]\n"
#define PKFMA(i) "v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[46:47]\n"
#define PKMUL(i) "v_pk_mul_f32 v[40:41], v[42:43], v[44:45]\n"
#define PKADD(i) "v_pk_add_f32 v[40:41], v[42:43], v[44:45]\n"
#define FMAX(i) "v_fma_f32 v40, v42, v44, v46\n"
#define MULX(i) "v_mul_f32 v40, v42, v44\n"
#define LSHL(i) "v_lshlrev_b32 %" S(i) ", 2, %" S(i) "\n"
#define OR3(i) "v_or3_b32 %" S(i) ", %" S(i) ", %8, %9\n"
#define CMPSDWA(i) "v_cmp_ne_u16_sdwa s[20:21], %" S(i) ", %8 src0_sel:BYTE_0 src1_sel:DWORD\n"
#define DOT4(i) "v_dot4_u32_u8 %" S(i) ", %" S(i) ", %8, %9\n"
#define XORB(i) "v_xor_b32 %" S(i) ", %" S(i) ", %8\n"
#define ORB(i) "v_or_b32 %" S(i) ", %" S(i) ", %8\n"
#define DPPMOV(i) "v_mov_b32_dpp %" S(i) ", %" S(i) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define SUBREV(i) "v_subrev_u32 %" S(i) ", %" S(i) ", %8\n"
#define MINU(i) "v_min_u32 %" S(i) ", %" S(i) ", %8\n"
#define ALIGNBIT(i) "v_alignbit_b32 %" S(i) ", %" S(i) ", %8, 7\n"
#define SNOP(i) "s_nop 0\n"
#define SAND(i) "s_and_b32 s20, s20, s21\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = a + i * b;
    float m = 1.0000001f * a, c = 1e-9f * b;
    float sm = __builtin_amdgcn_readfirstlane(m);
    asm volatile("s_mov_b64 s[22:23], exec\n" ::: "s22", "s23");
    for (int i = 0; i < iters; ++i) {
#define CASE(N, OP) if (MODE == N) { REP8(OP); REP8(OP); }
        CASE(0, FMA) CASE(1, MUL) CASE(2, ADD) CASE(3, MULS) CASE(4, FMAC) CASE(5, CMPV) CASE(6, CMPS) CASE(7, CMPCLS)
        CASE(8, CNDV) CASE(9, CNDS) CASE(10, CVTI) CASE(11, CVTF) CASE(12, AND) CASE(13, LSHR) CASE(14, ADDU)
        CASE(15, MUL24) CASE(16, MAD24) CASE(17, MULLO) CASE(18, MULHI) CASE(19, LSHLADD) CASE(20, ADD3) CASE(21, BFE)
        CASE(22, ANDOR) CASE(23, RCP) CASE(24, MAXF) CASE(25, MED3) CASE(26, LDEXP) CASE(27, DIVFIX) CASE(28, MOV)
        CASE(29, TRUNC) CASE(30, FLOOR) CASE(31, CVTU) CASE(32, PERM) CASE(33, XAD) CASE(34, LSHLOR) CASE(35, SNOP) CASE(36, SAND)
        CASE(37, CNDE64V) CASE(38, CMPU) CASE(39, CMPUV) CASE(40, MINABS) CASE(41, MAX3) CASE(42, BFI) CASE(43, FRACT) CASE(44, SUBF)
        CASE(45, MADU64X) CASE(46, LSHLADD64X) CASE(47, PKFMA) CASE(48, PKMUL) CASE(49, PKADD) CASE(50, FMAX) CASE(51, MULX) CASE(52, LSHL) CASE(53, OR3) CASE(54, CMPSDWA) CASE(55, DOT4) CASE(56, XORB) CASE(57, ORB) CASE(58, DPPMOV) CASE(59, SUBREV) CASE(60, MINU) CASE(61, ALIGNBIT)
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
float run(float *out, int iters, int waves_per_simd) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(256 * waves_per_simd), block(256);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

template <int MODE>
void row(float *out, const char *name, float ref4, float ref8) {
    const int iters = 4000;
    float t4 = run<MODE>(out, iters, 4), t8 = run<MODE>(out, iters, 8);
    // cycles per wave-instruction per SIMD at 2.4 GHz
    double c4 = t4 * 1e-3 * 2.4e9 / (iters * 16.0) / 4, c8 = t8 * 1e-3 * 2.4e9 / (iters * 16.0) / 8;
    printf("%-22s 4 waves/SIMD %.2f cyc (x%.2f of fma)   8 waves/SIMD %.2f cyc (x%.2f)\n", name, c4, ref4 > 0 ? t4 / ref4 : 1.0, c8,
           ref8 > 0 ? t8 / ref8 : 1.0);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float *out; hipMalloc(&out, 64);
    const int iters = 4000;
    run<0>(out, iters, 4);
    float r4 = run<0>(out, iters, 4), r8 = run<0>(out, iters, 8);
#define ROW(N, NAME) row<N>(out, NAME, r4, r8);
    ROW(0, "v_fma_f32") ROW(1, "v_mul_f32") ROW(2, "v_add_f32") ROW(3, "v_mul_f32 s,v") ROW(4, "v_fmac_f32")
    ROW(5, "v_cmp_lt_f32 vcc") ROW(6, "v_cmp_lt_f32 sgpr") ROW(7, "v_cmp_class_f32 sgpr") ROW(8, "v_cndmask vcc") ROW(9, "v_cndmask sgpr")
    ROW(10, "v_cvt_i32_f32") ROW(11, "v_cvt_f32_i32") ROW(12, "v_and_b32") ROW(13, "v_lshrrev_b32") ROW(14, "v_add_u32")
    ROW(15, "v_mul_u32_u24") ROW(16, "v_mad_u32_u24") ROW(17, "v_mul_lo_u32") ROW(18, "v_mul_hi_u32") ROW(19, "v_lshl_add_u32")
    ROW(20, "v_add3_u32") ROW(21, "v_bfe_u32") ROW(22, "v_and_or_b32") ROW(23, "v_rcp_f32") ROW(24, "v_max_f32") ROW(25, "v_med3_f32")
    ROW(26, "v_ldexp_f32") ROW(27, "v_div_fixup_f32") ROW(28, "v_mov_b32") ROW(29, "v_trunc_f32") ROW(30, "v_floor_f32") ROW(31, "v_cvt_u32_f32")
    ROW(32, "v_perm_b32") ROW(33, "v_xad_u32") ROW(34, "v_lshl_or_b32") ROW(35, "s_nop 0") ROW(36, "s_and_b32")
    ROW(37, "v_cndmask_e64 vcc") ROW(38, "v_cmp_lt_u32 sgpr") ROW(39, "v_cmp_lt_u32 vcc") ROW(40, "v_min_f32 |a|,|b|") ROW(41, "v_max3_f32 abs") ROW(42, "v_bfi_b32") ROW(43, "v_fract_f32") ROW(44, "v_sub_f32")
    ROW(45, "v_mad_u64_u32") ROW(46, "v_lshl_add_u64") ROW(47, "v_pk_fma_f32 (same regs)") ROW(48, "v_pk_mul_f32 (same regs)") ROW(49, "v_pk_add_f32 (same regs)") ROW(50, "v_fma_f32 (same regs)") ROW(51, "v_mul_f32 (same regs)") ROW(52, "v_lshlrev_b32") ROW(53, "v_or3_b32") ROW(54, "v_cmp_ne_u16_sdwa") ROW(55, "v_dot4_u32_u8") ROW(56, "v_xor_b32") ROW(57, "v_or_b32") ROW(58, "v_mov_b32_dpp quad_perm") ROW(59, "v_subrev_u32") ROW(60, "v_min_u32") ROW(61, "v_alignbit_b32")
    return 0;
}
