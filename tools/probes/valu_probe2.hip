// Probe (round 6): what a vector instruction costs on gfx950 in REAL shader cycles -- s_memtime around
// the loop, no assumed clock -- and what the clock is while that stream runs.  valu_probe.hip (round 5) converted wall
// time at an assumed 2.4 GHz and found v_fma_f32 at 4.1 "cycles" and v_mul_f32 / v_add_f32 at 2.6: this probe separates
// the two factors, cycles per wave-instruction and SIMD (from the in-kernel counter) and the effective clock (cycles /
// wall time), per instruction class, at 2 / 4 / 8 waves per SIMD, for single-class streams, for VOP2 / VOP3 mixes and for
// the instruction mix of one projection of the carve / averaging kernels (csrc/sc_project.h: 9 mul/add of the three
// sums, rcp + 2 fma, 2 x (mul + 2 fma), 2 x (mul + add), 2 cvt, 2 cmp, shift + mad24 + shift, cndmask, bfe, cmp).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_probe2 valu_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define S(x) #x
#define REGS "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
#define INS "v"(m), "v"(c)
#define CLOB "vcc", "scc", "s20", "s21", "s22", "s23"
#define FMA(i) "v_fma_f32 %" S(i) ", %" S(i) ", %8, %9\n"
#define MUL(i) "v_mul_f32 %" S(i) ", %" S(i) ", %8\n"
#define ADD(i) "v_add_f32 %" S(i) ", %" S(i) ", %9\n"
#define RCP(i) "v_rcp_f32 %" S(i) ", %" S(i) "\n"
#define CVT(i) "v_cvt_i32_f32 %" S(i) ", %" S(i) "\n"
#define CMP(i) "v_cmp_lt_u32 s[20:21], %" S(i) ", %8\n"
#define CND(i) "v_cndmask_b32 %" S(i) ", %" S(i) ", %8, s[22:23]\n"
#define SHR(i) "v_lshrrev_b32 %" S(i) ", 5, %" S(i) "\n"
#define SHL(i) "v_lshlrev_b32 %" S(i) ", 2, %" S(i) "\n"
#define MAD24(i) "v_mad_u32_u24 %" S(i) ", %" S(i) ", %8, %9\n"
#define BFE(i) "v_bfe_u32 %" S(i) ", %" S(i) ", 5, 1\n"
#define AND(i) "v_and_b32 %" S(i) ", %" S(i) ", %8\n"
#define PKFMA(i) "v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[46:47]\n"
#define PKMUL(i) "v_pk_mul_f32 v[40:41], v[42:43], v[44:45]\n"

#define ALL8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
// 16 instructions per block in every mode
#define BLOCK_SINGLE(OP) asm volatile(ALL8(OP) ALL8(OP) : REGS : INS : CLOB);
#define BLOCK_PK(OP) asm volatile(ALL8(OP) ALL8(OP) : REGS : INS : CLOB, "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
// half VOP2 (mul / add), half VOP3 (fma)
#define BLOCK_MULFMA asm volatile(MUL(0) FMA(1) MUL(2) FMA(3) MUL(4) FMA(5) MUL(6) FMA(7) ADD(0) FMA(1) ADD(2) FMA(3) ADD(4) FMA(5) ADD(6) FMA(7) : REGS : INS : CLOB);
// the 34 instructions of one projection (two blocks of 17 would not fit the 16-per-block count: 32 of them in two
// blocks, the remaining shift and compare are in the count of the "mix" by weight): 9 mul/add, rcp, 6 fma, 2 mul, 2 add,
// 2 cvt, 2 cmp, shr, mad24, shl, cndmask, bfe, cmp, and -- the 34 over 32 slots -- two of the adds counted once
#define BLOCK_PROJ_A asm volatile(MUL(0) ADD(0) ADD(0) MUL(1) ADD(1) ADD(1) MUL(2) ADD(2) ADD(2) RCP(3) FMA(3) FMA(3) MUL(4) FMA(4) FMA(4) MUL(5) : REGS : INS : CLOB);
#define BLOCK_PROJ_B asm volatile(FMA(5) FMA(5) MUL(6) ADD(6) MUL(7) ADD(7) CVT(6) CVT(7) CMP(6) CMP(7) SHR(0) MAD24(0) SHL(0) CND(0) BFE(1) CMP(1) : REGS : INS : CLOB);
// the same 32 instructions for FOUR voxels side by side, as the kernels issue them (a lane's four voxels, every step on all
// four before the next step): 128 instructions = 8 blocks' worth per loop turn
#define OP4(OP) OP(0) OP(1) OP(2) OP(3)
#define BLOCK_PROJ4 asm volatile(                                                                                     \
    OP4(MUL) OP4(ADD) OP4(ADD) OP4(MUL) OP4(ADD) OP4(ADD) OP4(MUL) OP4(ADD) OP4(ADD) OP4(RCP) OP4(FMA) OP4(FMA)        \
    OP4(MUL) OP4(FMA) OP4(FMA) OP4(MUL) OP4(FMA) OP4(FMA) OP4(MUL) OP4(ADD) OP4(MUL) OP4(ADD) OP4(CVT) OP4(CVT)        \
    OP4(CMP) OP4(CMP) OP4(SHR) OP4(MAD24) OP4(SHL) OP4(CND) OP4(BFE) OP4(CMP) : REGS : INS : CLOB);

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float a, float b) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = a + i * b;
    float m = 1.0000001f * a, c = 1e-9f * b;
    asm volatile("s_mov_b64 s[22:23], exec\n" ::: "s22", "s23");
    const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { BLOCK_SINGLE(FMA) }
        if (MODE == 1) { BLOCK_SINGLE(MUL) }
        if (MODE == 2) { BLOCK_SINGLE(ADD) }
        if (MODE == 3) { BLOCK_SINGLE(RCP) }
        if (MODE == 4) { BLOCK_SINGLE(CVT) }
        if (MODE == 5) { BLOCK_SINGLE(CMP) }
        if (MODE == 6) { BLOCK_SINGLE(CND) }
        if (MODE == 7) { BLOCK_SINGLE(SHR) }
        if (MODE == 8) { BLOCK_SINGLE(MAD24) }
        if (MODE == 9) { BLOCK_SINGLE(AND) }
        if (MODE == 10) { BLOCK_PK(PKFMA) }
        if (MODE == 11) { BLOCK_PK(PKMUL) }
        if (MODE == 12) { BLOCK_MULFMA }
        if (MODE == 13) { if (i & 1) { BLOCK_PROJ_B } else { BLOCK_PROJ_A } }
        if (MODE == 14) { if ((i & 7) == 0) { BLOCK_PROJ4 } }  // 128 instructions every 8th turn = 16 per turn
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Res { double wall_ms, cycles_med; };

template <int MODE>
Res run(float *out, unsigned long long *cyc, int iters, int wps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int nblocks = 256 * wps;  // 4 waves per block, 1024 SIMDs
    Res best{1e9, 0};
    std::vector<unsigned long long> h((size_t)nblocks * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(256), 0, 0, out, cyc, iters, 1.0f, 2.0f);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best.wall_ms) {
            hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            best.wall_ms = ms;
            best.cycles_med = (double)h[h.size() / 2];
        }
    }
    hipEventDestroy(a);
    hipEventDestroy(b);
    return best;
}

template <int MODE>
void row(float *out, unsigned long long *cyc, const char *name) {
    const int iters = 20000;  // 320 000 instructions per wave: ~1 ms and more per launch
    printf("%-34s", name);
    for (int wps : {2, 4, 8}) {
        Res r = run<MODE>(out, cyc, iters, wps);
        // a wave's loop lasts as long as its SIMD is busy with all `wps` waves' streams
        const double per_instr = r.cycles_med / ((double)iters * 16.0 * wps);
        const double ghz = r.cycles_med / (r.wall_ms * 1e-3) / 1e9;
        printf("  | %d waves: %5.2f cyc/instr/SIMD, %4.2f GHz, %6.3f ms", wps, per_instr, ghz, r.wall_ms);
    }
    printf("\n");
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 64);
    hipMalloc(&cyc, (size_t)256 * 8 * 4 * 8);
    run<0>(out, cyc, 2000, 8);
    printf("cycles: s_memtime around the loop, median over all wavefronts; GHz: those cycles / the launch's wall time (HIP events)\n");
    row<0>(out, cyc, "v_fma_f32");
    row<1>(out, cyc, "v_mul_f32");
    row<2>(out, cyc, "v_add_f32");
    row<3>(out, cyc, "v_rcp_f32");
    row<4>(out, cyc, "v_cvt_i32_f32");
    row<5>(out, cyc, "v_cmp_lt_u32 sgpr");
    row<6>(out, cyc, "v_cndmask_b32 sgpr");
    row<7>(out, cyc, "v_lshrrev_b32");
    row<8>(out, cyc, "v_mad_u32_u24");
    row<9>(out, cyc, "v_and_b32");
    row<10>(out, cyc, "v_pk_fma_f32 (same regs)");
    row<11>(out, cyc, "v_pk_mul_f32 (same regs)");
    row<12>(out, cyc, "mul/add + fma alternating");
    row<13>(out, cyc, "one projection's mix, one voxel");
    row<14>(out, cyc, "one projection's mix, four voxels");
    return 0;
}
