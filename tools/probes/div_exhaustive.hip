// Probe: is a SHORTER division sequence than project()'s still the correctly rounded quotient?
// EXHAUSTIVE over the significands: every pair (1.m_n, 1.m_d), 2^23 x 2^23 = 7.0e13 pairs, against hipcc's IEEE
// division (-fhip-fp32-correctly-rounded-divide-sqrt).  Every operation of the candidates is a multiplication or an
// FMA, so as long as nothing leaves the normal range a power-of-two scale of n or d scales every intermediate exactly
// and the rounding decisions depend on the significands alone: a clean exhaustive pass proves the sequence for all
// normal operands whose intermediates stay normal (project()'s certified range, sc_project.h).  A second, sampled pass
// spreads the exponents over that range as a cross-check of the scaling argument (v_rcp_f32 included).
//   A: r = rcp(d) refined by one Newton step;  q = n r;  ONE correction  q += r (n - d q)
//   B: r = rcp(d) raw;                          q = n r;  TWO corrections
//   C: r refined, TWO corrections -- the sequence project() uses today (must show 0)
//   D: r raw, ONE correction (expected to fail: shows the probe can see a failure)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//        -fno-gpu-flush-denormals-to-zero -o div_exhaustive div_exhaustive.hip
// usage: div_exhaustive [first_slice [nslices]]   (32 slices of 2^18 denominators each; default: all)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

struct Tally {
    unsigned long long bad[4];
    unsigned long long pairs;
    uint32_t ex_n[4][8], ex_d[4][8];  // first few failing pairs per candidate (bit patterns)
    uint32_t nex[4];
};

__device__ __forceinline__ float corr(float n, float d, float r, float q) {
    const float e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ void note(Tally *t, int c, float n, float d) {
    const uint32_t k = atomicAdd(&t->nex[c], 1u);
    if (k < 8) {
        t->ex_n[c][k] = __float_as_uint(n);
        t->ex_d[c][k] = __float_as_uint(d);
    }
}

// one thread = one denominator significand; loops over `ncount` numerator significands from n0 (stride 1)
__global__ __launch_bounds__(256) void sweep(uint32_t d0, uint32_t n0, uint32_t ncount, int dexp, int nexp, Tally *t) {
    const uint32_t md = d0 + blockIdx.x * 256u + threadIdx.x;
    const float d = __uint_as_float(((uint32_t)(127 + dexp) << 23) | (md & 0x7fffffu));
    const float r1 = __builtin_amdgcn_rcpf(d);
    const float r2 = __builtin_fmaf(__builtin_fmaf(-d, r1, 1.0f), r1, r1);
    unsigned long long bad[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < ncount; ++i) {
        const float n = __uint_as_float(((uint32_t)(127 + nexp) << 23) | ((n0 + i) & 0x7fffffu));
        const float ref = n / d;
        const float qa = corr(n, d, r2, n * r2);
        const float qb = corr(n, d, r1, corr(n, d, r1, n * r1));
        const float qc = corr(n, d, r2, qa);
        const float qd = corr(n, d, r1, n * r1);
        if (qa != ref) { if (bad[0]++ == 0) note(t, 0, n, d); }
        if (qb != ref) { if (bad[1]++ == 0) note(t, 1, n, d); }
        if (qc != ref) { if (bad[2]++ == 0) note(t, 2, n, d); }
        if (qd != ref) { if (bad[3]++ == 0) note(t, 3, n, d); }
    }
    for (int c = 0; c < 4; ++c) {
        unsigned long long v = bad[c];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&t->bad[c], v);
    }
    if ((threadIdx.x & 63u) == 0) atomicAdd(&t->pairs, 64ull * ncount);
}

// sampled pass: hashed significands AND exponents (d in 2^-10 .. 2^30, n in 2^-40 .. 2^30: the certified range)
__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void sampled(uint64_t count, uint32_t seed, Tally *t) {
    unsigned long long bad[4] = {0, 0, 0, 0}, done = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t a = mix((uint32_t)i ^ seed), b = mix((uint32_t)(i >> 32) + a + 0x9e3779b9u), c = mix(a ^ (b * 3u + 1u));
        const int dexp = -10 + (int)(c % 40u), nexp = -40 + (int)((c >> 8) % 70u);
        const float d = __uint_as_float(((uint32_t)(127 + dexp) << 23) | (a & 0x7fffffu));
        float n = __uint_as_float(((uint32_t)(127 + nexp) << 23) | (b & 0x7fffffu));
        if (c >> 31) n = -n;
        const float r1 = __builtin_amdgcn_rcpf(d);
        const float r2 = __builtin_fmaf(__builtin_fmaf(-d, r1, 1.0f), r1, r1);
        const float ref = n / d;
        const float qa = corr(n, d, r2, n * r2);
        const float qb = corr(n, d, r1, corr(n, d, r1, n * r1));
        const float qc = corr(n, d, r2, qa);
        const float qd = corr(n, d, r1, n * r1);
        if (qa != ref) { if (bad[0]++ == 0) note(t, 0, n, d); }
        if (qb != ref) { if (bad[1]++ == 0) note(t, 1, n, d); }
        if (qc != ref) { if (bad[2]++ == 0) note(t, 2, n, d); }
        if (qd != ref) { if (bad[3]++ == 0) note(t, 3, n, d); }
        ++done;
    }
    for (int c = 0; c < 4; ++c)
        if (bad[c]) atomicAdd(&t->bad[c], bad[c]);
    atomicAdd(&t->pairs, done);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
    const int first = argc > 1 ? atoi(argv[1]) : 0;
    const int nsl = argc > 2 ? atoi(argv[2]) : 32 - first;
    Tally *dt = nullptr, ht;
    CHECK(hipMalloc(reinterpret_cast<void **>(&dt), sizeof ht));
    CHECK(hipMemset(dt, 0, sizeof ht));
    const auto t0 = std::chrono::steady_clock::now();
    // a slice = 2^18 denominators x all 2^23 numerators, in 8 launches of 2^20 numerators (each well under a second)
    for (int s = first; s < first + nsl && s < 32; ++s) {
        for (int part = 0; part < 8; ++part)
            hipLaunchKernelGGL(sweep, dim3(1024), dim3(256), 0, 0, (uint32_t)s << 18, (uint32_t)part << 20, 1u << 20, 0, 0, dt);
        CHECK(hipDeviceSynchronize());
        {
            CHECK(hipMemcpy(&ht, dt, sizeof ht, hipMemcpyDeviceToHost));
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("slice %3d: %.4e pairs, mismatches A %llu B %llu C %llu D %llu  (%.1f s)\n", s, (double)ht.pairs, ht.bad[0],
                   ht.bad[1], ht.bad[2], ht.bad[3], sec);
            fflush(stdout);
        }
    }
    CHECK(hipMemcpy(&ht, dt, sizeof ht, hipMemcpyDeviceToHost));
    const Tally ex = ht;
    CHECK(hipMemset(dt, 0, sizeof ht));
    hipLaunchKernelGGL(sampled, dim3(4096), dim3(256), 0, 0, (uint64_t)1 << 34, 12345u, dt);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&ht, dt, sizeof ht, hipMemcpyDeviceToHost));
    printf("{\"exhaustive\": {\"slices\": [%d, %d], \"pairs\": %llu, \"mismatches\": {\"A_refined_one_correction\": %llu, "
           "\"B_raw_two_corrections\": %llu, \"C_refined_two_corrections\": %llu, \"D_raw_one_correction\": %llu}},\n",
           first, first + nsl, ex.pairs, ex.bad[0], ex.bad[1], ex.bad[2], ex.bad[3]);
    printf(" \"sampled_exponents\": {\"pairs\": %llu, \"mismatches\": {\"A\": %llu, \"B\": %llu, \"C\": %llu, \"D\": %llu}},\n",
           ht.pairs, ht.bad[0], ht.bad[1], ht.bad[2], ht.bad[3]);
    printf(" \"examples\": {");
    const char *names = "ABCD";
    for (int c = 0; c < 4; ++c) {
        printf("%s\"%c\": [", c ? ", " : "", names[c]);
        for (uint32_t k = 0; k < ex.nex[c] && k < 8; ++k) printf("%s[\"0x%08x\", \"0x%08x\"]", k ? ", " : "", ex.ex_n[c][k], ex.ex_d[c][k]);
        printf("]");
    }
    printf("}}\n");
    (void)hipFree(dt);
    return 0;
}
