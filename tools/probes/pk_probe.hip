// Probe: issue cost of v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 against their scalar forms on gfx950.
// One block of N waves per SIMD-filling grid; each wave runs ITER x 8 independent instructions.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_probe pk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float2v r0 = {a, b}, r1 = {b, a}, r2 = {a + 1, b}, r3 = {b, a + 1}, r4 = {a, b + 2}, r5 = {b + 3, a}, r6 = {a, a}, r7 = {b, b};
    float2v m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 16 scalar mul (8 pairs)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(r0.x), "+v"(r1.x), "+v"(r2.x), "+v"(r3.x), "+v"(r4.x), "+v"(r5.x), "+v"(r6.x), "+v"(r7.x) : "v"(m.x));
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(r0.y), "+v"(r1.y), "+v"(r2.y), "+v"(r3.y), "+v"(r4.y), "+v"(r5.y), "+v"(r6.y), "+v"(r7.y) : "v"(m.y));
        } else if (MODE == 1) {  // 8 packed mul = the same 16 multiplications
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m));
        } else if (MODE == 2) {  // 8 packed fma
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m), "v"(c));
        } else if (MODE == 3) {  // 16 scalar fma
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0.x), "+v"(r1.x), "+v"(r2.x), "+v"(r3.x), "+v"(r4.x), "+v"(r5.x), "+v"(r6.x), "+v"(r7.x) : "v"(m.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0.y), "+v"(r1.y), "+v"(r2.y), "+v"(r3.y), "+v"(r4.y), "+v"(r5.y), "+v"(r6.y), "+v"(r7.y) : "v"(m.y), "v"(c.y));
        } else {  // 8 packed add
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        }
    }
    float2v s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (s.x + s.y == 123.456f) out[0] = s.x;
}

int main() {
    float *out; hipMalloc(&out, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    const char *names[] = {"16 x v_mul_f32", "8 x v_pk_mul_f32", "8 x v_pk_fma_f32", "16 x v_fma_f32", "8 x v_pk_add_f32"};
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        dim3 grid(256 * waves_per_simd), block(256);  // 4 waves per block = one per SIMD of a CU
        for (int mode = 0; mode < 5; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a, 0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, 1.0f, 2.0f); break;
                    case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, 1.0f, 2.0f); break;
                    case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, iters, 1.0f, 2.0f); break;
                    case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, iters, 1.0f, 2.0f); break;
                    default: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, out, iters, 1.0f, 2.0f); break;
                }
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            // cycles per wave per 16 multiplications at 2.4 GHz, waves_per_simd waves sharing a SIMD
            double cyc = best * 1e-3 * 2.4e9 / iters / waves_per_simd;
            printf("waves/SIMD %d  %-18s %.3f ms  -> %.1f cycles per 16 lane-ops group per wave\n", waves_per_simd, names[mode], best, cyc);
        }
    }
    return 0;
}
