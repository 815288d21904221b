// Probe: what a kernel that finds nothing to do costs on the stream, by grid shape, behind a kernel that wrote
// 128 MB (the position of brick_confirm_kernel / carve_resume_kernel in a fused batch).
// build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip ; run under rocprofv3 --kernel-trace --stats
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void work_kernel(int4 *out, size_t n) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v4i v = {-1, -1, -1, -1};
        __builtin_nontemporal_store(v, reinterpret_cast<v4i *>(out + i));
    }
}

template <int G, int B>
__global__ __launch_bounds__(B) void idle_kernel(const uint32_t *flag, uint32_t *out) {
    if (*flag == 0) return;
    out[blockIdx.x * B + threadIdx.x] = 1;
}

// a kernel whose every block reads 256 counters on lines of their own and scans them in LDS (the preamble of the
// list / unit kernels), then leaves
template <int G>
__global__ __launch_bounds__(256) void scan_kernel(const uint32_t *counters, uint32_t *out) {
    __shared__ uint32_t pref[257];
    const uint32_t tid = threadIdx.x;
    pref[tid + 1] = counters[tid * 32];
    if (tid == 0) pref[0] = 0;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t val = pref[tid + 1], add = tid >= off ? pref[tid + 1 - off] : 0u;
        __syncthreads();
        pref[tid + 1] = val + add;
        __syncthreads();
    }
    if (pref[256] != 0) out[blockIdx.x] = pref[256];
}

int main() {
    int4 *buf; hipMalloc(&buf, (size_t)128 << 20);
    uint32_t *flag; hipMalloc(&flag, 256 * 128 + 4096); hipMemset(flag, 0, 256 * 128 + 4096);
    uint32_t *out; hipMalloc(&out, 4096 * 512 * 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t n = ((size_t)128 << 20) / 16;
#define RUN(G, B) for (int r = 0; r < 40; ++r) { \
        hipLaunchKernelGGL(work_kernel, dim3(2048), dim3(256), 0, s, buf, n); \
        hipLaunchKernelGGL((idle_kernel<G, B>), dim3(G), dim3(B), 0, s, flag, out); }
    RUN(64, 256) RUN(256, 256) RUN(1024, 256) RUN(2048, 256) RUN(4096, 256)
    RUN(64, 512) RUN(256, 512) RUN(1024, 512) RUN(2048, 512) RUN(4096, 512)
#define RUNS(G) for (int r = 0; r < 40; ++r) { \
        hipLaunchKernelGGL(work_kernel, dim3(2048), dim3(256), 0, s, buf, n); \
        hipLaunchKernelGGL((scan_kernel<G>), dim3(G), dim3(256), 0, s, flag, out); }
    RUNS(256) RUNS(512) RUNS(1280) RUNS(2048)
    // three idle kernels in a row behind the work kernel: does the second one cost what the first does?
    for (int r = 0; r < 40; ++r) {
        hipLaunchKernelGGL(work_kernel, dim3(2048), dim3(256), 0, s, buf, n);
        hipLaunchKernelGGL((idle_kernel<255, 256>), dim3(255), dim3(256), 0, s, flag, out);
        hipLaunchKernelGGL((idle_kernel<254, 256>), dim3(254), dim3(256), 0, s, flag, out);
        hipLaunchKernelGGL((idle_kernel<253, 256>), dim3(253), dim3(256), 0, s, flag, out);
    }
    hipStreamSynchronize(s);
    printf("done\n");
    return 0;
}
