// Probe (round 6): what it costs to hand work from one role to the next INSIDE a launch -- consumer
// blocks behind the producer blocks of the same grid, the producers ending with a release-increment at agent scope, the
// consumers spinning on the count and acquiring -- against the kernel boundary it would replace (4.7 us behind a
// writer: launch_floor.hip).  Workgroups are dispatched in index order, so every producer is resident or finished when a
// consumer starts and no producer waits for a consumer; the spin is BOUNDED all the same (a consumer gives up after
// ~20 ms and raises a flag the host prints).
//   two launches : producer kernel (P blocks write `mb` MB in address order), consumer kernel (C blocks read one word of
//                  every producer's last line and add them up)
//   one launch   : the same two roles in one grid of P + C blocks, handed over through the counter
// Both check the sum (every producer's last store must be visible to every consumer).
// build: hipcc --offload-arch=gfx950 -O3 -o handoff_probe handoff_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void produce(v4i *out, size_t n16, uint32_t bid, uint32_t nprod, int stamp) {
    // block `bid` writes its contiguous share; the last 16 bytes of the share carry the stamp
    const size_t per = n16 / nprod, lo = (size_t)bid * per;
    for (size_t i = lo + threadIdx.x; i < lo + per; i += blockDim.x) {
        v4i v = {stamp, stamp, stamp, stamp};
        out[i] = v;
    }
}

__device__ __forceinline__ void consume(const v4i *in, size_t n16, uint32_t nprod, int stamp, uint32_t *bad) {
    const size_t per = n16 / nprod;
    int wrong = 0;
    for (uint32_t p = threadIdx.x; p < nprod; p += blockDim.x) wrong += in[(size_t)p * per + per - 1].x != stamp;
    if (wrong) atomicAdd(bad, (uint32_t)wrong);
}

__global__ __launch_bounds__(256) void producer_kernel(v4i *out, size_t n16, uint32_t nprod, int stamp) {
    produce(out, n16, blockIdx.x, nprod, stamp);
}
__global__ __launch_bounds__(256) void consumer_kernel(const v4i *in, size_t n16, uint32_t nprod, int stamp, uint32_t *bad) {
    consume(in, n16, nprod, stamp, bad);
}

// FENCE 1: acq_rel increments, a release store of the gate by the last arriver, relaxed polls of the gate, ONE acquire fence
// behind the open gate (the memory model's way).  FENCE 0 (the first version of this probe, relaxed everywhere): every
// consumer read stale stamps -- cross-XCD visibility needs the L2 write-back and invalidate the fences stand for.
template <int FENCE, int SLEEP>
__global__ __launch_bounds__(256) void fused_kernel(v4i *buf, size_t n16, uint32_t nprod, int stamp, uint32_t *done,
                                                    uint32_t target, uint32_t *bad, uint32_t *timeouts) {
    if (blockIdx.x < nprod) {
        produce(buf, n16, blockIdx.x, nprod, stamp);
        __syncthreads();
        if (threadIdx.x == 0) {
            // the counter and the flag the consumers poll are on lines of their own: the producers' increments do not
            // queue behind the polls (with both on one line the hand-over took 29-147 us: the first version of this probe)
            uint32_t prev;
            if (FENCE) prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            else prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1u == target) {  // the last arriver opens the gate
                if (FENCE) __hip_atomic_store(done + 32, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(done + 32, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 0;
        for (int spin = 0; spin < 400000; ++spin) {  // ~20 ms at 50 ns a turn
            const uint32_t d = __hip_atomic_load(done + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(SLEEP);
        }
        if (!good) atomicAdd(timeouts, 1u);
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one acquire behind the open gate, none per poll
        ok = good;
    }
    __syncthreads();
    if (ok) consume(buf, n16, nprod, stamp, bad);
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int reps = 40;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t *ctl;
    hipMalloc(&ctl, 4096);
    hipMemset(ctl, 0, 4096);
    uint32_t *done = ctl, *bad = ctl + 64, *timeouts = ctl + 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-8s %-6s %-6s | two launches (us) | one launch, poll every ~0.4 us    | ... every ~3 us (us)    | bad sums / timeouts\n", "MB", "prod", "cons");
    for (int mb : {0, 16, 128}) {
        for (uint32_t nprod : {256u, 1024u}) {
            for (uint32_t ncons : {256u, 1024u}) {
                const size_t bytes = std::max<size_t>((size_t)mb << 20, (size_t)nprod * 256 * 16);
                const size_t n16 = bytes / 16;
                v4i *buf;
                hipMalloc(&buf, bytes);
                float t[3] = {0, 0, 0};
                uint32_t target = 0;
                hipMemsetAsync(ctl, 0, 4096, s);
                for (int mode = 0; mode < 3; ++mode) {
                    std::vector<float> ms;
                    for (int r = 0; r < reps + 5; ++r) {
                        const int stamp = mode * 1000 + r + 1;
                        hipEventRecord(e0, s);
                        if (mode == 0) {
                            hipLaunchKernelGGL(producer_kernel, dim3(nprod), dim3(256), 0, s, buf, n16, nprod, stamp);
                            hipLaunchKernelGGL(consumer_kernel, dim3(ncons), dim3(256), 0, s, buf, n16, nprod, stamp, bad);
                        } else {
                            target += nprod;
                            if (mode == 1)
                                hipLaunchKernelGGL((fused_kernel<1, 8>), dim3(nprod + ncons), dim3(256), 0, s, buf, n16, nprod, stamp, done, target, bad, timeouts);
                            else
                                hipLaunchKernelGGL((fused_kernel<1, 64>), dim3(nprod + ncons), dim3(256), 0, s, buf, n16, nprod, stamp, done, target, bad + 1, timeouts);
                        }
                        hipEventRecord(e1, s);
                        hipEventSynchronize(e1);
                        float x;
                        hipEventElapsedTime(&x, e0, e1);
                        if (r >= 5) ms.push_back(x);
                    }
                    std::sort(ms.begin(), ms.end());
                    t[mode] = ms[ms.size() / 2] * 1e3f;
                }
                uint32_t h[192];
                hipMemcpy(h, ctl, sizeof h, hipMemcpyDeviceToHost);
                printf("%-8d %-6u %-6u | %8.2f          | %8.2f                         | %8.2f                | %u, %u / %u\n", mb, nprod,
                       ncons, t[0], t[1], t[2], h[64], h[65], h[128]);
                hipFree(buf);
            }
        }
    }
    return 0;
}
