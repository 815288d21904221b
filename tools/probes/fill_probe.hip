// Probe: how fast can persistent blocks write the 512^3 int32 state (537 MB) on gfx950, by store pattern and grid size?
// The fused batch writes every label exactly once; 75 % of those bytes are the -1 of bricks found empty, written by
// "store blocks" beside the survivor stages (store_culled_bricks).  A strip = 16 columns x 512 voxels = 32 KB contiguous.
//   pattern 0: linear, grid-stride, 16 B per lane (what a plain fill kernel does)
//   pattern 1: today's store blocks -- a block takes a strip, wave w its columns 4w..4w+3 (lane >> 4), 16 lanes x 16 B =
//              one brick's 256-byte run of a column per instruction, 8 instructions walk the column (bz loop)
//   pattern 2: a block takes a strip and writes it linearly, 4 KB per round (256 lanes x 16 B), 8 rounds
//   pattern 3: a wave takes a quarter strip (8 KB) and writes it linearly, 1 KB per instruction, 8 instructions
//   pattern 4: like 2, the strips of a block CONSECUTIVE (block b owns strips [b S/G, (b+1) S/G)) instead of strided
// nt = 1: nontemporal stores.  Reports TB/s per (pattern, nt, blocks).
// build: hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr uint32_t kStrips = 512 * 32, kStripBytes = 32768;

template <int PAT, bool NT>
__global__ __launch_bounds__(256) void fill(int32_t *__restrict__ p, uint64_t bytes, const uint8_t *__restrict__ flags) {
    const v4i val = {-1, -1, -1, -1};
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    char *base = reinterpret_cast<char *>(p);
    auto st = [&](char *a) {
        if (NT) __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(a));
        else *reinterpret_cast<v4i *>(a) = val;
    };
    if (PAT == 0) {
        for (uint64_t off = ((uint64_t)blockIdx.x * 256u + tid) * 16u; off < bytes; off += (uint64_t)gridDim.x * 4096u) st(base + off);
    } else if (PAT == 1) {
        for (uint32_t s = blockIdx.x; s < kStrips; s += gridDim.x) {
            const uint32_t f = lane < 8u ? flags[s * 8u + lane] : 0u;
            const unsigned long long culled = __ballot(f == 1u);
            char *col = base + (uint64_t)s * kStripBytes + (wave * 4u + (lane >> 4)) * 2048u;
            for (uint32_t bz = 0; bz < 8; ++bz) {
                if (!((culled >> bz) & 1ull)) continue;
                st(col + bz * 256u + (lane & 15u) * 16u);
            }
        }
    } else if (PAT == 2 || PAT == 4) {
        const uint32_t per = (kStrips + gridDim.x - 1) / gridDim.x;
        const uint32_t s0 = PAT == 4 ? blockIdx.x * per : blockIdx.x, s1 = PAT == 4 ? min(kStrips, s0 + per) : kStrips;
        const uint32_t step = PAT == 4 ? 1u : gridDim.x;
        for (uint32_t s = s0; s < s1; s += step) {
            const uint32_t f = lane < 8u ? flags[s * 8u + lane] : 0u;
            const unsigned long long culled = __ballot(f == 1u);
            char *sb = base + (uint64_t)s * kStripBytes;
#pragma unroll
            for (uint32_t r = 0; r < 8; ++r) {
                const uint32_t off = r * 4096u + tid * 16u;
                if ((culled >> ((off >> 8) & 7u)) & 1ull) st(sb + off);
            }
        }
    } else if (PAT == 5) {
        // pattern 0's order (4 KB chunks, block b takes chunks b, b + G, ...) with the brick flags looked up per
        // 256-byte run: run r of the volume = column r / bricks_z (columns are contiguous: (plane, y) rows of nzp
        // voxels), brick bz = r % bricks_z of strip (plane, y / 16); the indices advance by carries, not divisions;
        // 8 chunks' flags are fetched before their 8 stores
        const uint32_t bricks_z = 8, bricks_y = 32, ny = 512;
        const uint64_t nruns = bytes / 256u;
        const uint32_t step = gridDim.x * 16u;  // runs per round of the grid
        const uint32_t dbz = step % bricks_z, dcol = step / bricks_z, dj = dcol % ny, dil = dcol / ny;
        uint64_t r = (uint64_t)blockIdx.x * 16u + (tid >> 4);
        uint32_t bz = (uint32_t)(r % bricks_z), col = (uint32_t)(r / bricks_z), j = col % ny, il = col / ny;
        char *a = base + r * 256u + (tid & 15u) * 16u;
        while (r < nruns) {
            uint32_t f[8];
            char *addr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                addr[u] = a;
                f[u] = r < nruns ? flags[(il * bricks_y + (j >> 4)) * bricks_z + bz] : 0u;
                r += step;
                a += (uint64_t)step * 256u;
                bz += dbz;
                if (bz >= bricks_z) { bz -= bricks_z; ++j; }
                j += dj;
                il += dil;
                if (j >= ny) { j -= ny; ++il; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (f[u] == 1u) st(addr[u]);
        }
    } else if (PAT == 6) {
        // pattern 0's order with a per-chunk 16-bit map (bit s: store run s of the chunk) read through SCALAR loads --
        // their counter (lgkmcnt) is not the stores' (vmcnt), so nothing in the loop waits for a store; eight chunks'
        // words are fetched per wait.  `flags` is used as the map here (all ones).
        typedef const __attribute__((address_space(4))) uint32_t *cmap_t;  // constant address space: scalar loads
        const cmap_t map = (cmap_t)(uintptr_t)flags;
        const uint32_t nchunks = (uint32_t)(bytes / 4096u), G = gridDim.x;
        const uint32_t slotbit = 1u << (tid >> 4), voff = tid * 16u;
        v4i minus = val;
        asm volatile("" : "+v"(minus));
        for (uint32_t c = blockIdx.x; c < nchunks; c += 8u * G) {
            uint32_t m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t cc = c + (uint32_t)u * G;
                const uint32_t word = cc < nchunks ? map[cc >> 1] : 0u;
                m[u] = (cc & 1u) ? word >> 16 : word & 0xffffu;
            }
            const char *ap = base + (uint64_t)c * 4096u;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (m[u] & slotbit) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(voff), "v"(minus), "s"(ap) : "memory");
                ap += (uint64_t)G * 4096u;
            }
        }
        asm volatile("s_nop 1" ::"v"(minus));
    } else if (PAT == 3) {
        const uint32_t nw = gridDim.x * 4u, w = blockIdx.x * 4u + wave;
        for (uint32_t q = w; q < kStrips * 4u; q += nw) {
            const uint32_t s = q >> 2;
            const uint32_t f = lane < 8u ? flags[s * 8u + lane] : 0u;
            const unsigned long long culled = __ballot(f == 1u);
            char *qb = base + (uint64_t)q * 8192u;
#pragma unroll
            for (uint32_t r = 0; r < 8; ++r) {
                const uint32_t off = r * 1024u + lane * 16u;
                if ((culled >> ((off >> 8) & 7u)) & 1ull) st(qb + off);
            }
        }
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <int PAT, bool NT>
static int run(int32_t *buf, uint64_t bytes, const uint8_t *flags, int blocks, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill<PAT, NT>), dim3(blocks), dim3(256), 0, 0, buf, bytes, flags);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fill<PAT, NT>), dim3(blocks), dim3(256), 0, 0, buf, bytes, flags);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    printf("pattern %d nt %d blocks %5d: %7.1f us  %.2f TB/s\n", PAT, (int)NT, blocks, us, bytes / us * 1e-6);
    fflush(stdout);
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return 0;
}

int main(int argc, char **argv) {
    const int PATFLAG = argc > 1 ? atoi(argv[1]) : 1;  // 1: brick flags 'EMPTY' (patterns 1-5); 255: all-ones map (pattern 6)
    const uint64_t bytes = (uint64_t)kStrips * kStripBytes;
    int32_t *buf = nullptr;
    uint8_t *flags = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void **>(&buf), bytes));
    CHECK(hipMalloc(reinterpret_cast<void **>(&flags), kStrips * 16));  // (pattern 6: 2 bytes per 4 KB chunk = 8 per strip... 16 for room)
    CHECK(hipMemset(flags, PATFLAG, kStrips * 16));
    const int grids[] = {64, 128, 192, 256, 384, 512, 1024};
    const int reps = 20;
    for (int g : grids) {
        if (run<0, false>(buf, bytes, flags, g, reps)) return 2;
        if (run<0, true>(buf, bytes, flags, g, reps)) return 2;
        if (run<1, false>(buf, bytes, flags, g, reps)) return 2;
        if (run<1, true>(buf, bytes, flags, g, reps)) return 2;
        if (run<2, false>(buf, bytes, flags, g, reps)) return 2;
        if (run<2, true>(buf, bytes, flags, g, reps)) return 2;
        if (run<3, true>(buf, bytes, flags, g, reps)) return 2;
        if (run<4, true>(buf, bytes, flags, g, reps)) return 2;
        if (run<5, true>(buf, bytes, flags, g, reps)) return 2;
        if (run<6, true>(buf, bytes, flags, g, reps)) return 2;
    }
    (void)hipFree(buf);
    (void)hipFree(flags);
    return 0;
}
