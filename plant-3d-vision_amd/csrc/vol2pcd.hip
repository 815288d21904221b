// vol2pcd.hip -- the carve's immediate consumer on the GPU.
//
// Replaces plant3dvision/proc3d.py::vol2pcd (:490-570), which tasks/proc3d.py:134 calls on the
// Voxels output: binarise (> 0.5), two exact Euclidean distance transforms, signed distance,
// np.gradient, three Gaussian filters (sigma 1), pick the voxels of the level-set shell and
// move each along its normalised gradient.  Running it here means the 4N-byte volume never
// has to cross PCIe: only the shell's points and normals come back.
//
// Arithmetic follows the reference's float64 operations and their order:
//   * EDT: squared distances are integers, sqrt is correctly rounded -> identical to
//     scipy.ndimage.distance_transform_edt.  Only voxels within RADIUS of the surface can
//     influence the output (shell |d| <= lsv + sqrt(3); gradient reach 1 + Gaussian reach 4 per
//     axis), so the transform is exact up to RADIUS and clamped beyond ("far", never used).
//   * np.gradient (unit spacing, edge_order 1): (f[i+1]-f[i-1])/2 inside, one-sided at the ends.
//   * scipy.ndimage.gaussian_filter(sigma=1): radius 4, mode "reflect", axes 0,1,2 in turn,
//     and scipy's symmetric correlate1d order  tmp = f[l]*w0; for j=4..1: tmp += (f[l-j]+f[l+j])*wj
//     (reproduces scipy 1.15 bit for bit; weights are computed by the host with NumPy).
//   * per point: n = g/|g|, p = idx - n*(d + lsv - sqrt(3)/2), normal = -n; the reference takes
//     |g| from BLAS (np.linalg.norm) and open3d renormalises, so the last bits of points and
//     normals are machine-dependent there; here |g| = sqrt((gx*gx+gy*gy)+gz*gz).
// Built with -ffp-contract=off like the rest.

#include <hip/hip_runtime.h>


#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "spacecarve.h"

namespace {

constexpr int kB = 256;
constexpr uint32_t kFar = 0xffffu;
constexpr int kBlk = 8;  // activity is tracked per 8x8x8 block of voxels

int fail_v(int code, const char *msg);  // defined below

// The work buffers (49 bytes per voxel) are kept between calls while they are small (up to 1 GiB: volumes up
// to ~280^3) -- allocating and freeing them was 2.8 of a call's 4.6 ms -- one per device, handed out to one
// caller at a time (a second caller on the same device takes its own, freed at the end of the call).
// sc_vol2pcd_release() gives them back.
struct ScratchSlot { char *base = nullptr; size_t cap = 0; bool busy = false; };
std::mutex g_scratch_mu;
ScratchSlot g_scratch[64];

char *scratch_take(int device, size_t bytes, bool *cached) {
    *cached = false;
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        ScratchSlot &sl = g_scratch[device];
        if (!sl.busy) {
            if (sl.cap < bytes) {
                if (sl.base) (void)hipFree(sl.base);
                sl.base = nullptr;
                sl.cap = 0;
                char *p = nullptr;
                if (hipMalloc(reinterpret_cast<void **>(&p), bytes) != hipSuccess) return nullptr;
                sl.base = p;
                sl.cap = bytes;
            }
            sl.busy = true;
            *cached = true;
            return sl.base;
        }
    }
    char *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), bytes) != hipSuccess) return nullptr;
    return p;
}

// Buffers above this size go back to the device when the call ends: 49 bytes per voxel is 6.6 GB at 512^3 and
// 52 GB at 1024^3 -- HBM a Voxels -> PointCloud worker would hold while later engines, survivor lists and
// two-engine runs allocate -- against 1 ms of a 4.6 ms call saved by keeping them.
constexpr size_t kScratchKeepMax = (size_t)1 << 30;

void scratch_give(int device, char *p, bool cached) {
    if (!p) return;
    if (cached) {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        ScratchSlot &sl = g_scratch[device];
        sl.busy = false;
        if (sl.cap > kScratchKeepMax) {
            (void)hipFree(sl.base);
            sl.base = nullptr;
            sl.cap = 0;
        }
    } else {
        (void)hipFree(p);
    }
}

// Geometry shared by the per-voxel kernels.  They run on the ACTIVE 8^3 blocks only (a list built
// on the device: a few per cent of a plant's volume): thread block b takes list entry b, thread t
// the voxels t and t + 256 of its 512.
struct Vol {
    int nx, ny, nz;          // the planes at hand: the whole volume, or a slab of it with its halo
    int nby, nbz;            // blocks along y and z
    const uint8_t *active;   // [nbx][nby][nbz], 1 = within reach of the surface
    const uint32_t *list;    // linear ids of the blocks a kernel walks (active, or halo)
    const uint32_t *cols;    // columns of blocks (bx * nby + by) holding an active block (shell kernels)
    int xoff;                // plane 0 at hand is plane xoff of the volume (a point's x index is the volume's)
    int cx0, cx1;            // only shell voxels of planes [cx0, cx1) at hand are put out (the slab without its halo)
};

__device__ __forceinline__ bool voxel(const Vol &v, int q, int &x, int &y, int &z, int64_t &i) {
    const uint32_t b = v.list[blockIdx.x];
    const int bz = (int)(b % (uint32_t)v.nbz), by = (int)((b / (uint32_t)v.nbz) % (uint32_t)v.nby),
              bx = (int)(b / ((uint32_t)v.nbz * (uint32_t)v.nby));
    const int l = (int)threadIdx.x + q * kB;  // 0..511: dx = l >> 6, dy = (l >> 3) & 7, dz = l & 7
    x = bx * kBlk + (l >> 6);
    y = by * kBlk + ((l >> 3) & 7);
    z = bz * kBlk + (l & 7);
    if (x >= v.nx || y >= v.ny || z >= v.nz) return false;
    i = ((int64_t)x * v.ny + y) * v.nz + z;
    return true;
}

// binarise (proc3d.py:515): 16 voxels per thread, one 16-byte store
template <typename T>
__global__ __launch_bounds__(kB) void occ_kernel(const T *__restrict__ vol, uint8_t *__restrict__ occ,
                                                 int64_t n) {
    int64_t i0 = ((int64_t)blockIdx.x * kB + threadIdx.x) * 16;
    if (i0 >= n) return;
    if (i0 + 16 <= n && (reinterpret_cast<uintptr_t>(occ) & 15) == 0) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t bits = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) bits |= ((double)vol[i0 + q * 4 + e] > 0.5 ? 1u : 0u) << (8 * e);
            w[q] = bits;
        }
        *reinterpret_cast<uint4 *>(occ + i0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int64_t i = i0; i < min(n, i0 + 16); ++i) occ[i] = (double)vol[i] > 0.5 ? 1 : 0;
    }
}

// The same from carve labels as the ranks of a sharded run leave them after an all-gather: `world` ranks' planes at
// 2 bits (label & 3) or 1 bit (label == 1) per voxel, rank-major ([world][rank_words] words; plane i of the grid is
// plane i / world of rank i % world, or plane i - first(r) of the rank whose slab holds it).  A label is one of
// -1, 0, 1, so `volume > 0.5` (proc3d.py:515) is `label == 1`: the occupancy is read off the packed form and the
// full-size int8 / int32 grid is never written.  occ[0] is voxel 0 of global plane x0.
struct PackedIn {
    const uint32_t *recv;
    uint64_t rank_words;
    uint32_t world, nx_total;
    int32_t cyclic, bits;
};

template <int BITS>
__global__ __launch_bounds__(kB) void occ_packed_kernel(PackedIn pk, uint8_t *__restrict__ occ, int64_t n, uint64_t plane,
                                                        uint32_t x0) {
    constexpr uint32_t PER = 32u / BITS;
    const int64_t i0 = ((int64_t)blockIdx.x * kB + threadIdx.x) * 16;
    if (i0 >= n) return;
    auto locate = [&](uint64_t v, uint32_t &r, uint64_t &src) {  // v: voxel of the GLOBAL grid
        const uint32_t i = (uint32_t)(v / plane);
        const uint64_t within = v - (uint64_t)i * plane;
        uint32_t p;
        if (pk.cyclic) {
            r = i % pk.world;
            p = i / pk.world;
        } else {  // slabs [nx r / world, nx (r + 1) / world)
            r = (uint32_t)(((uint64_t)i * pk.world + pk.world - 1) / pk.nx_total);
            while ((uint64_t)pk.nx_total * r / pk.world > i) --r;
            while ((uint64_t)pk.nx_total * (r + 1) / pk.world <= i) ++r;
            p = i - (uint32_t)((uint64_t)pk.nx_total * r / pk.world);
        }
        src = (uint64_t)p * plane + within;
    };
    const uint64_t g0 = (uint64_t)x0 * plane + (uint64_t)i0;
    if (plane % PER == 0 && i0 + 16 <= n && (reinterpret_cast<uintptr_t>(occ) & 15) == 0) {  // the 16 voxels share a plane and a word
        uint32_t r;
        uint64_t src;
        locate(g0, r, src);
        const uint32_t word = pk.recv[(uint64_t)r * pk.rank_words + src / PER] >> (BITS * (uint32_t)(src % PER));
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t bits = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t lab = (word >> (BITS * (q * 4 + e))) & (BITS == 2 ? 3u : 1u);
                bits |= (lab == 1u ? 1u : 0u) << (8 * e);
            }
            w[q] = bits;
        }
        *reinterpret_cast<uint4 *>(occ + i0) = make_uint4(w[0], w[1], w[2], w[3]);
        return;
    }
    for (int64_t i = i0; i < min(n, i0 + 16); ++i) {
        uint32_t r;
        uint64_t src;
        locate(g0 + (uint64_t)(i - i0), r, src);
        const uint32_t lab = (pk.recv[(uint64_t)r * pk.rank_words + src / PER] >> (BITS * (uint32_t)(src % PER))) & (BITS == 2 ? 3u : 1u);
        occ[i] = lab == 1u ? 1 : 0;
    }
}

// per 8^3 block: bit 0 = holds a foreground voxel, bit 1 = holds a background voxel
__global__ __launch_bounds__(kB) void block_class_kernel(const uint8_t *__restrict__ occ, int nx, int ny,
                                                         int nz, int nbx, int nby, int nbz,
                                                         uint8_t *__restrict__ cls) {
    int64_t b = (int64_t)blockIdx.x * kB + threadIdx.x;
    if (b >= (int64_t)nbx * nby * nbz) return;
    int bz = (int)(b % nbz), by = (int)((b / nbz) % nby), bx = (int)(b / ((int64_t)nbz * nby));
    uint32_t fg = 0, bg = 0;
    for (int dx = 0; dx < kBlk; ++dx) {
        int x = bx * kBlk + dx;
        if (x >= nx) break;
        for (int dy = 0; dy < kBlk; ++dy) {
            int y = by * kBlk + dy;
            if (y >= ny) break;
            const uint8_t *row = occ + ((int64_t)x * ny + y) * nz;
            for (int dz = 0; dz < kBlk; ++dz) {
                int z = bz * kBlk + dz;
                if (z >= nz) break;
                uint8_t o = row[z];
                fg |= o;
                bg |= (uint8_t)(o ^ 1);
            }
        }
    }
    cls[b] = (uint8_t)(fg | (bg << 1));
}

// A voxel can matter only if a voxel of the other class lies within R of it.  Block-level,
// conservative: the block holds class c and some block within rb = ceil(R/8) blocks holds the
// other class.
__global__ __launch_bounds__(kB) void block_active_kernel(const uint8_t *__restrict__ cls, int nbx, int nby,
                                                          int nbz, int rb, uint8_t *__restrict__ active) {
    int64_t b = (int64_t)blockIdx.x * kB + threadIdx.x;
    if (b >= (int64_t)nbx * nby * nbz) return;
    int bz = (int)(b % nbz), by = (int)((b / nbz) % nby), bx = (int)(b / ((int64_t)nbz * nby));
    uint32_t near = 0;
    for (int x = max(0, bx - rb); x <= min(nbx - 1, bx + rb); ++x)
        for (int y = max(0, by - rb); y <= min(nby - 1, by + rb); ++y)
            for (int z = max(0, bz - rb); z <= min(nbz - 1, bz + rb); ++z)
                near |= cls[((int64_t)x * nby + y) * nbz + z];
    uint32_t own = cls[b];
    // own has fg and a bg block is near, or own has bg and an fg block is near
    active[b] = (uint8_t)((((own & 1u) && (near & 2u)) || ((own & 2u) && (near & 1u))) ? 1 : 0);
}

// Compact the block map into two lists: ACTIVE blocks (the per-voxel kernels walk these) and HALO
// blocks (inactive, but within rb blocks of an active one: the EDT passes of active voxels read
// them, so they must hold "far").  counts[0] / counts[1] receive the list lengths.
__global__ __launch_bounds__(kB) void block_lists_kernel(const uint8_t *__restrict__ active, int nbx, int nby,
                                                         int nbz, int rb, uint32_t *__restrict__ act_list,
                                                         uint32_t *__restrict__ halo_list,
                                                         uint32_t *__restrict__ counts) {
    int64_t b = (int64_t)blockIdx.x * kB + threadIdx.x;
    bool is_act = false, is_halo = false;
    if (b < (int64_t)nbx * nby * nbz) {
        is_act = active[b] != 0;
        if (!is_act) {
            int bz = (int)(b % nbz), by = (int)((b / nbz) % nby), bx = (int)(b / ((int64_t)nbz * nby));
            uint32_t near = 0;
            for (int x = max(0, bx - rb); x <= min(nbx - 1, bx + rb); ++x)
                for (int y = max(0, by - rb); y <= min(nby - 1, by + rb); ++y)
                    for (int z = max(0, bz - rb); z <= min(nbz - 1, bz + rb); ++z)
                        near |= active[((int64_t)x * nby + y) * nbz + z];
            is_halo = near != 0;
        }
    }
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    unsigned long long ma = __ballot(is_act), mh = __ballot(is_halo);
    uint32_t ba = 0, bh = 0;
    if (lane == 0) {
        if (ma) ba = atomicAdd(&counts[0], (uint32_t)__popcll(ma));
        if (mh) bh = atomicAdd(&counts[1], (uint32_t)__popcll(mh));
    }
    ba = __shfl(ba, 0);
    bh = __shfl(bh, 0);
    if (is_act) act_list[ba + (uint32_t)__popcll(ma & below)] = (uint32_t)b;
    if (is_halo) halo_list[bh + (uint32_t)__popcll(mh & below)] = (uint32_t)b;
}

// halo blocks: (far, far) in both EDT buffers
__global__ __launch_bounds__(kB) void halo_fill_kernel(uint32_t *__restrict__ g0, uint32_t *__restrict__ g1, Vol v) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int x, y, z;
        int64_t i;
        if (!voxel(v, q, x, y, z, i)) continue;
        g0[i] = 0xffffffffu;
        g1[i] = 0xffffffffu;
    }
}

// columns of blocks (bx, by) that hold an active block: the shell kernels visit only their rows
__global__ __launch_bounds__(kB) void column_list_kernel(const uint8_t *__restrict__ active, int ncols, int nbz,
                                                         uint32_t *__restrict__ col_list,
                                                         uint32_t *__restrict__ count) {
    int c = (int)(blockIdx.x * kB + threadIdx.x);
    bool any = false;
    if (c < ncols)
        for (int bz = 0; bz < nbz; ++bz) any |= active[(int64_t)c * nbz + bz] != 0;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(any), below = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t base = 0;
    if (lane == 0 && m) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    if (any) col_list[base + (uint32_t)__popcll(m & below)] = (uint32_t)c;
}

// EDT pass along z (the contiguous axis).  Two channels per voxel, packed lo/hi 16 bits:
// A = squared distance to the nearest BACKGROUND voxel of the line, B = to the nearest FOREGROUND
// voxel; a voxel's own class gives 0 in the other channel.  Exact up to R, kFar beyond.
// Inactive voxels count as (far, far): no active voxel of the other class lies within R of them,
// so they are never anybody's nearest site (halo blocks are filled with that, the rest is never read).
__global__ __launch_bounds__(kB) void edt_z_kernel(const uint8_t *__restrict__ occ,
                                                   uint32_t *__restrict__ g, Vol v, int R) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int x, y, k;
        int64_t i;
        if (!voxel(v, q, x, y, k, i)) continue;
        uint8_t c = occ[i];
        uint32_t best = kFar;
        for (int d = 1; d <= R; ++d) {
            bool hit = (k - d >= 0 && occ[i - d] != c) || (k + d < v.nz && occ[i + d] != c);
            if (hit) { best = (uint32_t)(d * d); break; }
        }
        g[i] = c ? best : (best << 16);  // fg: A=best,B=0 ; bg: A=0,B=best
    }
}

// EDT pass along y: both channels,  H(p) = min_j ( j^2 + G(p + j*stride) ), |j| <= R.
__global__ __launch_bounds__(kB) void edt_y_kernel(const uint32_t *__restrict__ g,
                                                   uint32_t *__restrict__ h, Vol v, int R) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int x, p, z;
        int64_t i;
        if (!voxel(v, q, x, p, z, i)) continue;
        const int64_t stride = v.nz;
        uint32_t w0 = g[i];
        uint32_t a = w0 & 0xffffu, b = w0 >> 16;
        for (int j = 1; j <= R; ++j) {
            uint32_t jj = (uint32_t)(j * j);
            if (jj >= a && jj >= b) break;  // nothing farther can improve either channel
            if (p - j >= 0) {
                uint32_t w = g[i - j * stride];
                a = min(a, (w & 0xffffu) + jj);
                b = min(b, (w >> 16) + jj);
            }
            if (p + j < v.ny) {
                uint32_t w = g[i + j * stride];
                a = min(a, (w & 0xffffu) + jj);
                b = min(b, (w >> 16) + jj);
            }
        }
        h[i] = min(a, kFar) | (min(b, kFar) << 16);
    }
}

// last EDT pass (along x) fused with the signed distance of proc3d.py:518-522:
//   dist = where(dist > 0.5, dist - 0.5, -mdist + 0.5)
__global__ __launch_bounds__(kB) void edt_x_kernel(const uint32_t *__restrict__ g,
                                                   const uint8_t *__restrict__ occ,
                                                   double *__restrict__ sd, Vol v, int R) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int p, y, z;
        int64_t i;
        if (!voxel(v, q, p, y, z, i)) continue;
        const int64_t stride = (int64_t)v.ny * v.nz;
        bool fg = occ[i] != 0;
        int sh = fg ? 0 : 16;
        uint32_t a = (g[i] >> sh) & 0xffffu;
        for (int j = 1; j <= R; ++j) {
            uint32_t jj = (uint32_t)(j * j);
            if (jj >= a) break;
            if (p - j >= 0) a = min(a, ((g[i - j * stride] >> sh) & 0xffffu) + jj);
            if (p + j < v.nx) a = min(a, ((g[i + j * stride] >> sh) & 0xffffu) + jj);
        }
        double d = sqrt((double)a);  // exact integer in, correctly rounded sqrt
        sd[i] = fg ? d - 0.5 : -d + 0.5;
    }
}

// np.gradient along one axis (unit spacing, edge_order 1).  At the rim of the active region a
// neighbour may be an inactive voxel holding anything: such values can only reach voxels farther
// from the shell than anything the shell's points depend on (see R).
template <int AXIS>
__global__ __launch_bounds__(kB) void gradient_kernel(const double *__restrict__ f,
                                                      double *__restrict__ out, Vol v) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int x, y, z;
        int64_t i;
        if (!voxel(v, q, x, y, z, i)) continue;
        const int64_t stride = AXIS == 0 ? (int64_t)v.ny * v.nz : AXIS == 1 ? v.nz : 1;
        const int p = AXIS == 0 ? x : AXIS == 1 ? y : z;
        const int len = AXIS == 0 ? v.nx : AXIS == 1 ? v.ny : v.nz;
        double r;
        if (p == 0) r = f[i + stride] - f[i];
        else if (p == len - 1) r = f[i] - f[i - stride];
        else r = (f[i + stride] - f[i - stride]) / 2.0;
        out[i] = r;
    }
}

__device__ __forceinline__ int reflect_index(int q, int len) {  // scipy "reflect": d c b a | a b c d | d c b a
    int period = 2 * len;
    q %= period;
    if (q < 0) q += period;
    return q < len ? q : period - 1 - q;
}

struct GaussW { double w[5]; };  // w[0] centre .. w[4] farthest

// scipy.ndimage.correlate1d, symmetric branch, radius 4, along one axis
template <int AXIS>
__global__ __launch_bounds__(kB) void gauss_kernel(const double *__restrict__ f, double *__restrict__ out,
                                                   Vol v, GaussW gw) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int x, y, z;
        int64_t i;
        if (!voxel(v, q, x, y, z, i)) continue;
        const int64_t stride = AXIS == 0 ? (int64_t)v.ny * v.nz : AXIS == 1 ? v.nz : 1;
        const int p = AXIS == 0 ? x : AXIS == 1 ? y : z;
        const int len = AXIS == 0 ? v.nx : AXIS == 1 ? v.ny : v.nz;
        int64_t base = i - (int64_t)p * stride;
        double tmp = f[i] * gw.w[0];
        if (p >= 4 && p + 4 < len) {
#pragma unroll
            for (int j = 4; j >= 1; --j) tmp += (f[i - j * stride] + f[i + j * stride]) * gw.w[j];
        } else {
#pragma unroll
            for (int j = 4; j >= 1; --j)
                tmp += (f[base + (int64_t)reflect_index(p - j, len) * stride] +
                        f[base + (int64_t)reflect_index(p + j, len) * stride]) * gw.w[j];
        }
        out[i] = tmp;
    }
}

// shell test of proc3d.py:535: (dist > -lsv) * (dist <= -lsv + sqrt(3)), on active voxels only (the
// signed distance of the others is never computed).  A chunk is up to 1024 voxels of one (x, y)
// row, chunk id = (x * ny + y) * nzseg + segment -- C order, like the reference's np.argwhere.  Only
// rows of block columns that hold an active block are visited (the counts of the others stay 0).
constexpr int kChunk = 1024;

__device__ __forceinline__ bool on_shell(const double *__restrict__ sd, const Vol &v, int x, int y, int z,
                                         double lo, double hi, int64_t &i, double &d) {
    if (z >= v.nz || x < v.cx0 || x >= v.cx1) return false;
    if (v.active[((int64_t)(x / kBlk) * v.nby + (y / kBlk)) * v.nbz + (z / kBlk)] == 0) return false;
    i = ((int64_t)x * v.ny + y) * v.nz + z;
    d = sd[i];
    return (d > lo) & (d <= hi);
}

// blockIdx.x = 64 * (entry of the active-column list) + row inside the column, blockIdx.y = z segment
__device__ __forceinline__ bool shell_row(const Vol &v, int &x, int &y) {
    const uint32_t c = v.cols[blockIdx.x >> 6], l = blockIdx.x & 63u;
    x = (int)(c / (uint32_t)v.nby) * kBlk + (int)(l >> 3);
    y = (int)(c % (uint32_t)v.nby) * kBlk + (int)(l & 7u);
    return x < v.nx && y < v.ny;
}

__global__ __launch_bounds__(kB) void shell_count_kernel(const double *__restrict__ sd, Vol v,
                                                         double lo, double hi,
                                                         uint32_t *__restrict__ counts) {
    __shared__ uint32_t s[kB / 64];
    int x, y;
    if (!shell_row(v, x, y)) return;  // block-uniform
    uint32_t c = 0;
    for (int q = 0; q < kChunk / kB; ++q) {
        int z = (int)blockIdx.y * kChunk + q * kB + (int)threadIdx.x;
        int64_t i;
        double d;
        c += on_shell(sd, v, x, y, z, lo, hi, i, d) ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        counts[((int64_t)x * v.ny + y) * gridDim.y + blockIdx.y] = s[0] + s[1] + s[2] + s[3];
}

// C-order compaction of the shell + the per-point step of proc3d.py:539-553,563
__global__ __launch_bounds__(kB) void shell_points_kernel(const double *__restrict__ sd,
                                                          const double *__restrict__ gx,
                                                          const double *__restrict__ gy,
                                                          const double *__restrict__ gz, Vol v,
                                                          double lo, double hi,
                                                          double lsv, double ox, double oy, double oz,
                                                          double vs, const uint32_t *__restrict__ counts,
                                                          const uint64_t *__restrict__ offsets,
                                                          double *__restrict__ pts,
                                                          double *__restrict__ nrm) {
    __shared__ uint32_t wsum[kB / 64];
    int x, y;
    if (!shell_row(v, x, y)) return;  // block-uniform
    const int64_t chunk = ((int64_t)x * v.ny + y) * gridDim.y + blockIdx.y;
    if (counts[chunk] == 0) return;  // block-uniform
    uint64_t off = offsets[chunk];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = 0; q < kChunk / kB; ++q) {
        int z = (int)blockIdx.y * kChunk + q * kB + (int)threadIdx.x;
        int64_t i = 0;
        double d = 0.0;
        bool on = on_shell(sd, v, x, y, z, lo, hi, i, d);
        unsigned long long b = __ballot(on);
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < kB / 64; ++w) {
            if (w < wave) before += wsum[w];
            total += wsum[w];
        }
        if (on) {
            unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
            uint64_t r = off + before + (uint32_t)__popcll(b & below);
            double a = gx[i], bq = gy[i], c = gz[i];
            double nn = sqrt((a * a + bq * bq) + c * c);
            double px = NAN, py = NAN, pz = NAN, n0 = NAN, n1 = NAN, n2 = NAN;
            if (nn > 0.0) {
                double u0 = a / nn, u1 = bq / nn, u2 = c / nn;
                double val = d + lsv - sqrt(3.0) / 2.0;
                double xi = (double)(x + v.xoff), yi = (double)y, zi = (double)z;
                // index2point (proc3d.py:45): voxel_size * idx + origin
                px = vs * (xi - u0 * val) + ox;
                py = vs * (yi - u1 * val) + oy;
                pz = vs * (zi - u2 * val) + oz;
                // -grad_normalized, then open3d's normalize_normals
                double m0 = -u0, m1 = -u1, m2 = -u2;
                double mm = sqrt((m0 * m0 + m1 * m1) + m2 * m2);
                n0 = m0 / mm; n1 = m1 / mm; n2 = m2 / mm;
            }
            pts[3 * r] = px; pts[3 * r + 1] = py; pts[3 * r + 2] = pz;
            nrm[3 * r] = n0; nrm[3 * r + 1] = n1; nrm[3 * r + 2] = n2;
        }
        off += total;
        __syncthreads();
    }
}

// Exclusive prefix sum of the chunk counts (C order) on the device, so that only the total
// crosses PCIe: per x-plane sums, then every plane scans its own counts from its base.
__global__ __launch_bounds__(kB) void plane_sums_kernel(const uint32_t *__restrict__ counts, int64_t per_plane,
                                                        uint64_t *__restrict__ plane_sum) {
    __shared__ uint64_t s[kB / 64];
    const uint32_t *c = counts + (int64_t)blockIdx.x * per_plane;
    uint64_t sum = 0;
    for (int64_t i = threadIdx.x; i < per_plane; i += kB) sum += c[i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) plane_sum[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(kB) void chunk_offsets_kernel(const uint32_t *__restrict__ counts, int64_t per_plane,
                                                           const uint64_t *__restrict__ plane_sum,
                                                           uint64_t *__restrict__ offsets,
                                                           uint64_t *__restrict__ total) {
    __shared__ uint64_t s[kB / 64];
    __shared__ uint64_t carry;
    // base of this plane = sum of the planes before it
    uint64_t before = 0;
    for (uint32_t x = threadIdx.x; x < blockIdx.x; x += kB) before += plane_sum[x];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = before;
    __syncthreads();
    if (threadIdx.x == 0) carry = s[0] + s[1] + s[2] + s[3];
    __syncthreads();
    const uint32_t *c = counts + (int64_t)blockIdx.x * per_plane;
    uint64_t *o = offsets + (int64_t)blockIdx.x * per_plane;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t t = 0; t < per_plane; t += kB) {
        const int64_t i = t + threadIdx.x;
        const uint64_t mine = i < per_plane ? c[i] : 0;
        uint64_t inc = mine;  // inclusive scan inside the wavefront
        for (int d = 1; d < 64; d <<= 1) {
            uint64_t up = __shfl_up(inc, d);
            if ((int)lane >= d) inc += up;
        }
        if (lane == 63) s[wave] = inc;
        __syncthreads();
        uint64_t base = carry;
        for (uint32_t w = 0; w < wave; ++w) base += s[w];
        if (i < per_plane) o[i] = base + inc - mine;
        __syncthreads();
        if (threadIdx.x == kB - 1) carry = base + inc;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = carry;
}

thread_local char g_verr[256];
int fail_v(int code, const char *msg) {
    strncpy(g_verr, msg, sizeof g_verr - 1);
    g_verr[sizeof g_verr - 1] = 0;
    return code;
}

#define V_TRY(expr)                                                                \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) { rc = fail_v(_e == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE, hipGetErrorString(_e)); goto done; } \
    } while (0)

inline uint32_t blocks_for(int64_t n) { return (uint32_t)((n + kB - 1) / kB); }

}  // namespace

extern "C" {

const char *sc_vol2pcd_last_error(void) { return g_verr; }

void sc_free_host(void *p) { free(p); }

// What can reach a shell voxel: |d| <= |lsv| + sqrt(3) there, gradient 1 + Gaussian 4 per axis.
static int reach_of(double level_set_value) {
    return (int)std::ceil(std::fabs(level_set_value) + 1.7321 + 5.0 * 1.7321 + 3.0);
}

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// bytes of the work buffers for `planes` x-planes of ny x nz voxels (the layout below)
static size_t scratch_bytes(int64_t planes, int64_t ny, int64_t nz) {
    const int64_t n = planes * ny * nz;
    const int64_t nbx = (planes + kBlk - 1) / kBlk, nby = (ny + kBlk - 1) / kBlk, nbz = (nz + kBlk - 1) / kBlk;
    const int64_t nb = nbx * nby * nbz;
    const int64_t nchunks = planes * ny * ((nz + 1023) / 1024);
    return al256((size_t)n) + 2 * al256((size_t)nb) + 2 * al256((size_t)nb * 4) + al256((size_t)nbx * nby * 4) + al256(64) +
           al256((size_t)planes * 8) + al256((size_t)nchunks * 4) + al256((size_t)nchunks * 8) + 6 * al256((size_t)n * 8);
}

// The pipeline on x-planes [0, nx) AT HAND -- the whole volume, or a slab of it with its halo: `volume` points at
// plane `xoff` of the volume, and only the shell voxels of planes [cx0, cx1) at hand are put out.
static int vol2pcd_range(const void *volume, int on_device, int dtype, int64_t nx, int64_t ny, int64_t nz, int xoff,
                         int cx0, int cx1, const double origin[3], double voxel_size, double level_set_value,
                         const double gauss_w[5], int device, double **points_out, double **normals_out,
                         int64_t *count, const PackedIn *pk = nullptr) {  // pk: packed labels instead of `volume`
    *points_out = *normals_out = nullptr;
    *count = 0;
    const int64_t n = nx * ny * nz;
    const size_t esz = dtype == 0 ? 4 : dtype == 1 ? 4 : dtype == 2 ? 8 : 1;
    const int R = reach_of(level_set_value);
    const int rb = (R + kBlk - 1) / kBlk;
    const int nbx = (int)((nx + kBlk - 1) / kBlk), nby = (int)((ny + kBlk - 1) / kBlk),
              nbz = (int)((nz + kBlk - 1) / kBlk);
    const int64_t nb = (int64_t)nbx * nby * nbz;
    const uint32_t nzseg = (uint32_t)((nz + kChunk - 1) / kChunk);
    const int64_t nchunks = nx * ny * (int64_t)nzseg;
    const double lo = -level_set_value, hi = -level_set_value + std::sqrt(3.0);
    int rc = SC_OK;
    void *vol_d = nullptr;
    char *scratch = nullptr;
    bool scratch_cached = false;
    double *pts_d = nullptr, *nrm_d = nullptr;
    GaussW gw;
    memcpy(gw.w, gauss_w, sizeof gw.w);
    uint64_t total = 0;
    uint32_t nlist[3] = {0, 0, 0};
    hipStream_t st = nullptr;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // one scratch allocation:
    // occ | cls | active | block lists (active, halo) + their lengths + shell total | counts | offsets |
    // sd | A (g0,g1 then ga) | gb | gx | gy | gz
    const size_t o_occ = 0, o_cls = o_occ + al((size_t)n), o_act = o_cls + al((size_t)nb),
                 o_la = o_act + al((size_t)nb), o_lh = o_la + al((size_t)nb * 4), o_lc = o_lh + al((size_t)nb * 4),
                 o_ln = o_lc + al((size_t)nbx * nby * 4),
                 o_pl = o_ln + al(64), o_cnt = o_pl + al((size_t)nx * 8), o_off = o_cnt + al((size_t)nchunks * 4),
                 o_sd = o_off + al((size_t)nchunks * 8), o_a = o_sd + al((size_t)n * 8),
                 o_b = o_a + al((size_t)n * 8), o_gx = o_b + al((size_t)n * 8),
                 o_gy = o_gx + al((size_t)n * 8), o_gz = o_gy + al((size_t)n * 8),
                 bytes = o_gz + al((size_t)n * 8);
    uint8_t *occ, *cls, *act;
    uint32_t *counts, *g0, *g1, *list_act, *list_halo, *list_cols, *list_n;
    uint64_t *offs_d, *total_d, *plane_d;
    double *sd, *ga, *gb, *gx, *gy, *gz;
    Vol v, vh;
    dim3 block(kB);

    if (nb > 0xffffffffLL || nchunks > 0x7fffffffLL) return fail_v(SC_ERR_INVALID, "volume too large");
    V_TRY(hipSetDevice(device));
    scratch = scratch_take(device, bytes, &scratch_cached);
    if (!scratch) { rc = fail_v(SC_ERR_NOMEM, "device allocation of the work buffers failed"); goto done; }
    occ = reinterpret_cast<uint8_t *>(scratch + o_occ);
    cls = reinterpret_cast<uint8_t *>(scratch + o_cls);
    act = reinterpret_cast<uint8_t *>(scratch + o_act);
    list_act = reinterpret_cast<uint32_t *>(scratch + o_la);
    list_halo = reinterpret_cast<uint32_t *>(scratch + o_lh);
    list_cols = reinterpret_cast<uint32_t *>(scratch + o_lc);
    list_n = reinterpret_cast<uint32_t *>(scratch + o_ln);
    total_d = reinterpret_cast<uint64_t *>(scratch + o_ln + 16);
    plane_d = reinterpret_cast<uint64_t *>(scratch + o_pl);
    counts = reinterpret_cast<uint32_t *>(scratch + o_cnt);
    offs_d = reinterpret_cast<uint64_t *>(scratch + o_off);
    sd = reinterpret_cast<double *>(scratch + o_sd);
    ga = reinterpret_cast<double *>(scratch + o_a);
    g0 = reinterpret_cast<uint32_t *>(scratch + o_a);
    g1 = g0 + n;
    gb = reinterpret_cast<double *>(scratch + o_b);
    gx = reinterpret_cast<double *>(scratch + o_gx);
    gy = reinterpret_cast<double *>(scratch + o_gy);
    gz = reinterpret_cast<double *>(scratch + o_gz);
    v = Vol{(int)nx, (int)ny, (int)nz, nby, nbz, act, list_act, list_cols, xoff, cx0, cx1};
    vh = Vol{(int)nx, (int)ny, (int)nz, nby, nbz, act, list_halo, list_cols, xoff, cx0, cx1};
    if (pk != nullptr) {
        on_device = 1;  // (nothing of ours to free)
    } else if (on_device) {
        vol_d = const_cast<void *>(volume);
    } else {
        V_TRY(hipMalloc(&vol_d, (size_t)n * esz));
        V_TRY(hipMemcpy(vol_d, volume, (size_t)n * esz, hipMemcpyHostToDevice));
    }
    if (pk != nullptr) {
        if (pk->bits == 2) hipLaunchKernelGGL(occ_packed_kernel<2>, dim3(blocks_for((n + 15) / 16)), block, 0, st, *pk, occ, n, (uint64_t)ny * (uint64_t)nz, (uint32_t)xoff);
        else hipLaunchKernelGGL(occ_packed_kernel<1>, dim3(blocks_for((n + 15) / 16)), block, 0, st, *pk, occ, n, (uint64_t)ny * (uint64_t)nz, (uint32_t)xoff);
    } else
    switch (dtype) {
        case 0: hipLaunchKernelGGL(occ_kernel<int32_t>, dim3(blocks_for((n + 15) / 16)), block, 0, st, (const int32_t *)vol_d, occ, n); break;
        case 1: hipLaunchKernelGGL(occ_kernel<float>, dim3(blocks_for((n + 15) / 16)), block, 0, st, (const float *)vol_d, occ, n); break;
        case 2: hipLaunchKernelGGL(occ_kernel<double>, dim3(blocks_for((n + 15) / 16)), block, 0, st, (const double *)vol_d, occ, n); break;
        default: hipLaunchKernelGGL(occ_kernel<uint8_t>, dim3(blocks_for((n + 15) / 16)), block, 0, st, (const uint8_t *)vol_d, occ, n); break;
    }
    hipLaunchKernelGGL(block_class_kernel, dim3(blocks_for(nb)), block, 0, st, occ, (int)nx, (int)ny, (int)nz, nbx, nby, nbz, cls);
    hipLaunchKernelGGL(block_active_kernel, dim3(blocks_for(nb)), block, 0, st, cls, nbx, nby, nbz, rb, act);
    V_TRY(hipMemsetAsync(list_n, 0, 64, st));
    hipLaunchKernelGGL(block_lists_kernel, dim3(blocks_for(nb)), block, 0, st, act, nbx, nby, nbz, rb, list_act,
                       list_halo, list_n);
    hipLaunchKernelGGL(column_list_kernel, dim3(blocks_for((int64_t)nbx * nby)), block, 0, st, act, nbx * nby, nbz,
                       list_cols, list_n + 2);
    V_TRY(hipMemsetAsync(counts, 0, (size_t)nchunks * 4, st));  // rows the shell kernels skip count nothing
    V_TRY(hipGetLastError());
    // the only host round trip before the result: how many blocks the per-voxel kernels walk
    V_TRY(hipMemcpy(nlist, list_n, sizeof nlist, hipMemcpyDeviceToHost));
    if (nlist[0] > 0) {
        const dim3 ga_(nlist[0]);
        // halo blocks hold (far, far) in both EDT buffers; other inactive voxels are never read
        if (nlist[1] > 0) hipLaunchKernelGGL(halo_fill_kernel, dim3(nlist[1]), block, 0, st, g0, g1, vh);
        hipLaunchKernelGGL(edt_z_kernel, ga_, block, 0, st, occ, g0, v, R);
        hipLaunchKernelGGL(edt_y_kernel, ga_, block, 0, st, g0, g1, v, R);
        hipLaunchKernelGGL(edt_x_kernel, ga_, block, 0, st, g1, occ, sd, v, R);
        V_TRY(hipGetLastError());
        // gradient along each axis, then gaussian_filter: axes 0, 1, 2 in turn (proc3d.py:525-531)
        hipLaunchKernelGGL(gradient_kernel<0>, ga_, block, 0, st, sd, ga, v);
        hipLaunchKernelGGL(gauss_kernel<0>, ga_, block, 0, st, ga, gb, v, gw);
        hipLaunchKernelGGL(gauss_kernel<1>, ga_, block, 0, st, gb, ga, v, gw);
        hipLaunchKernelGGL(gauss_kernel<2>, ga_, block, 0, st, ga, gx, v, gw);
        hipLaunchKernelGGL(gradient_kernel<1>, ga_, block, 0, st, sd, ga, v);
        hipLaunchKernelGGL(gauss_kernel<0>, ga_, block, 0, st, ga, gb, v, gw);
        hipLaunchKernelGGL(gauss_kernel<1>, ga_, block, 0, st, gb, ga, v, gw);
        hipLaunchKernelGGL(gauss_kernel<2>, ga_, block, 0, st, ga, gy, v, gw);
        hipLaunchKernelGGL(gradient_kernel<2>, ga_, block, 0, st, sd, ga, v);
        hipLaunchKernelGGL(gauss_kernel<0>, ga_, block, 0, st, ga, gb, v, gw);
        hipLaunchKernelGGL(gauss_kernel<1>, ga_, block, 0, st, gb, ga, v, gw);
        hipLaunchKernelGGL(gauss_kernel<2>, ga_, block, 0, st, ga, gz, v, gw);
        V_TRY(hipGetLastError());
        const dim3 rows(nlist[2] * 64u, nzseg);
        hipLaunchKernelGGL(shell_count_kernel, rows, block, 0, st, sd, v, lo, hi, counts);
        hipLaunchKernelGGL(plane_sums_kernel, dim3((uint32_t)nx), block, 0, st, counts, ny * (int64_t)nzseg, plane_d);
        hipLaunchKernelGGL(chunk_offsets_kernel, dim3((uint32_t)nx), block, 0, st, counts, ny * (int64_t)nzseg, plane_d,
                           offs_d, total_d);
        V_TRY(hipGetLastError());
        V_TRY(hipMemcpy(&total, total_d, sizeof total, hipMemcpyDeviceToHost));
    }
    if (total > 0) {
        V_TRY(hipMalloc(reinterpret_cast<void **>(&pts_d), (size_t)total * 48));
        nrm_d = pts_d + total * 3;
        hipLaunchKernelGGL(shell_points_kernel, dim3(nlist[2] * 64u, nzseg), block, 0, st, sd, gx, gy, gz, v, lo, hi, level_set_value,
                           origin[0], origin[1], origin[2], voxel_size, counts, offs_d, pts_d, nrm_d);
        V_TRY(hipGetLastError());
        *points_out = static_cast<double *>(malloc((size_t)total * 24));
        *normals_out = static_cast<double *>(malloc((size_t)total * 24));
        if (!*points_out || !*normals_out) { rc = fail_v(SC_ERR_NOMEM, "host allocation failed"); goto done; }
        V_TRY(hipMemcpy(*points_out, pts_d, (size_t)total * 24, hipMemcpyDeviceToHost));
        V_TRY(hipMemcpy(*normals_out, nrm_d, (size_t)total * 24, hipMemcpyDeviceToHost));
    }
    *count = (int64_t)total;

done:
    if (rc != SC_OK) {
        free(*points_out); free(*normals_out);
        *points_out = *normals_out = nullptr;
        *count = 0;
    }
    (void)hipDeviceSynchronize();
    if (!on_device && vol_d) (void)hipFree(vol_d);
    scratch_give(device, scratch, scratch_cached);
    if (pts_d) (void)hipFree(pts_d);
    return rc;
}

// Largest work buffers a call may take (0: no limit).  49 bytes per voxel is 6.6 GB at 512^3 and 52 GB at
// 1024^3; above the limit the volume goes through in x-SLABS: every step of the pipeline has a finite reach
// along x (the distance transform is exact up to the reach R and clamped beyond, the gradient reaches 1 plane,
// the three Gaussians 4), so the planes [c0, c1) with H = R + 8 planes of halo on either side give the shell
// voxels of [c0, c1) the values the whole volume would give them -- the one-sided differences and the
// reflections at a slab's artificial ends stay inside the halo.  Slabs come in x order, which is the order of
// the output (C order of the shell voxels).
// (Default 8 GiB: a 512^3 volume still goes through in one piece, 4.6 ms against 7.5 in 1 GiB slabs; buffers above
// 1 GiB are given back when the call ends either way, see scratch_give.)
static int64_t g_scratch_limit = (int64_t)8 << 30;

void sc_vol2pcd_set_scratch_limit(int64_t bytes) { g_scratch_limit = bytes < 0 ? 0 : bytes; }

static int vol2pcd_driver(const void *volume, int on_device, int dtype, int64_t nx, int64_t ny, int64_t nz,
                          const double origin[3], double voxel_size, double level_set_value,
                          const double gauss_w[5], int device, double **points_out, double **normals_out,
                          int64_t *count, const PackedIn *pk) {
    if ((!volume && !pk) || !origin || !gauss_w || !points_out || !normals_out || !count)
        return fail_v(SC_ERR_INVALID, "null argument");
    if (nx < 2 || ny < 2 || nz < 2) return fail_v(SC_ERR_INVALID, "np.gradient needs at least 2 voxels per axis");
    if (nx > 65535 || ny > 65535) return fail_v(SC_ERR_INVALID, "x and y are limited to 65535 voxels");
    if (dtype < 0 || dtype > 3) return fail_v(SC_ERR_INVALID, "volume dtype: 0 int32, 1 float32, 2 float64, 3 uint8");
    if (!(std::fabs(level_set_value) < 200.0)) return fail_v(SC_ERR_INVALID, "level_set_value out of range");
    *points_out = *normals_out = nullptr;
    *count = 0;
    const int64_t limit = g_scratch_limit;
    const int H = reach_of(level_set_value) + 8;
    int64_t planes = nx;
    if (limit > 0 && scratch_bytes(nx, ny, nz) > (size_t)limit) {
        const size_t per_plane = scratch_bytes(64, ny, nz) / 64 + 1;
        planes = std::max<int64_t>((int64_t)((size_t)limit / per_plane), 2 * H + 8);  // at least 8 planes of its own per slab
    }
    if (planes >= nx)
        return vol2pcd_range(volume, on_device, dtype, nx, ny, nz, 0, 0, (int)nx, origin, voxel_size, level_set_value,
                             gauss_w, device, points_out, normals_out, count, pk);
    const int64_t S = planes - 2 * H;
    const size_t esz = dtype == 0 ? 4 : dtype == 1 ? 4 : dtype == 2 ? 8 : 1;
    std::vector<double *> ps, ns;
    std::vector<int64_t> cs;
    int rc = SC_OK;
    int64_t total = 0;
    for (int64_t c0 = 0; c0 < nx && rc == SC_OK; c0 += S) {
        const int64_t c1 = std::min(nx, c0 + S), a = std::max<int64_t>(0, c0 - H), b = std::min(nx, c1 + H);
        double *p = nullptr, *q = nullptr;
        int64_t c = 0;
        rc = vol2pcd_range(pk ? nullptr : static_cast<const char *>(volume) + (size_t)a * ny * nz * esz, on_device, dtype, b - a, ny, nz,
                           (int)a, (int)(c0 - a), (int)(c1 - a), origin, voxel_size, level_set_value, gauss_w, device, &p, &q, &c, pk);
        ps.push_back(p); ns.push_back(q); cs.push_back(c);
        total += c;
    }
    if (rc == SC_OK && total > 0) {
        double *P = static_cast<double *>(malloc((size_t)total * 24)), *N = static_cast<double *>(malloc((size_t)total * 24));
        if (!P || !N) {
            free(P); free(N);
            rc = fail_v(SC_ERR_NOMEM, "host allocation failed");
        } else {
            int64_t at = 0;
            for (size_t k = 0; k < cs.size(); ++k) {
                if (cs[k] > 0) {
                    memcpy(P + 3 * at, ps[k], (size_t)cs[k] * 24);
                    memcpy(N + 3 * at, ns[k], (size_t)cs[k] * 24);
                    at += cs[k];
                }
            }
            *points_out = P;
            *normals_out = N;
            *count = total;
        }
    }
    for (size_t k = 0; k < ps.size(); ++k) { free(ps[k]); free(ns[k]); }
    return rc;
}

int sc_vol2pcd(const void *volume, int on_device, int dtype, int64_t nx, int64_t ny, int64_t nz,
               const double origin[3], double voxel_size, double level_set_value,
               const double gauss_w[5], int device, double **points_out, double **normals_out,
               int64_t *count) {
    return vol2pcd_driver(volume, on_device, dtype, nx, ny, nz, origin, voxel_size, level_set_value, gauss_w, device,
                          points_out, normals_out, count, nullptr);
}

int sc_vol2pcd_packed(const void *recv_dev, int64_t rank_bytes, int world, int partition, int bits, int64_t nx,
                      int64_t ny, int64_t nz, const double origin[3], double voxel_size, double level_set_value,
                      const double gauss_w[5], int device, double **points_out, double **normals_out, int64_t *count) {
    if (!recv_dev) return fail_v(SC_ERR_INVALID, "null argument");
    if (bits != 1 && bits != 2) return fail_v(SC_ERR_INVALID, "bits must be 1 or 2");
    if (partition != 0 && partition != 1) return fail_v(SC_ERR_INVALID, "partition: 0 plane-cyclic, 1 slabs");
    if (world < 1 || nx < world || rank_bytes < 0 || (rank_bytes & 3)) return fail_v(SC_ERR_INVALID, "bad world / stride");
    const uint64_t pmax = (uint64_t)(nx + world - 1) / world;
    if ((uint64_t)rank_bytes * 8 < pmax * (uint64_t)ny * (uint64_t)nz * (uint64_t)bits)
        return fail_v(SC_ERR_INVALID, "rank stride too small for its planes");
    const PackedIn pk{static_cast<const uint32_t *>(recv_dev), (uint64_t)rank_bytes / 4, (uint32_t)world, (uint32_t)nx,
                      partition == 0 ? 1 : 0, bits};
    return vol2pcd_driver(nullptr, 1, 3, nx, ny, nz, origin, voxel_size, level_set_value, gauss_w, device, points_out,
                          normals_out, count, &pk);
}

void sc_vol2pcd_release(void) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    int current = -1;
    const bool restore = hipGetDevice(&current) == hipSuccess;  // the caller's current device stays what it was
    for (int d = 0; d < 64; ++d) {
        ScratchSlot &sl = g_scratch[d];
        if (sl.base && !sl.busy && hipSetDevice(d) == hipSuccess) {
            (void)hipFree(sl.base);
            sl.base = nullptr;
            sl.cap = 0;
        }
    }
    if (restore) (void)hipSetDevice(current);
}

}  // extern "C"
