// Part of spacecarve.hip (included there, inside its anonymous namespace, behind sc_misc.h) -- the BRICK-SPARSE
// transport form of carve labels (round 6; multi-GPU assembly; no reference counterpart: the reference is
// single-device, cl.py:29-30, and its consumer reads the whole int32 array, cl.py:229-232).
//
// A carved volume is almost everywhere uniform: on the benchmarked plant 94 % of the 16 x 64-voxel bricks are proved
// empty from four corner projections each and carry a 1-byte verdict in HBM already; 99.7 % of the labels are -1.
// The dense 2-bit form (pack_labels_kernel) still visits every word of the 512 MiB state's index space and puts
// 32 MiB per rank on the links.  This form sends, per rank:
//
//     SparseHeader (64 bytes)
//     codes   [nbricks]  one byte per brick: 0 / 1 / 3 = every voxel of the brick is 0 / 1 / -1 (the label & 3 of
//                        the dense form), 2 = MIXED: the brick's labels travel            (padded to 64 bytes)
//     ids     [cap]      the brick index of payload slot s                                  (cap % 16 == 0)
//     payload [cap][256] slot s: the brick's 16 x 64 labels at 2 bits each, voxel (column jl, depth kl) of the brick
//                        at bits 2 ((jl * 64 + kl) % 16) of word (jl * 64 + kl) / 16; voxels beyond ny / nz are 0
//
// 131 KB + 260 bytes per mixed brick -- 2 MB on the plant (7 577 live bricks) against 32 MiB -- and the pack kernel
// reads the labels of the bricks it does not already know only: the verdict bytes of the batch that made the labels
// (`level` 2: a single fused batch on a fresh volume -- 1 / 4: all -1, 2: FULL, 6: UNTOUCHED, see brick_flags_kernel),
// the dead bytes of earlier launches (`level` 1: all -1 until the next clear), or nothing (`level` 0: every brick is
// read).  A brick that is read and turns out uniform -- the inside of a solid, a live brick the views carved out --
// gets its code, not a slot: `mixed` is what a consumer has to look at.
// `cap` is the sender's capacity; header.nmixed > cap means slots were refused (the codes are still right): every
// receiver sees the same headers and asks for the gather again with a larger capacity (sharded.py).

constexpr uint32_t kSparseMagic = 0x50534353u;  // "SCSP"
constexpr uint32_t kSparseBrickBytes = kBrickY * kBrickZ / 4;  // 256
constexpr uint32_t kSparseMixed = 2u;

struct SparseHeader {
    uint32_t magic, version, bits, nbricks;
    uint32_t cap, nmixed, planes, ny;
    uint32_t nz, bricks_y, bricks_z, first;  // first / stride: the rank's planes are first, first + stride, ... of the grid
    uint32_t stride, pad[3];
};
static_assert(sizeof(SparseHeader) == 64, "SparseHeader layout");

// byte offsets inside a rank's buffer (the same arithmetic on both sides; spacecarve.h: sc_sparse_rank_bytes)
struct SparseLayout {
    uint64_t codes, ids, payload, total;
};
__host__ __device__ inline SparseLayout sparse_layout(uint32_t nbricks, uint32_t cap) {
    SparseLayout l;
    l.codes = 64;
    l.ids = l.codes + (((uint64_t)nbricks + 63u) & ~(uint64_t)63u);
    l.payload = l.ids + (((uint64_t)cap * 4u + 63u) & ~(uint64_t)63u);
    l.total = l.payload + (uint64_t)cap * kSparseBrickBytes;
    return l;
}

struct SparseCounters {  // one 128-byte line per call parity
    uint32_t nmixed, nwork, done, pad[29];
};

struct SparseScan {
    const uint8_t *flags;  // level 2: the last batch's verdict bytes; level 1: the dead bytes; level 0: unused
    uint32_t nbricks;
    int32_t level;
    uint32_t code_full, code_untouched;  // what a FULL (2) / an UNTOUCHED (6) brick holds (level 2)
    uint32_t *work;                      // unknown bricks are appended here (count in SparseCounters::nwork); null: they are
                                         // on the lists the pack role is given (the engine's live and late lists)
};
struct SparseLists {  // the bricks whose labels are read: entry i of list q at list[q][i * step[q]], i < *count[q]
    const uint32_t *list[3];
    const uint32_t *count[3];
    int32_t step[3];
};

// Blocks [0, nscan): one lane per brick, the code of every brick a verdict byte settles (and, with sc.work, the list of
// the others).  Blocks [nscan, gridDim): one WAVEFRONT per listed brick and turn -- lane l reads the 16 labels
// (lane & 3) * 16 .. + 15 of column lane >> 2, four 16-byte loads in one flight, and holds word l of the brick's 64 --
// three ballots say whether the brick is uniform; a mixed one takes a slot with one atomic and leaves with one
// coalesced 256-byte store.  The wavefront that finishes last (a counter, nobody waits) writes the header.
__global__ __launch_bounds__(kBlock) void sparse_pack_kernel(const int32_t *__restrict__ labels, GridDesc g, uint32_t bricks_y,
                                                            uint32_t bricks_z, SparseScan sc, uint32_t nscan, SparseLists sl,
                                                            char *__restrict__ wire, SparseHeader hdr, SparseCounters *cnt,
                                                            SparseCounters *cnt_next) {
    const SparseLayout lay = sparse_layout(hdr.nbricks, hdr.cap);
    uint8_t *codes = reinterpret_cast<uint8_t *>(wire + lay.codes);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (blockIdx.x < nscan) {
        const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
        bool unknown = false;
        if (b < sc.nbricks) {
            uint32_t code = kSparseMixed;
            if (sc.level == 2) {
                const uint32_t f = sc.flags[b];
                code = (f == 1u || f == 4u) ? 3u : (f == 2u ? sc.code_full : (f == 6u ? sc.code_untouched : kSparseMixed));
            } else if (sc.level == 1) {
                code = sc.flags[b] != 0 ? 3u : kSparseMixed;
            }
            unknown = code == kSparseMixed;
            if (!unknown) codes[b] = (uint8_t)code;
        }
        if (sc.work != nullptr) {
            const unsigned long long m = __ballot(unknown);
            if (m != 0) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&cnt->nwork, (uint32_t)__popcll(m));
                base = __shfl(base, 0);
                if (unknown) sc.work[base + (uint32_t)__popcll(m & below)] = b;
            }
        }
        return;
    }
    if (blockIdx.x == nscan && threadIdx.x == 0 && cnt_next != nullptr) {  // the next call's counters
        cnt_next->nmixed = 0u;
        cnt_next->nwork = 0u;
        cnt_next->done = 0u;
    }
    uint32_t *ids = reinterpret_cast<uint32_t *>(wire + lay.ids);
    uint32_t *payload = reinterpret_cast<uint32_t *>(wire + lay.payload);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nw = (gridDim.x - nscan) * (kBlock / 64), wv = (blockIdx.x - nscan) * (kBlock / 64) + wave;
    uint32_t n0 = 0, n1 = 0, n2 = 0;
    if (sl.list[0] != nullptr) n0 = *sl.count[0];
    if (sl.list[1] != nullptr) n1 = *sl.count[1];
    if (sl.list[2] != nullptr) n2 = *sl.count[2];
    n0 = min(n0, hdr.nbricks); n1 = min(n1, hdr.nbricks); n2 = min(n2, hdr.nbricks);  // (a list holds bricks: never more)
    const uint32_t total = n0 + n1 + n2;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t jl = lane >> 2, kq = (lane & 3u) * 16u;
    for (uint32_t t = wv; t < total; t += nw) {  // wave-uniform
        uint32_t b;
        if (t < n0) b = sl.list[0][(int64_t)t * sl.step[0]];
        else if (t < n0 + n1) b = sl.list[1][(int64_t)(t - n0) * sl.step[1]];
        else b = sl.list[2][(int64_t)(t - n0 - n1) * sl.step[2]];
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= hdr.nbricks) continue;
        const uint32_t il = b / per_plane, rem = b - il * per_plane;
        const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
        const uint32_t j = by * kBrickY + jl, k0 = bz * kBrickZ + kq;
        uint32_t word = 0, vmw = 0;
        if (j < g.ny) {
            const int32_t *p = labels + ((uint64_t)il * g.ny + j) * g.nzp + k0;  // rows are padded to whole 16-byte groups
            int4 q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                q[e] = make_int4(0, 0, 0, 0);
                if (k0 + 4u * e < g.nz) q[e] = *reinterpret_cast<const int4 *>(p + 4 * e);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t k = k0 + 4u * e;
                const uint32_t nvalid = k < g.nz ? min(4u, g.nz - k) : 0u;
                const uint32_t vm = nvalid >= 4u ? 0xffu : ((1u << (2u * nvalid)) - 1u);
                const uint32_t byte = ((uint32_t)q[e].x & 3u) | (((uint32_t)q[e].y & 3u) << 2) | (((uint32_t)q[e].z & 3u) << 4) |
                                      (((uint32_t)q[e].w & 3u) << 6);
                word |= (byte & vm) << (8 * e);
                vmw |= vm << (8 * e);
            }
        }
        const bool all3 = __ballot(((word ^ 0xffffffffu) & vmw) != 0u) == 0ull;
        const bool all1 = __ballot(((word ^ 0x55555555u) & vmw) != 0u) == 0ull;
        const bool all0 = __ballot(word != 0u) == 0ull;
        uint32_t code = all3 ? 3u : (all1 ? 1u : (all0 ? 0u : kSparseMixed));
        if (code == kSparseMixed) {
            uint32_t slot = 0;
            if (lane == 0) slot = atomicAdd(&cnt->nmixed, 1u);
            slot = __shfl(slot, 0);
            if (slot < hdr.cap) {
                if (lane == 0) ids[slot] = b;
                payload[(uint64_t)slot * 64u + lane] = word;
            }
        }
        if (lane == 0) codes[b] = (uint8_t)code;
    }
    if (lane == 0) {
        __threadfence();
        const uint32_t prev = atomicAdd(&cnt->done, 1u);
        if (prev == nw - 1u) {  // the last wavefront to finish: every slot has been asked for
            __threadfence();
            hdr.nmixed = atomicAdd(&cnt->nmixed, 0u);
            *reinterpret_cast<SparseHeader *>(wire) = hdr;
        }
    }
}

// The other end: `world` ranks' buffers `rank_bytes` apart, as an all-gather leaves them, into ONE grid in global
// order -- int8 / int32 labels, or (OCC) the uint8 occupancy label == 1 that vol2pcd binarises to (proc3d.py:515).
// Blocks [0, nfill): one wavefront per (rank, brick) with a uniform code; the others: one wavefront per (rank, slot).
struct SparseIn {
    const char *recv;
    uint64_t rank_bytes;
    uint32_t world, nx;
    int32_t cyclic;
    uint32_t ny, nz, bricks_y, bricks_z;
    uint32_t nbricks_max;  // bricks of the rank with the most planes
    uint32_t cap_max;      // largest capacity of a rank
};

template <typename OUT, bool OCC>
__device__ __forceinline__ void sparse_put_brick(const SparseIn &in, OUT *__restrict__ out, uint32_t r, uint32_t first,
                                                 uint32_t stride, uint32_t b, uint32_t word, uint32_t lane) {
    const uint32_t per_plane = in.bricks_y * in.bricks_z;
    const uint32_t il = b / per_plane, rem = b - il * per_plane;
    const uint32_t by = rem / in.bricks_z, bz = rem - by * in.bricks_z;
    const uint32_t i = first + il * stride;
    const uint32_t j = by * kBrickY + (lane >> 2), k0 = bz * kBrickZ + (lane & 3u) * 16u;
    if (i >= in.nx || j >= in.ny || k0 >= in.nz) return;
    OUT vals[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t lab = (word >> (2 * q)) & 3u;
        vals[q] = OCC ? (OUT)(lab == 1u ? 1 : 0) : (OUT)(lab == 3u ? -1 : (int)lab);
    }
    OUT *p = out + ((uint64_t)i * in.ny + j) * in.nz + k0;
    const uint32_t nvalid = min(16u, in.nz - k0);
    if (nvalid == 16u && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < (int)(16 * sizeof(OUT) / 16); ++q)
            reinterpret_cast<uint4 *>(p)[q] = reinterpret_cast<const uint4 *>(vals)[q];
    } else {
        for (uint32_t q = 0; q < nvalid; ++q) p[q] = vals[q];
    }
}

template <typename OUT, bool OCC>
__global__ __launch_bounds__(kBlock) void sparse_unpack_kernel(SparseIn in, OUT *__restrict__ out, uint32_t nfill) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x < nfill) {
        const uint64_t t = (uint64_t)blockIdx.x * (kBlock / 64) + wave;
        const uint32_t r = (uint32_t)(t / in.nbricks_max), b = (uint32_t)(t - (uint64_t)r * in.nbricks_max);
        if (r >= in.world) return;
        const char *base = in.recv + (uint64_t)r * in.rank_bytes;
        const SparseHeader h = *reinterpret_cast<const SparseHeader *>(base);
        if (b >= h.nbricks) return;
        const uint32_t code = reinterpret_cast<const uint8_t *>(base + 64)[b];
        if (code == kSparseMixed) return;
        sparse_put_brick<OUT, OCC>(in, out, r, h.first, h.stride, b, code * 0x55555555u, lane);
        return;
    }
    const uint64_t t = (uint64_t)(blockIdx.x - nfill) * (kBlock / 64) + wave;
    const uint32_t r = (uint32_t)(t / in.cap_max), s = (uint32_t)(t - (uint64_t)r * in.cap_max);
    if (r >= in.world) return;
    const char *base = in.recv + (uint64_t)r * in.rank_bytes;
    const SparseHeader h = *reinterpret_cast<const SparseHeader *>(base);
    if (s >= min(h.nmixed, h.cap)) return;
    const SparseLayout lay = sparse_layout(h.nbricks, h.cap);
    const uint32_t b = reinterpret_cast<const uint32_t *>(base + lay.ids)[s];
    if (b >= h.nbricks) return;
    const uint32_t word = reinterpret_cast<const uint32_t *>(base + lay.payload)[(uint64_t)s * 64u + lane];
    sparse_put_brick<OUT, OCC>(in, out, r, h.first, h.stride, b, word, lane);
}
