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
    uint32_t stride;
    uint32_t nread;  // bricks whose labels the sender read to make this buffer (diagnostic: the others were settled by a verdict byte)
    uint32_t pad[2];
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

struct SparseCounters {  // per call parity; every counter on a 128-byte line of its own (returning atomics on one line
                         // serialise at ~11 ns each: the first version had three on one line and one atomic per brick
                         // and wavefront -- 15 700 of them on a plant, 187 us for a kernel that moves 33 MB)
    uint32_t nmixed, pad0[31];
    uint32_t nwork, pad1[31];
    uint32_t done, pad2[31];
};

struct SparseScan {
    const uint8_t *flags;  // level 2: the last batch's verdict bytes; level 1: the dead bytes; level 0: unused
    uint32_t nbricks;
    int32_t level;
    uint32_t code_full, code_untouched;  // what a FULL (2) / an UNTOUCHED (6) brick holds (level 2)
    uint32_t *work;                      // unknown bricks are appended here (count in SparseCounters::nwork); null: they are
                                         // on the lists the pack role is given (the engine's live and late lists)
};
struct SparseLists {  // the bricks whose labels are read: entry i of list q at listq[i * stepq], i < *countq (null: no list q)
    // (plain members, not arrays: the compiler merges `q == 0 ? list[0] : ...` into an indexed access of a private copy
    // of the struct, which it keeps in LDS -- and a kernel whose private data is in LDS reads its dispatch packet from
    // host memory to number its threads: 11 us per launch, measured)
    const uint32_t *list0, *list1, *list2;
    const uint32_t *count0, *count1, *count2;
    int32_t step0, step1, step2;
};

// Blocks [0, nscan): one lane per brick, the code of every brick a verdict byte settles (and, with sc.work, the list of
// the others).  Blocks [nscan, gridDim): a block takes 32 consecutive list entries per turn, a WAVEFRONT 8 of them --
// lane l reads the 16 labels (lane & 3) * 16 .. + 15 of column lane >> 2 of a brick, four 16-byte loads, the eight bricks
// in one flight, and holds word l of each brick's 64 in a register -- three ballots per brick say whether it is
// uniform; the block's mixed bricks of the turn take their slots with ONE atomic (returning atomics on one line
// serialise at ~11 ns: one per brick was 84 us of them on a plant) and leave with a coalesced 256-byte store each;
// lanes 0..7 write the wavefront's 8 codes.  The block that finishes last (a counter, nobody waits) writes the
// header.
constexpr uint32_t kSparseTurn = 8;  // bricks per wavefront and turn: ONE flight of loads (16 in two flights measured 3.5 us slower per batch)

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
    __shared__ uint32_t s_mixed[kBlock / 64], s_base;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t npack = gridDim.x - nscan;
    uint32_t n0 = 0, n1 = 0, n2 = 0;
    if (sl.list0 != nullptr) n0 = *sl.count0;
    if (sl.list1 != nullptr) n1 = *sl.count1;
    if (sl.list2 != nullptr) n2 = *sl.count2;
    n0 = min(n0, hdr.nbricks); n1 = min(n1, hdr.nbricks); n2 = min(n2, hdr.nbricks);  // (a list holds bricks: never more)
    const uint32_t total = n0 + n1 + n2;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t jl = lane >> 2, kq = (lane & 3u) * 16u;
    constexpr uint32_t kBlockTurn = kSparseTurn * (kBlock / 64);  // 32 entries per block and turn
    for (uint32_t tb = (blockIdx.x - nscan) * kBlockTurn; tb < total; tb += npack * kBlockTurn) {  // block-uniform
        // lane i < 16 holds entry tb + 16 wave + i: its brick (0xffffffff: none), its first column and voxel, and the
        // element offset of that corner in the state -- the divisions once per lane, not once per brick and wavefront
        uint32_t mine = 0xffffffffu, j0 = 0, kb = 0, off_lo = 0, off_hi = 0;
        {
            const uint32_t t = tb + wave * kSparseTurn + lane;
            if (lane < kSparseTurn && t < total) {
                const uint32_t *lp = t < n0 ? sl.list0 : (t < n0 + n1 ? sl.list1 : sl.list2);
                const int64_t at = t < n0 ? (int64_t)t * sl.step0
                                          : (t < n0 + n1 ? (int64_t)(t - n0) * sl.step1 : (int64_t)(t - n0 - n1) * sl.step2);
                mine = lp[at];
                if (mine >= hdr.nbricks) mine = 0xffffffffu;
            }
            if (mine != 0xffffffffu) {
                const uint32_t il = mine / per_plane, rem = mine - il * per_plane;
                const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
                j0 = by * kBrickY;
                kb = bz * kBrickZ;
                const uint64_t off = ((uint64_t)il * g.ny + j0) * g.nzp + kb;
                off_lo = (uint32_t)off;
                off_hi = (uint32_t)(off >> 32);
            }
        }
        uint32_t w[kSparseTurn];
        uint32_t codes2 = 0, mixed = 0;  // wave-uniform: 2 bits / 1 bit per entry of the turn
        constexpr uint32_t kFlight = 8;   // bricks whose loads are in flight together (32 x 16 bytes per lane)
#pragma unroll
        for (uint32_t s8 = 0; s8 < kSparseTurn; s8 += kFlight) {
            int4 q[kFlight][4];
            uint32_t bj[kFlight], bk[kFlight];
            bool have[kFlight];
#pragma unroll
            for (uint32_t i = 0; i < kFlight; ++i) {
                have[i] = __builtin_amdgcn_readlane(mine, s8 + i) != 0xffffffffu;  // wave-uniform
                bj[i] = __builtin_amdgcn_readlane(j0, s8 + i);
                bk[i] = __builtin_amdgcn_readlane(kb, s8 + i);
                const uint64_t off = ((uint64_t)__builtin_amdgcn_readlane(off_hi, s8 + i) << 32) | __builtin_amdgcn_readlane(off_lo, s8 + i);
                // No branch around a load (the compiler waits for every load it has put behind one: 64 round trips in a
                // row, 34 us): a lane without a brick, beyond ny or beyond nz reads the first labels of the volume instead,
                // and the validity masks below drop what it read
                const int32_t *p = labels + off + (uint64_t)jl * g.nzp + kq;  // rows are padded to whole 16-byte groups
                const bool row_ok = have[i] && bj[i] + jl < g.ny;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int32_t *pe = (row_ok && bk[i] + kq + 4u * e < g.nz) ? p + 4 * e : labels;
                    q[i][e] = *reinterpret_cast<const int4 *>(pe);
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < kFlight; ++i) {
                uint32_t word = 0, vmw = 0;
                if (have[i] && bj[i] + jl < g.ny) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t k = bk[i] + kq + 4u * e;
                        const uint32_t nvalid = k < g.nz ? min(4u, g.nz - k) : 0u;
                        const uint32_t vm = nvalid >= 4u ? 0xffu : ((1u << (2u * nvalid)) - 1u);
                        const uint32_t byte = ((uint32_t)q[i][e].x & 3u) | (((uint32_t)q[i][e].y & 3u) << 2) |
                                              (((uint32_t)q[i][e].z & 3u) << 4) | (((uint32_t)q[i][e].w & 3u) << 6);
                        word |= (byte & vm) << (8 * e);
                        vmw |= vm << (8 * e);
                    }
                }
                w[s8 + i] = word;
                if (!have[i]) continue;
                const bool all3 = __ballot(((word ^ 0xffffffffu) & vmw) != 0u) == 0ull;
                const bool all1 = __ballot(((word ^ 0x55555555u) & vmw) != 0u) == 0ull;
                const bool all0 = __ballot(word != 0u) == 0ull;
                const uint32_t code = all3 ? 3u : (all1 ? 1u : (all0 ? 0u : kSparseMixed));
                codes2 |= code << (2u * (s8 + i));
                if (code == kSparseMixed) mixed |= 1u << (s8 + i);
            }
        }
        // the block's mixed bricks of this turn take their slots with ONE atomic
        if (lane == 0) s_mixed[wave] = (uint32_t)__popc(mixed);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t n = 0;
#pragma unroll
            for (int q = 0; q < kBlock / 64; ++q) n += s_mixed[q];
            s_base = n ? atomicAdd(&cnt->nmixed, n) : 0u;
        }
        __syncthreads();
        uint32_t base = s_base;
        for (uint32_t q = 0; q < wave; ++q) base += s_mixed[q];
        base = __builtin_amdgcn_readfirstlane(base);
        __syncthreads();  // (s_mixed / s_base have been read before the next turn writes them)
#pragma unroll
        for (uint32_t i = 0; i < kSparseTurn; ++i) {
            if (!((mixed >> i) & 1u)) continue;  // wave-uniform
            const uint32_t slot = base + (uint32_t)__popc(mixed & ((1u << i) - 1u));
            if (slot < hdr.cap) {
                if (lane == 0) ids[slot] = __builtin_amdgcn_readlane(mine, i);
                payload[(uint64_t)slot * 64u + lane] = w[i];
            }
        }
        if (mine != 0xffffffffu) codes[mine] = (uint8_t)((codes2 >> (2u * lane)) & 3u);  // lanes 0..7
    }
    // No fence here: a block's slot reservations are RETURNING atomics (their values were used above), so they have been
    // performed at the device's coherence point before this barrier, and the counter below is incremented behind it --
    // the block that reads `prev == npack - 1` reads every reservation.  (__threadfence() here is an agent-scope release:
    // a write-back of the XCD's L2, 4 MB of it dirty with the labels the batch just wrote -- 20 us for 256 blocks that had
    // nothing else to do, measured.)  What the kernel stores for later kernels and copies is released by its end.
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t prev = atomicAdd(&cnt->done, 1u);
        if (prev == npack - 1u) {  // the last block to finish: every slot has been asked for
            uint32_t *h = reinterpret_cast<uint32_t *>(wire);
            const uint32_t nm = atomicAdd(&cnt->nmixed, 0u);
            h[0] = hdr.magic; h[1] = hdr.version; h[2] = hdr.bits; h[3] = hdr.nbricks; h[4] = hdr.cap; h[5] = nm;
            h[6] = hdr.planes; h[7] = hdr.ny; h[8] = hdr.nz; h[9] = hdr.bricks_y; h[10] = hdr.bricks_z; h[11] = hdr.first;
            h[12] = hdr.stride; h[13] = total; h[14] = 0u; h[15] = 0u;
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
    auto decode = [](uint32_t lab) -> uint32_t {  // the output element as its bit pattern
        return OCC ? (lab == 1u ? 1u : 0u) : (lab == 3u ? (sizeof(OUT) == 1 ? 0xffu : 0xffffffffu) : lab);
    };
    OUT *p = out + ((uint64_t)i * in.ny + j) * in.nz + k0;
    const uint32_t nvalid = min(16u, in.nz - k0);
    // (no private array: one that is indexed by a loop counter lives in LDS, see SparseLists)
    if (nvalid == 16u && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        if (sizeof(OUT) == 1) {
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                o[q] = decode((word >> (8 * q)) & 3u) | (decode((word >> (8 * q + 2)) & 3u) << 8) |
                       (decode((word >> (8 * q + 4)) & 3u) << 16) | (decode((word >> (8 * q + 6)) & 3u) << 24);
            *reinterpret_cast<uint4 *>(p) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                reinterpret_cast<uint4 *>(p)[q] = make_uint4(decode((word >> (8 * q)) & 3u), decode((word >> (8 * q + 2)) & 3u),
                                                             decode((word >> (8 * q + 4)) & 3u), decode((word >> (8 * q + 6)) & 3u));
        }
    } else {
        for (uint32_t q = 0; q < nvalid; ++q) p[q] = (OUT)decode((word >> (2u * q)) & 3u);
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
