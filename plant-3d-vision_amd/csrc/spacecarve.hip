// spacecarve.hip -- MI355X (gfx950 / CDNA4) voxel back-projection engine + its C ABI.
//
// Replaces, behind include/spacecarve.h, the device layer of the reference:
//   plant3dvision/kernels/backprojection.c  carve :57-84, average :36-55,
//                                           backproject_point :3-34
//   plant3dvision/kernels/common.h          unravel_index :1-12
//   plant3dvision/cl.py                     Backprojection buffer lifecycle :118-311
//
// Design:
//   * State stays in the reference's layout: C-order [nx][ny][nz], int32 / float32.
//   * One lane owns 4 consecutive z-voxels of one (i,j) column: one 16-byte load and one
//     16-byte store per lane, 1 KiB per wavefront instruction; one owner per voxel, so
//     there are no atomics and no races (as in the reference, one work-item per voxel).
//   * A launch applies a CHUNK of views to the state it holds in registers (1 view per
//     launch = the reference's schedule).  The carve update is order-independent, so
//     dead lanes drop out and a wavefront leaves the view loop as soon as a ballot says
//     every one of its voxels is carved.
//   * A fused carve (many views) first settles whole 16x64-voxel BRICKS from four corner
//     projections each: bricks some view sees entirely over background are EMPTY (-1, filled
//     by store blocks beside the final stage), bricks every view sees entirely over
//     foreground are FULL (0 -> 1).  Only the remaining LIVE bricks are projected voxel by
//     voxel, for two views; the voxels still alive are compacted into survivor lists
//     (wave-aggregated atomics on 256 sharded counters) and finished by persistent kernels
//     with one lane per survivor.
//   * Carve masks live in HBM as 1 bit per pixel in 32x32-pixel tiles (one 128-byte line
//     per tile): the 64..256 z-neighbours a wavefront projects land on a short image
//     segment of arbitrary orientation, i.e. on a handful of lines, whatever the camera roll.
//   * The x/y partial sums of every dot product are hoisted per column WITHOUT changing
//     the reference's left-to-right rounding: ((R0*x + R1*y) + R2*z) + t0.
//   * Arithmetic contract: IEEE binary32, no FMA contraction (built with
//     -ffp-contract=off), correctly rounded division, and the (int) cast guarded so
//     that NaN / inf / out-of-range are rejected exactly like x86 cvttss2si -> INT_MIN.
//
// gfx950 only.  No fallback path: every entry point fails with SC_ERR_DEVICE when HIP
// cannot run the kernels.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "spacecarve.h"
#include "spacecarve_tuning.h"

namespace {

// ------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------

struct ViewDesc {   // 128 bytes, read with scalar loads (the view index is wave-uniform)
    float K[4];     // fx fy cx cy
    float R[9];     // row-major
    float t[3];
    const void *mask;  // carve: tiled bit words; average: float32 [H][W]
    int32_t W, H;
    int32_t tiles_x;
    int32_t pad;
    float Wf, Hf;
    const uint8_t *occ;  // carve: one byte per 32x32 tile: bit 0 some foreground, bit 1 only foreground
    int32_t safe;        // certify_view(): every voxel centre of the grid has 2^-10 < pz and |px|, |py|, pz < 2^30
                         // under this pose, and the intrinsics are finite and below 2^30 (see project())
    int32_t pad2;
    const uint32_t *cmask;  // carve, 16-byte pack form: per 32x32 tile the 4x4 map of its 8x8-pixel CELLS -- bits 0..15
                            // "cell holds some foreground", bits 16..31 "cell holds some background" (cell (cx, cy) of
                            // the tile at bit cy * 4 + cx; padding counts as background); null: no cell level
    uint64_t reserved;
};
static_assert(sizeof(ViewDesc) == 128, "ViewDesc layout");

struct GridDesc {
    float ox, oy, oz, vs;
    uint32_t ny, nz;
    uint32_t i0;        // global x index of the engine's first plane
    uint32_t gpc;       // 4-voxel groups per row = nzp / 4 (the last ones of a padded row own fewer than 4 voxels, or none)
    uint64_t ngroups;   // columns owned * gpc
    uint32_t istride;   // global x step between the engine's planes (1: slab, W: plane-cyclic)
    uint32_t nzp;       // row pitch of the state in voxels: nz rounded up to a multiple of 64 (a row = one
                        // (plane, column) run of nz voxels, 256-byte aligned; the padding is never read back)
};

constexpr int kBlock = 256;
constexpr int kTile = 32;        // mask tile edge in pixels (32 rows x 32 bits = 128 B)
constexpr int kSub = 256;        // sharded append counters = sub-lists of a survivor list
constexpr int kStreamGroups = 2; // 16-byte groups per lane in the per-view streaming kernel
                                 // (measured with streaming loads: 2 -> 0.0803, 3 -> 0.0811, 4 -> 0.0860 ms)
constexpr int kXcdRun = 16;      // consecutive logical blocks kept on one XCD

// Survivor lists of the fused carve (see carve_list_kernel).  Zeroed before every fused launch.
// Every counter sits on a 128-byte line of its own: returning device-scope atomics on one
// line serialise (~90 per microsecond measured), on different lines they do not.
struct alignas(128) ListCounter {
    uint32_t n;
    uint32_t pad[31];
};
struct ListCtl {
    ListCounter count[5][kSub];  // entries appended per sub-list, one set per list stage; set 3: bulk units, set 4:
                                 // their work items
    uint32_t overflow;           // a sub-list ran out of room: the dense resume kernel takes over
    uint32_t nlive[2];           // brick form: bricks no view found empty (entries of the live list); the flags
                                 // kernel of launch q counts in word q & 1 and zeroes the other one, so launches
                                 // that keep the same block (fewer than 6 views: no survivor stages) need no memset
    uint32_t nlate;              // FULL candidates a later view did not keep whole (entries of the late list)
    uint32_t nfill[2];           // settled bricks that need a fill (entries of the fill list), same alternation
    uint32_t pad[26];
    ListCounter xcd_next[64];    // dense stage: ticket counters for the live list, 8 per XCD (index xcd * 8 + c: the
                                 // wavefronts of XCD k whose number ends in c share one; see carve_brick_kernel)
    ListCounter cand;            // .n non-zero: the flags kernel left FULL candidates open (the confirm kernel has work).
                                 // A flag on a line of its own, read before it is written: as a count (an atomic per
                                 // block, 11 ns each on one address) it cost 22 us when every brick is a candidate,
                                 // and as a plain store on the line of the live-brick counter it doubled that
                                 // kernel's time on a bulky object (the atomics on `nlive` waited behind the stores)
};

// Bricks whose -1 fill is left to the final list stage (see carve_list_kernel).
struct CullStores {
    const uint8_t *flags;  // null: nothing deferred
    uint32_t bricks_y, bricks_z, nstrips, first;  // strips [first, nstrips) are filled there
    int32_t kept, fresh;   // see Fill
    uint32_t fill_blocks;  // 0: one store block per strip; n: n persistent store blocks
    int32_t init;          // see Fill
};

// XCD-aware block remap.  Blocks b and b+8 share an XCD (round-robin dispatch); runs of
// kXcdRun consecutive logical blocks (neighbouring columns, which project onto the same mask
// lines) stay on one XCD's L2.  Which XCD takes which run of a group of 8 rotates from group
// to group: a grid plane is a whole number of runs, so a fixed deal would hand the busy stripe
// of every plane (the columns under the object) to the same few XCDs -- measured 12 % slower
// on the fused carve and 10 % on the streaming kernel.  Speed only: any placement gives the
// same result.
__device__ __forceinline__ uint32_t spread_block(uint32_t bid, uint32_t nblocks) {
    uint32_t full = nblocks - nblocks % (8u * kXcdRun);
    if (bid >= full) return bid;
    uint32_t xcd = bid & 7u, seq = bid >> 3;
    uint32_t grp = seq / kXcdRun;
    xcd = (xcd + grp * 3u + (grp >> 3) * 5u) & 7u;  // rotate: no XCD owns a fixed stripe of y
    return (grp * 8u + xcd) * kXcdRun + (seq % kXcdRun);
}

// Correctly rounded p/pz for BOTH image coordinates from ONE reciprocal.
// hipcc expands an IEEE f32 division into  div_scale x2, rcp, 2 fma (Newton step on the
// reciprocal), mul, 4 fma (two corrections of the quotient), div_fmas, div_fixup.  When the
// operands are in a range where v_div_scale scales nothing and v_div_fixup fixes nothing
// (denominator and both numerators normal, within 2^+-40: see the V_DIV_SCALE_F32 rules), that
// expansion is exactly the plain-FMA sequence below, so running it by hand with the refined
// reciprocal SHARED between the two numerators gives bit-identical quotients with 13
// instructions instead of 22 (and one quarter-rate v_rcp_f32 instead of two).  Any lane outside
// the range sends its whole wavefront through the compiler's division.
// sc_selftest_division() compares the two bit-for-bit on 2^32 operand pairs.
__device__ __forceinline__ bool div_fast_range(float px, float py, float pz) {
    // fmin/fmax drop a NaN operand, so NaNs are excluded by explicit (ordered) comparisons
    bool ordered = !__builtin_isunordered(px, py);
    float lo = fminf(fabsf(px), fabsf(py));
    float hi = fmaxf(fmaxf(fabsf(px), fabsf(py)), pz);
    return ordered & (pz > 0x1p-40f) & (lo > 0x1p-40f) & (hi < 0x1p40f);  // also false for
                                                            // zero numerators, pz <= 0, inf
}
__device__ __forceinline__ float refined_rcp(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_by_rcp(float n, float d, float r) {
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

// backproject_point (backprojection.c:3-34) with the x/y partial sums hoisted.
// a{x,y,z} = R[0]*x + R[1]*y etc. (rounded as the reference rounds them).
//
// What the instructions cost on gfx950 (tools/probes/valu_probe.hip, cycles of a SIMD per wavefront
// instruction, independent instructions, 8 wavefronts per SIMD): v_mul_f32 / v_add_f32 / v_sub_f32 /
// v_and / v_lshrrev / v_add_u32 / v_mov 2.6; every three-operand or VOP3-only form (v_fma_f32, v_cmp_*,
// v_cvt_*, v_min/max, v_bfi, v_mad_*) 4.3-4.7; v_rcp_f32 8.3.  Where the host has certified the pose
// (d.safe: every voxel of the grid has 2^-10 < pz and |px|, |py|, pz < 2^30, intrinsics finite and below
// 2^30) the range test of the fast division is the two comparisons left of it and the picture test is
// two unsigned comparisons of the truncated coordinates -- uf, vf are finite there, v_cvt_i32_f32
// truncates toward zero ((-1, 0) -> 0, accepted like the reference's cast) and saturates, so
// (unsigned)u < W is exactly  uf > -1 && uf < W.
__device__ __forceinline__ bool project(float ax, float ay, float az, float z,
                                        const ViewDesc &d, int &u, int &v) {
    float pz = (az + d.R[8] * z) + d.t[2];  // :11
    float px = (ax + d.R[2] * z) + d.t[0];  // :17
    float py = (ay + d.R[5] * z) + d.t[1];  // :18
    const bool safe = d.safe != 0;          // wave-uniform
    unsigned long long outside;             // lanes whose operands the fast division does not cover
    if (safe) {
        outside = __builtin_amdgcn_ballot_w64(!(fabsf(px) > 0x1p-40f)) | __builtin_amdgcn_ballot_w64(!(fabsf(py) > 0x1p-40f));
        asm volatile("" : "+s"(outside));  // keeps the two ballots apart: merged, the lane predicate
    } else {                               // travels through a vector register and back (two more instructions)
        outside = __builtin_amdgcn_ballot_w64(!div_fast_range(px, py, pz));
        asm volatile("" : "+s"(outside));
    }
    if (outside == 0) {
        const float r = refined_rcp(pz);
        // (the packed forms v_pk_mul/fma_f32 were tried for the two chains: no faster in these kernels)
        const float uf = div_by_rcp(px, pz, r) * d.K[0] + d.K[2];  // :20
        const float vf = div_by_rcp(py, pz, r) * d.K[1] + d.K[3];  // :21
        u = (int)uf;
        v = (int)vf;
        if (safe) return ((uint32_t)u < (uint32_t)d.W) & ((uint32_t)v < (uint32_t)d.H);
        return (uf > -1.0f) & (uf < d.Wf) & (vf > -1.0f) & (vf < d.Hf);  // pz > 0 here
    }
    const float qx = px / pz, qy = py / pz;
    float uf = qx * d.K[0] + d.K[2];  // :20
    float vf = qy * d.K[1] + d.K[3];  // :21
    // :13 rejects pz < 0 (not NaN, not -0); :23-31 reject (int)uf outside [0, W-1].
    // Truncation toward zero accepts uf in (-1, 0); NaN/inf/huge fail the comparisons,
    // which is what the cvttss2si INT_MIN result does in the canonical restatement.
    // (bitwise &: one straight-line predicate, no short-circuit branches)
    bool ok = !(pz < 0.0f) & (uf > -1.0f) & (uf < d.Wf) & (vf > -1.0f) & (vf < d.Hf);
    u = (int)uf;
    v = (int)vf;
    return ok;
}

__device__ __forceinline__ uint32_t mask_word_index(int u, int v, int tiles_x) {
    // all factors are < 2^24 for in-image pixels: the 24-bit multiply is full rate
    return (__umul24((uint32_t)(v >> 5), (uint32_t)tiles_x) + (uint32_t)(u >> 5)) * 32u +
           (uint32_t)(v & 31);
}

// The mask pointer comes out of a descriptor, so the compiler cannot tell its address space and
// would emit flat loads; it is always global memory.
typedef const __attribute__((address_space(1))) uint32_t *gmask_t;
__device__ __forceinline__ uint32_t load_mask_word(const void *mask, uint32_t word) {
    return ((gmask_t)(uintptr_t)mask)[word];
}

struct Vox4 {
    uint64_t elem;   // offset of the group's first voxel in the slab state
    uint32_t k0;     // z index of that voxel
    uint32_t nvalid; // 0..4 voxels of this group that exist (nz tail, row padding)
    float x, y;
};

// group index -> column and z run (common.h:6-8: z fastest), voxel centre x, y
__device__ __forceinline__ void decode_group(const GridDesc &g, uint64_t grp, Vox4 &vx) {
    uint32_t col, kq;
    if (g.ngroups <= 0xffffffffull) {
        uint32_t g32 = (uint32_t)grp;
        col = g32 / g.gpc;
        kq = g32 - col * g.gpc;
    } else {
        col = (uint32_t)(grp / g.gpc);
        kq = (uint32_t)(grp - (uint64_t)col * g.gpc);
    }
    uint32_t il = col / g.ny;
    uint32_t j = col - il * g.ny;
    vx.k0 = kq * 4u;
    vx.nvalid = vx.k0 < g.nz ? min(4u, g.nz - vx.k0) : 0u;
    vx.elem = (uint64_t)col * g.nzp + vx.k0;  // == grp * 4: rows are whole groups
    // backprojection.c:71-72 -- origin + (float)index * voxel_size, GLOBAL x index of the plane
    vx.x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
    vx.y = g.oy + (float)(int)j * g.vs;
}

// Where a fused launch appends the voxels that are still alive after its dense views.
struct Append {
    uint32_t *list;   // nullptr: no append
    ListCtl *ctl;
    uint32_t subcap;  // entries per sub-list
    uint32_t sub;     // sub-list of this block
    // brick form: a wavefront's share of a brick (16 columns x 16 voxels, a UNIT) with at least `bulk_min`
    // voxels alive after the dense views goes on the bulk list as a whole instead of voxel by voxel
    // (unit_verdicts: the views are asked about the unit, one view per lane, before any projects its voxels)
    uint32_t *bulk;   // nullptr: no such list
    uint32_t bulkcap; // units per sub-list
    uint32_t bulk_min;
};

// carve (backprojection.c:57-84) of one 4-voxel group over views[0..nviews), state in
// registers.  FRESH: the state is known to be `init` everywhere (nothing applied since
// create/clear) and is not read.  VEC: nz % 4 == 0, state accessed as int4 (`pre` holds the
// group's state, already loaded by the caller so that loads of several groups overlap).
template <bool FRESH, bool VEC>
__device__ __forceinline__ void carve_group(int32_t *__restrict__ labels, const GridDesc &g,
                                            const ViewDesc *__restrict__ views, int nviews,
                                            int32_t init, uint64_t grp, int4 pre,
                                            const Append &ap) {
    Vox4 vx;
    int32_t lab[4], was[4];
    // a grid whose rows are padded (nz not a multiple of 64) has groups that own fewer than four voxels:
    // they are told apart up front; on an unpadded grid a group is decoded only if something in it lives
    const bool padded = g.nzp != g.nz;  // grid-uniform
    if (!VEC || padded) decode_group(g, grp, vx);
    int32_t *p = labels + (VEC ? grp * 4 : vx.elem);
    if (FRESH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) lab[e] = init;
    } else if (VEC) {
        lab[0] = pre.x; lab[1] = pre.y; lab[2] = pre.z; lab[3] = pre.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) lab[e] = (e < (int)vx.nvalid) ? p[e] : -1;
    }
    uint32_t alive = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (VEC && padded && e >= (int)vx.nvalid) lab[e] = -1;  // padding counts as carved
        was[e] = lab[e];
        if ((VEC || e < (int)vx.nvalid) && lab[e] != -1) alive |= 1u << e;  // :67
    }
    if (!FRESH && alive == 0) return;  // nothing to do and nothing to write
    if (VEC && !padded) decode_group(g, grp, vx);

    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = g.oz + (float)(int)(vx.k0 + e) * g.vs;  // :73

    for (int vi = 0; vi < nviews; ++vi) {
        if (__ballot(alive != 0) == 0) break;  // whole wavefront carved
        const ViewDesc d = views[vi];          // wave-uniform: scalar loads, once per view
        float ax = d.R[0] * vx.x + d.R[1] * vx.y;
        float ay = d.R[3] * vx.x + d.R[4] * vx.y;
        float az = d.R[6] * vx.x + d.R[7] * vx.y;
        const uint32_t *bits = static_cast<const uint32_t *>(d.mask);
        bool ok[4];
        uint32_t w[4];
        int sh[4];
        // all four projections first, then the four gathers in flight together
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int u, v;
            ok[e] = project(ax, ay, az, z[e], d, u, v) & ((alive >> e) & 1u);
            sh[e] = u & 31;
            // unconditional gather (word 0 when the voxel is out): no branch per element
            w[e] = load_mask_word(bits, ok[e] ? mask_word_index(u, v, d.tiles_x) : 0u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ok[e]) {
                if (((w[e] >> sh[e]) & 1u) == 0) {  // :79
                    lab[e] = -1;
                    alive &= ~(1u << e);
                } else if (lab[e] == 0) {  // :81
                    lab[e] = 1;
                }
            }
        }
    }

    if (VEC) {
        bool changed = FRESH || lab[0] != was[0] || lab[1] != was[1] || lab[2] != was[2] ||
                       lab[3] != was[3];
        if (changed) *reinterpret_cast<int4 *>(p) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < (int)vx.nvalid && (FRESH || lab[e] != was[e])) p[e] = lab[e];
    }

    if (ap.list != nullptr) {
        // survivors -> sub-list `ap.sub`: one atomic per wavefront, entries = slab-local voxel
        // index, bit 31 = "label is still 0" (a later foreground hit must write 1)
        unsigned long long b[4];
        uint32_t total = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b[e] = __ballot((alive >> e) & 1u);
            total += (uint32_t)__popcll(b[e]);
        }
        if (total != 0) {  // wave-uniform
            unsigned long long act = __ballot(1);
            uint32_t lane = __lane_id();
            uint32_t base = 0;
            if (lane == (uint32_t)(__ffsll((long long)act) - 1))
                base = atomicAdd(&ap.ctl->count[0][ap.sub].n, total);
            base = __shfl(base, __ffsll((long long)act) - 1);
            if (base + total > ap.subcap) {
                if (lane == (uint32_t)(__ffsll((long long)act) - 1)) ap.ctl->overflow = 1u;
            } else {
                uint32_t *dst = ap.list + (size_t)ap.sub * ap.subcap + base;
                unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                uint32_t off = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if ((alive >> e) & 1u) {
                        uint32_t rank = off + (uint32_t)__popcll(b[e] & below);
                        dst[rank] = (uint32_t)(vx.elem + e) | (lab[e] == 0 ? 0x80000000u : 0u);
                    }
                    off += (uint32_t)__popcll(b[e]);
                }
            }
        }
    }
}

// A chunk of views per launch: one group per lane; optionally appends the survivors.
template <bool FRESH, bool VEC>
__global__ __launch_bounds__(kBlock) void carve_kernel(int32_t *__restrict__ labels, GridDesc g,
                                                       const ViewDesc *__restrict__ views,
                                                       int nviews, int32_t init, Append ap) {
    uint32_t lb = spread_block(blockIdx.x, gridDim.x);
    uint64_t grp = (uint64_t)lb * kBlock + threadIdx.x;
    if (grp >= g.ngroups) return;
    int4 pre = make_int4(0, 0, 0, 0);
    if (!FRESH && VEC) pre = *reinterpret_cast<const int4 *>(labels + grp * 4);
    ap.sub = (lb * 0x9E3779B1u) >> 24;  // kSub == 256: hashed, so a dense region loads every sub-list alike
    carve_group<FRESH, VEC>(labels, g, views, nviews, init, grp, pre, ap);
}

// Mask ingest, fast form for 1-byte masks whose rows are 16-byte aligned multiples of 16 px.
// A block turns 128-pixel x 32-row panels into 32x32 tiles.  Lane l of wavefront w loads 16
// pixels: row 8w + l/8 of the panel, 16-byte chunk l%8 of that row's 128-byte line -- so every
// wavefront load instruction reads 8 whole lines, and kPackRows of them are in flight per lane.
// 16 bytes -> 16 bits in-lane (SWAR non-zero test + one multiply per dword), neighbouring lanes
// join their halves with one shuffle, and the even lanes store the tile words.
__device__ __forceinline__ uint32_t nonzero_nibble(uint32_t w) {
    uint32_t t = (w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;  // bit 7 of every non-zero byte
    return (t * 0x00204081u) >> 28;  // gathers bits 7,15,23,31 into a nibble (no carries collide)
}

// (Measured: 44-50 us for 72 masks of 1440x1080 whatever ROWS is, and the same for a band form
// reading whole rows contiguously.  tools/probes/read_probe.hip: a plain read of those 112 MB
// takes 41 us when they come from HBM -- every step writes 0.5 GB of labels in between, so they
// do -- and 19 us from the Infinity Cache.  The kernel sits on the cold-read floor.)
// A batch of 1-byte masks to pack: slots [slot0, slot0 + nslots) of the packed arena, slot s taking
// the raw view order[s] (the views of a fused carve are packed in the order they will be applied,
// so that the first few can be packed ahead and the rest beside the dense stage).
constexpr int kPackOrderMax = 256;
struct PackJob {
    const uint8_t *raw;
    int64_t row_stride, view_stride;
    int32_t W, H, tiles_x, tiles_y;
    uint32_t *out;
    int64_t out_view_words;
    uint32_t flip;      // 0 plain, 0xffffffff for np.invert on uint8, 0x01010101 for np.invert on bool bytes
    int32_t use_order;  // 0: slot s takes raw view s
    uint8_t *occ;
    uint32_t *cmask;    // per tile: the 4x4 map of its 8x8-pixel cells, [slot][tiles_y][tiles_x] (see ViewDesc)
    int32_t slot0, nslots;
    uint16_t order[kPackOrderMax];
};

template <int ROWS>  // tile rows per block: that many 16-byte loads in flight per lane
__device__ __forceinline__ void pack16_block(const PackJob &pj, uint32_t b) {
    __shared__ uint32_t cm_s[ROWS * 4];  // per tile of the block: cells with some foreground | cells with some background << 16
    const int W = pj.W, H = pj.H, tiles_x = pj.tiles_x, tiles_y = pj.tiles_y;
    const uint32_t flip = pj.flip;
    const int lane = threadIdx.x & 63;
    const int txb = (tiles_x + 3) >> 2;            // panels per tile row
    const int tyb = (tiles_y + ROWS - 1) / ROWS;   // block rows per view
    int bx = (int)(b % (uint32_t)txb);
    uint32_t r = b / (uint32_t)txb;
    int by = (int)(r % (uint32_t)tyb);
    int slot = (int)(r / (uint32_t)tyb);
    if (slot >= pj.nslots) return;  // block-uniform
    slot += pj.slot0;
    const int64_t view = pj.use_order ? (int64_t)pj.order[slot] : (int64_t)slot;
    const uint8_t *raw = pj.raw + view * pj.view_stride;
    if (threadIdx.x < ROWS * 4) cm_s[threadIdx.x] = 0;
    const int wave = (int)(threadIdx.x >> 6);
    int row = wave * 8 + (lane >> 3);  // row inside the tile
    int c = lane & 7;                  // 16-pixel chunk inside the panel
    int u0 = bx * 128 + c * 16;
    int tx = bx * 4 + (c >> 1);
    uint4 q[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        int v = (by * ROWS + k) * 32 + row;
        q[k] = make_uint4(flip, flip, flip, flip);  // padding stays background after the flip
        if (v < H && u0 < W)  // W % 16 == 0: a 16-pixel run is inside the row or outside it
            q[k] = *reinterpret_cast<const uint4 *>(raw + (int64_t)v * pj.row_stride + u0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        int ty = by * ROWS + k;
        uint32_t half = nonzero_nibble(q[k].x ^ flip) | (nonzero_nibble(q[k].y ^ flip) << 4) |
                        (nonzero_nibble(q[k].z ^ flip) << 8) | (nonzero_nibble(q[k].w ^ flip) << 12);
        uint32_t other = __shfl_xor(half, 1);
        uint32_t word = half | (other << 16);
        if ((c & 1) == 0 && tx < tiles_x && ty < tiles_y)
            pj.out[(int64_t)slot * pj.out_view_words + ((int64_t)ty * tiles_x + tx) * 32 + row] = word;
        // 8x8-pixel cells: this wavefront holds rows 8w .. 8w + 7 of the tile (cell row w), lane 8r + c the
        // pixels 16c .. 16c + 15 of row r -- two cells' worth.  Four ballots; bit c + 8r of each belongs to
        // lane 8r + c, so lane c < 8 reads off its two cells over the eight rows.
        const unsigned long long any_a = __ballot((half & 0xffu) != 0u), any_b = __ballot((half >> 8) != 0u);
        const unsigned long long all_a = __ballot((half & 0xffu) == 0xffu), all_b = __ballot((half >> 8) == 0xffu);
        if (lane < 8) {
            constexpr unsigned long long M = 0x0101010101010101ull;
            const uint32_t fa = ((any_a >> c) & M) != 0 ? 1u : 0u, fb = ((any_b >> c) & M) != 0 ? 1u : 0u;
            const uint32_t ha = ((all_a >> c) & M) != M ? 1u : 0u, hb = ((all_b >> c) & M) != M ? 1u : 0u;  // padding: background
            const int bit = wave * 4 + (c & 1) * 2;  // cell (2 (c & 1), w) of tile c >> 1
            atomicOr(&cm_s[k * 4 + (c >> 1)], ((fa | (fb << 1)) << bit) | ((ha | (hb << 1)) << (16 + bit)));
        }
    }
    __syncthreads();
    // tile occupancy: bit 0 = some foreground, bit 1 = nothing but foreground; every byte is
    // written here, nothing for the host to clear
    if (threadIdx.x < ROWS * 4) {
        int ty = by * ROWS + (int)(threadIdx.x >> 2), txo = bx * 4 + (int)(threadIdx.x & 3);
        if (ty < tiles_y && txo < tiles_x) {
            const uint32_t cm = cm_s[threadIdx.x];
            const int64_t tile = (int64_t)slot * tiles_x * tiles_y + (int64_t)ty * tiles_x + txo;
            pj.occ[tile] = ((cm & 0xffffu) ? 1 : 0) | ((cm >> 16) ? 0 : 2);
            if (pj.cmask != nullptr) pj.cmask[tile] = cm;
        }
    }
}

template <int ROWS>
__global__ __launch_bounds__(kBlock) void pack16_kernel(PackJob pj) {
    pack16_block<ROWS>(pj, blockIdx.x);
}

// The same ingest in BAND form (SC_OPT_PACK_ROWS 0, the default for pictures up to kBandTiles tiles wide): a
// block takes one tile row of a view -- 32 picture rows, W bytes each, one contiguous run of memory when the rows
// are not padded -- as a list of 16-pixel tasks in row-major order, 256 at a time, so that a wavefront load reads
// 1 KB in one piece, and writes the band's tiles (contiguous in the packed arena) from LDS in 16-byte pieces.
// Measured on one MI355X, 72 masks resident in the Infinity Cache (tools/probes/pack_shape.py): the panel form
// takes 36 us on 1440 x 1080 pictures (a block's 16 KB lie in 128 pieces 1440 B apart, and 12 % of the panel
// blocks hang over the picture's edges) and 24 us on the same bytes as 128 x 12150 pictures, where a block's
// bytes are one run.  Tasks per row are rounded up to an even number: the two halves of a tile row word sit in
// neighbouring lanes.
constexpr int kBandTiles = 64;   // widest picture of the band form: 2048 pixels
constexpr int kBandPhase = 6;    // 16-byte loads in flight per lane (3, 4, 6, 12: the same within a microsecond)

__device__ __forceinline__ void pack_band_block(const PackJob &pj, uint32_t b) {
    __shared__ alignas(16) uint32_t band_s[kBandTiles * 32];  // the band's tile words, as they lie in the packed arena
    __shared__ uint32_t cmb_s[kBandTiles];        // per tile: cells with some foreground | with some background << 16
    const int W = pj.W, H = pj.H, tiles_x = pj.tiles_x, tiles_y = pj.tiles_y;
    const uint32_t flip = pj.flip;
    const uint32_t tid = threadIdx.x;
    const int ty = (int)(b % (uint32_t)tiles_y);
    int slot = (int)(b / (uint32_t)tiles_y);
    if (slot >= pj.nslots) return;  // block-uniform
    slot += pj.slot0;
    const int64_t view = pj.use_order ? (int64_t)pj.order[slot] : (int64_t)slot;
    const uint8_t *raw = pj.raw + view * pj.view_stride + (int64_t)ty * 32 * pj.row_stride;
    const int cpr = W >> 4;                    // 16-pixel chunks per row (W % 16 == 0)
    const uint32_t cprp = 2u * (uint32_t)tiles_x;  // ... rounded up to an even number
    const uint32_t ntasks = 32u * cprp;
    const int rows_here = min(32, H - ty * 32);
    if (tid < (uint32_t)tiles_x) cmb_s[tid] = 0u;
    // task q = tid + 256 i: row q / cprp, chunk q % cprp, stepped without a division
    uint32_t row = tid / cprp, c = tid - row * cprp;
    const uint32_t drow = (uint32_t)kBlock / cprp, dc = (uint32_t)kBlock - drow * cprp;
    bool synced = false;
    for (uint32_t base = 0; base < ntasks; base += (uint32_t)kBlock * kBandPhase) {
        uint4 q[kBandPhase];
        uint32_t rr[kBandPhase], cc[kBandPhase];
#pragma unroll
        for (int i = 0; i < kBandPhase; ++i) {
            rr[i] = row; cc[i] = c;
            q[i] = make_uint4(flip, flip, flip, flip);  // padding stays background after the flip
            if ((int)row < rows_here && (int)c < cpr)
                q[i] = *reinterpret_cast<const uint4 *>(raw + (int64_t)row * pj.row_stride + (int64_t)c * 16);
            row += drow; c += dc;
            if (c >= cprp) { c -= cprp; ++row; }
        }
        if (!synced) { __syncthreads(); synced = true; }  // cmb_s is zero for everybody (block-uniform branch)
#pragma unroll
        for (int i = 0; i < kBandPhase; ++i) {
            const uint32_t half = nonzero_nibble(q[i].x ^ flip) | (nonzero_nibble(q[i].y ^ flip) << 4) |
                                  (nonzero_nibble(q[i].z ^ flip) << 8) | (nonzero_nibble(q[i].w ^ flip) << 12);
            const uint32_t other = __shfl_xor(half, 1);  // the task next door: same row, the tile's other half
            if (rr[i] < 32u) {                           // (tasks past the band's end belong to nobody)
                if ((cc[i] & 1u) == 0u) band_s[(cc[i] >> 1) * 32u + rr[i]] = half | (other << 16);
                // the task's 16 pixels are two 8-pixel cells of cell row rr >> 3
                const uint32_t fa = (half & 0xffu) != 0u, fb = (half >> 8) != 0u;
                const uint32_t ha = (half & 0xffu) != 0xffu, hb = (half >> 8) != 0xffu;  // padding: background
                const uint32_t bit = (rr[i] >> 3) * 4u + (cc[i] & 1u) * 2u;
                atomicOr(&cmb_s[cc[i] >> 1], ((fa | (fb << 1)) << bit) | ((ha | (hb << 1)) << (16u + bit)));
            }
        }
    }
    __syncthreads();
    // the band's tiles: tiles_x * 32 words in a row in the packed arena
    uint4 *dst = reinterpret_cast<uint4 *>(pj.out + (int64_t)slot * pj.out_view_words + (int64_t)ty * tiles_x * 32);
    const uint4 *src = reinterpret_cast<const uint4 *>(band_s);
    for (uint32_t i = tid; i < (uint32_t)tiles_x * 8u; i += kBlock) dst[i] = src[i];
    if (tid < (uint32_t)tiles_x) {
        const uint32_t cm = cmb_s[tid];
        const int64_t tile = (int64_t)slot * tiles_x * tiles_y + (int64_t)ty * tiles_x + tid;
        pj.occ[tile] = ((cm & 0xffffu) ? 1 : 0) | ((cm >> 16) ? 0 : 2);
        if (pj.cmask != nullptr) pj.cmask[tile] = cm;
    }
}

__global__ __launch_bounds__(kBlock) void pack_band_kernel(PackJob pj) { pack_band_block(pj, blockIdx.x); }

// ---- brick form of the dense stage -------------------------------------------------------
// A block takes a BRICK of 16 columns (along y) x 64 voxels (along z) instead of 1024 consecutive
// voxels (bricks at the far y / z faces may stick out of the grid; any ny, nz with nz <= 4096): wavefront w owns columns 4w..4w+3, lane l the
// 4-voxel group (l & 15) of column (l >> 4).  A brick projects onto a small image patch, which
// makes a conservative emptiness test worthwhile (brick_flags_kernel, ahead of the dense
// kernel): project the brick's four corners, widen their bounding box by a rigorous bound on
// the float32 rounding of corners AND interior voxels, and if that box lies inside the image,
// in front of the camera, and only over 32x32 tiles that hold no foreground, then the
// reference would find every voxel of the brick in-image on a zero pixel
// (backprojection.c:26-31,79): the whole block carves its voxels without projecting them.
// Any doubt -> no culling.  Measured on the 512^3 plant scene: 72 % of the bricks are culled
// in the first view.
constexpr int kBrickY = 16, kBrickZ = 64;

// One lane per (view, brick).  A brick lies in one x-plane, so it is a planar rectangle: with
// every corner in front of the camera its image is the convex hull of the images of its four
// corners, and |R[..] * coordinate| terms are largest at a corner, so bounds taken over the four
// corners hold for every voxel of the brick.
// The image of a RECTANGLE of voxels of one x-plane (columns j0..j1, voxels k0..k1): a box in pixel
// coordinates that contains the pixel every voxel of the rectangle is projected to by the reference
// arithmetic, or the knowledge that no voxel of it is touched by the view at all.
struct PixelBox {
    float umin, umax, vmin, vmax;  // widened by the bound of DESIGN.md 4b
    bool inside;   // every voxel is in front of the camera and lands inside the picture, on a pixel of the box
    bool outside;  // every voxel is behind the camera or projects out of the picture: the view does nothing to
                   // it (backprojection.c:13,23-31)
};

__device__ __forceinline__ PixelBox rect_box(const ViewDesc &d, const GridDesc &g, float x, int j0, int j1, int k0, int k1) {
    PixelBox bx{0.0f, 0.0f, 0.0f, 0.0f, false, false};
    float ez = 0.0f, ex = 0.0f, ey = 0.0f, qxm = 0.0f, qym = 0.0f;
    float pzmin = INFINITY, pzmax = -INFINITY, umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    bool nan = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float y = g.oy + (float)((c >> 1) ? j1 : j0) * g.vs;  // backprojection.c:72
        float z = g.oz + (float)((c & 1) ? k1 : k0) * g.vs;   // :73
        float rzx = d.R[6] * x, rzy = d.R[7] * y, rzz = d.R[8] * z;
        float rxx = d.R[0] * x, rxy = d.R[1] * y, rxz = d.R[2] * z;
        float ryx = d.R[3] * x, ryy = d.R[4] * y, ryz = d.R[5] * z;
        float pz = ((rzx + rzy) + rzz) + d.t[2];
        float px = ((rxx + rxy) + rxz) + d.t[0];
        float py = ((ryx + ryy) + ryz) + d.t[1];
        // absolute rounding-error bounds of the three dot products (8x the worst case)
        ez = fmaxf(ez, (fabsf(rzx) + fabsf(rzy) + fabsf(rzz) + fabsf(d.t[2])) * 0x1p-19f);
        ex = fmaxf(ex, (fabsf(rxx) + fabsf(rxy) + fabsf(rxz) + fabsf(d.t[0])) * 0x1p-19f);
        ey = fmaxf(ey, (fabsf(ryx) + fabsf(ryy) + fabsf(ryz) + fabsf(d.t[1])) * 0x1p-19f);
        // an ESTIMATE of the voxel kernels' correctly rounded quotients is enough here (v_rcp_f32
        // is good to 1 ulp, the product adds half of one); the slack below pays for it
        float rz = __builtin_amdgcn_rcpf(pz);
        float qx = px * rz, qy = py * rz;
        float u = qx * d.K[0] + d.K[2], v = qy * d.K[1] + d.K[3];
        // fminf/fmaxf drop NaN operands: track them explicitly
        nan |= __builtin_isunordered(u, v) | __builtin_isunordered(pz, pz);
        pzmin = fminf(pzmin, pz);
        pzmax = fmaxf(pzmax, pz);
        qxm = fmaxf(qxm, fabsf(qx)); qym = fmaxf(qym, fabsf(qy));
        umin = fminf(umin, u); umax = fmaxf(umax, u);
        vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
    }
    if (nan) return bx;
    // depth is affine over the rectangle and its rounding error is below ez: with every corner below
    // -4 ez every voxel has p_z < 0 and is rejected (:13)
    if (pzmax < -4.0f * ez) { bx.outside = true; return bx; }
    bool front = pzmin > 4.0f * ez;  // depth is affine over the rectangle: all voxels in front
    // pixel-space slack: 2 px + propagated dot-product error + 8 ulp of the largest magnitude in
    // q * f + c (quotient estimate above, the voxel kernels' own division, product and sum
    // roundings -- relative to the OPERANDS, so that a principal point far outside the picture,
    // where q * f and c cancel, cannot make the bound too small)
    float inv = 2.0f / pzmin;
    float mu = 2.0f + fabsf(d.K[0]) * (ex + qxm * ez) * inv +
               (fabsf(d.K[0]) * qxm + fabsf(d.K[2]) + fmaxf(fabsf(umin), fabsf(umax))) * 0x1p-20f;
    float mv = 2.0f + fabsf(d.K[1]) * (ey + qym * ez) * inv +
               (fabsf(d.K[1]) * qym + fabsf(d.K[3]) + fmaxf(fabsf(vmin), fabsf(vmax))) * 0x1p-20f;
    umin -= mu; umax += mu; vmin -= mv; vmax += mv;
    // a NaN anywhere makes a comparison false -> no verdict
    bx.inside = front & (umin >= 0.0f) & (umax <= d.Wf - 1.0f) & (vmin >= 0.0f) & (vmax <= d.Hf - 1.0f);
    // the widened box holds every voxel's uf, vf: all of it at or left of -1, at or right of W, above or
    // below the picture means (int)uf is outside [0, W - 1] (or (int)vf outside [0, H - 1]) for all of them
    bx.outside = front & ((umax <= -1.0f) | (umin >= d.Wf) | (vmax <= -1.0f) | (vmin >= d.Hf));
    bx.umin = umin; bx.umax = umax; bx.vmin = vmin; bx.vmax = vmax;
    return bx;
}

struct Footprint {  // 32x32-pixel tiles the brick's image may touch; ok == false: no verdict from the tiles
    int tx0, tx1, ty0, ty1;
    bool ok;
    bool outside;  // see PixelBox
};

__device__ __forceinline__ Footprint brick_footprint(const ViewDesc &d, const GridDesc &g, float x, int j0, int k0) {
    Footprint fpr{0, 0, 0, 0, false, false};
    const PixelBox bx = rect_box(d, g, x, j0, j0 + kBrickY - 1, k0, k0 + kBrickZ - 1);
    fpr.outside = bx.outside;
    if (!bx.inside) return fpr;
    fpr.tx0 = (int)bx.umin >> 5; fpr.tx1 = (int)bx.umax >> 5; fpr.ty0 = (int)bx.vmin >> 5; fpr.ty1 = (int)bx.vmax >> 5;
    fpr.ok = (fpr.tx1 - fpr.tx0 + 1) * (fpr.ty1 - fpr.ty0 + 1) <= 64;
    return fpr;
}

// Verdict of a view about a rectangle of voxels at the CELL level (8x8 pixels, ViewDesc::cmask): every voxel
// of the rectangle lands in-image on a pixel of the box, so when no cell under the box holds foreground the
// view carves them all (EMPTY, 1: backprojection.c:79), when none holds background it keeps them all (FULL,
// 2: :81); 4 OUTSIDE (rect_box); else 0.  Up to 16 32x32 tiles are looked at (one word each).
constexpr int kCellShift = 3;  // 8x8-pixel cells
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {  // bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
template <int NX, int NY>  // the cells of the box among the tiles of an NX x NY window: foreground / background seen
__device__ __forceinline__ void window_cells(const ViewDesc &d, int cx0, int cx1, int cy0, int cy1, int tx0, int ty0,
                                             int nxw, int nyw, uint32_t &fg, uint32_t &bg) {
    // every word first (one lane asks for one view: its loads hit nothing another lane's do, and a loop
    // would wait for each in turn), then the masks
    uint32_t w[NY][NX];
#pragma unroll
    for (int a = 0; a < NY; ++a)
#pragma unroll
        for (int b = 0; b < NX; ++b) {
            w[a][b] = 0u;
            if (a < nyw && b < nxw) w[a][b] = d.cmask[(ty0 + a) * d.tiles_x + tx0 + b];
        }
    uint32_t cols[NX];
#pragma unroll
    for (int b = 0; b < NX; ++b) {  // columns of cells of tile column b inside the box
        const int ox = (tx0 + b) * 4;
        const int c0 = min(max(cx0 - ox, 0), 3), c1 = max(min(cx1 - ox, 3), c0);
        cols[b] = ((0xfu >> (3 - (c1 - c0))) << c0) * 0x1111u;
    }
#pragma unroll
    for (int a = 0; a < NY; ++a) {  // rows of cells of tile row a inside the box: bits 4 r0 .. 4 r1 + 3
        const int oy = (ty0 + a) * 4;
        const int r0 = min(max(cy0 - oy, 0), 3), r1 = max(min(cy1 - oy, 3), r0);
        const uint32_t rows = (0xffffu >> (12 - 4 * (r1 - r0))) << (4 * r0);
#pragma unroll
        for (int b = 0; b < NX; ++b) {  // (tiles beyond the window hold 0)
            const uint32_t m = rows & cols[b];
            fg |= w[a][b] & m;
            bg |= (w[a][b] >> 16) & m;
        }
    }
}

__device__ __forceinline__ uint32_t rect_verdict_cells(const ViewDesc &d, const GridDesc &g, float x, int j0, int j1,
                                                       int k0, int k1) {
    const PixelBox bx = rect_box(d, g, x, j0, j1, k0, k1);
    if (bx.outside) return 4u;
    if (!bx.inside) return 0u;
    const int cx0 = (int)bx.umin >> kCellShift, cx1 = (int)bx.umax >> kCellShift;
    const int cy0 = (int)bx.vmin >> kCellShift, cy1 = (int)bx.vmax >> kCellShift;
    const int tx0 = cx0 >> 2, ty0 = cy0 >> 2;
    const int nxw = (cx1 >> 2) - tx0 + 1, nyw = (cy1 >> 2) - ty0 + 1;  // the window of tiles under the box
    // a 3 x 3 window or one of three shapes of 16 tiles; the lanes of a wavefront ask about one rectangle of
    // voxels from cameras of one rig, so they mostly agree on the shape and one of the four runs
    const int shape = (nxw <= 3 && nyw <= 3) ? 4 : ((nxw <= 2 && nyw <= 8) ? 1 : ((nxw <= 4 && nyw <= 4) ? 2 : ((nxw <= 8 && nyw <= 2) ? 3 : 0)));
    if (shape == 0) return 0u;
    uint32_t fg = 0, bg = 0;
    if (shape == 4) window_cells<3, 3>(d, cx0, cx1, cy0, cy1, tx0, ty0, nxw, nyw, fg, bg);  // the usual one: a square unit
    if (shape == 1) window_cells<2, 8>(d, cx0, cx1, cy0, cy1, tx0, ty0, nxw, nyw, fg, bg);
    if (shape == 2) window_cells<4, 4>(d, cx0, cx1, cy0, cy1, tx0, ty0, nxw, nyw, fg, bg);
    if (shape == 3) window_cells<8, 2>(d, cx0, cx1, cy0, cy1, tx0, ty0, nxw, nyw, fg, bg);
    return fg == 0u ? 1u : (bg == 0u ? 2u : 0u);
}

__device__ __forceinline__ uint32_t brick_verdict(const ViewDesc &d, const GridDesc &g, float x, int j0,
                                                  int k0, int occ_tx) {
    const Footprint fpr = brick_footprint(d, g, x, j0, k0);
    if (fpr.outside) return 4u;  // OUTSIDE: the view does nothing to the brick
    if (!fpr.ok) return 0u;
    uint32_t any = 0, all = 3;
    for (int ty = fpr.ty0; ty <= fpr.ty1; ++ty)
        for (int tx = fpr.tx0; tx <= fpr.tx1; ++tx) {
            uint32_t o = d.occ[ty * occ_tx + tx];
            any |= o;
            all &= o;
        }
    // every voxel of the brick lands in-image on a pixel of these tiles: all of them background
    // (EMPTY: the view carves the whole brick) or all of them foreground (FULL: the view keeps it)
    return (any & 1u) == 0 ? 1u : ((all & 2u) != 0 ? 2u : 0u);
}

// float32 masks of the averaging kernel (tiled form, ViewDesc::pad == 2): behind the per-region flags
// (d.occ: 1 = every pixel of the 32x32 region holds the same float, bit for bit) come the regions'
// values.  A footprint over regions that all hold ONE value adds that value to every voxel of the brick
// (backprojection.c:54) without projecting any: returns 3 and the value's bits, else 0.
__device__ __forceinline__ uint32_t brick_flat_f32(const ViewDesc &d, const GridDesc &g, float x, int j0, int k0,
                                                   uint32_t &bits) {
    const Footprint fpr = brick_footprint(d, g, x, j0, k0);
    bits = 0u;
    if (fpr.outside) return 4u;  // the view adds nothing to the brick
    if (!fpr.ok) return 0u;
    const int otx = (d.W + 31) >> 5, oty = (d.H + 31) >> 5;
    const uint32_t *val = reinterpret_cast<const uint32_t *>(d.occ + (((size_t)otx * oty + 3) & ~(size_t)3));
    const uint32_t first = val[fpr.ty0 * otx + fpr.tx0];
    bool flat = true;
    for (int ty = fpr.ty0; ty <= fpr.ty1; ++ty)
        for (int tx = fpr.tx0; tx <= fpr.tx1; ++tx)
            flat &= d.occ[ty * otx + tx] != 0 && val[ty * otx + tx] == first;
    bits = first;
    return flat ? 3u : 0u;
}

// The emptiness verdict of every brick ahead of the dense kernel: flags[brick] = 1 when ANY of the
// first `nviews` views of the batch finds the brick empty (carve is order-independent: one
// in-image zero pixel in any view carves a voxel, backprojection.c:79, so the views tested here
// need not be the dense stage's).  The bricks no view found empty are appended to the LIVE list:
// the dense kernel walks that list (a few per cent of the bricks on a plant), the -1 fill of the
// others needs the flags only.
// (History: a first brick kernel had its wavefront 0 run the test on 32 column end points while
// the other three waited behind a barrier, 61 % of its wave cycles; a second one started one
// block per strip of bricks and most of those found nothing to do, ~3 us each, 8 rounds deep.)
// A block is 8 wavefronts over the same 64 bricks: wavefront w tests views w, w + 8, ... (a
// wave-uniform view, so its descriptor stays in scalar registers), the verdicts meet in LDS.
constexpr int kFlagWaves = 8;

// Its own descriptors may travel in the kernel arguments (`own`, when `views` is null); block 0
// then also copies the batch's descriptors from the host's page-locked staging buffer to the
// device array the later kernels read -- no separate host-to-device copy on the stream.
struct FlagViews { ViewDesc v[kFlagWaves]; };
struct DescCopy { const uint32_t *src; uint32_t *dst; uint32_t words; };

__global__ __launch_bounds__(64 * kFlagWaves) void brick_flags_kernel(
    GridDesc g, const ViewDesc *__restrict__ views, int nviews, uint32_t bricks_y, uint32_t bricks_z,
    uint32_t nbricks, uint8_t *__restrict__ flags, uint32_t *__restrict__ live, ListCtl *ctl,
    FlagViews own, DescCopy dc, const ViewDesc *__restrict__ allviews, int nall, int nbatch,
    uint8_t *__restrict__ dead, int dead_stale, uint32_t parity, uint32_t *__restrict__ fill_list) {
    __shared__ unsigned long long s_empty[kFlagWaves], s_full[kFlagWaves], s_seen[kFlagWaves];
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < dc.words; i += 64 * kFlagWaves) dc.dst[i] = dc.src[i];
        if (threadIdx.x == 0) ctl->nlive[parity ^ 1u] = ctl->nfill[parity ^ 1u] = 0u;  // the next launch's counters
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t lb = blockIdx.x * 64u + lane;
    // DEAD bricks: an earlier launch found the brick empty, every voxel is -1 and stays so whatever
    // is carved later (backprojection.c:67) -- until the next clear.  They get no verdict, no fill and
    // no place on the live list (flag 4): the reference's cadence of one launch per view touches a few
    // per cent of the volume after the first views instead of streaming all of it through.
    const bool inb = lb < nbricks;
    // (dead_stale: the labels went back to default_value since the flags were written -- nothing is dead,
    // and this launch rewrites every flag instead of a memset on the stream)
    const bool isdead = inb && dead != nullptr && !dead_stale && dead[lb] != 0;
    const bool valid = inb && !isdead;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
    const int j0 = (int)(by * kBrickY), k0 = (int)(bz * kBrickZ);
    // round 0: the first `nviews` views, one per wavefront (more: strided)
    // `full`: every view so far keeps the brick as it is -- sees all of it over foreground (verdict 2) or
    // does not see it at all (4, OUTSIDE); `seen`: at least one of them was a 2, so a label 0 becomes 1
    bool empty = false, full = true, seen = false;
    if (valid) {
        if (views == nullptr) {  // grid-uniform: nviews <= kFlagWaves, one view per wavefront
            if ((int)wave < nviews) {
                const uint32_t v = brick_verdict(own.v[wave], g, x, j0, k0, own.v[wave].tiles_x);
                empty = v == 1u;
                full = v == 2u || v == 4u;
                seen = v == 2u;
            }
        } else {
            for (int vi = (int)wave; vi < nviews; vi += kFlagWaves) {
                const ViewDesc d = views[vi];
                const uint32_t v = brick_verdict(d, g, x, j0, k0, d.tiles_x);
                empty |= v == 1u;
                full &= v == 2u || v == 4u;
                seen |= v == 2u;
            }
        }
    }
    unsigned long long any_empty = 0, cand = 0, any_seen = 0;
    {
        const unsigned long long me = __ballot(empty), mf = __ballot(full && valid), ms = __ballot(seen);
        if (lane == 0) { s_empty[wave] = me; s_full[wave] = mf; s_seen[wave] = ms; }
        __syncthreads();
        cand = ~0ull;
#pragma unroll
        for (int w = 0; w < kFlagWaves; ++w) { any_empty |= s_empty[w]; cand &= s_full[w]; any_seen |= s_seen[w]; }
        cand &= ~any_empty;
    }
    // FULL candidates (every view so far sees the whole brick over foreground) go through the
    // remaining views, 8 per round, until one view says otherwise: on a plant no brick gets past
    // round 0; inside a solid object this is what spares its voxels all their projections
    for (int base = nviews; base < nall && cand != 0; base += kFlagWaves) {  // block-uniform
        const int vi = base + (int)wave;
        bool e2 = false, f2 = true, s2 = false;
        if (vi < nall && ((cand >> lane) & 1ull)) {
            const ViewDesc d = allviews[vi];
            const uint32_t v = brick_verdict(d, g, x, j0, k0, d.tiles_x);
            e2 = v == 1u;
            f2 = v == 2u || v == 4u;
            s2 = v == 2u;
        }
        const unsigned long long me = __ballot(e2), mf = __ballot(f2), ms = __ballot(s2);
        __syncthreads();  // the previous round's masks have been read by everybody
        if (lane == 0) { s_empty[wave] = me; s_full[wave] = mf; s_seen[wave] = ms; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kFlagWaves; ++w) { any_empty |= s_empty[w]; cand &= s_full[w]; any_seen |= s_seen[w]; }
        cand &= ~any_empty;
    }
    if (nall <= 0) cand = 0;  // fullness not asked for
    if (wave != 0) return;
    const bool gone = (any_empty >> lane) & 1ull, kept = (cand >> lane) & 1ull, saw = (any_seen >> lane) & 1ull;
    // kept by every view of the batch: FULL (2: some view saw it, a 0 becomes 1) or UNTOUCHED (6: no view
    // sees any of it, the labels stay); by every view packed so far only: a candidate (3 seen / 7 unseen)
    if (inb) flags[lb] = isdead ? 4 : (gone ? 1 : (kept ? (nall >= nbatch ? (saw ? 2 : 6) : (saw ? 3 : 7)) : 0));
    if (nall < nbatch) {  // grid-uniform: later views are not packed yet, kept bricks are candidates
        const unsigned long long mc = __ballot(inb && !isdead && !gone && kept);
        if (mc != 0 && lane == 0 && ctl->cand.n == 0u) ctl->cand.n = 1u;
    }
    if (valid && dead != nullptr && (gone || dead_stale)) dead[lb] = gone ? 1 : 0;
    // the bricks left go on the live list, one atomic per block
    const bool alive = valid && !gone && !kept;
    const unsigned long long m = __ballot(alive);
    if (m != 0) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ctl->nlive[parity], (uint32_t)__popcll(m));
        base = __shfl(base, 0);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (alive) live[base + (uint32_t)__popcll(m & below)] = lb;
    }
    if (fill_list != nullptr) {  // launches whose dense kernel fills from a list (see carve_brick_light_kernel)
        const bool fillme = valid && (gone || kept);
        const unsigned long long mf = __ballot(fillme);
        if (mf != 0) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ctl->nfill[parity], (uint32_t)__popcll(mf));
            base = __shfl(base, 0);
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            // bit 31: kept whole (0 -> 1); bit 30: kept and unseen (nothing changes)
            if (fillme) fill_list[base + (uint32_t)__popcll(mf & below)] = lb | (gone ? 0u : (saw ? 0x80000000u : 0x40000000u));
        }
    }
}

// Two views applied to the four voxels of a lane: both projections first, then the eight gathers of a
// lane in one flight (the kernels that call this wait on memory, not on arithmetic), then
// backprojection.c:79-83 for the first view and, for what it left alive, for the second.
__device__ __forceinline__ void two_views(const ViewDesc &da, const ViewDesc &db, bool two, float x, float y,
                                          const float (&z)[4], int32_t (&lab)[4], uint32_t &alive) {
    const float aax = da.R[0] * x + da.R[1] * y, aay = da.R[3] * x + da.R[4] * y, aaz = da.R[6] * x + da.R[7] * y;
    const float bax = db.R[0] * x + db.R[1] * y, bay = db.R[3] * x + db.R[4] * y, baz = db.R[6] * x + db.R[7] * y;
    const uint32_t *bita = static_cast<const uint32_t *>(da.mask);
    const uint32_t *bitb = static_cast<const uint32_t *>(db.mask);
    bool oka[4], okb[4];
    uint32_t wa[4], wb[4];
    int sha[4], shb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int u, v;
        const bool live = (alive >> e) & 1u;
        oka[e] = project(aax, aay, aaz, z[e], da, u, v) & live;
        sha[e] = u & 31;
        wa[e] = load_mask_word(bita, oka[e] ? mask_word_index(u, v, da.tiles_x) : 0u);
        okb[e] = project(bax, bay, baz, z[e], db, u, v) & live & two;
        shb[e] = u & 31;
        wb[e] = load_mask_word(bitb, okb[e] ? mask_word_index(u, v, db.tiles_x) : 0u);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (oka[e]) {
            if (((wa[e] >> sha[e]) & 1u) == 0) {  // :79
                lab[e] = -1;
                alive &= ~(1u << e);
            } else if (lab[e] == 0) {  // :81
                lab[e] = 1;
            }
        }
        if (okb[e] && ((alive >> e) & 1u)) {  // a voxel the first view carved is skipped (:67)
            if (((wb[e] >> shb[e]) & 1u) == 0) {
                lab[e] = -1;
                alive &= ~(1u << e);
            } else if (lab[e] == 0) {
                lab[e] = 1;
            }
        }
    }
}

template <bool FRESH>
__device__ __forceinline__ void brick_voxels(int32_t *__restrict__ labels, const GridDesc &g,
                                             const ViewDesc *__restrict__ views, int nviews,
                                             int32_t init, Append ap, uint32_t il, uint32_t j,
                                             uint32_t k0, uint32_t lb, uint32_t lane, uint32_t unit = 0) {
    // bricks at the far y / z faces of the grid may stick out of it: lanes beyond ny or nz own
    // nothing (they still take part in the wave-wide ballots), a group at the end of a column
    // may be short, and when nz % 4 != 0 groups are not 16-byte aligned (element accesses)
    const bool inside = j < g.ny && k0 < g.nz;
    const int nvalid = inside ? (int)min(4u, g.nz - k0) : 0;
    const bool vec = (g.nzp & 3u) == 0;  // grid-uniform (the pitch is a multiple of 64: always)
    const uint64_t elem = ((uint64_t)il * g.ny + j) * g.nzp + k0;
    int32_t *p = labels + elem;
    int32_t lab[4], was[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) lab[e] = -1;  // what a lane does not own counts as carved
    if (FRESH) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) lab[e] = init;
    } else if (vec) {
        if (inside) {
            int4 q = *reinterpret_cast<const int4 *>(p);
            lab[0] = q.x; lab[1] = q.y; lab[2] = q.z; lab[3] = q.w;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e >= nvalid) lab[e] = -1;  // row padding behind the last voxel
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) lab[e] = p[e];
    }
    uint32_t alive = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        was[e] = lab[e];
        if (lab[e] != -1) alive |= 1u << e;  // :67
    }
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // :71, global plane index
    const float y = g.oy + (float)(int)j * g.vs;
    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = g.oz + (float)(int)(k0 + e) * g.vs;  // :73

    for (int vi = 0; vi < nviews; vi += 2) {
        if (__ballot(alive != 0) == 0) break;  // nothing left alive in this wavefront
        const bool two = vi + 1 < nviews;      // wave-uniform
        const ViewDesc da = views[vi];
        const ViewDesc db = views[two ? vi + 1 : vi];
        two_views(da, db, two, x, y, z, lab, alive);
    }

    if (vec) {
        bool changed = FRESH || lab[0] != was[0] || lab[1] != was[1] || lab[2] != was[2] || lab[3] != was[3];
        if (inside && changed) *reinterpret_cast<int4 *>(p) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid && (FRESH || lab[e] != was[e])) p[e] = lab[e];
    }

    if (ap.list != nullptr) {
        ap.sub = (lb * 0x9E3779B1u) >> 24;
        unsigned long long b[4];
        uint32_t total = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b[e] = __ballot((alive >> e) & 1u);
            total += (uint32_t)__popcll(b[e]);
        }
        bool bulked = false;
        if (ap.bulk != nullptr && total >= ap.bulk_min) {  // wave-uniform
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&ap.ctl->count[3][ap.sub].n, 1u);
            pos = __shfl(pos, 0);
            bulked = pos < ap.bulkcap;  // (a full sub-list: the voxels take the ordinary lists)
            if (bulked && lane == 0)
                ap.bulk[(size_t)ap.sub * ap.bulkcap + pos] = lb * 4u + unit;
        }
        if (total != 0 && !bulked) {  // wave-uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ap.ctl->count[0][ap.sub].n, total);
            base = __shfl(base, 0);
            if (base + total > ap.subcap) {
                if (lane == 0) ap.ctl->overflow = 1u;
            } else {
                uint32_t *dst = ap.list + (size_t)ap.sub * ap.subcap + base;
                unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                uint32_t off = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if ((alive >> e) & 1u) {
                        uint32_t rank = off + (uint32_t)__popcll(b[e] & below);
                        dst[rank] = (uint32_t)(elem + e) | (lab[e] == 0 ? 0x80000000u : 0u);
                    }
                    off += (uint32_t)__popcll(b[e]);
                }
            }
        }
    }
}

// The bricks of a strip the flags kernel has settled.  EMPTY (flag 1): live voxels become -1, dead
// ones are -1 already -- one 16-byte store per lane and brick, nothing else.  FULL (flag 2): every
// view keeps every voxel, so a label 0 becomes 1 and any other label stays (backprojection.c:81):
// `kept` is that value for a volume known to hold `init` everywhere (fresh), else the labels are
// read, patched and written back.
struct Fill {
    int32_t kept;   // label of a FULL brick's voxels when the volume is fresh: init == 0 ? 1 : init
    int32_t fresh;  // the volume holds `init` everywhere (nothing applied since clear)
    int32_t init;   // ... and this is what an UNTOUCHED brick (flag 6) of a fresh volume gets
};

__device__ __forceinline__ void store_culled_bricks(int32_t *__restrict__ labels, const GridDesc &g,
                                                    const uint8_t *__restrict__ flags, uint32_t strip,
                                                    uint32_t bricks_y, uint32_t bricks_z, Fill fill) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t il = strip / bricks_y, by = strip - il * bricks_y;
    const uint32_t j = by * kBrickY + wave * 4 + (lane >> 4);
    const uint32_t f = (lane < bricks_z) ? flags[strip * bricks_z + lane] : 0u;
    const unsigned long long culled = __ballot(f == 1u), full = __ballot(f == 2u);
    // UNTOUCHED bricks (6) keep their labels: only a fresh volume, whose labels exist as `init` in name
    // only, has something to write there
    const unsigned long long untouched = fill.fresh ? __ballot(f == 6u) : 0ull;
    if (j >= g.ny) return;  // a strip at the far y face may stick out of the grid
    int32_t *col = labels + ((uint64_t)il * g.ny + j) * g.nzp;
    const bool vec = (g.nzp & 3u) == 0;
    for (uint32_t bz = 0; bz < bricks_z; ++bz) {
        const bool isfull = (full >> bz) & 1ull, isunt = (untouched >> bz) & 1ull;
        if (!((culled >> bz) & 1ull) && !isfull && !isunt) continue;
        const uint32_t k0 = bz * kBrickZ + (lane & 15) * 4;
        if (k0 >= g.nz) continue;
        const uint32_t n = min(4u, g.nz - k0);
        if (!isfull || fill.fresh) {
            const int32_t val = isunt ? fill.init : (isfull ? fill.kept : -1);
            if (vec) {
                // streaming store: the fill is written once and not read again by this batch; kept
                // out of the caches it does not evict the masks the next batch packs
                typedef int v4i __attribute__((ext_vector_type(4)));
                v4i vv = {val, val, val, val};
                __builtin_nontemporal_store(vv, reinterpret_cast<v4i *>(col + k0));
            } else {
                for (uint32_t e = 0; e < n; ++e) col[k0 + e] = val;
            }
        } else {  // FULL brick of a stored volume: 0 -> 1, the rest as it is
            for (uint32_t e = 0; e < n; ++e)
                if (col[k0 + e] == 0) col[k0 + e] = 1;
        }
    }
}

// The UNITS of the bulk list (a wavefront's share of a live brick -- its 16 columns, voxels 16w .. 16w + 15 of
// each: a square patch of the plane -- with most of its voxels alive after the dense views) are asked about as a whole before anything projects
// their voxels: every remaining view at once, one view per lane, at the cell level (rect_verdict_cells).
//   some view sees the unit entirely over background (EMPTY): every voxel is carved, done;
//   views that see it entirely over foreground (FULL) make a label 0 a 1 (backprojection.c:81) here and now,
//   and like the views that do not see it at all (OUTSIDE) have nothing more to say;
//   the UNDECIDED views are the only ones that have to project its voxels: they become work items
//   (half a unit x up to 16 of those views, see UnitItems) for the final list stage -- or, when that would
//   be no cheaper than the ordinary survivor lists (few voxels alive, most views undecided), the unit's
//   voxels are appended to the first list like any other survivor.
struct UnitJob {
    const uint32_t *units;    // null: no bulk list.  [kSub][cap] unit ids (brick * 4 + wavefront), counts in ctl->count[3]
    uint32_t cap;
    uint4 *items;             // [kSub][icap] work items out, counts in ctl->count[4]
    uint32_t icap;
    const ViewDesc *views;    // every view of the batch
    int32_t nall, ndense;     // ... their number (<= 128), and how many of them the dense stage has applied
    uint32_t bricks_y, bricks_z;
    int32_t *labels;
    uint32_t *list;           // the first survivor list and the room of its sub-lists (counts in ctl->count[0])
    uint32_t subcap;
    uint32_t bias;            // items are chosen when their turns * 16 <= bias * the turns the lists would take
    uint32_t *stats;          // per unit block: {units that got their verdicts, turns those spared the survivor stages}
};

__device__ __forceinline__ void unit_verdicts(const UnitJob &uj, const GridDesc &g, ListCtl *ctl, uint32_t unit,
                                              uint32_t sub, uint32_t lane, uint32_t &saved) {
    const uint32_t lb = unit >> 2, w = unit & 3u;
    const uint32_t per_plane = uj.bricks_y * uj.bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / uj.bricks_z, bz = rem - by * uj.bricks_z;
    const int j0 = (int)(by * kBrickY), kb = (int)(bz * kBrickZ + w * 16u);  // 16 columns x 16 voxels
    const uint32_t j = (uint32_t)j0 + (lane >> 2), k0 = (uint32_t)kb + (lane & 3u) * 4u;
    const bool inside = j < g.ny && k0 < g.nz;
    const int nvalid = inside ? (int)min(4u, g.nz - k0) : 0;
    const uint32_t elem = (il * g.ny + j) * g.nzp + k0;
    int32_t *p = uj.labels + elem;  // the pitch is a multiple of 64: 16-byte groups
    int32_t lab[4] = {-1, -1, -1, -1};  // what a lane does not own counts as carved
    if (inside) {
        const int4 q = *reinterpret_cast<const int4 *>(p);
        lab[0] = q.x; lab[1] = q.y; lab[2] = q.z; lab[3] = q.w;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e >= nvalid) lab[e] = -1;  // row padding behind the last voxel
    }
    uint32_t alive = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (lab[e] != -1) alive |= 1u << e;  // :67
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // :71, global plane index
    unsigned long long need[2] = {0ull, 0ull};
    bool seen = false, empty = false;
    for (int h = 0; h < 2 && h * 64 < uj.nall; ++h) {
        const int vi = h * 64 + (int)lane;
        uint32_t v = 8u;  // no such view, or one the dense stage has applied
        if (vi < uj.nall && vi >= uj.ndense) {
            const ViewDesc d = uj.views[vi];  // one descriptor per lane
            v = d.cmask != nullptr ? rect_verdict_cells(d, g, x, j0, j0 + kBrickY - 1, kb, kb + 15) : 0u;
        }
        empty |= __ballot(v == 1u) != 0;
        seen |= __ballot(v == 2u) != 0;
        need[h] = __ballot(v == 0u);
    }
    unsigned long long b[4];
    uint32_t nalive = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        b[e] = __ballot((alive >> e) & 1u);
        nalive += (uint32_t)__popcll(b[e]);
    }
    // turns of (128 voxels x 2 views) the unit's voxels would take in the survivor lists
    const uint32_t list_cost = ((nalive + 127u) >> 7) * (((uint32_t)(uj.nall - uj.ndense) + 1u) >> 1);
    if (empty) {  // some view carves every voxel of the unit
        if (inside && alive != 0) *reinterpret_cast<int4 *>(p) = make_int4(-1, -1, -1, -1);
        saved += list_cost;
        return;
    }
    if (seen) {
        bool changed = false;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (lab[e] == 0) { lab[e] = 1; changed = true; }  // :81 by a view that keeps the whole unit
        if (inside && changed) *reinterpret_cast<int4 *>(p) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    }
    const uint32_t nneed = (uint32_t)__popcll(need[0]) + (uint32_t)__popcll(need[1]);
    const unsigned long long anyalive = __ballot(alive != 0);
    if (nneed == 0 || anyalive == 0) {  // wave-uniform: the labels are final
        saved += list_cost;
        return;
    }
    // the undecided views of each 64-view word in pieces of up to 16; one item per (half with something
    // alive, word, piece): lane = piece * 4 + word * 2 + half
    unsigned long long pm[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bool bit = (need[h] >> lane) & 1ull;
        const uint32_t piece = lanes_below(need[h]) >> 4;  // this lane's view is the (16 piece + ..)-th undecided one
#pragma unroll
        for (int q = 0; q < 4; ++q) pm[h][q] = __ballot(bit && piece == (uint32_t)q);
    }
    const uint32_t hq = lane & 1u, wq = (lane >> 1) & 1u, pq = lane >> 2;
    unsigned long long mymask = 0ull;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (wq == (uint32_t)h && pq == (uint32_t)q) mymask = pm[h][q];
    const uint32_t halves = ((uint32_t)(anyalive & 0xffffffffull) != 0u ? 1u : 0u) + ((uint32_t)(anyalive >> 32) != 0u ? 1u : 0u);
    const bool half_alive = ((anyalive >> (32u * hq)) & 0xffffffffull) != 0;
    const bool mine = lane < 16u && mymask != 0ull && half_alive;
    const unsigned long long im = __ballot(mine);
    const uint32_t nitems = (uint32_t)__popcll(im);
    // turns of (128 voxels x 2 views): the items' against what the unit's voxels would take in the lists
    const uint32_t item_cost = halves * ((nneed + 1u) / 2u) + nitems;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (item_cost * 16u <= uj.bias * list_cost) {
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&ctl->count[4][sub].n, nitems);
        pos = __shfl(pos, 0);
        if (pos + nitems <= uj.icap) {
            if (mine)
                uj.items[(size_t)sub * uj.icap + pos + (uint32_t)__popcll(im & below)] =
                    make_uint4(unit * 2u + hq, wq * 64u, (uint32_t)mymask, (uint32_t)(mymask >> 32));
            saved += list_cost - min(list_cost, item_cost);
            return;
        }
        // (no room: the count stays beyond the capacity, the reader clamps it; the voxels take the list)
    }
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ctl->count[0][sub].n, nalive);
    base = __shfl(base, 0);
    if (base + nalive > uj.subcap) {
        if (lane == 0) ctl->overflow = 1u;
        return;
    }
    uint32_t *dst = uj.list + (size_t)sub * uj.subcap + base;
    uint32_t off = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if ((alive >> e) & 1u) dst[off + (uint32_t)__popcll(b[e] & below)] = (elem + (uint32_t)e) | (lab[e] == 0 ? 0x80000000u : 0u);
        off += (uint32_t)__popcll(b[e]);
    }
}

// FULL candidates (flag 3: every view the flags kernel could see keeps the brick whole, but the masks
// of views [v0, v1) were packed only afterwards, beside the dense stage) put the question to those
// views: same organisation as the flags kernel's own FULL rounds (64 bricks per block, one view per
// wavefront and round, verdicts joined in LDS).  Kept by all: flag 2, filled like any FULL brick.
// Otherwise flag 5 and a place on the LATE list: the resume kernel carves such a brick over all the
// views of the batch, voxel by voxel.  A block without candidates leaves at once.
// The units of the bulk list get their verdicts, one wavefront per unit (unit_verdicts): a persistent grid of
// blocks of 8 wavefronts, launched behind the confirm kernel (the masks of every view are packed by then) and
// ahead of the list stages.  (A kernel of its own: inside the confirm kernel its registers cost that kernel's
// blocks three wavefronts per SIMD, 40 us on a batch of all-foreground masks.)
__global__ __launch_bounds__(64 * kFlagWaves) void unit_verdict_kernel(GridDesc g, ListCtl *ctl, UnitJob uj) {
    const uint32_t nunitblocks = gridDim.x;
    __shared__ uint32_t upref[kSub + 1];
    const uint32_t tid = threadIdx.x;
    {
        if (tid < kSub) upref[tid + 1] = min(ctl->count[3][tid].n, uj.cap);
        if (tid == 0) upref[0] = 0;
        __syncthreads();
        for (uint32_t off = 1; off < kSub; off <<= 1) {
            uint32_t val = 0, add = 0;
            if (tid < kSub) {
                val = upref[tid + 1];
                add = (tid >= off) ? upref[tid + 1 - off] : 0u;
            }
            __syncthreads();
            if (tid < kSub) upref[tid + 1] = val + add;
            __syncthreads();
        }
    }
    const uint32_t total = upref[kSub];
    const uint32_t uwave = __builtin_amdgcn_readfirstlane(tid >> 6), ulane = tid & 63u;
    const uint32_t nworkers = nunitblocks * kFlagWaves;
    uint32_t nunits = 0, saved = 0;
    for (uint32_t i = blockIdx.x * kFlagWaves + uwave; i < total; i += nworkers) {
        uint32_t lo = 0, hi = kSub;  // largest s with upref[s] <= i (wave-uniform)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (upref[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t unit = __builtin_amdgcn_readfirstlane(uj.units[(size_t)lo * uj.cap + (i - upref[lo])]);
        unit_verdicts(uj, g, ctl, unit, lo, ulane, saved);
        ++nunits;
    }
    // what the host's on / off decision reads (see flush): one pair per block, summed by a list kernel (ReportJob)
    // (atomics on one address from every wavefront of the grid would take longer than the verdicts)
    __shared__ uint32_t s_stat[2];
    if (tid < 2) s_stat[tid] = 0u;
    __syncthreads();
    if (ulane == 0 && nunits != 0) {
        atomicAdd(&s_stat[0], nunits);
        atomicAdd(&s_stat[1], saved);
    }
    __syncthreads();
    if (tid < 2) uj.stats[blockIdx.x * 2u + tid] = s_stat[tid];
}

__global__ __launch_bounds__(64 * kFlagWaves) void brick_confirm_kernel(
    GridDesc g, const ViewDesc *__restrict__ views, int v0, int v1, uint32_t bricks_y, uint32_t bricks_z,
    uint32_t nbricks, uint8_t *__restrict__ flags, uint32_t *__restrict__ late, ListCtl *ctl) {
    if (v0 >= v1 || ctl->cand.n == 0) return;  // no view was packed late, or the flags kernel left no candidate open
    __shared__ unsigned long long s_full[kFlagWaves], s_seen[kFlagWaves];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t per_plane = bricks_y * bricks_z;
    // a persistent grid over the groups of 64 bricks
    for (uint32_t grp = blockIdx.x; grp * 64u < nbricks; grp += gridDim.x) {
        const uint32_t lb = grp * 64u + lane;
        const uint32_t fl = lb < nbricks ? flags[lb] : 0u;
        const bool isc = fl == 3u || fl == 7u;  // candidates: some view so far saw the brick whole / none sees it
        unsigned long long any_seen = __ballot(fl == 3u);
        unsigned long long cand = __ballot(isc);
        if (cand == 0) continue;  // block-uniform: every wavefront read the same 64 flags
        const uint32_t il = lb / per_plane;
        const uint32_t rem = lb - il * per_plane;
        const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
        const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
        for (int base = v0; base < v1 && cand != 0; base += kFlagWaves) {  // block-uniform
            const int vi = base + (int)wave;
            bool keeps = true, sees = false;
            if (vi < v1 && ((cand >> lane) & 1ull)) {
                const ViewDesc d = views[vi];
                const uint32_t v = brick_verdict(d, g, x, (int)(by * kBrickY), (int)(bz * kBrickZ), d.tiles_x);
                keeps = v == 2u || v == 4u;
                sees = v == 2u;
            }
            const unsigned long long mf = __ballot(keeps), ms = __ballot(sees);
            __syncthreads();  // the previous round's masks have been read by everybody
            if (lane == 0) { s_full[wave] = mf; s_seen[wave] = ms; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kFlagWaves; ++w) { cand &= s_full[w]; any_seen |= s_seen[w]; }
        }
        if (wave != 0) continue;
        if (isc) flags[lb] = ((cand >> lane) & 1ull) ? (((any_seen >> lane) & 1ull) ? 2 : 6) : 5;
        const bool failed = isc && !((cand >> lane) & 1ull);
        const unsigned long long m = __ballot(failed);
        if (m != 0) {
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&ctl->nlate, (uint32_t)__popcll(m));
            pos = __shfl(pos, 0);
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            if (failed) late[pos + (uint32_t)__popcll(m & below)] = lb;
        }
    }
}

// The dense kernel proper: a persistent grid walks the live list, one brick per block and turn
// (wavefront w owns columns 4w..4w+3 of the brick); runs of kXcdRun consecutive entries
// (neighbouring bricks, which project onto the same mask lines) stay on one XCD.  Blocks behind
// the walkers, one per strip, fill the bricks found empty of strips [0, nstore) (the final list
// stage fills the others, see carve_list_kernel).
#ifdef SC_TRACE_DENSE  // diagnostic builds only (tools/probes/dense_trace.py): what every walker wavefront did, and when
__device__ uint32_t g_dense_trace[8192 * 8];
#endif
template <bool FRESH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void carve_brick_kernel(int32_t *__restrict__ labels, GridDesc g,
                                                             const ViewDesc *__restrict__ views,
                                                             int nviews, int32_t init, Append ap,
                                                             uint32_t bricks_y, uint32_t bricks_z,
                                                             const uint8_t *__restrict__ flags,
                                                             const uint32_t *__restrict__ live,
                                                             ListCtl *ctl, uint32_t nwalkers,
                                                             uint32_t nstore, PackJob ride, int pack_rows,
                                                             uint32_t parity, int nverd_arg, uint32_t verd_max_live) {
    if (blockIdx.x >= nwalkers + nstore) {
        // riders: the masks of the views the later stages apply are packed here, beside the walkers
        // (this stage waits on gathers and arithmetic, the packing on HBM reads).  One short block per
        // panel: persistent riders measured the same or slower.
        const uint32_t b = blockIdx.x - nwalkers - nstore;
        if (pack_rows == 0) pack_band_block(ride, b);
        else if (pack_rows == 1) pack16_block<1>(ride, b);
        else if (pack_rows == 2) pack16_block<2>(ride, b);
        else if (pack_rows == 8) pack16_block<8>(ride, b);
        else pack16_block<4>(ride, b);
        return;
    }
    if (blockIdx.x >= nwalkers) {
        store_culled_bricks(labels, g, flags, blockIdx.x - nwalkers, bricks_y, bricks_z,
                            Fill{init == 0 ? 1 : init, FRESH ? 1 : 0, init});
        return;
    }
    // Walkers are WAVEFRONTS: each takes the next live brick of its XCD's runs (runs of kXcdRun consecutive entries --
    // neighbouring bricks, which project onto the same mask lines -- stay on one XCD; a ticket counter per XCD), asks
    // the views packed ahead about the brick's four UNITS (16 columns x 16 voxels) at the cell level, one (unit, view)
    // pair per lane, carves the units some view finds empty without projecting a voxel -- two thirds of a plant's:
    // the brick is live because a 32x32 tile under it touches the plant, the unit lies beside it -- and projects the
    // others.  (A block of four wavefronts per brick, one unit each, left three in four idle once units are culled;
    // tickets keep every wavefront busy whatever the bricks hold.)
    const uint32_t nlive = ctl->nlive[parity];
    // masks whose tiles settled less than half of the bricks (noise: none) have no structure for the cells to find
    const int nverd = nlive <= verd_max_live ? nverd_arg : 0;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t xcd = blockIdx.x & 7u;
    // (the first ticket of a wavefront is its own number among the XCD's: a thousand atomics on one address at the
    // kernel's start would take longer than the first bricks)
    const uint32_t per_xcd = (nwalkers >> 3) * (kBlock / 64);
    bool first = true;
    uint32_t misses = 0, turn = 0;
#ifdef SC_TRACE_DENSE
    const uint64_t tr0 = wall_clock64();
    uint32_t tr_bricks = 0, tr_units = 0, tr_verd = 0, tr_unit = 0, tr_tick = 0;
#endif
    for (;;) {
#ifdef SC_TRACE_DENSE
        const uint64_t tra = wall_clock64();
#endif
        uint32_t t = (blockIdx.x >> 3) * (kBlock / 64) + (threadIdx.x >> 6);
        if (!first) {
            // eight counters per XCD, each dealing every eighth run of the XCD's entries to the wavefronts whose
            // number ends in c: returning atomics on one address take 11 ns each, and with one counter per XCD the
            // 1 500 tickets of a plant's batch were 16 us of them in a row
            const uint32_t c = t & 7u;
            uint32_t n = 0;
            if (lane == 0) n = atomicAdd(&ctl->xcd_next[xcd * 8u + c].n, 1u);
            n = __builtin_amdgcn_readfirstlane(n);
            t = per_xcd + ((n / kXcdRun) * 8u + c) * kXcdRun + (n % kXcdRun);
        }
        first = false;
        t = __builtin_amdgcn_readfirstlane(t);
        const uint32_t entry = ((t / kXcdRun) * 8u + xcd) * kXcdRun + (t % kXcdRun);
        if ((t / kXcdRun) * 8u * kXcdRun >= nlive) break;  // past the last run for every XCD
        if (entry >= nlive) continue;
        const uint32_t lb = __builtin_amdgcn_readfirstlane(live[entry]);
        const uint32_t il = lb / per_plane;
        const uint32_t rem = lb - il * per_plane;
        const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
        uint32_t culled = 0;
#ifdef SC_TRACE_DENSE
        const uint64_t trb = wall_clock64();
        tr_tick += (uint32_t)(trb - tra);
        ++tr_bricks;
#endif
        // (a wavefront whose last 8 bricks had no unit to cull -- masks without structure -- asks only about every
        // eighth brick from then on: the verdicts cost a tenth of the projections they cannot spare there)
        const bool ask = nverd > 0 && (misses < 8u || (turn & 7u) == 0u);
        ++turn;
        if (ask) {  // wave-uniform
            const uint32_t u = lane >> 4, vq = lane & 15u;
            uint32_t v = 0u;
            if ((int)vq < nverd) {
                const ViewDesc d = views[vq];  // one descriptor per lane
                const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
                v = rect_verdict_cells(d, g, x, (int)(by * kBrickY), (int)(by * kBrickY) + kBrickY - 1,
                                       (int)(bz * kBrickZ + u * 16u), (int)(bz * kBrickZ + u * 16u) + 15);
            }
            const unsigned long long e = __ballot(v == 1u);  // some view carves the whole unit
            culled = ((e & 0xffffull) ? 1u : 0u) | (((e >> 16) & 0xffffull) ? 2u : 0u) |
                     (((e >> 32) & 0xffffull) ? 4u : 0u) | ((e >> 48) ? 8u : 0u);
            misses = culled ? 0u : misses + 1u;
        }
#ifdef SC_TRACE_DENSE
        const uint64_t trc = wall_clock64();
        tr_verd += (uint32_t)(trc - trb);
#endif
        // lane = column * 4 + group of 4 voxels: a square patch of the plane, the UNIT the bulk list speaks of (see Append)
        const uint32_t j = by * kBrickY + (lane >> 2);
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t k0 = bz * kBrickZ + u * 16u + (lane & 3u) * 4u;
            if ((culled >> u) & 1u) {
                if (j < g.ny && k0 < g.nz)
                    *reinterpret_cast<int4 *>(labels + ((uint64_t)il * g.ny + j) * g.nzp + k0) = make_int4(-1, -1, -1, -1);
                continue;
            }
            brick_voxels<FRESH>(labels, g, views, nviews, init, ap, il, j, k0, lb, lane, u);
#ifdef SC_TRACE_DENSE
            ++tr_units;
#endif
        }
#ifdef SC_TRACE_DENSE
        tr_unit += (uint32_t)(wall_clock64() - trc);
#endif
    }
#ifdef SC_TRACE_DENSE
    if (lane == 0) {
        const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
        if (w < 8192u) {
            uint32_t *o = g_dense_trace + w * 8u;
            o[0] = (uint32_t)tr0; o[1] = (uint32_t)wall_clock64(); o[2] = tr_bricks; o[3] = tr_units;
            o[4] = tr_verd; o[5] = tr_unit; o[6] = tr_tick; o[7] = 0;
        }
    }
#endif
}
#ifdef SC_TRACE_DENSE
}  // namespace
extern "C" int sc_debug_dense_trace(uint32_t *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dense_trace), sizeof(uint32_t) * 8192 * 8);
}
namespace {
#endif

// The dense kernel of a launch WITHOUT survivor stages (fewer than 6 views; a single view in the
// reference's cadence, cl.py:223-226): walkers on the live list as above, and persistent FILLERS on
// the fill list the flags kernel wrote (settled bricks that are not dead yet) instead of one store
// block per strip of the grid -- after the first views nearly every brick is dead and a launch costs
// what its few live and newly settled bricks cost, not a pass over the grid.
template <bool FRESH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void carve_brick_light_kernel(
    int32_t *__restrict__ labels, GridDesc g, const ViewDesc *__restrict__ views, int nviews, int32_t init,
    uint32_t bricks_y, uint32_t bricks_z, const uint32_t *__restrict__ live, const uint32_t *__restrict__ fill_list,
    const ListCtl *ctl, uint32_t nwalkers, uint32_t parity) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t per_plane = bricks_y * bricks_z;
    if (blockIdx.x >= nwalkers) {
        const uint32_t nfill = ctl->nfill[parity], nfillers = gridDim.x - nwalkers;
        const bool vec = (g.nzp & 3u) == 0;
        const int32_t kept = init == 0 ? 1 : init;
        // 64 entries per load (one per lane), handed out with v_readlane: one round trip per 64 bricks
        for (uint32_t base = (blockIdx.x - nwalkers) * 64u; base < nfill; base += nfillers * 64u) {
            const uint32_t mine = (base + lane < nfill) ? fill_list[base + lane] : 0u;
            const uint32_t n = min(64u, nfill - base);
            for (uint32_t q = 0; q < n; ++q) {
                const uint32_t ent = __builtin_amdgcn_readlane(mine, q);
                const bool isfull = (ent >> 31) != 0, isunt = ((ent >> 30) & 1u) != 0;
                if (isunt && !FRESH) continue;  // kept and unseen: the labels stay
                const uint32_t lb = ent & 0x3fffffffu;
                const uint32_t il = lb / per_plane;
                const uint32_t rem = lb - il * per_plane;
                const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
                const uint32_t j = by * kBrickY + wave * 4 + (lane >> 4), k0 = bz * kBrickZ + (lane & 15) * 4;
                if (j >= g.ny || k0 >= g.nz) continue;
                int32_t *p = labels + ((uint64_t)il * g.ny + j) * g.nzp + k0;
                const uint32_t nv4 = min(4u, g.nz - k0);
                if (!isfull || FRESH) {
                    const int32_t val = isunt ? init : (isfull ? kept : -1);
                    if (vec) {
                        typedef int v4i __attribute__((ext_vector_type(4)));
                        v4i vv = {val, val, val, val};
                        __builtin_nontemporal_store(vv, reinterpret_cast<v4i *>(p));
                    } else {
                        for (uint32_t e = 0; e < nv4; ++e) p[e] = val;
                    }
                } else {  // kept whole: 0 -> 1, the rest as it is (backprojection.c:81)
                    for (uint32_t e = 0; e < nv4; ++e)
                        if (p[e] == 0) p[e] = 1;
                }
            }
        }
        return;
    }
    const uint32_t nlive = ctl->nlive[parity];
    const Append none{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
    const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3, per_xcd = nwalkers >> 3;
    for (uint32_t t = seq; ; t += per_xcd) {
        const uint32_t entry = ((t / kXcdRun) * 8u + xcd) * kXcdRun + (t % kXcdRun);
        if ((t / kXcdRun) * 8u * kXcdRun >= nlive) break;  // past the last run for every XCD
        if (entry >= nlive) continue;
        const uint32_t lb = live[entry];
        const uint32_t il = lb / per_plane;
        const uint32_t rem = lb - il * per_plane;
        const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
        const uint32_t j = by * kBrickY + wave * 4 + (lane >> 4);
        brick_voxels<FRESH>(labels, g, views, nviews, init, none, il, j, bz * kBrickZ + (lane & 15) * 4, lb, lane);
    }
}

// One view per launch (the reference's schedule, cl.py:223-226): the descriptor travels in
// the kernel arguments (no copy, no host-side wait), and each lane walks kStreamGroups
// 16-byte groups with the next group's state load already in flight -- after the first view
// nearly every wavefront only streams its state through and leaves.
template <bool FRESH, bool VEC>
__global__ __launch_bounds__(kBlock) void carve_kernel_1(int32_t *__restrict__ labels, GridDesc g,
                                                         ViewDesc view, int32_t init) {
    constexpr int G = (!FRESH && VEC) ? kStreamGroups : 1;
    uint32_t lb = spread_block(blockIdx.x, gridDim.x);
    uint64_t grp = (uint64_t)lb * (kBlock * G) + threadIdx.x;
    Append none{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
    int4 cur = make_int4(-1, -1, -1, -1);
    // streaming loads: the state (512 MiB) is far bigger than the Infinity Cache, every view
    // reads all of it once
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto stream_load = [&](uint64_t gidx) {
        v4i q = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(labels + gidx * 4));
        return make_int4(q.x, q.y, q.z, q.w);
    };
    if (!FRESH && VEC && grp < g.ngroups) cur = stream_load(grp);  // carve_group takes the stored labels from `cur`
#pragma unroll 1
    for (int s = 0; s < G; ++s, grp += kBlock) {
        int4 nxt = make_int4(-1, -1, -1, -1);
        if (G > 1 && s + 1 < G && grp + kBlock < g.ngroups)
            nxt = stream_load(grp + kBlock);
        if (grp < g.ngroups) carve_group<FRESH, VEC>(labels, g, &view, 1, init, grp, cur, none);
        cur = nxt;
    }
}

// Work items of the bulk units (see unit_verdicts): (half a unit = 8 columns x 16 voxels, up to 16 of the
// views that have to project its voxels, as a mask over 64 consecutive views).  The final list stage's
// wavefronts take them after their own spans; items of one unit may run side by side, which is exact for the
// same reason as the spans of one chunk: -1 is a plain store, 0 -> 1 a compare-and-swap on 0.
// What the bulk units' verdicts of this batch were worth, for the host's on / off decision (see flush): the
// sums over the unit blocks' pairs (UnitJob::stats), written as ONE 8-byte word to page-locked memory by
// block 0 of the first list kernel behind the verdicts, when it is through with its own work.
struct ReportJob {
    unsigned long long *report;  // null: nothing to report.  seq << 48 | min(units, 2^24 - 1) << 24 | min(turns spared / 16, 2^24 - 1)
    const uint32_t *stats;
    uint32_t nstats, seq;
};

struct UnitItems {
    const uint4 *items;       // null: none.  .x = unit * 2 + half, .y = first view of the mask, .z / .w = the mask
    uint32_t cap;             // items per sub-list (counts in ctl->count[4])
    const ViewDesc *views;    // every view of the batch (the items' view numbers index this)
    uint32_t bricks_y, bricks_z;
};

// Fused carve, sparse phase: one lane per SURVIVOR.  Reads the survivor sub-lists a previous
// stage appended and applies views with every lane busy, two views per iteration (two
// independent projection chains and two gathers in flight per lane).  A persistent grid of
// wavefronts walks the work items; the counts live in device memory, so the host never waits
// to learn how many survivors there are.
//   FINAL == false: an item is a 64-entry chunk and ALL `nviews` views; labels that change
//     are written (carved -> -1 at once, 0 -> 1 at the end) and what is still alive is appended
//     to `lout` for the next stage.
//   FINAL == true : an item is a 64-entry chunk times a GROUP of `vgsize` views, so that there
//     are many more items than wavefronts (no tail); groups of one chunk may run concurrently
//     on different wavefronts, which is exact because a carve is a plain store of -1 (final,
//     idempotent) and a 0 -> 1 promotion is a compare-and-swap on 0 (it can never undo a -1).
// (at most 80 SGPRs: with 82-96 the CU admits 7 such blocks instead of 8, with 98+ only 6 --
// MI355X_MICROARCH.md, "Residency" -- and the store blocks need the slots the list blocks leave)
template <bool FINAL, int P>  // P voxels per lane (an item is a chunk of 64 * P entries)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void carve_list_kernel(int32_t *__restrict__ labels, GridDesc g,
                                                            const ViewDesc *__restrict__ views,
                                                            int nviews,
                                                            const uint32_t *__restrict__ lin,
                                                            uint32_t *__restrict__ lout,
                                                            ListCtl *ctl, int sin, int sout,
                                                            uint32_t subcap, int vgsize, CullStores cs, UnitItems ui,
                                                            ReportJob rj) {
    __shared__ uint32_t pref[kSub + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t bx = blockIdx.x, gdim = gridDim.x;
    // A final stage with deferred stores has cs.nstrips STORE blocks behind its persistent list
    // blocks: this stage is bound by projection arithmetic and the -1 fill of the bricks the flags
    // kernel found empty by HBM writes, so the two run side by side instead of one after the
    // other.  The list blocks leave wavefront slots free; short store blocks stream through them.
    const bool split = cs.flags != nullptr;
    const uint32_t nstore = split ? (cs.fill_blocks ? cs.fill_blocks : cs.nstrips - cs.first) : 0u;
    const uint32_t nbid = gdim - nstore;
    if (split && bx >= nbid) {
        // one short block per strip, or (fill_blocks > 0) that many blocks walking the strips: a
        // wavefront's stores do not hold it up, so few of them keep the write path busy and the
        // wavefront slots go to the list blocks
        for (uint32_t strip = cs.first + (bx - nbid); strip < cs.nstrips; strip += nstore)
            store_culled_bricks(labels, g, cs.flags, strip, cs.bricks_y, cs.bricks_z, Fill{cs.kept, cs.fresh, cs.init});
        return;
    }
    if (ctl->overflow) return;  // the dense resume kernel does the remaining views instead
    const uint32_t bid = bx;
    constexpr uint32_t CH = 64u * P;
    {
        uint32_t c = (min(ctl->count[sin][tid].n, subcap) + CH - 1u) / CH;  // kSub == kBlock
        if (tid == 0) pref[0] = 0;
        pref[tid + 1] = c;
        __syncthreads();
        for (uint32_t off = 1; off < kSub; off <<= 1) {
            uint32_t val = pref[tid + 1];
            uint32_t add = (tid >= off) ? pref[tid + 1 - off] : 0u;
            __syncthreads();
            pref[tid + 1] = val + add;
            __syncthreads();
        }
    }
    const uint32_t chunks = pref[kSub];
    const uint32_t lane = tid & 63u;
    __shared__ uint32_t ipref[FINAL ? kSub + 1 : 1];
    const bool with_items = FINAL && ui.items != nullptr;  // grid-uniform
    if (with_items) {
        const uint32_t c = min(ctl->count[4][tid].n, ui.cap);
        if (tid == 0) ipref[0] = 0;
        ipref[tid + 1] = c;
        __syncthreads();
        for (uint32_t off = 1; off < kSub; off <<= 1) {
            const uint32_t val = ipref[tid + 1];
            const uint32_t add = (tid >= off) ? ipref[tid + 1 - off] : 0u;
            __syncthreads();
            ipref[tid + 1] = val + add;
            __syncthreads();
        }
    }
    const uint64_t nworkers = (uint64_t)nbid * (kBlock / 64);
    // the wavefront index must be a scalar for the compiler, or everything derived from the
    // item (view range, descriptors) is treated as divergent and fetched with vector loads
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // FINAL: the (chunk, view) pairs, chunk-major, are cut into one SPAN per wavefront -- every
    // wavefront gets the same number of projections whatever the counts are (with whole
    // (chunk, view group) items 17 k items over 4096 wavefronts meant 5 items for some and 4 for
    // others), and a span crosses a chunk boundary once or twice, so the decode of the entries is
    // paid once or twice per wavefront.  `vgsize` only rounds the span length.
    // Not FINAL: an item is a chunk and all the views, dealt round-robin.
    const uint64_t total = FINAL ? (uint64_t)chunks * (uint32_t)nviews : (uint64_t)chunks;
    uint64_t per = 1;
    if (FINAL) {
        per = (total + nworkers - 1) / nworkers;
        const uint64_t r = (uint64_t)max(vgsize, 1);
        per = (per + r - 1) / r * r;
    }
    const uint64_t wid = (uint64_t)bid * (kBlock / 64) + wave;
    uint64_t pos = FINAL ? min(total, wid * per) : wid;
    const uint64_t end = FINAL ? min(total, pos + per) : total;
    while (pos < end) {
        uint32_t c;
        int v0, v1;
        if (FINAL) {
            c = (uint32_t)(pos / (uint32_t)nviews);
            v0 = (int)(pos - (uint64_t)c * (uint32_t)nviews);
            v1 = (int)min((uint64_t)nviews, (uint64_t)v0 + (end - pos));
            pos += (uint64_t)(v1 - v0);
        } else {
            c = (uint32_t)pos;
            v0 = 0;
            v1 = nviews;
            pos += nworkers;
        }
        uint32_t lo = 0, hi = kSub;  // largest s with pref[s] <= c (wave-uniform)
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (pref[mid] <= c) lo = mid; else hi = mid;
        }
        const uint32_t s = lo;
        const uint32_t cnt = min(ctl->count[sin][s].n, subcap);
        // P voxels per lane: the descriptor traffic and the scalar bookkeeping of a view are shared,
        // and a lane has P * U independent projection chains and gathers in flight
        uint32_t idx[P];
        bool zero[P], flipped[P], alive[P];
        float x[P], y[P], z[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t e = (c - pref[s]) * CH + (uint32_t)p * 64u + lane;
            alive[p] = e < cnt;
            uint32_t entry = 0;
            if (alive[p]) entry = lin[(size_t)s * subcap + e];
            idx[p] = entry & 0x7fffffffu;
            zero[p] = (entry >> 31) != 0;  // label is still 0
            flipped[p] = false;
            const uint32_t col = idx[p] / g.nzp;  // entries index the padded rows
            const uint32_t k = idx[p] - col * g.nzp;
            const uint32_t il = col / g.ny;
            const uint32_t j = col - il * g.ny;
            x[p] = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // backprojection.c:71-73
            y[p] = g.oy + (float)(int)j * g.vs;
            z[p] = g.oz + (float)(int)k * g.vs;
        }
        // U views per iteration.  The final stage is bound by arithmetic (U = 2); the stages before
        // it wait on memory and most of their voxels die within a few views (U = 4).
        constexpr int U = P >= 4 ? 1 : (FINAL ? 2 : 4);
        for (int vi = v0; vi < v1; vi += U) {
            bool any = false;
#pragma unroll
            for (int p = 0; p < P; ++p) any |= alive[p];
            if (__ballot(any) == 0) break;
            bool ok[U][P], fg[U][P];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                // past the end of the range the last view is applied once more: a view applied twice
                // changes nothing (a carve is final, a kept 0 is already 1), and nothing per lane has
                // to know whether the slot was real
                const ViewDesc d = views[vi + q < v1 ? vi + q : vi];
                // every field in scalar registers NOW: left alone the compiler fetches Wf/Hf,
                // tiles_x and the mask pointer one by one where they are first used, three
                // more scalar-load round trips inside each projection
                asm volatile("" ::"s"(d.Wf), "s"(d.Hf), "s"(d.tiles_x), "s"(d.mask));
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    int uu, vv;
                    // dead lanes project along (their carve below is masked): cheaper than a per-lane test here
                    ok[q][p] = project(d.R[0] * x[p] + d.R[1] * y[p], d.R[3] * x[p] + d.R[4] * y[p],
                                       d.R[6] * x[p] + d.R[7] * y[p], z[p], d, uu, vv);
                    uint32_t w = 0;
                    if (ok[q][p]) w = load_mask_word(d.mask, mask_word_index(uu, vv, d.tiles_x));
                    fg[q][p] = ((w >> (uu & 31)) & 1u) != 0;
                }
            }
            // U applications of backprojection.c:79-83; a zero pixel in any of the views wins
#pragma unroll
            for (int p = 0; p < P; ++p) {
                bool carve = false, keep = false;
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    carve |= ok[q][p] & !fg[q][p];
                    keep |= ok[q][p] & fg[q][p];
                }
                if (carve & alive[p]) {
                    alive[p] = false;
                    zero[p] = false;
                    labels[idx[p]] = -1;
                } else if (zero[p] & keep) {
                    zero[p] = false;
                    if (FINAL) atomicCAS(&labels[idx[p]], 0, 1); else flipped[p] = true;
                }
            }
        }
        if (!FINAL) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (alive[p] && flipped[p]) labels[idx[p]] = 1;
                unsigned long long b = __ballot(alive[p]);
                if (b != 0) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&ctl->count[sout][s].n, (uint32_t)__popcll(b));
                    base = __shfl(base, 0);
                    // survivors of sub-list s never outnumber its entries: no overflow here
                    if (alive[p]) {
                        unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                        lout[(size_t)s * subcap + base + (uint32_t)__popcll(b & below)] =
                            idx[p] | (zero[p] ? 0x80000000u : 0u);
                    }
                }
            }
        }
    }
    if (with_items) {
        // the bulk units' work items, dealt round-robin: two voxels per lane, the views the item names, two
        // per turn as above
        const uint32_t itotal = ipref[kSub];
        const uint32_t per_plane = ui.bricks_y * ui.bricks_z;
        for (uint32_t i = (uint32_t)wid; i < itotal; i += (uint32_t)nworkers) {
            uint32_t lo = 0, hi = kSub;  // largest s with ipref[s] <= i (wave-uniform)
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ipref[mid] <= i) lo = mid; else hi = mid;
            }
            uint4 it = ui.items[(size_t)lo * ui.cap + (i - ipref[lo])];
            it.x = __builtin_amdgcn_readfirstlane(it.x);
            it.y = __builtin_amdgcn_readfirstlane(it.y);
            it.z = __builtin_amdgcn_readfirstlane(it.z);
            it.w = __builtin_amdgcn_readfirstlane(it.w);
            const uint32_t unit = it.x >> 1, lb = unit >> 2;
            const uint32_t il = lb / per_plane, rem = lb - il * per_plane;
            const uint32_t by = rem / ui.bricks_z, bz = rem - by * ui.bricks_z;
            // half h of a unit: its columns 8 h .. 8 h + 7; lane = (column & 3) * 16 + voxel, p = column >> 2
            const uint32_t j0 = by * kBrickY + (it.x & 1u) * 8u + (lane >> 4), k = bz * kBrickZ + (unit & 3u) * 16u + (lane & 15u);
            const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // backprojection.c:71-73
            const float z = g.oz + (float)(int)k * g.vs;
            uint32_t idx[2];
            bool alive[2], zero[2];
            float y[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint32_t j = j0 + 4u * (uint32_t)p;
                const bool inside = j < g.ny && k < g.nz;
                idx[p] = (il * g.ny + j) * g.nzp + k;
                int32_t lab = -1;
                if (inside) lab = labels[idx[p]];
                alive[p] = lab != -1;
                zero[p] = lab == 0;
                y[p] = g.oy + (float)(int)j * g.vs;
            }
            unsigned long long m = ((unsigned long long)it.w << 32) | it.z;
            const uint32_t vbase = it.y;
            while (m != 0) {
                if (__ballot(alive[0] | alive[1]) == 0) break;
                const uint32_t a = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                uint32_t b = a;  // a lone view is applied twice: nothing changes the second time
                if (m != 0) {
                    b = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                }
                bool ok[2][2], fg[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const ViewDesc d = ui.views[vbase + (q ? b : a)];
                    asm volatile("" ::"s"(d.Wf), "s"(d.Hf), "s"(d.tiles_x), "s"(d.mask));
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        int uu, vv;
                        ok[q][p] = project(d.R[0] * x + d.R[1] * y[p], d.R[3] * x + d.R[4] * y[p],
                                           d.R[6] * x + d.R[7] * y[p], z, d, uu, vv);
                        uint32_t w = 0;
                        if (ok[q][p]) w = load_mask_word(d.mask, mask_word_index(uu, vv, d.tiles_x));
                        fg[q][p] = ((w >> (uu & 31)) & 1u) != 0;
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const bool carve = (ok[0][p] & !fg[0][p]) | (ok[1][p] & !fg[1][p]);
                    const bool keep = (ok[0][p] & fg[0][p]) | (ok[1][p] & fg[1][p]);
                    if (carve & alive[p]) {
                        alive[p] = false;
                        zero[p] = false;
                        labels[idx[p]] = -1;
                    } else if (zero[p] & keep) {
                        zero[p] = false;
                        atomicCAS(&labels[idx[p]], 0, 1);
                    }
                }
            }
        }
    }
    if (rj.report != nullptr && bid == 0) {  // block-uniform
        __shared__ uint32_t s_sum[2];
        if (tid < 2) s_sum[tid] = 0u;
        __syncthreads();
        uint32_t a = 0, b = 0;
        for (uint32_t q = tid; q < rj.nstats; q += kBlock) {
            a += rj.stats[q * 2u];
            b += rj.stats[q * 2u + 1u];
        }
        if (a | b) {
            atomicAdd(&s_sum[0], a);
            atomicAdd(&s_sum[1], b);
        }
        __syncthreads();
        if (tid == 0)
            *rj.report = ((unsigned long long)(rj.seq & 0xffffu) << 48) | ((unsigned long long)min(s_sum[0], 0xffffffu) << 24) |
                         (unsigned long long)min(s_sum[1] >> 4, 0xffffffu);
    }
}

// Fused carve, safety net: when a survivor sub-list overflowed (e.g. masks that carve
// nothing), a persistent grid applies the remaining views densely instead.
struct LateBricks {           // FULL candidates that turned out not to be (see brick_confirm_kernel)
    const uint32_t *late;     // null: the batch had no open candidates
    const ViewDesc *allviews; // every view of the batch
    const uint8_t *flags;
    int32_t nall, init, fresh;
    uint32_t bricks_y, bricks_z;
};

// A unit of a LATE brick (a FULL candidate some later view did not keep whole after all) through every view of
// the batch, one wavefront: the views are first asked about the unit as a whole, 64 at a time, one view per
// lane, at the cell level -- a view that sees it entirely over background carves all of it, views that see it
// entirely over foreground or not at all have nothing to say about its voxels one by one -- and only the
// others project them, two per turn.  (A brick inside a solid object lies over foreground in nearly all the
// views, one at the edge of the pictures outside nearly all.)
template <bool FRESH>
__device__ __forceinline__ void late_unit(int32_t *__restrict__ labels, const GridDesc &g,
                                          const ViewDesc *__restrict__ views, int nall, int32_t init, uint32_t unit,
                                          uint32_t bricks_y, uint32_t bricks_z, uint32_t lane) {
    const uint32_t lb = unit >> 2, w = unit & 3u;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
    const int j0 = (int)(by * kBrickY), kb = (int)(bz * kBrickZ + w * 16u);  // 16 columns x 16 voxels
    const uint32_t j = (uint32_t)j0 + (lane >> 2), k0 = (uint32_t)kb + (lane & 3u) * 4u;
    const bool inside = j < g.ny && k0 < g.nz;
    const int nvalid = inside ? (int)min(4u, g.nz - k0) : 0;
    int32_t *p = labels + ((uint64_t)il * g.ny + j) * g.nzp + k0;  // the pitch is a multiple of 64: 16-byte groups
    int32_t lab[4] = {-1, -1, -1, -1}, was[4];  // what a lane does not own counts as carved
    if (FRESH) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) lab[e] = init;
    } else if (inside) {
        const int4 q = *reinterpret_cast<const int4 *>(p);
        lab[0] = q.x; lab[1] = q.y; lab[2] = q.z; lab[3] = q.w;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e >= nvalid) lab[e] = -1;  // row padding behind the last voxel
    }
    uint32_t alive = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        was[e] = lab[e];
        if (lab[e] != -1) alive |= 1u << e;  // :67
    }
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // :71, global plane index
    const float y = g.oy + (float)(int)j * g.vs;
    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = g.oz + (float)(int)(k0 + e) * g.vs;  // :73
    bool seen = false;
    for (int base = 0; base < nall; base += 64) {
        if (__ballot(alive != 0) == 0) break;
        const int vi = base + (int)lane;
        uint32_t v = 8u;  // no such view
        if (vi < nall) {
            const ViewDesc d = views[vi];  // one descriptor per lane
            v = d.cmask != nullptr ? rect_verdict_cells(d, g, x, j0, j0 + kBrickY - 1, kb, kb + 15) : 0u;
        }
        const unsigned long long empty = __ballot(v == 1u), full = __ballot(v == 2u);
        unsigned long long need = __ballot(v == 0u);
        if (empty != 0) {  // some view carves every voxel of the unit
#pragma unroll
            for (int e = 0; e < 4; ++e) lab[e] = -1;
            alive = 0;
            break;
        }
        seen |= full != 0;
        while (need != 0) {
            if (__ballot(alive != 0) == 0) break;
            const int a = __builtin_ctzll(need);
            need &= need - 1;
            int b = a;
            const bool two = need != 0;
            if (two) {
                b = __builtin_ctzll(need);
                need &= need - 1;
            }
            const ViewDesc da = views[base + a];
            const ViewDesc db = views[base + b];
            two_views(da, db, two, x, y, z, lab, alive);
        }
    }
    if (seen) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (lab[e] == 0) lab[e] = 1;  // :81 by a view that kept the whole unit
    }
    const bool changed = FRESH || lab[0] != was[0] || lab[1] != was[1] || lab[2] != was[2] || lab[3] != was[3];
    if (inside && changed) *reinterpret_cast<int4 *>(p) = make_int4(lab[0], lab[1], lab[2], lab[3]);
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void carve_resume_kernel(int32_t *__restrict__ labels, GridDesc g,
                                                              const ViewDesc *__restrict__ views,
                                                              int nviews, const ListCtl *ctl,
                                                              ListCtl *next, LateBricks lb) {
    // last kernel of a batch: leave the counters of the NEXT batch zeroed (the two blocks
    // alternate; nobody else touches that one now), so no memset sits on the stream
    if (next != nullptr) {
        uint32_t *z = reinterpret_cast<uint32_t *>(next);
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < sizeof(ListCtl) / 4; i += gridDim.x * kBlock) z[i] = 0u;
    }
    Append none{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
    if (lb.late != nullptr) {
        // bricks some later view does not keep whole after all: every view, one wavefront per unit
        const uint32_t nlate = ctl->nlate;
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        const uint32_t nworkers = gridDim.x * (kBlock / 64);
        for (uint32_t t = blockIdx.x * (kBlock / 64) + wave; t < nlate * 4u; t += nworkers) {
            const uint32_t unit = lb.late[t >> 2] * 4u + (t & 3u);
            if (lb.fresh) late_unit<true>(labels, g, lb.allviews, lb.nall, lb.init, unit, lb.bricks_y, lb.bricks_z, lane);
            else late_unit<false>(labels, g, lb.allviews, lb.nall, lb.init, unit, lb.bricks_y, lb.bricks_z, lane);
        }
    }
    if (!ctl->overflow) return;
    uint64_t nblk = (g.ngroups + kBlock - 1) / kBlock;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        uint64_t grp = blk * kBlock + threadIdx.x;
        bool skip = grp >= g.ngroups;
        if (!skip && lb.late != nullptr) {
            // bricks every view keeps whole have nothing to gain from this pass, and late bricks are carved
            // above over ALL views by a block that may not have written them yet: not this pass's voxels
            Vox4 vx;
            decode_group(g, grp, vx);
            const uint32_t col = (uint32_t)(vx.elem / g.nzp), il = col / g.ny, j = col - il * g.ny;
            const uint32_t fl = lb.flags[(il * lb.bricks_y + j / kBrickY) * lb.bricks_z + vx.k0 / kBrickZ];
            skip = fl == 5u || fl == 2u || fl == 6u;
        }
        // (a wavefront's lanes leave carve_group's view loop together: skipped lanes still vote)
        if (grp < g.ngroups && !skip) {
            int4 pre = make_int4(0, 0, 0, 0);
            if (VEC) pre = *reinterpret_cast<const int4 *>(labels + grp * 4);
            carve_group<false, VEC>(labels, g, views, nviews, 0, grp, pre, none);
        }
    }
}

// average (backprojection.c:36-55): value += mask[v][u] for every in-image view, in the
// order given (float32 sum, order matters).  Two mask forms (ViewDesc::pad):
//   0  float32 [H][W] row-major, the value itself (what cl.py:205-215 hands the kernel);
//   1  the ORIGINAL uint8 mask in 16x8-pixel tiles (one 128-byte line per tile) plus a
//      256-entry float table: table[b] is what the host conversion (img_as_float32, log)
//      makes of byte b, so table[mask] is the same float32 the reference would upload, at
//      a quarter of the bytes and with tile-coherent gathers.  The table sits in LDS.
constexpr int kATileW = 16, kATileH = 8;
//   2  float32 in 8x4-pixel tiles (32 floats = one 128-byte line per tile; tilef_kernel)
constexpr int kFTileW = 8, kFTileH = 4;
__device__ __forceinline__ uint32_t ftile_offset(int u, int v, int tiles_x) {
    return (__umul24((uint32_t)(v >> 2), (uint32_t)tiles_x) + (uint32_t)(u >> 3)) * 32u + (uint32_t)((v & 3) * 8 + (u & 7));
}

template <bool FRESH, bool VEC>
__device__ __forceinline__ void average_body(float *__restrict__ values, const GridDesc &g,
                                             const ViewDesc *__restrict__ views, int nviews,
                                             float init, const float *__restrict__ lut) {
    __shared__ float lut_s[256];
    if (lut != nullptr) {  // block-uniform
        lut_s[threadIdx.x] = lut[threadIdx.x];  // kBlock == 256
        __syncthreads();
    }
    uint32_t lb = spread_block(blockIdx.x, gridDim.x);
    uint64_t grp = (uint64_t)lb * kBlock + threadIdx.x;
    if (grp >= g.ngroups) return;
    Vox4 vx;
    decode_group(g, grp, vx);
    float val[4];
    float *p = values + vx.elem;
    if (FRESH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = init;
    } else if (VEC) {
        float4 q = *reinterpret_cast<const float4 *>(p);
        val[0] = q.x; val[1] = q.y; val[2] = q.z; val[3] = q.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = (e < (int)vx.nvalid) ? p[e] : 0.0f;
    }
    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = g.oz + (float)(int)(vx.k0 + e) * g.vs;

    for (int vi = 0; vi < nviews; ++vi) {
        const ViewDesc d = views[vi];
        float ax = d.R[0] * vx.x + d.R[1] * vx.y;
        float ay = d.R[3] * vx.x + d.R[4] * vx.y;
        float az = d.R[6] * vx.x + d.R[7] * vx.y;
        bool ok[4];
        float add[4];
        if (d.pad == 1) {  // wave-uniform
            const uint8_t *m = static_cast<const uint8_t *>(d.mask);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int u, v;
                ok[e] = project(ax, ay, az, z[e], d, u, v) & (e < (int)vx.nvalid);
                uint32_t off = (__umul24((uint32_t)(v >> 3), (uint32_t)d.tiles_x) + (uint32_t)(u >> 4)) * 128u +
                               (uint32_t)((v & 7) * 16 + (u & 15));
                uint32_t b = 0;
                if (ok[e]) b = m[off];
                add[e] = lut_s[b];
            }
        } else if (d.pad == 2) {  // float32 in 8x4 tiles
            const float *m = static_cast<const float *>(d.mask);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int u, v;
                ok[e] = project(ax, ay, az, z[e], d, u, v) & (e < (int)vx.nvalid);
                add[e] = 0.0f;
                if (ok[e]) add[e] = m[ftile_offset(u, v, d.tiles_x)];
            }
        } else {
            const float *m = static_cast<const float *>(d.mask);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int u, v;
                ok[e] = project(ax, ay, az, z[e], d, u, v) & (e < (int)vx.nvalid);
                add[e] = 0.0f;
                if (ok[e]) add[e] = m[(int64_t)v * d.W + u];  // nearest texel
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ok[e]) val[e] = val[e] + add[e];  // :54
    }
    if (VEC) {
        *reinterpret_cast<float4 *>(p) = make_float4(val[0], val[1], val[2], val[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < (int)vx.nvalid) p[e] = val[e];
    }
}

template <bool FRESH, bool VEC>
__global__ __launch_bounds__(kBlock) void average_kernel(float *__restrict__ values, GridDesc g,
                                                         const ViewDesc *__restrict__ views,
                                                         int nviews, float init,
                                                         const float *__restrict__ lut) {
    average_body<FRESH, VEC>(values, g, views, nviews, init, lut);
}

template <bool FRESH, bool VEC>
__global__ __launch_bounds__(kBlock) void average_kernel_1(float *__restrict__ values, GridDesc g,
                                                           ViewDesc view, float init,
                                                           const float *__restrict__ lut) {
    average_body<FRESH, VEC>(values, g, &view, 1, init, lut);
}

// uint8 [V][H][W] row-major -> 16x8-pixel tiles (128 B each) for the averaging gather.
// Fast form: W % 16 == 0 and 16-byte aligned rows -- every lane moves one 16-byte run.
__global__ __launch_bounds__(kBlock) void tile8_kernel(const uint8_t *__restrict__ raw,
                                                       int64_t row_stride, int64_t view_stride, int W,
                                                       int H, int nviews, int tiles_x, int tiles_y,
                                                       uint8_t *__restrict__ out, int fast) {
    int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int chunks = (W + 15) >> 4;
    int64_t total = (int64_t)nviews * H * chunks;
    if (idx >= total) return;
    int c = (int)(idx % chunks);
    int64_t r = idx / chunks;
    int v = (int)(r % H);
    int view = (int)(r / H);
    const uint8_t *src = raw + view * view_stride + (int64_t)v * row_stride + c * 16;
    uint8_t *dst = out + ((int64_t)view * tiles_y * tiles_x + (int64_t)(v >> 3) * tiles_x + c) * 128 + (v & 7) * 16;
    if (fast) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
    } else {
        int n = min(16, W - c * 16);
        for (int k = 0; k < n; ++k) dst[k] = src[k];
    }
}

// One byte per 32x32-pixel tile of a 16x8-tiled uint8 mask (W % 16 == 0): bit 0 = some byte is not
// 0, bit 1 = every byte is 255 -- what brick_verdict reads as "some / only foreground"
// (average_brick_kernel).  One wavefront per tile: lane l takes the 16 pixels (row l >> 1, half
// l & 1); pixels beyond the picture do not count.  (Setting the flags from the tiling kernel
// itself, with atomics on the shared bytes, cost 1 ms per 72 masks.)
__global__ __launch_bounds__(kBlock) void uniform_tiles_kernel(const uint8_t *__restrict__ tiled, int W, int H,
                                                               int nviews, int tiles_x, int tiles_y,
                                                               uint8_t *__restrict__ uni) {
    const int otx = (W + 31) >> 5, oty = (H + 31) >> 5;
    const int64_t tile = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (tile >= (int64_t)nviews * otx * oty) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const int tx = (int)(tile % otx), ty = (int)((tile / otx) % oty), view = (int)(tile / ((int64_t)otx * oty));
    const int v = ty * 32 + (lane >> 1), c = tx * 2 + (lane & 1);
    bool nz = false, hole = false;
    if (v < H && c * 16 < W) {
        const uint8_t *src = tiled + ((int64_t)view * tiles_y * tiles_x + (int64_t)(v >> 3) * tiles_x + c) * 128 + (v & 7) * 16;
        const uint4 q = *reinterpret_cast<const uint4 *>(src);
        nz = (q.x | q.y | q.z | q.w) != 0u;
        hole = (q.x & q.y & q.z & q.w) != 0xffffffffu;
    }
    const unsigned long long anynz = __ballot(nz), anyhole = __ballot(hole);
    if (lane == 0) uni[tile] = (uint8_t)((anynz ? 1u : 0u) | (anyhole ? 0u : 2u));
}

// float32 [V][H][W] row-major -> 8x4-pixel tiles (32 floats = one 128-byte line per tile): the voxels a
// wavefront projects land on a short image segment of any orientation, i.e. on a handful of lines,
// where row-major floats give one line per 32 pixels of ONE row.  Fast form: W % 4 == 0 and 16-byte
// aligned rows -- every lane moves four floats.
__global__ __launch_bounds__(kBlock) void tilef_kernel(const float *__restrict__ raw, int64_t row_stride,
                                                       int64_t view_stride, int W, int H, int nviews,
                                                       int tiles_x, int tiles_y, float *__restrict__ out, int fast) {
    int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int chunks = (W + 3) >> 2;
    int64_t total = (int64_t)nviews * H * chunks;
    if (idx >= total) return;
    int c = (int)(idx % chunks);
    int64_t r = idx / chunks;
    int v = (int)(r % H);
    int view = (int)(r / H);
    const float *src = reinterpret_cast<const float *>(reinterpret_cast<const char *>(raw) + view * view_stride +
                                                       (int64_t)v * row_stride) + c * 4;
    float *dst = out + (int64_t)view * tiles_y * tiles_x * 32 + ftile_offset(c * 4, v, tiles_x);
    if (fast) {
        *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
    } else {
        int n = min(4, W - c * 4);
        for (int k = 0; k < n; ++k) dst[k] = src[k];
    }
}

// Per 32x32-pixel region of a tiled float mask: is it one value, bit for bit (pixels beyond the
// picture do not count)?  flag byte + the value of its first pixel.  One wavefront per region.
__global__ __launch_bounds__(kBlock) void uniform_f32_kernel(const float *__restrict__ tiled, int W, int H, int nviews,
                                                             int tiles_x, int tiles_y, uint8_t *__restrict__ uni,
                                                             size_t uni_view_bytes) {
    const int otx = (W + 31) >> 5, oty = (H + 31) >> 5;
    const int64_t reg = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (reg >= (int64_t)nviews * otx * oty) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const int rx = (int)(reg % otx), ry = (int)((reg / otx) % oty), view = (int)(reg / ((int64_t)otx * oty));
    const uint32_t *base = reinterpret_cast<const uint32_t *>(tiled) + (int64_t)view * tiles_y * tiles_x * 32;
    const uint32_t first = base[ftile_offset(rx * 32, ry * 32, tiles_x)];
    bool same = true;
    // lane l: row ry*32 + l/2, half l & 1 of the 32 columns
    const int v = ry * 32 + (lane >> 1);
    if (v < H) {
        for (int q = 0; q < 16; ++q) {
            const int u = rx * 32 + (lane & 1) * 16 + q;
            if (u < W) same &= base[ftile_offset(u, v, tiles_x)] == first;
        }
    }
    const unsigned long long differ = __ballot(!same);
    if (lane == 0) {
        uint8_t *f = uni + (size_t)view * uni_view_bytes;
        f[ry * otx + rx] = differ == 0 ? 1 : 0;
        reinterpret_cast<uint32_t *>(f + (((size_t)otx * oty + 3) & ~(size_t)3))[ry * otx + rx] = first;
    }
}

// ---- brick form of the averaging kernel (uint8 masks + table) ---------------------------------
// Masks out of a segmentation are mostly flat: background 0, foreground 255.  Where a brick's
// footprint in a view (same conservative box as the carve's brick_verdict) lies over tiles of
// nothing but 0, every voxel of the brick is in-image and adds table[0] for that view; over tiles of
// nothing but 255, table[255]: the same float32 addition the reference performs
// (backprojection.c:54), in the same view order, without projecting anything.  Only views whose
// footprint is mixed are projected voxel by voxel; a view that does not see the brick at all (verdict 4,
// OUTSIDE) is skipped.
__global__ __launch_bounds__(kBlock) void avg_flags_kernel(GridDesc g, const ViewDesc *__restrict__ views,
                                                           int nviews, uint32_t bricks_y, uint32_t bricks_z,
                                                           uint32_t nbricks, uint8_t *__restrict__ verd,
                                                           uint32_t *__restrict__ verdf) {
    const uint32_t lb = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t vi = blockIdx.y;  // block-uniform view: scalar descriptor
    if (lb >= nbricks) return;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
    const ViewDesc d = views[vi];
    if (d.pad == 2) {  // tiled float32 mask: flat when every region under the brick holds one value
        uint32_t bits;
        const uint32_t v = brick_flat_f32(d, g, x, (int)(by * kBrickY), (int)(bz * kBrickZ), bits);
        verd[(size_t)lb * (uint32_t)nviews + vi] = (uint8_t)v;
        if (verdf != nullptr) verdf[(size_t)lb * (uint32_t)nviews + vi] = bits;
        return;
    }
    verd[(size_t)lb * (uint32_t)nviews + vi] =
        (uint8_t)brick_verdict(d, g, x, (int)(by * kBrickY), (int)(bz * kBrickZ), (d.W + 31) >> 5);
}

template <bool FRESH>
__global__ __launch_bounds__(kBlock) void average_brick_kernel(float *__restrict__ values, GridDesc g,
                                                               const ViewDesc *__restrict__ views, int nviews,
                                                               float init, const float *__restrict__ lut,
                                                               uint32_t bricks_y, uint32_t bricks_z,
                                                               const uint8_t *__restrict__ verd,
                                                               const uint32_t *__restrict__ verdf) {
    __shared__ float lut_s[256];
    lut_s[threadIdx.x] = lut != nullptr ? lut[threadIdx.x] : 0.0f;  // kBlock == 256
    __syncthreads();
    const uint32_t lb = spread_block(blockIdx.x, gridDim.x);
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t j = by * kBrickY + wave * 4 + (lane >> 4);
    const uint32_t k0 = bz * kBrickZ + (lane & 15) * 4;
    const bool inside = j < g.ny && k0 < g.nz;
    const int nvalid = inside ? (int)min(4u, g.nz - k0) : 0;
    const bool vec = (g.nzp & 3u) == 0;
    float *p = values + ((uint64_t)il * g.ny + j) * g.nzp + k0;
    float val[4] = {init, init, init, init};
    if (!FRESH) {
        if (vec) {
            if (inside) {
                float4 q = *reinterpret_cast<const float4 *>(p);
                val[0] = q.x; val[1] = q.y; val[2] = q.z; val[3] = q.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nvalid) val[e] = p[e];
        }
    }
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // backprojection.c:71-73
    const float y = g.oy + (float)(int)j * g.vs;
    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = g.oz + (float)(int)(k0 + e) * g.vs;
    const float add0 = lut_s[0], add255 = lut_s[255];
    const uint8_t *myverd = verd + (size_t)lb * (uint32_t)nviews;
    const uint32_t *myverdf = verdf != nullptr ? verdf + (size_t)lb * (uint32_t)nviews : nullptr;
    for (int v0 = 0; v0 < nviews; v0 += 64) {
        // the verdicts of up to 64 views, one per lane, handed out with v_readlane
        const int nv = min(64, nviews - v0);
        const uint32_t mine = ((int)lane < nv) ? myverd[v0 + (int)lane] : 0u;
        const uint32_t minef = (myverdf != nullptr && (int)lane < nv) ? myverdf[v0 + (int)lane] : 0u;
        for (int q = 0; q < nv; ++q) {
            const uint32_t c = __builtin_amdgcn_readlane(mine, q);  // wave-uniform (brick-uniform)
            if (c == 4u) continue;  // OUTSIDE: no voxel of the brick is in the picture, the view adds nothing (:50-52)
            if (c != 0u) {
                const float add = c == 1u ? add0 : (c == 2u ? add255 : __uint_as_float(__builtin_amdgcn_readlane(minef, q)));
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = val[e] + add;  // :54, every voxel is in-image
                continue;
            }
            const ViewDesc d = views[v0 + q];
            const float ax = d.R[0] * x + d.R[1] * y, ay = d.R[3] * x + d.R[4] * y, az = d.R[6] * x + d.R[7] * y;
            if (d.pad == 2) {  // tiled float32 mask (wave-uniform)
                const float *mf = static_cast<const float *>(d.mask);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int u, v;
                    const bool ok = project(ax, ay, az, z[e], d, u, v);
                    float add = 0.0f;
                    if (ok) add = mf[ftile_offset(u, v, d.tiles_x)];
                    if (ok) val[e] = val[e] + add;  // :54
                }
                continue;
            }
            const uint8_t *m = static_cast<const uint8_t *>(d.mask);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int u, v;
                const bool ok = project(ax, ay, az, z[e], d, u, v);
                const uint32_t off = (__umul24((uint32_t)(v >> 3), (uint32_t)d.tiles_x) + (uint32_t)(u >> 4)) * 128u +
                                     (uint32_t)((v & 7) * 16 + (u & 15));
                uint32_t b = 0;
                if (ok) b = m[off];
                const float add = lut_s[b];
                if (ok) val[e] = val[e] + add;  // :54
            }
        }
    }
    if (vec) {
        if (inside) *reinterpret_cast<float4 *>(p) = make_float4(val[0], val[1], val[2], val[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) p[e] = val[e];
    }
}

// ---- several labels at once --------------------------------------------------------------------------------
// The label volumes of one scan share their cameras (the reference's process_fileset runs the same poses once
// per label, cl.py:248-255): here a voxel is projected ONCE per view and the L labels' masks are gathered at
// that pixel into L sums -- each sum the same float32 additions in the same view order as its own launch would
// make (bit-identical per label by construction).  Brick form as above: a view is projected for a brick only
// if some label's footprint there is mixed; a label whose footprint is flat adds its table value.
#ifndef SC_MAXLABELS
#define SC_MAXLABELS 4
#endif
constexpr int kMaxLabels = SC_MAXLABELS;
struct MultiArgs {
    float *values[kMaxLabels];
    const ViewDesc *views[kMaxLabels];  // label l's descriptors (its own tiled masks; the poses are the same)
    const uint8_t *verd[kMaxLabels];    // [bricks][views] verdicts of label l (avg_flags_kernel)
    const float *lut[kMaxLabels];
    float init[kMaxLabels];
};

// The verdicts of the L labels about every (brick, view): the footprint -- a matter of the pose -- is worked out
// once, the labels differ in the uniformity flags under it (avg_flags_kernel, uint8 masks).
template <int L>
__global__ __launch_bounds__(kBlock) void avg_flags_multi_kernel(MultiArgs a, GridDesc g, int nviews, uint32_t bricks_y,
                                                                 uint32_t bricks_z, uint32_t nbricks, uint8_t *const *verd_out) {
    const uint32_t lb = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t vi = blockIdx.y;  // block-uniform view: scalar descriptor
    if (lb >= nbricks) return;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
    const ViewDesc d = a.views[0][vi];
    const Footprint fpr = brick_footprint(d, g, x, (int)(by * kBrickY), (int)(bz * kBrickZ));
    const int occ_tx = (d.W + 31) >> 5;
    uint32_t v[L];
#pragma unroll
    for (int l = 0; l < L; ++l) v[l] = fpr.outside ? 4u : 0u;
    if (!fpr.outside && fpr.ok) {
        uint32_t any[L], all[L];
#pragma unroll
        for (int l = 0; l < L; ++l) { any[l] = 0; all[l] = 3; }
        for (int ty = fpr.ty0; ty <= fpr.ty1; ++ty)
            for (int tx = fpr.tx0; tx <= fpr.tx1; ++tx) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const uint32_t o = a.views[l][vi].occ[ty * occ_tx + tx];
                    any[l] |= o;
                    all[l] &= o;
                }
            }
#pragma unroll
        for (int l = 0; l < L; ++l) v[l] = (any[l] & 1u) == 0 ? 1u : ((all[l] & 2u) != 0 ? 2u : 0u);
    }
#pragma unroll
    for (int l = 0; l < L; ++l) const_cast<uint8_t *>(a.verd[l])[(size_t)lb * (uint32_t)nviews + vi] = (uint8_t)v[l];
}

template <int L, bool FRESH>
__global__ __launch_bounds__(kBlock) void average_multi_kernel(MultiArgs a, GridDesc g, int nviews, uint32_t bricks_y,
                                                               uint32_t bricks_z) {
    __shared__ float lut_s[L][256];
#pragma unroll
    for (int l = 0; l < L; ++l) lut_s[l][threadIdx.x] = a.lut[l][threadIdx.x];  // kBlock == 256
    __syncthreads();
    const uint32_t lb = spread_block(blockIdx.x, gridDim.x);
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t j = by * kBrickY + wave * 4 + (lane >> 4);
    const uint32_t k0 = bz * kBrickZ + (lane & 15) * 4;
    const bool inside = j < g.ny && k0 < g.nz;
    const uint64_t elem = ((uint64_t)il * g.ny + j) * g.nzp + k0;  // the pitch is a multiple of 64: 16-byte groups
    float val[L][4];
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
        for (int e = 0; e < 4; ++e) val[l][e] = a.init[l];
        if (!FRESH && inside) {
            const float4 q = *reinterpret_cast<const float4 *>(a.values[l] + elem);
            val[l][0] = q.x; val[l][1] = q.y; val[l][2] = q.z; val[l][3] = q.w;
        }
    }
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // backprojection.c:71-73
    const float y = g.oy + (float)(int)j * g.vs;
    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = g.oz + (float)(int)(k0 + e) * g.vs;
    for (int v0 = 0; v0 < nviews; v0 += 64) {
        // the verdicts of up to 64 views per label, one per lane, handed out with v_readlane
        const int nv = min(64, nviews - v0);
        uint32_t mine[L];
#pragma unroll
        for (int l = 0; l < L; ++l) mine[l] = ((int)lane < nv) ? a.verd[l][(size_t)lb * (uint32_t)nviews + v0 + (int)lane] : 0u;
        for (int q = 0; q < nv; ++q) {
            uint32_t c[L];
            bool mixed = false;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                c[l] = __builtin_amdgcn_readlane(mine[l], q);  // wave-uniform (brick-uniform)
                mixed |= c[l] == 0u;
            }
            if (c[0] == 4u) continue;  // OUTSIDE is a matter of the pose: no label's picture holds a voxel of the brick (:50-52)
            if (!mixed) {  // every label's footprint is flat: the labels' table values, nothing projected
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const float add = c[l] == 1u ? lut_s[l][0] : lut_s[l][255];
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[l][e] = val[l][e] + add;  // :54, every voxel is in-image
                }
                continue;
            }
            const ViewDesc d = a.views[0][v0 + q];  // the pose, and the picture's geometry
            const float ax = d.R[0] * x + d.R[1] * y, ay = d.R[3] * x + d.R[4] * y, az = d.R[6] * x + d.R[7] * y;
            bool ok[4];
            uint32_t off[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int u, v;
                ok[e] = project(ax, ay, az, z[e], d, u, v);
                off[e] = (__umul24((uint32_t)(v >> 3), (uint32_t)d.tiles_x) + (uint32_t)(u >> 4)) * 128u +
                         (uint32_t)((v & 7) * 16 + (u & 15));
            }
#pragma unroll
            for (int l = 0; l < L; ++l) {
                if (c[l] != 0u) {  // flat for this label: every voxel is in-image and adds the one value
                    const float add = c[l] == 1u ? lut_s[l][0] : lut_s[l][255];
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[l][e] = val[l][e] + add;
                    continue;
                }
                const uint8_t *m = static_cast<const uint8_t *>(a.views[l][v0 + q].mask);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t b = 0;
                    if (ok[e]) b = m[off[e]];
                    const float add = lut_s[l][b];
                    if (ok[e]) val[l][e] = val[l][e] + add;  // :54
                }
            }
        }
    }
    if (inside) {
#pragma unroll
        for (int l = 0; l < L; ++l)
            *reinterpret_cast<float4 *>(a.values[l] + elem) = make_float4(val[l][0], val[l][1], val[l][2], val[l][3]);
    }
}

// Self-test of the shared-reciprocal division against the compiler's IEEE division.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(kBlock) void div_selftest_kernel(uint64_t count, uint32_t seed, int mode,
                                                              unsigned long long *out) {
    uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * kBlock;
    unsigned long long bad = 0, fast = 0;
    for (; i < count; i += stride) {
        uint32_t a = mix32((uint32_t)i ^ seed), b = mix32((uint32_t)(i >> 32) + a + seed * 0x9e3779b9u);
        uint32_t c = mix32(a ^ (b * 0x85ebca6bu) ^ 0x1234567u), e = mix32(c + b);
        float n1, n2, dd;
        if (mode == 0) {  // raw bit patterns: every class of float
            n1 = __uint_as_float(a); n2 = __uint_as_float(b); dd = __uint_as_float(c);
        } else {          // projection-like magnitudes, random mantissas and signs
            n1 = __uint_as_float((a & 0x807fffffu) | ((110u + (e & 31u)) << 23));
            n2 = __uint_as_float((b & 0x807fffffu) | ((110u + ((e >> 5) & 31u)) << 23));
            dd = __uint_as_float((c & 0x007fffffu) | ((118u + ((e >> 10) & 15u)) << 23));
        }
        if (div_fast_range(n1, n2, dd)) {
            ++fast;
            float r = refined_rcp(dd);
            float q1 = div_by_rcp(n1, dd, r), q2 = div_by_rcp(n2, dd, r);
            float w1 = n1 / dd, w2 = n2 / dd;
            bad += (__float_as_uint(q1) != __float_as_uint(w1)) + (__float_as_uint(q2) != __float_as_uint(w2));
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    if (fast) atomicAdd(&out[1], fast);
}

// Self-test of project() itself -- the only place where bit-exactness with the reference's
// backproject_point (backprojection.c:3-34) can break -- on explicit or hashed samples.  A sample is
// (pose record, voxel index); its result word is  v * W + u + 1  when the reference would touch
// mask[v][u] and 0 when it rejects the point.  Hashed samples (ijk == nullptr): the pose record is
// drawn per wavefront (as in the voxel kernels, where the view is wave-uniform and the shared-
// reciprocal division is taken when every lane is in range), the voxel per lane inside that
// record's grid.  The CPU oracle generates the same samples (oracle_selftest_project) and the
// test compares the words, or a digest per 65536 samples: sum of mix32(word ^ index).
struct PoseRec {   // 28 words; sc_selftest_project's `poses`
    float K[4], R[9], t[3];
    float ox, oy, oz, vs;
    int32_t W, H, nx, ny, nz, pad[3];
};
static_assert(sizeof(PoseRec) == 112, "PoseRec layout");

__global__ __launch_bounds__(kBlock) void project_selftest_kernel(uint64_t count, uint32_t seed, uint32_t nposes,
                                                                  const PoseRec *__restrict__ poses,
                                                                  const int32_t *__restrict__ ijk,
                                                                  const int32_t *__restrict__ pose_idx,
                                                                  uint32_t *__restrict__ words,
                                                                  unsigned long long *__restrict__ digests) {
    const uint64_t nwaves = (count + 63) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t w = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); w < nwaves;
         w += (uint64_t)gridDim.x * (kBlock / 64)) {
        const uint64_t i = w * 64 + lane;
        unsigned long long part = 0;
        if (i < count) {
            uint32_t p;
            int vi, vj, vk;
            if (ijk != nullptr) {
                p = pose_idx ? (uint32_t)pose_idx[i] : 0u;
                vi = ijk[3 * i]; vj = ijk[3 * i + 1]; vk = ijk[3 * i + 2];
            } else {
                uint32_t h = mix32((uint32_t)w ^ seed);
                h = mix32(h + (uint32_t)(w >> 32) * 0x9e3779b9u);
                p = h % nposes;
                const uint32_t h2 = mix32((uint32_t)i * 0x9e3779b9u + seed + (uint32_t)(i >> 32));
                const uint32_t h3 = mix32(h2 ^ 0x85ebca6bu), h4 = mix32(h3 + 0xc2b2ae35u);
                vi = (int)(h2 % (uint32_t)poses[p].nx);
                vj = (int)(h3 % (uint32_t)poses[p].ny);
                vk = (int)(h4 % (uint32_t)poses[p].nz);
            }
            const PoseRec r = poses[p];
            ViewDesc d;
#pragma unroll
            for (int q = 0; q < 4; ++q) d.K[q] = r.K[q];
#pragma unroll
            for (int q = 0; q < 9; ++q) d.R[q] = r.R[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) d.t[q] = r.t[q];
            d.mask = nullptr; d.occ = nullptr; d.W = r.W; d.H = r.H; d.tiles_x = 0; d.pad = 0;
            d.safe = r.pad[0]; d.pad2 = 0;  // certified by the host for the box the samples come from
            d.Wf = (float)r.W; d.Hf = (float)r.H;
            // exactly what the voxel kernels do: coordinates as backprojection.c:71-73, the x / y
            // partial sums of the three dot products first (the reference's own association)
            const float x = r.ox + (float)vi * r.vs, y = r.oy + (float)vj * r.vs, z = r.oz + (float)vk * r.vs;
            const float ax = d.R[0] * x + d.R[1] * y, ay = d.R[3] * x + d.R[4] * y, az = d.R[6] * x + d.R[7] * y;
            int u, v;
            const bool ok = project(ax, ay, az, z, d, u, v);
            const uint32_t word = ok ? (uint32_t)v * (uint32_t)r.W + (uint32_t)u + 1u : 0u;
            if (words != nullptr) words[i] = word;
            part = (unsigned long long)mix32(word ^ (uint32_t)i);
        }
        if (digests != nullptr) {  // 64 consecutive samples share a digest: one atomic per wavefront
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (lane == 0) atomicAdd(&digests[w >> 10], part);
        }
    }
}

// int32 labels -> int8 (sc_get_values_i8): 16 labels per lane, 4 coalesced 16-byte loads in flight,
// one 16-byte store.
__global__ __launch_bounds__(kBlock) void narrow_i8_kernel(const int32_t *__restrict__ src, int8_t *__restrict__ dst,
                                                           uint64_t n) {
    const uint64_t base = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 16;
    if (base + 16 <= n && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int4 v = *reinterpret_cast<const int4 *>(src + base + 4 * q);
            w[q] = ((uint32_t)v.x & 0xffu) | (((uint32_t)v.y & 0xffu) << 8) | (((uint32_t)v.z & 0xffu) << 16) |
                   (((uint32_t)v.w & 0xffu) << 24);
        }
        *reinterpret_cast<uint4 *>(dst + base) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (uint64_t i = base; i < n && i < base + 16; ++i) dst[i] = (int8_t)src[i];
    }
}

// The state without its row padding (rows of nz of nzp elements), as 4-byte elements or narrowed to int8:
// one wavefront per row and pass, consecutive lanes on consecutive elements.
template <typename OUT>
__global__ __launch_bounds__(kBlock) void depitch_kernel(const uint32_t *__restrict__ src, OUT *__restrict__ dst,
                                                         uint64_t rows, uint32_t nz, uint32_t nzp) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t row = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < rows;
         row += (uint64_t)gridDim.x * (kBlock / 64)) {
        const uint32_t *in = src + row * nzp;
        OUT *out = dst + row * nz;
        for (uint32_t k = lane; k < nz; k += 64u) out[k] = (OUT)in[k];  // int8: the low byte, as the narrowing kernel
    }
}

// Carve labels packed for the wire (multi-GPU assembly): BITS = 2 keeps the three states (label & 3:
// -1 -> 3, 0 -> 0, 1 -> 1), BITS = 1 the occupancy the consumer binarises to (label == 1: proc3d.py:515 reads
// `volume > 0.5`); voxel v of the engine's planes * ny * nz voxels (no row padding) sits at bit BITS * (v % (32 /
// BITS)) of word v / (32 / BITS).  One lane makes one word.  Rows that are whole words (nz % (32 / BITS) == 0)
// are read as 16-byte groups, and a word that lies in a DEAD brick -- some launch found the brick empty, every
// voxel is -1 until the next clear -- is written without reading its labels: on a plant 94 % of the volume.
template <int BITS>
__global__ __launch_bounds__(kBlock) void pack_labels_kernel(const int32_t *__restrict__ labels, uint32_t *__restrict__ out,
                                                             uint64_t n, uint32_t nz, uint32_t nzp, uint32_t ny,
                                                             const uint8_t *__restrict__ dead, uint32_t bricks_y,
                                                             uint32_t bricks_z) {
    constexpr uint32_t PER = 32u / BITS;
    const uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t v0 = w * PER;
    if (v0 >= n) return;
    uint32_t word = 0;
    const uint64_t row = v0 / nz;
    const uint32_t k = (uint32_t)(v0 - row * nz);
    if (nz % PER == 0 && v0 + PER <= n) {  // the word lies inside one row, 16-byte aligned in the pitched state
        if (dead != nullptr) {
            const uint32_t il = (uint32_t)(row / ny), j = (uint32_t)(row - (uint64_t)il * ny);
            if (dead[(il * bricks_y + j / kBrickY) * bricks_z + k / kBrickZ]) {  // PER <= 32 divides 64: one brick
                out[w] = BITS == 2 ? 0xffffffffu : 0u;
                return;
            }
        }
        const int4 *src = reinterpret_cast<const int4 *>(labels + row * nzp + k);
#pragma unroll
        for (uint32_t q = 0; q < PER / 4; ++q) {
            const int4 a = src[q];
            if (BITS == 2)
                word |= (((uint32_t)a.x & 3u) | (((uint32_t)a.y & 3u) << 2) | (((uint32_t)a.z & 3u) << 4) | (((uint32_t)a.w & 3u) << 6)) << (8 * q);
            else
                word |= ((a.x == 1 ? 1u : 0u) | (a.y == 1 ? 2u : 0u) | (a.z == 1 ? 4u : 0u) | (a.w == 1 ? 8u : 0u)) << (4 * q);
        }
    } else {
        uint64_t r = row;
        uint32_t kk = k;
        for (uint32_t q = 0; q < PER && v0 + q < n; ++q) {
            const int32_t a = labels[r * nzp + kk];
            word |= (BITS == 2 ? ((uint32_t)a & 3u) : (a == 1 ? 1u : 0u)) << (BITS * q);
            if (++kk == nz) { kk = 0; ++r; }
        }
    }
    out[w] = word;
}

// The other end of the wire: `world` ranks' packed planes, as an all-gather leaves them ([world][rank_words]
// words, rank r's planes in its own order), into ONE grid in global order -- plane i of the grid is plane i / world
// of rank i % world (plane-cyclic) or plane i - first(r) of the rank whose slab holds it -- unpacked to int8 or
// int32 on the way.  One lane makes 16 consecutive voxels of the output (one 16-byte store as int8, four as
// int32); planes of whole words (ny * nz % (32 / BITS) == 0) take one word each, other shapes voxel by voxel.
template <int BITS, typename OUT>
__global__ __launch_bounds__(kBlock) void unpack_labels_kernel(const uint32_t *__restrict__ recv, OUT *__restrict__ out,
                                                               uint64_t rank_words, uint32_t world, uint32_t nx,
                                                               uint64_t plane, int cyclic) {
    constexpr uint32_t PER = 32u / BITS;
    const uint64_t n = (uint64_t)nx * plane;
    const uint64_t v0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 16u;
    if (v0 >= n) return;
    auto locate = [&](uint64_t v, uint32_t &r, uint64_t &src) {
        const uint32_t i = (uint32_t)(v / plane);
        const uint64_t within = v - (uint64_t)i * plane;
        uint32_t p;
        if (cyclic) {
            r = i % world;
            p = i / world;
        } else {  // slabs [nx r / world, nx (r + 1) / world)
            r = (uint32_t)(((uint64_t)i * world + world - 1) / nx);
            while ((uint64_t)nx * r / world > i) --r;
            while ((uint64_t)nx * (r + 1) / world <= i) ++r;
            p = i - (uint32_t)((uint64_t)nx * r / world);
        }
        src = (uint64_t)p * plane + within;
    };
    auto decode = [](uint32_t bits) -> OUT {
        if (BITS == 2) return (OUT)((bits & 3u) == 3u ? -1 : (int)(bits & 3u));
        return (OUT)(bits & 1u);
    };
    OUT vals[16];
    if (plane % PER == 0 && v0 + 16 <= n) {  // 16 | PER: the 16 voxels share a plane and a word
        uint32_t r;
        uint64_t src;
        locate(v0, r, src);
        const uint32_t word = recv[(uint64_t)r * rank_words + src / PER] >> (BITS * (uint32_t)(src % PER));
#pragma unroll
        for (int q = 0; q < 16; ++q) vals[q] = decode(word >> (BITS * q));
        if (sizeof(OUT) == 1) {
            *reinterpret_cast<uint4 *>(out + v0) = *reinterpret_cast<const uint4 *>(vals);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint4 *>(out + v0 + 4 * q) = *reinterpret_cast<const uint4 *>(vals + 4 * q);
        }
        return;
    }
    for (uint32_t q = 0; q < 16 && v0 + q < n; ++q) {
        uint32_t r;
        uint64_t src;
        locate(v0 + q, r, src);
        out[v0 + q] = decode(recv[(uint64_t)r * rank_words + src / PER] >> (BITS * (uint32_t)(src % PER)));
    }
}

__global__ __launch_bounds__(kBlock) void fill_kernel(uint32_t *__restrict__ dst, uint64_t n,
                                                      uint32_t bits) {
    uint64_t idx = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (idx + 4 <= n) {
        *reinterpret_cast<uint4 *>(dst + idx) = make_uint4(bits, bits, bits, bits);
    } else {
        for (; idx < n; ++idx) dst[idx] = bits;
    }
}

// Mask ingest, general form: raw [V][H][W] pixels (u8 or i32) -> 1 bit/pixel (pixel !=
// background; background 0 is the test at backprojection.c:79 on the cast of cl.py:215, 255 / 1
// fold the fileset loop's np.invert of a uint8 / bool mask, cl.py:300-301), in 32x32 tiles.  One wavefront votes
// 64 consecutive pixels of a row with a ballot and writes the two 32-bit tile words.
template <typename T>
__global__ __launch_bounds__(kBlock) void pack_kernel(const T *__restrict__ raw,
                                                      int64_t row_stride, int64_t view_stride,
                                                      int W, int H, int nviews, int tiles_x,
                                                      uint32_t *__restrict__ out,
                                                      int64_t out_view_words, T background,
                                                      uint8_t *__restrict__ occ, int tiles_y) {
    const int lane = threadIdx.x & 63;
    const int segs = (W + 63) >> 6;
    int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    int64_t total = (int64_t)nviews * H * segs;
    if (wave >= total) return;
    int seg = (int)(wave % segs);
    int64_t r = wave / segs;
    int v = (int)(r % H);
    int view = (int)(r / H);
    int u = seg * 64 + lane;
    bool fg = false;
    if (u < W) {
        const char *row = reinterpret_cast<const char *>(raw) + view * view_stride + v * row_stride;
        fg = reinterpret_cast<const T *>(row)[u] != background;
    }
    unsigned long long vote = __ballot(fg);
    uint32_t *o = out + view * out_view_words;
    uint32_t base = (uint32_t)(v >> 5) * (uint32_t)tiles_x;
    uint32_t rowin = (uint32_t)(v & 31);
    uint8_t *oc = occ + (int64_t)view * tiles_x * tiles_y;  // zeroed by the host; racing stores all write 1
    if (lane == 0) {
        o[(base + seg * 2) * 32u + rowin] = (uint32_t)vote;
        if ((uint32_t)vote) oc[base + seg * 2] = 1;
    } else if (lane == 32 && seg * 2 + 1 < tiles_x) {
        o[(base + seg * 2 + 1) * 32u + rowin] = (uint32_t)(vote >> 32);
        if ((uint32_t)(vote >> 32)) oc[base + seg * 2 + 1] = 1;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(_e == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE,            \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                        __LINE__);                                                           \
    } while (0)

struct Chunk {
    char *base = nullptr;
    size_t cap = 0, used = 0;
};

struct TimedLaunch {
    hipEvent_t start, stop;
};

constexpr int kSlots = 4;
constexpr int kNumKernels = 7;

}  // namespace

struct sc_engine {
    int device = 0;
    int mode = SC_MODE_CARVE;
    int64_t nx = 0, ny = 0, nz = 0, i0 = 0, istride = 1, planes = 0, n = 0;
    int64_t nzp = 0;     // row pitch of the state in voxels (nz rounded up to a multiple of 64)
    int64_t npitch = 0;  // planes * ny * nzp: elements of the state as it lies in memory
    void *dense = nullptr;  // planes * ny * nz elements: the state without the row padding, made on demand
                            // for read-backs and device consumers when nzp != nz
    float origin[3] = {0, 0, 0};
    float vs = 1.0f;
    float default_value = 0.0f;
    void *state = nullptr;
    bool fresh = true;

    hipStream_t own_stream = nullptr, stream = nullptr;

    // deferred views
    std::vector<ViewDesc> pending;
    ViewDesc *views_dev = nullptr;  // ring of descriptors, consumed in stream order
    ViewDesc *views_pin = nullptr;
    size_t views_cap = 0, views_head = 0;

    // mask storage for pending views
    std::vector<Chunk> chunks;

    uint8_t *flags = nullptr;  // fused carve, brick form: one emptiness verdict per brick (inside ctl's allocation)
    uint32_t *live = nullptr;  // ... and the bricks no view found empty (count in ctl->nlive)
    ListCtl *ctl2[2] = {nullptr, nullptr};  // counter blocks of alternate batches (ctl points at the current one)
    bool ctl_clean[2] = {false, false};     // known to be all zero
    int ctl_idx = 0;
    int64_t full_bricks = 1;      // bricks every view sees whole over foreground get their label without projections
    int64_t avg_brick = 1;        // averaging: brick form with uniform-footprint verdicts
    int64_t avg_tile_f32 = 1;     // averaging: float32 masks are re-laid in 8x4-pixel tiles (0: read row-major)
    uint8_t *verd = nullptr;      // ... its [bricks][views] verdicts
    size_t verd_cap = 0;
    uint32_t *verdf = nullptr;    // ... and, for tiled float32 masks, the value a flat footprint adds
    size_t verdf_cap = 0;
    int64_t stage1_store_share = 4;  // sixteenths of the deferred strips filled beside the FIRST list stage
    int64_t stage1_list_blocks = 1280; // ... and that stage's persistent list blocks then
    int64_t defer_share = 16;     // sixteenths of the strips whose empty bricks the final list stage fills
    int64_t defer_stores = 1280;  // list blocks of a final stage that also fills the empty bricks (0: the dense stage fills them);
                                  // 5 per CU beside 2 store blocks: 1024 -> 1280 is worth 4-5 % on bulky scenes, nothing on a plant; 1536 loses
    int64_t pack_rows = 0;     // 0: the band form of the 16-byte pack kernel; 1, 2, 4, 8: the panel form, tile rows per block
    int64_t view_brick = 1;    // a single-view carve launch goes through the brick kernels too (0: streaming kernel)
    uint8_t *dead = nullptr;   // per brick: an earlier launch found it empty, every voxel is -1 (until the next clear)
    bool dead_clean = false;   // `dead` describes the labels (false after a clear: the next flags kernel rewrites it)
    int64_t final_voxels = 2;  // voxels per lane in the final survivor stage (1 or 2)
    int64_t stage1_voxels = 2; // ... in the stages before it
    int64_t fill_blocks = 512; // persistent store blocks of a list stage (0: one short block per strip)
    int64_t pack_ride = 1;     // a device batch is packed at flush, in view order: the first views ahead of
                               // the flags kernel, the others beside the dense stage (0: all ahead)
    int64_t brick_walkers = 1024;  // persistent blocks of the dense stage when packing rides with it
    int8_t *narrow = nullptr;  // scratch of sc_get_values_i8
    uint32_t *packed_labels = nullptr;  // sc_values_packed: the labels at 2 or 1 bits each
    int64_t unit_cull = 1;     // the dense stage asks the views packed ahead about every live brick's units (0: not;
                               // 2: even when the tiles settled less than half of the bricks)
    uint32_t *late = nullptr;  // FULL candidates a later view rejected (count in ctl->nlate)
    uint32_t *bulk = nullptr;  // units (a wavefront's share of a live brick) finished as a whole (counts in ctl->count[3])
    uint32_t bulkcap = 0;      // ... per sub-list
    int64_t bulk_min = 128;    // voxels of a unit (of 256) alive after the dense views for it to go there (0: never)
    bool last_bulk = false;    // the last fused launch had a bulk list
    uint4 *items = nullptr;    // the bulk units' work items (counts in ctl->count[4])
    uint32_t itemcap = 0;      // ... per sub-list
    int64_t item_bias = 12;    // sixteenths: items are chosen over the lists when they cost at most this share
    int64_t unit_blocks = 512; // blocks of 8 wavefronts walking the bulk list behind the confirm kernel
    // Whether the bulk list pays is a property of the scene (a bulky object or a grid wider than the pictures:
    // yes; a thin plant: the verdicts settle next to nothing and cost ~10 us), so the engine looks at what the
    // verdicts of its last batches spared the survivor stages and leaves the bulk list out for a while when
    // that was less than they cost.  Results never depend on it.
    int64_t bulk_adapt = 1;
    volatile unsigned long long *report = nullptr;  // page-locked: see ReportJob
    uint32_t report_seq = 0, report_seen = 0;
    int bulk_hold = 0;         // batches still to run without the bulk list
    uint32_t *unit_stats = nullptr;  // [unit blocks][2], see UnitJob
    size_t unit_stats_cap = 0;
    uint32_t *fill_list = nullptr;  // launches without survivor stages: settled bricks to fill (count in ctl->nfill)
    uint64_t flag_launches = 0;     // parity of the counters a flags kernel uses (see ListCtl)
    uint32_t last_parity = 0;
    struct DeferredBatch {     // sc_process_views_device batch whose packing waits for the flush
        bool on = false;
        const void *raw = nullptr;
        int V = 0, H = 0, W = 0, dtype = 0;
        int64_t row_stride = 0, view_stride = 0;
    } deferred;
    int64_t flag_views = 8;    // views that may veto a brick (0 = all of the batch)
    float *lut_dev = nullptr;  // averaging: 256-entry byte -> float32 table (SC_MASK_U8_LUT)

    // survivor lists of the fused carve
    uint32_t *lists = nullptr;  // 2 x (kSub * subcap) entries
    ListCtl *ctl = nullptr;
    uint32_t subcap = 0;

    // host-mask staging ring
    void *pin[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    void *raw[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t slot_ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool slot_armed[kSlots] = {false, false, false, false};
    size_t slot_bytes = 0;
    int next_slot = 0;

    // options
    int64_t views_per_launch = 0;
    int64_t view_order = 1;
    int64_t time_kernels = 0;
    int64_t max_pending = 256;
    int64_t compact = 1;
    int64_t brick = 1;
    int64_t dense_views = 2;     // views applied to every voxel before compaction
    int64_t stage1_views = 8;    // views applied to the first survivor list
    int64_t stage2_views = 0;    // views applied to the second survivor list (0: no such stage)
    int64_t list_blocks = 2048;  // persistent grid of the list / resume kernels
    int64_t view_group = 2;      // the spans of the final list stage are a multiple of this many views

    std::vector<TimedLaunch> timed[kNumKernels];
    hipEvent_t step_start = nullptr;
    bool step_open = false;
    hipEvent_t span_start = nullptr;  // sc_span_begin .. sc_span_end
    bool span_open = false;
    std::vector<hipEvent_t> event_pool;
};

namespace {

int use_device(sc_engine *e) {
    HIP_TRY(hipSetDevice(e->device));
    return SC_OK;
}

int get_event(sc_engine *e, hipEvent_t *ev) {
    if (!e->event_pool.empty()) {
        *ev = e->event_pool.back();
        e->event_pool.pop_back();
        return SC_OK;
    }
    HIP_TRY(hipEventCreate(ev));
    return SC_OK;
}

// SC_KERNEL_STEP: one event pair around everything a fused batch puts on the stream, from the
// packing of its masks to its last kernel.
int step_begin(sc_engine *e) {
    if (!e->time_kernels || e->step_open || e->views_per_launch == 1) return SC_OK;
    int rc = get_event(e, &e->step_start);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e->step_start, e->stream));
    e->step_open = true;
    return SC_OK;
}

int step_end(sc_engine *e, bool fused) {
    if (!e->step_open) return SC_OK;
    e->step_open = false;
    if (!fused) {
        e->event_pool.push_back(e->step_start);
        return SC_OK;
    }
    TimedLaunch tl{};
    tl.start = e->step_start;
    int rc = get_event(e, &tl.stop);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(tl.stop, e->stream));
    e->timed[SC_KERNEL_STEP].push_back(tl);
    return SC_OK;
}

struct LaunchTimer {
    sc_engine *e;
    int kid;
    TimedLaunch tl{};
    bool on = false;
    int begin() {
        if (!e->time_kernels) return SC_OK;
        if (e->time_kernels == 2 && kid != SC_KERNEL_CARVE && kid != SC_KERNEL_AVERAGE) return SC_OK;
        if (e->time_kernels == 2 && kid == SC_KERNEL_CARVE && e->step_open) return SC_OK;  // SC_KERNEL_STEP covers it
        int rc = get_event(e, &tl.start);
        if (rc) return rc;
        rc = get_event(e, &tl.stop);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(tl.start, e->stream));
        on = true;
        return SC_OK;
    }
    int end() {
        if (!on) return SC_OK;
        HIP_TRY(hipEventRecord(tl.stop, e->stream));
        e->timed[kid].push_back(tl);
        return SC_OK;
    }
};

GridDesc grid_desc(const sc_engine *e) {
    GridDesc g;
    g.ox = e->origin[0];
    g.oy = e->origin[1];
    g.oz = e->origin[2];
    g.vs = e->vs;
    g.ny = (uint32_t)e->ny;
    g.nz = (uint32_t)e->nz;
    g.i0 = (uint32_t)e->i0;
    g.istride = (uint32_t)e->istride;
    g.nzp = (uint32_t)e->nzp;
    g.gpc = (uint32_t)(e->nzp / 4);
    g.ngroups = (uint64_t)e->planes * (uint64_t)e->ny * g.gpc;
    return g;
}

int32_t init_bits_i32(const sc_engine *e) { return (int32_t)e->default_value; }

// The state as planes * ny * nz contiguous elements on the device: the state itself when its rows are not
// padded, else a copy without the padding (made on the engine's stream, valid until the state changes).
int dense_state(sc_engine *e, void **ptr) {
    if (e->nzp == e->nz) {
        *ptr = e->state;
        return SC_OK;
    }
    if (!e->dense) HIP_TRY(hipMalloc(&e->dense, (size_t)e->n * 4));
    const uint64_t rows = (uint64_t)e->planes * (uint64_t)e->ny;
    hipLaunchKernelGGL(depitch_kernel<uint32_t>, dim3((uint32_t)std::min<uint64_t>((rows + 3) / 4, 65536)), dim3(kBlock), 0,
                       e->stream, static_cast<const uint32_t *>(e->state), static_cast<uint32_t *>(e->dense), rows,
                       (uint32_t)e->nz, (uint32_t)e->nzp);
    HIP_TRY(hipGetLastError());
    *ptr = e->dense;
    return SC_OK;
}

int materialize(sc_engine *e) {
    if (!e->fresh) return SC_OK;
    uint32_t bits;
    if (e->mode == SC_MODE_CARVE) {
        int32_t v = init_bits_i32(e);
        memcpy(&bits, &v, 4);
    } else {
        memcpy(&bits, &e->default_value, 4);
    }
    uint64_t n = (uint64_t)e->npitch;  // padding included
    uint64_t blocks = (n + (uint64_t)kBlock * 4 - 1) / ((uint64_t)kBlock * 4);
    LaunchTimer lt{e, SC_KERNEL_FILL};
    int rc = lt.begin();
    if (rc) return rc;
    hipLaunchKernelGGL(fill_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, e->stream,
                       static_cast<uint32_t *>(e->state), n, bits);
    HIP_TRY(hipGetLastError());
    rc = lt.end();
    if (rc) return rc;
    e->fresh = false;
    return SC_OK;
}

// device storage for one pending view's mask, alive until the flush that consumes it
int arena_alloc(sc_engine *e, size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto &c : e->chunks) {
        if (c.cap - c.used >= bytes) {
            *out = c.base + c.used;
            c.used += bytes;
            return SC_OK;
        }
    }
    Chunk c;
    size_t last = e->chunks.empty() ? 0 : e->chunks.back().cap;
    c.cap = std::max(bytes, std::max<size_t>(last * 2, (size_t)16 << 20));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.base), c.cap));
    c.used = bytes;
    e->chunks.push_back(c);
    *out = c.base;
    return SC_OK;
}

void arena_reset(sc_engine *e) {
    // stream order protects reuse: later pack kernels / copies run after the launch that
    // read the old contents
    for (auto &c : e->chunks) c.used = 0;
}

int ensure_slots(sc_engine *e, size_t bytes) {
    if (bytes <= e->slot_bytes) return SC_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int s = 0; s < kSlots; ++s) {
        if (e->pin[s]) (void)hipHostFree(e->pin[s]);
        if (e->raw[s]) (void)hipFree(e->raw[s]);
        e->pin[s] = e->raw[s] = nullptr;
        e->slot_armed[s] = false;
    }
    e->slot_bytes = 0;
    for (int s = 0; s < kSlots; ++s) {
        HIP_TRY(hipHostMalloc(&e->pin[s], bytes, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&e->raw[s], bytes));
        if (!e->slot_ev[s]) HIP_TRY(hipEventCreateWithFlags(&e->slot_ev[s], hipEventDisableTiming));
    }
    e->slot_bytes = bytes;
    return SC_OK;
}

size_t elem_size(int dtype) {
    return (dtype == SC_MASK_U8 || dtype == SC_MASK_U8_INV || dtype == SC_MASK_BOOL_INV ||
            dtype == SC_MASK_U8_LUT) ? 1 : 4;
}

int check_dtype(const sc_engine *e, int dtype) {
    if (e->mode == SC_MODE_CARVE && (dtype == SC_MASK_U8 || dtype == SC_MASK_I32 ||
                                     dtype == SC_MASK_U8_INV || dtype == SC_MASK_BOOL_INV))
        return SC_OK;
    if (e->mode == SC_MODE_AVERAGE && dtype == SC_MASK_F32) return SC_OK;
    if (e->mode == SC_MODE_AVERAGE && dtype == SC_MASK_U8_LUT) {
        if (!e->lut_dev) return fail(SC_ERR_STATE, "SC_MASK_U8_LUT needs sc_set_lut first");
        return SC_OK;
    }
    return fail(SC_ERR_INVALID, "mask dtype %d does not fit engine mode %d", dtype, e->mode);
}

// Sufficient (not necessary) conditions, in double precision with room to spare, for what project()
// takes for granted of a view with `safe` set: over the voxel centres  o + i * vs,  ilo <= i <= ihi per axis,  2^-10 < pz  and  |px|, |py|, pz < 2^30;  K finite and below 2^30 in magnitude.  M_r bounds the
// magnitude of every partial sum of row r, so the float evaluation (six roundings, coordinates rounded
// twice) is within 2^-20 M_r of the real value; the margins below are 2^-18 M_r and factors of 2^10.
int32_t certify_view(const float *K, const float *R, const float *t, const float *o, float vs, const int64_t *ilo,
                     const int64_t *ihi) {  // voxel indices ilo[a] .. ihi[a] along axis a
    double lo[3], hi[3], amax[3];
    for (int a = 0; a < 3; ++a) {
        const double a0 = (double)o[a] + (double)ilo[a] * (double)vs, a1 = (double)o[a] + (double)ihi[a] * (double)vs;
        if (!std::isfinite(a0) || !std::isfinite(a1)) return 0;
        lo[a] = std::min(a0, a1);
        hi[a] = std::max(a0, a1);
        amax[a] = std::max(std::fabs(a0), std::fabs(a1)) * (1.0 + 0x1p-20) + 0x1p-100;
    }
    double M[3];
    for (int r = 0; r < 3; ++r) {
        M[r] = std::fabs((double)R[3 * r]) * amax[0] + std::fabs((double)R[3 * r + 1]) * amax[1] +
               std::fabs((double)R[3 * r + 2]) * amax[2] + std::fabs((double)t[r]);
        if (!(M[r] < 0x1p30)) return 0;  // also NaN
    }
    double pzmin = (double)t[2];
    for (int a = 0; a < 3; ++a) pzmin += std::min((double)R[6 + a] * lo[a], (double)R[6 + a] * hi[a]);
    if (!(pzmin - M[2] * 0x1p-18 > 0x1p-10)) return 0;
    for (int q = 0; q < 4; ++q)
        if (!(std::fabs((double)K[q]) < 0x1p30)) return 0;
    return 1;
}

void fill_desc(const sc_engine *e, ViewDesc &d, const float *K, const float *R, const float *t, const void *mask,
               int H, int W, const uint8_t *occ = nullptr) {
    memcpy(d.K, K, sizeof d.K);
    memcpy(d.R, R, sizeof d.R);
    memcpy(d.t, t, sizeof d.t);
    d.mask = mask;
    d.W = W;
    d.H = H;
    d.tiles_x = (W + kTile - 1) / kTile;
    d.pad = 0;
    d.occ = occ;
    d.Wf = (float)W;
    d.Hf = (float)H;
    const int64_t first[3] = {0, 0, 0}, last[3] = {e->nx - 1, e->ny - 1, e->nz - 1};  // the global grid: any partition of it is inside
    d.safe = certify_view(K, R, t, e->origin, e->vs, first, last);
    d.pad2 = 0;
    d.cmask = nullptr;
    d.reserved = 0;
}

size_t packed_words(int H, int W) {
    size_t tx = (size_t)(W + kTile - 1) / kTile, ty = (size_t)(H + kTile - 1) / kTile;
    return tx * ty * 32;
}

PackJob make_pack_job(const void *raw_dev, int64_t row_stride, int64_t view_stride, int W, int H,
                      uint32_t *packed, int64_t words, uint32_t flip, uint8_t *occ, uint32_t *cmask) {
    PackJob pj;
    memset(&pj, 0, sizeof pj);
    pj.raw = static_cast<const uint8_t *>(raw_dev);
    pj.row_stride = row_stride;
    pj.view_stride = view_stride;
    pj.W = W;
    pj.H = H;
    pj.tiles_x = (W + kTile - 1) / kTile;
    pj.tiles_y = (H + kTile - 1) / kTile;
    pj.out = packed;
    pj.out_view_words = words;
    pj.flip = flip;
    pj.occ = occ;
    pj.cmask = cmask;
    return pj;
}

// 0: the band form (pictures up to kBandTiles tiles wide, packed arena 16-byte aligned per band); else the panel
// form with that many tile rows per block
int pack_form(const sc_engine *e, const PackJob &pj) {
    // (narrow pictures make bands of a few hundred tasks, less than a block's worth: 128-pixel pictures took 61 us
    // in bands against 24 in panels)
    if (e->pack_rows == 0 && pj.tiles_x >= 16 && pj.tiles_x <= kBandTiles) return 0;
    return e->pack_rows == 0 ? 4 : (int)e->pack_rows;
}

int64_t pack16_blocks(const sc_engine *e, const PackJob &pj) {
    const int rows = pack_form(e, pj);
    if (rows == 0) return (int64_t)pj.nslots * pj.tiles_y;
    return (int64_t)pj.nslots * ((pj.tiles_y + rows - 1) / rows) * ((pj.tiles_x + 3) / 4);
}

// slots [pj.slot0, pj.slot0 + pj.nslots) as a launch of their own
int launch_pack16(sc_engine *e, const PackJob &pj) {
    if (pj.nslots <= 0) return SC_OK;
    const int rows = pack_form(e, pj);
    int64_t blocks = pack16_blocks(e, pj);
    if (blocks > 0x7fffffffLL) return fail(SC_ERR_INVALID, "mask batch too large");
#define LAUNCH_PACK16(ROWS) \
    hipLaunchKernelGGL(pack16_kernel<ROWS>, dim3((uint32_t)blocks), dim3(kBlock), 0, e->stream, pj)
    if (rows == 0) hipLaunchKernelGGL(pack_band_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, e->stream, pj);
    else if (rows == 1) LAUNCH_PACK16(1);
    else if (rows == 2) LAUNCH_PACK16(2);
    else if (rows == 8) LAUNCH_PACK16(8);
    else LAUNCH_PACK16(4);
#undef LAUNCH_PACK16
    HIP_TRY(hipGetLastError());
    return SC_OK;
}

bool pack16_eligible(const void *raw_dev, int W, int dtype, int64_t row_stride, int64_t view_stride) {
    return dtype != SC_MASK_I32 && (W % 16) == 0 && (row_stride % 16) == 0 && (view_stride % 16) == 0 &&
           (reinterpret_cast<uintptr_t>(raw_dev) % 16) == 0;
}

uint32_t pack_flip(int dtype) {
    return dtype == SC_MASK_U8_INV ? 0xffffffffu : dtype == SC_MASK_BOOL_INV ? 0x01010101u : 0u;
}

// raw device pixels [V][H][W] -> packed tiles in the arena; appends V pending views
int enqueue_pack(sc_engine *e, int V, const float *K, const float *R, const float *t,
                 const void *raw_dev, int H, int W, int dtype, int64_t row_stride,
                 int64_t view_stride) {
    size_t words = packed_words(H, W);
    void *packed = nullptr;
    int rc = arena_alloc(e, words * 4 * (size_t)V, &packed);
    if (rc) return rc;
    int tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile;
    size_t occ_bytes = (size_t)tiles_x * tiles_y;
    void *occ_v = nullptr;
    rc = arena_alloc(e, occ_bytes * (size_t)V, &occ_v);
    if (rc) return rc;
    uint8_t *occ = static_cast<uint8_t *>(occ_v);
    rc = step_begin(e);
    if (rc) return rc;
    LaunchTimer lt{e, SC_KERNEL_PACK};
    bool bytes = dtype != SC_MASK_I32;
    uint32_t flip = pack_flip(dtype);
    bool fast = pack16_eligible(raw_dev, W, dtype, row_stride, view_stride);
    uint32_t *cmask = nullptr;
    if (fast && (e->bulk_min > 0 || e->unit_cull)) {  // the cell level behind the units' verdicts: one word per tile
        void *cv = nullptr;
        rc = arena_alloc(e, occ_bytes * 4 * (size_t)V, &cv);
        if (rc) return rc;
        cmask = static_cast<uint32_t *>(cv);
    }
    if (fast) {
        PackJob pj = make_pack_job(raw_dev, row_stride, view_stride, W, H, static_cast<uint32_t *>(packed),
                                   (int64_t)words, flip, occ, cmask);
        pj.slot0 = 0;
        pj.nslots = V;
        rc = lt.begin();
        if (rc) return rc;
        rc = launch_pack16(e, pj);
        if (rc) return rc;
    } else {
        // the slow forms only ever set occupancy bytes
        HIP_TRY(hipMemsetAsync(occ, 0, occ_bytes * (size_t)V, e->stream));
        int segs = (W + 63) / 64;
        int64_t waves = (int64_t)V * H * segs;
        int64_t blocks = (waves + (kBlock / 64) - 1) / (kBlock / 64);
        if (blocks > 0x7fffffffLL) return fail(SC_ERR_INVALID, "mask batch too large");
        rc = lt.begin();
        if (rc) return rc;
        if (bytes) {
            // background byte: 0, or 255 / 1 when the mask is to be inverted first
            uint8_t bg = dtype == SC_MASK_U8_INV ? 255 : dtype == SC_MASK_BOOL_INV ? 1 : 0;
            hipLaunchKernelGGL(pack_kernel<uint8_t>, dim3((uint32_t)blocks), dim3(kBlock), 0,
                               e->stream, static_cast<const uint8_t *>(raw_dev), row_stride,
                               view_stride, W, H, V, tiles_x, static_cast<uint32_t *>(packed),
                               (int64_t)words, bg, occ, tiles_y);
        } else {
            hipLaunchKernelGGL(pack_kernel<int32_t>, dim3((uint32_t)blocks), dim3(kBlock), 0,
                               e->stream, static_cast<const int32_t *>(raw_dev), row_stride,
                               view_stride, W, H, V, tiles_x, static_cast<uint32_t *>(packed),
                               (int64_t)words, (int32_t)0, occ, tiles_y);
        }
    }
    HIP_TRY(hipGetLastError());
    rc = lt.end();
    if (rc) return rc;
    for (int q = 0; q < V; ++q) {
        ViewDesc d;
        fill_desc(e, d, K + 4 * q, R + 9 * q, t + 3 * q,
                  static_cast<uint32_t *>(packed) + (size_t)q * words, H, W, occ + (size_t)q * occ_bytes);
        if (cmask) d.cmask = cmask + (size_t)q * occ_bytes;
        e->pending.push_back(d);
    }
    return SC_OK;
}

// averaging, uint8 + table form: raw device bytes [V][H][W] -> 16x8 tiles; appends V pending views
int enqueue_tile8(sc_engine *e, int V, const float *K, const float *R, const float *t,
                  const void *raw_dev, int H, int W, int64_t row_stride, int64_t view_stride) {
    int tiles_x = (W + kATileW - 1) / kATileW, tiles_y = (H + kATileH - 1) / kATileH;
    size_t per_view = (size_t)tiles_x * tiles_y * 128;
    void *tiled = nullptr;
    int rc = arena_alloc(e, per_view * (size_t)V, &tiled);
    if (rc) return rc;
    int fast = (W % 16) == 0 && (row_stride % 16) == 0 && (view_stride % 16) == 0 &&
               (reinterpret_cast<uintptr_t>(raw_dev) % 16) == 0;
    int64_t total = (int64_t)V * H * ((W + 15) / 16);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return fail(SC_ERR_INVALID, "mask batch too large");
    // per 32x32-pixel tile: is it all 0 / all 255?  (brick form of the averaging kernel)
    const size_t uni_per_view = (size_t)((W + 31) / 32) * (size_t)((H + 31) / 32);
    uint8_t *uni = nullptr;
    if (fast && e->avg_brick) {
        void *u = nullptr;
        size_t bytes = (uni_per_view * (size_t)V + 3) & ~(size_t)3;
        rc = arena_alloc(e, bytes, &u);
        if (rc) return rc;
        uni = static_cast<uint8_t *>(u);
    }
    LaunchTimer lt{e, SC_KERNEL_PACK};
    rc = lt.begin();
    if (rc) return rc;
    hipLaunchKernelGGL(tile8_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, e->stream,
                       static_cast<const uint8_t *>(raw_dev), row_stride, view_stride, W, H, V, tiles_x,
                       tiles_y, static_cast<uint8_t *>(tiled), fast);
    if (uni) {
        int64_t ntiles = (int64_t)V * (int64_t)uni_per_view;
        hipLaunchKernelGGL(uniform_tiles_kernel, dim3((uint32_t)((ntiles + 3) / 4)), dim3(kBlock), 0, e->stream,
                           static_cast<const uint8_t *>(tiled), W, H, V, tiles_x, tiles_y, uni);
    }
    HIP_TRY(hipGetLastError());
    rc = lt.end();
    if (rc) return rc;
    for (int q = 0; q < V; ++q) {
        ViewDesc d;
        fill_desc(e, d, K + 4 * q, R + 9 * q, t + 3 * q, static_cast<uint8_t *>(tiled) + (size_t)q * per_view, H, W,
                  uni ? uni + (size_t)q * uni_per_view : nullptr);
        d.tiles_x = tiles_x;
        d.pad = 1;
        e->pending.push_back(d);
    }
    return SC_OK;
}

// averaging, float32 masks: raw device floats [V][H][W] -> 8x4 tiles + per-region uniformity; appends V
// pending views (ViewDesc::pad == 2)
int enqueue_tilef32(sc_engine *e, int V, const float *K, const float *R, const float *t, const void *raw_dev,
                    int H, int W, int64_t row_stride, int64_t view_stride) {
    const int tiles_x = (W + kFTileW - 1) / kFTileW, tiles_y = (H + kFTileH - 1) / kFTileH;
    const size_t per_view = (size_t)tiles_x * tiles_y * 128;
    void *tiled = nullptr;
    int rc = arena_alloc(e, per_view * (size_t)V, &tiled);
    if (rc) return rc;
    const int fast = (W % 4) == 0 && (row_stride % 16) == 0 && (view_stride % 16) == 0 &&
                     (reinterpret_cast<uintptr_t>(raw_dev) % 16) == 0;
    const int64_t total = (int64_t)V * H * ((W + 3) / 4);
    const int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return fail(SC_ERR_INVALID, "mask batch too large");
    const size_t nreg = (size_t)((W + 31) / 32) * (size_t)((H + 31) / 32);
    const size_t uni_view = ((nreg + 3) & ~(size_t)3) + nreg * 4;  // flags, then the regions' values
    void *u = nullptr;
    rc = arena_alloc(e, uni_view * (size_t)V, &u);
    if (rc) return rc;
    uint8_t *uni = static_cast<uint8_t *>(u);
    LaunchTimer lt{e, SC_KERNEL_PACK};
    rc = lt.begin();
    if (rc) return rc;
    hipLaunchKernelGGL(tilef_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, e->stream, static_cast<const float *>(raw_dev),
                       row_stride, view_stride, W, H, V, tiles_x, tiles_y, static_cast<float *>(tiled), fast);
    const int64_t regs = (int64_t)V * (int64_t)nreg;
    hipLaunchKernelGGL(uniform_f32_kernel, dim3((uint32_t)((regs + 3) / 4)), dim3(kBlock), 0, e->stream,
                       static_cast<const float *>(tiled), W, H, V, tiles_x, tiles_y, uni, uni_view);
    HIP_TRY(hipGetLastError());
    rc = lt.end();
    if (rc) return rc;
    for (int q = 0; q < V; ++q) {
        ViewDesc d;
        fill_desc(e, d, K + 4 * q, R + 9 * q, t + 3 * q, static_cast<char *>(tiled) + (size_t)q * per_view, H, W,
                  uni + (size_t)q * uni_view);
        d.tiles_x = tiles_x;
        d.pad = 2;
        e->pending.push_back(d);
    }
    return SC_OK;
}

// Order of the views inside a fused carve launch: greedily pick the view whose optical axis
// (third row of R) is most perpendicular to every axis already chosen (|cos| ignores the
// sign: opposite cameras see mirrored silhouettes).  Perpendicular silhouettes intersect in the
// smallest volume, so almost everything is carved by the first two views.  Legal because the
// carve state is order-independent; `average` never re-orders.
void order_views(std::vector<ViewDesc> &v, std::vector<uint32_t> *perm = nullptr) {
    size_t n = v.size();
    if (perm) {
        perm->resize(n);
        for (size_t q = 0; q < n; ++q) (*perm)[q] = (uint32_t)q;
    }
    if (n < 3 || n > 4096) return;
    std::vector<float> worst(n, 0.0f);
    std::vector<char> used(n, 0);
    std::vector<ViewDesc> out;
    out.reserve(n);
    size_t cur = 0;
    for (size_t step = 0; step < n; ++step) {
        used[cur] = 1;
        out.push_back(v[cur]);
        if (perm) (*perm)[step] = (uint32_t)cur;
        const float *a = v[cur].R + 6;
        size_t best = n;
        for (size_t q = 0; q < n; ++q) {
            if (used[q]) continue;
            const float *b = v[q].R + 6;
            float c = std::fabs(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
            if (c > worst[q]) worst[q] = c;
            if (best == n || worst[q] < worst[best]) best = q;
        }
        cur = best;
    }
    v.swap(out);
}

constexpr int kMinFusedViews = 6;  // below this a fused launch stays dense

int ensure_lists(sc_engine *e) {
    if (e->lists) return SC_OK;
    // room for 5/16 of the voxels: two views of coin-flip masks leave a quarter alive, which the hashed
    // sub-lists must hold with a margin for their unevenness (an overflow sends the batch down the dense
    // resume path, 10 x slower)
    uint64_t total = std::max<uint64_t>((uint64_t)e->n / 4 + (uint64_t)e->n / 16, (uint64_t)kSub * 1024);
    e->subcap = (uint32_t)((total + kSub - 1) / kSub);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->lists), (size_t)2 * kSub * e->subcap * sizeof(uint32_t)));
    return SC_OK;
}

// List counters, brick verdicts and the live-brick list share one allocation; one memset of the
// counters per fused launch.
int ensure_ctl(sc_engine *e) {
    if (e->ctl) return SC_OK;
    size_t nbricks = 0;
    if ((e->nz + kBrickZ - 1) / kBrickZ <= 64)
        nbricks = (size_t)e->planes * (size_t)((e->ny + kBrickY - 1) / kBrickY) * (size_t)((e->nz + kBrickZ - 1) / kBrickZ);
    char *base = nullptr;
    size_t flag_bytes = (nbricks + 15) & ~(size_t)15;
    // bulk units: four per brick, hashed over the sub-lists; twice the even share each (a full one sends its
    // units' voxels down the ordinary lists)
    e->bulkcap = (uint32_t)((nbricks * 4 * 2 + kSub - 1) / kSub + 64);
    const size_t bulk_words = nbricks ? (size_t)kSub * e->bulkcap : 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&base),
                      2 * sizeof(ListCtl) + flag_bytes + (3 * nbricks + bulk_words) * sizeof(uint32_t) + 16));
    HIP_TRY(hipMemsetAsync(base, 0, 2 * sizeof(ListCtl), e->stream));
    e->ctl2[0] = reinterpret_cast<ListCtl *>(base);
    e->ctl2[1] = e->ctl2[0] + 1;
    e->ctl_clean[0] = e->ctl_clean[1] = true;
    e->ctl_idx = 0;
    e->ctl = e->ctl2[0];
    e->flags = reinterpret_cast<uint8_t *>(base + 2 * sizeof(ListCtl));
    e->live = reinterpret_cast<uint32_t *>(base + 2 * sizeof(ListCtl) + flag_bytes);
    e->late = e->live + nbricks;
    e->fill_list = e->late + nbricks;
    e->bulk = bulk_words ? e->fill_list + nbricks : nullptr;
    if (bulk_words) {
        // up to 2 halves x 2 words x 4 pieces per unit; room for a third of that on average (a full sub-list
        // sends the unit's voxels down the ordinary lists)
        e->itemcap = e->bulkcap * 5u;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->items), (size_t)kSub * e->itemcap * sizeof(uint4)));
    }
    return SC_OK;
}

// Arena storage for the deferred device batch: slot q of the packed tiles / occupancy bytes goes to
// pending view q.  Returns the job that packs it (no slots chosen yet).
int deferred_job(sc_engine *e, PackJob *out) {
    const auto &db = e->deferred;
    size_t words = packed_words(db.H, db.W);
    void *packed = nullptr, *occ_v = nullptr;
    int rc = arena_alloc(e, words * 4 * (size_t)db.V, &packed);
    if (rc) return rc;
    const size_t occ_bytes = (size_t)((db.W + kTile - 1) / kTile) * (size_t)((db.H + kTile - 1) / kTile);
    rc = arena_alloc(e, occ_bytes * (size_t)db.V, &occ_v);
    if (rc) return rc;
    uint32_t *cmask = nullptr;
    if (e->bulk_min > 0 || e->unit_cull) {
        void *cv = nullptr;
        rc = arena_alloc(e, occ_bytes * 4 * (size_t)db.V, &cv);
        if (rc) return rc;
        cmask = static_cast<uint32_t *>(cv);
    }
    for (int q = 0; q < db.V; ++q) {
        e->pending[(size_t)q].mask = static_cast<uint32_t *>(packed) + (size_t)q * words;
        e->pending[(size_t)q].occ = static_cast<uint8_t *>(occ_v) + (size_t)q * occ_bytes;
        if (cmask) e->pending[(size_t)q].cmask = cmask + (size_t)q * occ_bytes;
    }
    *out = make_pack_job(db.raw, db.row_stride, db.view_stride, db.W, db.H, static_cast<uint32_t *>(packed),
                         (int64_t)words, pack_flip(db.dtype), static_cast<uint8_t *>(occ_v), cmask);
    return SC_OK;
}

// Pack the deferred batch now, in the order given, all of it ahead of any carve kernel.
int materialize_deferred(sc_engine *e) {
    if (!e->deferred.on) return SC_OK;
    PackJob pj;
    int rc = deferred_job(e, &pj);
    if (rc) return rc;
    e->deferred.on = false;
    pj.slot0 = 0;
    pj.nslots = e->deferred.V;
    rc = step_begin(e);
    if (rc) return rc;
    LaunchTimer lt{e, SC_KERNEL_PACK};
    rc = lt.begin();
    if (rc) return rc;
    rc = launch_pack16(e, pj);
    if (rc) return rc;
    return lt.end();
}

// What a fused carve of `nv` views will look like (see flush): decided before anything is launched,
// because a deferred batch is packed according to it.
struct FusedPlan {
    int ndense, nstage1, s1, flag_views;
    bool compact, brick, defer_stores;
    uint32_t bys, bzs, nbricks, nstrips, dense_store_strips;
};

FusedPlan fused_plan(const sc_engine *e, size_t nv, bool has_occ) {
    FusedPlan p{};
    p.ndense = (int)e->dense_views;
    p.nstage1 = (int)e->stage1_views;
    p.compact = e->compact && nv >= (size_t)kMinFusedViews && nv > (size_t)p.ndense &&
                (uint64_t)e->npitch < 0x80000000ull;
    p.bys = (uint32_t)((e->ny + kBrickY - 1) / kBrickY);
    p.bzs = (uint32_t)((e->nz + kBrickZ - 1) / kBrickZ);
    p.brick = (nv > 1 || e->view_brick) && e->brick && p.bzs <= 64 && (uint64_t)e->npitch < 0x80000000ull &&
              (uint64_t)e->planes * p.bys * p.bzs < 0x40000000ull && has_occ;  // brick ids carry two flag bits in the fill list
    p.nbricks = p.brick ? (uint32_t)((uint64_t)e->planes * p.bys * p.bzs) : 0u;
    p.flag_views = (int)nv;  // every view of the batch may veto a brick, not only the dense stage's
    if (e->flag_views > 0 && e->flag_views < (int64_t)p.flag_views) p.flag_views = (int)e->flag_views;
    p.s1 = (int)std::min<size_t>(nv, (size_t)p.ndense + (size_t)p.nstage1);
    // the -1 fill of empty bricks rides along with the list stages when there are any: strips
    // [0, dense_store_strips) are filled by the dense kernel's store blocks, the others by the list
    // stages' (defer_share sixteenths of them)
    p.nstrips = p.brick ? (uint32_t)((uint64_t)e->planes * p.bys) : 0u;
    p.dense_store_strips = p.nstrips;
    if (p.brick && p.compact && e->defer_stores > 0 && e->defer_share > 0)
        p.dense_store_strips = (uint32_t)((uint64_t)p.nstrips * (uint64_t)(16 - e->defer_share) / 16u);
    p.defer_stores = p.dense_store_strips < p.nstrips;
    return p;
}

// Launch the first `count` pending views (count == 0: all of them).
int flush(sc_engine *e, size_t count = 0) {
    if (e->pending.empty()) return SC_OK;
    size_t nv = count ? std::min(count, e->pending.size()) : e->pending.size();
    // A device batch whose packing was deferred is packed here, in the order its views will be
    // applied: the views the flags kernel, the dense stage and the first survivor stage need go
    // ahead, the others ride beside the dense stage (brick form).  Any other shape of launch packs
    // the whole batch first, in the order given.
    bool ordered = false;
    PackJob ride;
    memset(&ride, 0, sizeof ride);
    uint32_t ride_blocks = 0;
    int packed_ahead = (int)nv;
    if (e->deferred.on) {
        const bool whole = nv == e->pending.size() && nv == (size_t)e->deferred.V && e->mode == SC_MODE_CARVE && nv > 1;
        if (!whole) {
            int rcd = materialize_deferred(e);
            if (rcd) return rcd;
        } else {
            int rcd = step_begin(e);
            if (rcd) return rcd;
            std::vector<uint32_t> perm;
            if (e->view_order == 1) order_views(e->pending, &perm);
            else { perm.resize(nv); for (size_t q = 0; q < nv; ++q) perm[q] = (uint32_t)q; }
            ordered = true;
            PackJob pj;
            rcd = deferred_job(e, &pj);
            if (rcd) return rcd;
            e->deferred.on = false;
            pj.use_order = 1;
            for (size_t q = 0; q < nv; ++q) pj.order[q] = (uint16_t)perm[q];
            const FusedPlan fp = fused_plan(e, nv, true);
            int ahead = (int)nv;
            if (e->pack_ride && fp.brick && fp.compact && fp.defer_stores && fp.dense_store_strips == 0)
                ahead = std::min<int>((int)nv, std::max(fp.flag_views, fp.s1));
            pj.slot0 = 0;
            pj.nslots = ahead;
            LaunchTimer ltp{e, SC_KERNEL_PACK};
            rcd = ltp.begin();
            if (rcd) return rcd;
            rcd = launch_pack16(e, pj);
            if (rcd) return rcd;
            rcd = ltp.end();
            if (rcd) return rcd;
            if (ahead < (int)nv) {
                ride = pj;
                ride.slot0 = ahead;
                ride.nslots = (int)nv - ahead;
                int64_t rb = pack16_blocks(e, ride);
                if (rb > 0x3fffffffLL) return fail(SC_ERR_INVALID, "mask batch too large");
                ride_blocks = (uint32_t)rb;
                packed_ahead = ahead;
            }
        }
    }
    GridDesc g = grid_desc(e);
    uint64_t blocks = (g.ngroups + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffULL) return fail(SC_ERR_INVALID, "grid too large for one launch");
    // (rows are whole 16-byte groups -- the pitch is a multiple of 64 voxels -- so every kernel takes its
    // vector form, VEC = true; the element-wise forms remain in the templates for a layout without padding)
    dim3 grid((uint32_t)blocks), block(kBlock);
    const ViewDesc *vd = nullptr, *vpin = nullptr;
    if (nv > 1) {
        int rcs = step_begin(e);
        if (rcs) return rcs;
    }
    // a single view in brick form goes through the same kernels as a batch: it needs its descriptor
    // in the device array too
    const bool single_brick = nv == 1 && e->mode == SC_MODE_CARVE &&
                              fused_plan(e, nv, e->pending[0].occ != nullptr).brick;
    if (nv > 1 || single_brick) {
        if (!ordered && e->mode == SC_MODE_CARVE && e->view_order == 1 && nv == e->pending.size())
            order_views(e->pending);
        // descriptor ring: slots are reused only after a wrap, which waits for the stream
        if (nv > e->views_cap || e->views_head + nv > e->views_cap) {
            HIP_TRY(hipStreamSynchronize(e->stream));
            e->views_head = 0;
        }
        if (nv > e->views_cap) {
            if (e->views_dev) (void)hipFree(e->views_dev);
            if (e->views_pin) (void)hipHostFree(e->views_pin);
            e->views_dev = e->views_pin = nullptr;
            e->views_cap = 0;
            size_t cap = std::max<size_t>(nv * 4, 1024);
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->views_dev), cap * sizeof(ViewDesc)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->views_pin), cap * sizeof(ViewDesc),
                                  hipHostMallocDefault));
            e->views_cap = cap;
        }
        ViewDesc *pin = e->views_pin + e->views_head, *dev = e->views_dev + e->views_head;
        memcpy(pin, e->pending.data(), nv * sizeof(ViewDesc));
        e->views_head += nv;
        vd = dev;
        vpin = pin;
    }
    const ViewDesc &one = e->pending[0];
    int rc;
    // the descriptors reach the device array either by a copy on the stream, or -- brick form of
    // the fused carve -- through the flags kernel, which gets its own in its arguments
    bool desc_uploaded = false;
    auto upload_desc = [&]() -> int {
        if (desc_uploaded || vd == nullptr) return SC_OK;
        desc_uploaded = true;
        HIP_TRY(hipMemcpyAsync(const_cast<ViewDesc *>(vd), vpin, nv * sizeof(ViewDesc), hipMemcpyHostToDevice, e->stream));
        return SC_OK;
    };
    if (e->mode == SC_MODE_CARVE) {
        int32_t *st = static_cast<int32_t *>(e->state);
        int32_t init = init_bits_i32(e);
        // fused carve with survivor compaction: dense for the first `ndense` views, then lists
        const FusedPlan fp = fused_plan(e, nv, one.occ != nullptr);
        const int ndense = fp.ndense, nstage1 = fp.nstage1, flag_views = fp.flag_views;
        const uint32_t list_blocks = (uint32_t)e->list_blocks;
        const bool compact = fp.compact, brick = fp.brick, defer_stores = fp.defer_stores;
        Append ap{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
        int dense_views = (int)nv;
        const uint32_t bys = fp.bys, bzs = fp.bzs, nbricks = fp.nbricks, nstrips = fp.nstrips;
        const uint32_t dense_store_strips = fp.dense_store_strips;
        const bool desc_by_flags = brick && flag_views <= kFlagWaves;
        if (!desc_by_flags) {
            rc = upload_desc();
            if (rc) return rc;
        }
        if (compact || brick) {
            rc = ensure_ctl(e);
            if (rc) return rc;
            // list counters, overflow flag, live-brick count: this batch takes the block the previous
            // batch's final stage left zeroed (a memset only if there was no such stage)
            if (compact) {
                e->ctl_idx ^= 1;
                e->ctl = e->ctl2[e->ctl_idx];
                if (!e->ctl_clean[e->ctl_idx]) HIP_TRY(hipMemsetAsync(e->ctl, 0, sizeof(ListCtl), e->stream));
            }
            // (a launch without survivor stages keeps the block: its two counters alternate, see ListCtl)
            e->ctl_clean[e->ctl_idx] = false;
        }
        const uint32_t parity = (uint32_t)(e->flag_launches & 1u);
        if (brick) ++e->flag_launches;
        if (compact) {
            rc = ensure_lists(e);
            if (rc) return rc;
            ap.list = e->lists;
            ap.ctl = e->ctl;
            ap.subcap = e->subcap;
            dense_views = ndense;
        }
        // bulk units: brick form with survivor stages, every view with its cell level
        bool bulk_on = compact && brick && e->bulk_min > 0 && e->bulk != nullptr;
        if (nv > 128) bulk_on = false;  // the units' verdict masks cover 128 views
        for (size_t q = 0; q < nv && bulk_on; ++q) bulk_on = e->pending[q].cmask != nullptr;
        if (bulk_on && e->bulk_adapt) {
            if (!e->report) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(const_cast<unsigned long long **>(&e->report)), 64, hipHostMallocDefault));
                e->report[0] = 0ull;
            }
            const unsigned long long rep = e->report[0];
            const uint32_t seq = (uint32_t)(rep >> 48);
            if (seq != e->report_seen) {  // a batch has reported since the last look
                e->report_seen = seq;
                const uint64_t units = (rep >> 24) & 0xffffffu, spared = (rep & 0xffffffu) << 4;
                // Measured on one MI355X: the verdicts take max(10 us, 3.3 ns per unit) and a turn spared is worth
                // 0.28 ns of the survivor stages (2.3 us per turn over 8192 wavefronts)
                // (no unit at all: the verdict kernel's launch and the dense stage's bookkeeping bought nothing)
                if (spared < std::max<uint64_t>(36000, units * 12)) e->bulk_hold = 64;
            }
            if (e->bulk_hold > 0) {
                --e->bulk_hold;
                bulk_on = false;
            }
        }
        if (bulk_on && (size_t)e->unit_blocks > e->unit_stats_cap) {
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (e->unit_stats) (void)hipFree(e->unit_stats);
            e->unit_stats = nullptr;
            e->unit_stats_cap = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->unit_stats), (size_t)e->unit_blocks * 2 * sizeof(uint32_t)));
            e->unit_stats_cap = (size_t)e->unit_blocks;
        }
        if (bulk_on) {
            ap.bulk = e->bulk;
            ap.bulkcap = e->bulkcap;
            ap.bulk_min = (uint32_t)e->bulk_min;
        }
        e->last_bulk = bulk_on;
        LaunchTimer lt{e, SC_KERNEL_CARVE};
        if (!brick) {  // the brick form starts the timer after its flags kernel
            rc = lt.begin();
            if (rc) return rc;
        }
        if (nv == 1 && !brick) {
            // kStreamGroups groups per lane when the state is streamed through (see kernel)
            uint32_t per_block = !e->fresh ? kBlock * kStreamGroups : kBlock;
            dim3 grid1((uint32_t)((g.ngroups + per_block - 1) / per_block));
#define LAUNCH_CARVE1(F, V) \
    hipLaunchKernelGGL((carve_kernel_1<F, V>), grid1, block, 0, e->stream, st, g, one, init)
            if (e->fresh) {
                LAUNCH_CARVE1(true, true);
            } else {
                LAUNCH_CARVE1(false, true);
            }
#undef LAUNCH_CARVE1
        } else {
            if (brick) {
                // live-list walkers (whole groups of 8 XCDs), then store blocks, then packing riders; with
                // riders the walkers leave wavefront slots free for them
                const uint32_t nwalkers = ((uint32_t)(ride_blocks ? e->brick_walkers : e->list_blocks) + 7u) & ~7u;
                dim3 bgrid(nwalkers + dense_store_strips + ride_blocks);
                if (!e->dead) {
                    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->dead), (size_t)nbricks));
                    e->dead_clean = false;
                }
                const int dead_stale = e->dead_clean ? 0 : 1;  // the flags kernel rewrites them all
                e->dead_clean = true;
                // unit verdicts (cell level) by the views packed ahead, inside the dense stage
                int nverd = 0;
                const uint32_t verd_max_live = e->unit_cull == 2 ? 0xffffffffu : (uint32_t)(nbricks / 2);
                if (compact && e->unit_cull) {
                    nverd = std::min(packed_ahead, 16);
                    for (int q = 0; q < nverd; ++q)
                        if (e->pending[(size_t)q].cmask == nullptr) nverd = 0;
                }
                LaunchTimer ltf{e, SC_KERNEL_FLAGS};
                rc = ltf.begin();
                if (rc) return rc;
                FlagViews own{};
                DescCopy dc{nullptr, nullptr, 0u};
                if (desc_by_flags) {
                    for (int q = 0; q < flag_views; ++q) own.v[q] = e->pending[(size_t)q];
                    dc = DescCopy{reinterpret_cast<const uint32_t *>(vpin),
                                  reinterpret_cast<uint32_t *>(const_cast<ViewDesc *>(vd)),
                                  (uint32_t)(nv * sizeof(ViewDesc) / 4)};
                    desc_uploaded = true;
                }
                hipLaunchKernelGGL(brick_flags_kernel, dim3((nbricks + 63u) / 64u), dim3(64 * kFlagWaves), 0,
                                   e->stream, g, desc_by_flags ? static_cast<const ViewDesc *>(nullptr) : vd,
                                   flag_views, bys, bzs, nbricks, e->flags, e->live, e->ctl, own, dc,
                                   desc_by_flags ? vpin : vd, e->full_bricks ? packed_ahead : 0, (int)nv, e->dead,
                                   dead_stale, parity, compact ? static_cast<uint32_t *>(nullptr) : e->fill_list);
                e->last_parity = parity;
                rc = ltf.end();
                if (rc) return rc;
                rc = lt.begin();  // SC_KERNEL_CARVE times the dense kernel alone
                if (rc) return rc;
                if (!compact) {
                    // no survivor stages: walkers on the live list, fillers on the fill list
                    const dim3 lgrid(nwalkers + (uint32_t)std::max<int64_t>(e->fill_blocks, 64));
                    if (e->fresh)
                        hipLaunchKernelGGL((carve_brick_light_kernel<true>), lgrid, block, 0, e->stream, st, g, vd,
                                           dense_views, init, bys, bzs, e->live, e->fill_list, e->ctl, nwalkers, parity);
                    else
                        hipLaunchKernelGGL((carve_brick_light_kernel<false>), lgrid, block, 0, e->stream, st, g, vd,
                                           dense_views, init, bys, bzs, e->live, e->fill_list, e->ctl, nwalkers, parity);
                } else if (e->fresh)
                    hipLaunchKernelGGL((carve_brick_kernel<true>), bgrid, block, 0, e->stream, st, g, vd,
                                       dense_views, init, ap, bys, bzs, e->flags, e->live, e->ctl, nwalkers,
                                       dense_store_strips, ride, pack_form(e, ride), parity, nverd, verd_max_live);
                else
                    hipLaunchKernelGGL((carve_brick_kernel<false>), bgrid, block, 0, e->stream, st, g, vd,
                                       dense_views, init, ap, bys, bzs, e->flags, e->live, e->ctl, nwalkers,
                                       dense_store_strips, ride, pack_form(e, ride), parity, nverd, verd_max_live);
            } else {
#define LAUNCH_CARVE(F, V)                                                                    \
    hipLaunchKernelGGL((carve_kernel<F, V>), grid, block, 0, e->stream, st, g, vd, dense_views, \
                       init, ap)
                if (e->fresh) {
                    LAUNCH_CARVE(true, true);
                } else {
                    LAUNCH_CARVE(false, true);
                }
#undef LAUNCH_CARVE
            }
        }
        HIP_TRY(hipGetLastError());
        rc = lt.end();
        if (rc) return rc;
        if (compact) {
            int s1 = (int)std::min<size_t>(nv, (size_t)ndense + nstage1);
            uint32_t *l0 = e->lists, *l1 = e->lists + (size_t)kSub * e->subcap;
            LaunchTimer lt2{e, SC_KERNEL_LIST};
            rc = lt2.begin();
            if (rc) return rc;
            int vg = (int)e->view_group;
            // open FULL candidates exist only when packing rode beside the dense stage
            CullStores none{nullptr, 0u, 0u, 0u, 0u, 0, 0, 0u, 0}, cs = none;
            if (ride_blocks) {
                // the riders have packed the rest of the masks: open FULL candidates get their answer
                // (one block per 64 bricks up to 4096 blocks; without candidates a block leaves after one scalar load)
                const uint32_t nconfirm = std::min<uint32_t>((nbricks + 63u) / 64u, 4096u);
                hipLaunchKernelGGL(brick_confirm_kernel, dim3(nconfirm), dim3(64 * kFlagWaves), 0, e->stream, g, vd,
                                   packed_ahead, (int)nv, bys, bzs, nbricks, e->flags, e->late, e->ctl);
            }
            if (bulk_on) {
                // ... and the units of the bulk list their verdicts
                const UnitJob uj{e->bulk, e->bulkcap, e->items, e->itemcap, vd, (int32_t)nv, ndense, bys, bzs, st,
                                 e->lists, e->subcap, (uint32_t)e->item_bias, e->unit_stats};
                hipLaunchKernelGGL(unit_verdict_kernel, dim3((uint32_t)e->unit_blocks), dim3(64 * kFlagWaves), 0,
                                   e->stream, g, e->ctl, uj);
            }
            // final stage with deferred stores: e->defer_stores persistent list blocks (they leave
            // wavefront slots free) and one short store block per strip behind them
            dim3 fgrid(list_blocks);
            CullStores cs1 = none;
            dim3 grid1(list_blocks);
            if (defer_stores) {
                // the first list stage may take a share of the fill as well (it waits on memory)
                uint32_t mid = dense_store_strips;
                if ((size_t)s1 < nv && e->stage1_store_share > 0) {
                    mid += (uint32_t)((uint64_t)(nstrips - dense_store_strips) * (uint64_t)e->stage1_store_share / 16u);
                    const uint32_t n1 = mid - dense_store_strips, f1 = std::min<uint32_t>((uint32_t)e->fill_blocks, n1);
                    cs1 = CullStores{e->flags, bys, bzs, mid, dense_store_strips, init == 0 ? 1 : init, e->fresh ? 1 : 0, f1, init};
                    grid1 = dim3((uint32_t)e->stage1_list_blocks + (f1 ? f1 : n1));
                }
                const uint32_t nf = nstrips - mid, ff = std::min<uint32_t>((uint32_t)e->fill_blocks, nf);
                cs = CullStores{e->flags, bys, bzs, nstrips, mid, init == 0 ? 1 : init, e->fresh ? 1 : 0, ff, init};
                fgrid = dim3((uint32_t)e->defer_stores + (ff ? ff : nf));
            }
#define LAUNCH_LIST(FIN, GRID, ...)                                                                      \
    do {                                                                                                 \
        if ((FIN ? e->final_voxels : e->stage1_voxels) == 4) hipLaunchKernelGGL((carve_list_kernel<FIN, 4>), GRID, block, 0, e->stream, __VA_ARGS__); \
        else if ((FIN ? e->final_voxels : e->stage1_voxels) == 2) hipLaunchKernelGGL((carve_list_kernel<FIN, 2>), GRID, block, 0, e->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((carve_list_kernel<FIN, 1>), GRID, block, 0, e->stream, __VA_ARGS__);     \
    } while (0)
            // stage 1 (l0 -> l1), optional stage 2 (l1 -> l0), final stage on what is left
            int s2 = (int)std::min<size_t>(nv, (size_t)s1 + (size_t)e->stage2_views);
            uint32_t *nolist = nullptr;
            // the final stage also takes the work items of the bulk units
            const UnitItems noitems{nullptr, 0u, nullptr, 0u, 0u}, ui{bulk_on ? e->items : nullptr, e->itemcap, vd, bys, bzs};
            // ... and the first list kernel behind the verdicts tells the host what they were worth
            const ReportJob norep{nullptr, nullptr, 0u, 0u};
            ReportJob rj = norep;
            if (bulk_on && e->bulk_adapt)
                rj = ReportJob{const_cast<unsigned long long *>(e->report), e->unit_stats, (uint32_t)e->unit_blocks, ++e->report_seq};
            if ((size_t)s1 >= nv) {
                LAUNCH_LIST(true, fgrid, st, g, vd + ndense, s1 - ndense, l0, nolist, e->ctl, 0, 0, e->subcap, vg, cs, ui, rj);
            } else {
                LAUNCH_LIST(false, grid1, st, g, vd + ndense, s1 - ndense, l0, l1, e->ctl, 0, 1, e->subcap, vg, cs1, noitems, rj);
                if (s2 > s1 && (size_t)s2 < nv) {
                    LAUNCH_LIST(false, dim3(list_blocks), st, g, vd + s1, s2 - s1, l1, l0, e->ctl, 1, 2, e->subcap, vg, none, noitems, norep);
                    LAUNCH_LIST(true, fgrid, st, g, vd + s2, (int)nv - s2, l0, nolist, e->ctl, 2, 2, e->subcap, vg, cs, ui, norep);
                } else {
                    LAUNCH_LIST(true, fgrid, st, g, vd + s1, (int)nv - s1, l1, nolist, e->ctl, 1, 1, e->subcap, vg, cs, ui, norep);
                }
            }
#undef LAUNCH_LIST
            // the resume kernel is also what zeroes the next batch's counters
            e->ctl_clean[e->ctl_idx ^ 1] = true;
            const LateBricks late{ride_blocks ? e->late : nullptr, vd, e->flags, (int32_t)nv, init, e->fresh ? 1 : 0, bys, bzs};
            hipLaunchKernelGGL(carve_resume_kernel<true>, dim3(list_blocks), block, 0, e->stream,
                               st, g, vd + ndense, (int)nv - ndense, e->ctl, e->ctl2[e->ctl_idx ^ 1], late);
            HIP_TRY(hipGetLastError());
            rc = lt2.end();
            if (rc) return rc;
        }
    } else {
        float *st = static_cast<float *>(e->state);
        rc = upload_desc();
        if (rc) return rc;
        // brick form: uint8 masks with uniformity flags on every view of the batch, a table, a grid it fits
        const uint32_t abys = (uint32_t)((e->ny + kBrickY - 1) / kBrickY), abzs = (uint32_t)((e->nz + kBrickZ - 1) / kBrickZ);
        bool abrick = nv > 1 && e->avg_brick && (uint64_t)e->npitch < 0x80000000ull &&
                      (uint64_t)e->planes * abys * abzs < 0x80000000ull;
        bool any_f32 = false;
        for (size_t q = 0; q < nv && abrick; ++q) {
            const ViewDesc &pd = e->pending[q];
            abrick = (pd.pad == 1 && e->lut_dev != nullptr && pd.occ != nullptr) || (pd.pad == 2 && pd.occ != nullptr);
            any_f32 |= pd.pad == 2;
        }
        if (abrick) {
            const uint32_t anb = (uint32_t)((uint64_t)e->planes * abys * abzs);
            const size_t need = (size_t)anb * nv;
            if (need > e->verd_cap) {
                HIP_TRY(hipStreamSynchronize(e->stream));
                if (e->verd) (void)hipFree(e->verd);
                e->verd = nullptr;
                e->verd_cap = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->verd), need));
                e->verd_cap = need;
            }
            if (any_f32 && need > e->verdf_cap) {  // the flat values of float32 views
                HIP_TRY(hipStreamSynchronize(e->stream));
                if (e->verdf) (void)hipFree(e->verdf);
                e->verdf = nullptr;
                e->verdf_cap = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->verdf), need * 4));
                e->verdf_cap = need;
            }
            uint32_t *verdf = any_f32 ? e->verdf : nullptr;
            LaunchTimer ltf{e, SC_KERNEL_FLAGS};
            rc = ltf.begin();
            if (rc) return rc;
            hipLaunchKernelGGL(avg_flags_kernel, dim3((anb + kBlock - 1) / kBlock, (uint32_t)nv), block, 0, e->stream,
                               g, vd, (int)nv, abys, abzs, anb, e->verd, verdf);
            rc = ltf.end();
            if (rc) return rc;
            LaunchTimer lta{e, SC_KERNEL_AVERAGE};
            rc = lta.begin();
            if (rc) return rc;
            if (e->fresh)
                hipLaunchKernelGGL(average_brick_kernel<true>, dim3(anb), block, 0, e->stream, st, g, vd, (int)nv,
                                   e->default_value, e->lut_dev, abys, abzs, e->verd, verdf);
            else
                hipLaunchKernelGGL(average_brick_kernel<false>, dim3(anb), block, 0, e->stream, st, g, vd, (int)nv,
                                   e->default_value, e->lut_dev, abys, abzs, e->verd, verdf);
            HIP_TRY(hipGetLastError());
            rc = lta.end();
            if (rc) return rc;
            rc = step_end(e, nv > 1);
            if (rc) return rc;
            e->fresh = false;
            e->pending.erase(e->pending.begin(), e->pending.begin() + (ptrdiff_t)nv);
            if (e->pending.empty()) arena_reset(e);
            return SC_OK;
        }
        LaunchTimer lt{e, SC_KERNEL_AVERAGE};
        rc = lt.begin();
        if (rc) return rc;
#define LAUNCH_AVG(F, V)                                                                         \
    do {                                                                                         \
        if (nv == 1)                                                                             \
            hipLaunchKernelGGL((average_kernel_1<F, V>), grid, block, 0, e->stream, st, g, one,  \
                               e->default_value, e->lut_dev);                                    \
        else                                                                                     \
            hipLaunchKernelGGL((average_kernel<F, V>), grid, block, 0, e->stream, st, g, vd,     \
                               (int)nv, e->default_value, e->lut_dev);                           \
    } while (0)
        if (e->fresh) {
            LAUNCH_AVG(true, true);
        } else {
            LAUNCH_AVG(false, true);
        }
#undef LAUNCH_AVG
        HIP_TRY(hipGetLastError());
        rc = lt.end();
        if (rc) return rc;
    }
    rc = step_end(e, nv > 1);
    if (rc) return rc;
    e->fresh = false;
    e->pending.erase(e->pending.begin(), e->pending.begin() + (ptrdiff_t)nv);
    if (e->pending.empty()) arena_reset(e);  // masks of launched views are dead in stream order
    return SC_OK;
}

int after_enqueue(sc_engine *e) {
    if (e->views_per_launch > 0) {
        while ((int64_t)e->pending.size() >= e->views_per_launch) {
            int rc = flush(e, (size_t)e->views_per_launch);
            if (rc) return rc;
        }
        return SC_OK;
    }
    if ((int64_t)e->pending.size() >= e->max_pending) return flush(e);
    return SC_OK;
}

int check_view_args(const sc_engine *e, const float *K, const float *R, const float *t,
                    const void *mask, int H, int W) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    if (!K || !R || !t || !mask) return fail(SC_ERR_INVALID, "null view argument");
    if (H <= 0 || W <= 0 || H > (1 << 24) || W > (1 << 24))
        return fail(SC_ERR_INVALID, "bad mask shape %d x %d", H, W);
    return SC_OK;
}

// The engine owns the x-planes  i0, i0 + istride, ...  (`planes` of them) of the global grid.
int create(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t i0, int64_t istride,
           int64_t planes, const float *origin, float vs, int mode, float default_value, int device) {
    if (!out) return fail(SC_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!origin) return fail(SC_ERR_INVALID, "null origin");
    if (nx <= 0 || ny <= 0 || nz <= 0) return fail(SC_ERR_INVALID, "shape must be positive");
    // int -> float of an index must be exact
    if (nx > (1 << 24) || ny > (1 << 24) || nz > (1 << 24))
        return fail(SC_ERR_INVALID, "axis longer than 2^24 voxels");
    if (i0 < 0 || istride < 1 || planes < 1 || i0 + (planes - 1) * istride >= nx)
        return fail(SC_ERR_INVALID, "bad slab / plane set (first %lld, stride %lld, planes %lld of %lld)",
                    (long long)i0, (long long)istride, (long long)planes, (long long)nx);
    if (mode != SC_MODE_CARVE && mode != SC_MODE_AVERAGE)
        return fail(SC_ERR_INVALID, "unknown mode %d", mode);
    // `device` is a HIP ordinal; only that device has to be a gfx950
    int ndev = 0;
    hipError_t hq = hipGetDeviceCount(&ndev);
    if (hq != hipSuccess) return fail(SC_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(hq));
    if (device < 0 || device >= ndev)
        return fail(SC_ERR_DEVICE, "device %d not available (%d HIP device(s) visible)", device, ndev);
    {
        hipDeviceProp_t prop;
        hq = hipGetDeviceProperties(&prop, device);
        if (hq != hipSuccess) return fail(SC_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(hq));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SC_ERR_DEVICE, "device %d is %s; this engine is built for gfx950 only", device,
                        prop.gcnArchName);
    }
    sc_engine *e = new (std::nothrow) sc_engine();
    if (!e) return fail(SC_ERR_NOMEM, "host allocation failed");
    e->device = device;
    e->mode = mode;
    e->nx = nx; e->ny = ny; e->nz = nz; e->i0 = i0; e->istride = istride; e->planes = planes;
    e->n = planes * ny * nz;
    e->nzp = (nz + 63) / 64 * 64;
    e->npitch = planes * ny * e->nzp;
    memcpy(e->origin, origin, sizeof e->origin);
    e->vs = vs;
    e->default_value = default_value;
    hipError_t he = hipSetDevice(device);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMalloc(&e->state, (size_t)e->npitch * 4);
    if (he != hipSuccess) {
        int code = he == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE;
        fail(code, "engine setup failed: %s", hipGetErrorString(he));
        sc_destroy(e);
        return code;
    }
    e->stream = e->own_stream;
    e->fresh = true;
    *out = e;
    return SC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int sc_abi_version(void) { return SC_ABI_VERSION; }

const char *sc_last_error(void) { return g_err.c_str(); }

int sc_device_count(int *count) {
    if (!count) return fail(SC_ERR_INVALID, "null count");
    *count = 0;
    int n = 0;
    hipError_t he = hipGetDeviceCount(&n);
    if (he != hipSuccess) return fail(SC_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(he));
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;  // other devices are simply not ours
    }
    *count = ok;
    return SC_OK;
}

int sc_create(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, const float origin[3],
              float voxel_size, int mode, float default_value, int device) {
    return create(out, nx, ny, nz, 0, 1, nx, origin, voxel_size, mode, default_value, device);
}

int sc_create_slab(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t i0, int64_t i1,
                   const float origin[3], float voxel_size, int mode, float default_value,
                   int device) {
    if (i0 < 0 || i1 > nx || i0 >= i1) return fail(SC_ERR_INVALID, "bad slab [%lld, %lld)", (long long)i0, (long long)i1);
    return create(out, nx, ny, nz, i0, 1, i1 - i0, origin, voxel_size, mode, default_value, device);
}

int sc_create_cyclic(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t first, int64_t stride,
                     const float origin[3], float voxel_size, int mode, float default_value,
                     int device) {
    if (stride < 1 || first < 0 || first >= stride || first >= nx)
        return fail(SC_ERR_INVALID, "bad plane set (first %lld, stride %lld)", (long long)first, (long long)stride);
    int64_t planes = (nx - first + stride - 1) / stride;
    return create(out, nx, ny, nz, first, stride, planes, origin, voxel_size, mode, default_value, device);
}

void sc_destroy(sc_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (int k = 0; k < kNumKernels; ++k)
        for (auto &tl : e->timed[k]) {
            (void)hipEventDestroy(tl.start);
            (void)hipEventDestroy(tl.stop);
        }
    if (e->step_open) (void)hipEventDestroy(e->step_start);
    if (e->span_open) (void)hipEventDestroy(e->span_start);
    for (auto ev : e->event_pool) (void)hipEventDestroy(ev);
    for (auto &c : e->chunks) (void)hipFree(c.base);
    for (int s = 0; s < kSlots; ++s) {
        if (e->pin[s]) (void)hipHostFree(e->pin[s]);
        if (e->raw[s]) (void)hipFree(e->raw[s]);
        if (e->slot_ev[s]) (void)hipEventDestroy(e->slot_ev[s]);
    }
    if (e->views_dev) (void)hipFree(e->views_dev);
    if (e->views_pin) (void)hipHostFree(e->views_pin);
    if (e->narrow) (void)hipFree(e->narrow);
    if (e->packed_labels) (void)hipFree(e->packed_labels);
    if (e->dense) (void)hipFree(e->dense);
    if (e->verd) (void)hipFree(e->verd);
    if (e->verdf) (void)hipFree(e->verdf);
    if (e->dead) (void)hipFree(e->dead);
    if (e->lut_dev) (void)hipFree(e->lut_dev);
    if (e->lists) (void)hipFree(e->lists);
    if (e->ctl2[0]) (void)hipFree(e->ctl2[0]);
    if (e->items) (void)hipFree(e->items);
    if (e->report) (void)hipHostFree(const_cast<unsigned long long *>(e->report));
    if (e->unit_stats) (void)hipFree(e->unit_stats);
    if (e->state) (void)hipFree(e->state);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int sc_clear(sc_engine *e) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    e->pending.clear();
    e->deferred.on = false;
    e->dead_clean = false;  // the labels go back to default_value: no brick is known to be all -1
    arena_reset(e);
    if (e->step_open) {  // the views of an open SC_KERNEL_STEP window are gone: no sample for them
        e->event_pool.push_back(e->step_start);
        e->step_open = false;
    }
    e->fresh = true;  // materialised lazily: a fused launch never needs to read it
    return SC_OK;
}

int sc_set_option(sc_engine *e, int key, int64_t value) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    switch (key) {
        case SC_OPT_VIEWS_PER_LAUNCH:
            if (value < 0) return fail(SC_ERR_INVALID, "views_per_launch must be >= 0");
            e->views_per_launch = value;
            return SC_OK;
        case SC_OPT_VIEW_ORDER:
            if (value != 0 && value != 1) return fail(SC_ERR_INVALID, "view_order must be 0 or 1");
            e->view_order = value;
            return SC_OK;
        case SC_OPT_TIME_KERNELS:
            e->time_kernels = value == 2 ? 2 : (value ? 1 : 0);
            return SC_OK;
        case SC_OPT_COMPACT:
            e->compact = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_DENSE_VIEWS:
            if (value < 1 || value > 64) return fail(SC_ERR_INVALID, "dense_views must be in [1, 64]");
            e->dense_views = value;
            return SC_OK;
        case SC_OPT_STAGE1_VIEWS:
            if (value < 1 || value > 4096) return fail(SC_ERR_INVALID, "stage1_views must be in [1, 4096]");
            e->stage1_views = value;
            return SC_OK;
        case SC_OPT_LIST_BLOCKS:
            if (value < 1 || value > 65536) return fail(SC_ERR_INVALID, "list_blocks must be in [1, 65536]");
            e->list_blocks = value;
            return SC_OK;
        case SC_OPT_BRICK:
            e->brick = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_STAGE2_VIEWS:
            if (value < 0 || value > 4096) return fail(SC_ERR_INVALID, "stage2_views must be in [0, 4096]");
            e->stage2_views = value;
            return SC_OK;
        case SC_OPT_FULL_BRICKS:
            e->full_bricks = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_AVG_TILE_F32:
            e->avg_tile_f32 = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_AVG_BRICK:
            e->avg_brick = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_STAGE1_STORE_SHARE:
            if (value < 0 || value > 16) return fail(SC_ERR_INVALID, "stage1_store_share must be in [0, 16]");
            e->stage1_store_share = value;
            return SC_OK;
        case SC_OPT_STAGE1_LIST_BLOCKS:
            if (value < 1 || value > 65536) return fail(SC_ERR_INVALID, "stage1_list_blocks must be in [1, 65536]");
            e->stage1_list_blocks = value;
            return SC_OK;
        case SC_OPT_DEFER_SHARE:
            if (value < 0 || value > 16) return fail(SC_ERR_INVALID, "defer_share must be in [0, 16]");
            e->defer_share = value;
            return SC_OK;
        case SC_OPT_DEFER_STORES:
            if (value < 0 || value > 65536) return fail(SC_ERR_INVALID, "defer_stores must be in [0, 65536]");
            e->defer_stores = value;
            return SC_OK;
        case SC_OPT_PACK_ROWS:
            if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
                return fail(SC_ERR_INVALID, "pack_rows must be 0, 1, 2, 4 or 8");
            e->pack_rows = value;
            return SC_OK;
        case SC_OPT_VIEW_BRICK:
            e->view_brick = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_RESERVE_EVENTS: {
            if (value < 0 || value > 65536) return fail(SC_ERR_INVALID, "reserve_events must be in [0, 65536]");
            int rc = use_device(e);
            if (rc) return rc;
            while ((int64_t)e->event_pool.size() < value) {
                hipEvent_t ev;
                HIP_TRY(hipEventCreate(&ev));
                e->event_pool.push_back(ev);
            }
            return SC_OK;
        }
        case SC_OPT_STAGE1_VOXELS:
            if (value != 1 && value != 2 && value != 4) return fail(SC_ERR_INVALID, "stage1_voxels must be 1, 2 or 4");
            e->stage1_voxels = value;
            return SC_OK;
        case SC_OPT_FINAL_VOXELS:
            if (value != 1 && value != 2 && value != 4) return fail(SC_ERR_INVALID, "final_voxels must be 1, 2 or 4");
            e->final_voxels = value;
            return SC_OK;
        case SC_OPT_FILL_BLOCKS:
            if (value < 0 || value > 65536) return fail(SC_ERR_INVALID, "fill_blocks must be in [0, 65536]");
            e->fill_blocks = value;
            return SC_OK;
        case SC_OPT_PACK_RIDE:
            e->pack_ride = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_BRICK_WALKERS:
            if (value < 8 || value > 65536) return fail(SC_ERR_INVALID, "brick_walkers must be in [8, 65536]");
            e->brick_walkers = value;
            return SC_OK;
        case SC_OPT_FLAG_VIEWS:
            if (value < 0) return fail(SC_ERR_INVALID, "flag_views must be >= 0");
            e->flag_views = value;
            return SC_OK;
        case SC_OPT_VIEW_GROUP:
            if (value < 1 || value > 4096) return fail(SC_ERR_INVALID, "view_group must be in [1, 4096]");
            e->view_group = value;
            return SC_OK;
        case SC_OPT_BULK_MIN:
            if (value < 0 || value > 256) return fail(SC_ERR_INVALID, "bulk_min must be in [0, 256]");
            e->bulk_min = value;
            return SC_OK;
        case SC_OPT_ITEM_BIAS:
            if (value < 0 || value > 64) return fail(SC_ERR_INVALID, "item_bias must be in [0, 64]");
            e->item_bias = value;
            return SC_OK;
        case SC_OPT_UNIT_CULL:
            if (value < 0 || value > 2) return fail(SC_ERR_INVALID, "unit_cull must be 0, 1 or 2");
            e->unit_cull = value;
            return SC_OK;
        case SC_OPT_BULK_ADAPT:
            e->bulk_adapt = value ? 1 : 0;
            e->bulk_hold = 0;
            return SC_OK;
        case SC_OPT_UNIT_BLOCKS:
            if (value < 1 || value > 65536) return fail(SC_ERR_INVALID, "unit_blocks must be in [1, 65536]");
            e->unit_blocks = value;
            return SC_OK;
        case SC_OPT_MAX_PENDING:
            if (value < 1) return fail(SC_ERR_INVALID, "max_pending must be >= 1");
            e->max_pending = value;
            return SC_OK;
        default:
            return fail(SC_ERR_INVALID, "unknown option %d", key);
    }
}

int sc_set_lut(sc_engine *e, const float *lut256) {
    if (!e || !lut256) return fail(SC_ERR_INVALID, "null argument");
    if (e->mode != SC_MODE_AVERAGE) return fail(SC_ERR_STATE, "the table belongs to averaging engines");
    int rc = use_device(e);
    if (rc) return rc;
    rc = flush(e);  // views already enqueued keep the old table
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!e->lut_dev) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->lut_dev), 256 * sizeof(float)));
    HIP_TRY(hipMemcpy(e->lut_dev, lut256, 256 * sizeof(float), hipMemcpyHostToDevice));
    return SC_OK;
}

int sc_set_stream(sc_engine *e, void *hip_stream) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    rc = flush(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->own_stream;
    return SC_OK;
}

int sc_order_after(sc_engine *e, void *producer_stream) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    // NULL is the legacy default stream here (torch's default stream has handle 0); the engine's own
    // stream is non-blocking, so it does NOT synchronise with that stream by itself
    // (the handle 0 itself: hipStreamLegacy is not understood by every runtime this library meets --
    // torch's bundled one crashed on it)
    hipStream_t prod = static_cast<hipStream_t>(producer_stream);
    if (prod != nullptr && prod == e->stream) return SC_OK;  // same stream: already in order
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t he = hipEventRecord(ev, prod);
    if (he == hipSuccess) he = hipStreamWaitEvent(e->stream, ev, 0);
    (void)hipEventDestroy(ev);  // released once the wait has been satisfied
    if (he != hipSuccess) return fail(SC_ERR_DEVICE, "ordering after the producer stream failed: %s", hipGetErrorString(he));
    return SC_OK;
}

int sc_order_before(sc_engine *e, void *consumer_stream) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    hipStream_t cons = static_cast<hipStream_t>(consumer_stream);
    if (cons != nullptr && cons == e->stream) return SC_OK;  // same stream: already in order
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t he = hipEventRecord(ev, e->stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(cons, ev, 0);
    (void)hipEventDestroy(ev);  // released once the wait has been satisfied
    if (he != hipSuccess) return fail(SC_ERR_DEVICE, "ordering the consumer stream behind the engine failed: %s", hipGetErrorString(he));
    return SC_OK;
}

int sc_process_view(sc_engine *e, const float K[4], const float R[9], const float t[3],
                    const void *mask, int H, int W, int mask_dtype, int64_t row_stride_bytes) {
    int rc = check_view_args(e, K, R, t, mask, H, W);
    if (rc) return rc;
    rc = check_dtype(e, mask_dtype);
    if (rc) return rc;
    rc = use_device(e);
    if (rc) return rc;
    rc = materialize_deferred(e);  // a device batch waiting for its flush: packed now, order as given
    if (rc) return rc;
    size_t es = elem_size(mask_dtype);
    size_t row = (size_t)W * es;
    if (row_stride_bytes == 0) row_stride_bytes = (int64_t)row;
    if (row_stride_bytes < (int64_t)row) return fail(SC_ERR_INVALID, "row stride smaller than a row");
    size_t bytes = row * (size_t)H;
    rc = ensure_slots(e, bytes);
    if (rc) return rc;
    int s = e->next_slot;
    e->next_slot = (s + 1) % kSlots;
    if (e->slot_armed[s]) {
        HIP_TRY(hipEventSynchronize(e->slot_ev[s]));
        e->slot_armed[s] = false;
    }
    // consume the caller's buffer now (tight rows in the pinned slot)
    if (row_stride_bytes == (int64_t)row) {
        memcpy(e->pin[s], mask, bytes);
    } else {
        for (int r = 0; r < H; ++r)
            memcpy(static_cast<char *>(e->pin[s]) + (size_t)r * row,
                   static_cast<const char *>(mask) + (size_t)r * row_stride_bytes, row);
    }
    if (e->mode == SC_MODE_CARVE) {
        HIP_TRY(hipMemcpyAsync(e->raw[s], e->pin[s], bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->slot_ev[s], e->stream));
        e->slot_armed[s] = true;
        rc = enqueue_pack(e, 1, K, R, t, e->raw[s], H, W, mask_dtype, (int64_t)row, (int64_t)bytes);
        if (rc) return rc;
    } else if (mask_dtype == SC_MASK_U8_LUT) {
        HIP_TRY(hipMemcpyAsync(e->raw[s], e->pin[s], bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->slot_ev[s], e->stream));
        e->slot_armed[s] = true;
        rc = enqueue_tile8(e, 1, K, R, t, e->raw[s], H, W, (int64_t)row, (int64_t)bytes);
        if (rc) return rc;
    } else if (e->avg_tile_f32) {
        HIP_TRY(hipMemcpyAsync(e->raw[s], e->pin[s], bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->slot_ev[s], e->stream));
        e->slot_armed[s] = true;
        rc = enqueue_tilef32(e, 1, K, R, t, e->raw[s], H, W, (int64_t)row, (int64_t)bytes);
        if (rc) return rc;
    } else {
        void *dst = nullptr;
        rc = arena_alloc(e, bytes, &dst);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(dst, e->pin[s], bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->slot_ev[s], e->stream));
        e->slot_armed[s] = true;
        ViewDesc d;
        fill_desc(e, d, K, R, t, dst, H, W);
        e->pending.push_back(d);
    }
    return after_enqueue(e);
}

int sc_process_views(sc_engine *e, int V, const float *K, const float *R, const float *t,
                     const void *const *masks, int H, int W, int mask_dtype,
                     int64_t row_stride_bytes) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    if (V < 0 || (V > 0 && (!K || !R || !t || !masks))) return fail(SC_ERR_INVALID, "bad view batch");
    for (int q = 0; q < V; ++q) {
        int rc = sc_process_view(e, K + 4 * q, R + 9 * q, t + 3 * q, masks[q], H, W, mask_dtype,
                                 row_stride_bytes);
        if (rc) return rc;
    }
    return SC_OK;
}

int sc_process_views_device(sc_engine *e, int V, const float *K, const float *R, const float *t,
                            const void *masks_dev, int H, int W, int mask_dtype) {
    if (V == 0) return e ? SC_OK : fail(SC_ERR_INVALID, "null engine");
    int rc = check_view_args(e, K, R, t, masks_dev, H, W);
    if (rc) return rc;
    if (V < 0) return fail(SC_ERR_INVALID, "negative view count");
    rc = check_dtype(e, mask_dtype);
    if (rc) return rc;
    rc = use_device(e);
    if (rc) return rc;
    rc = materialize_deferred(e);
    if (rc) return rc;
    size_t es = elem_size(mask_dtype);
    int64_t row = (int64_t)W * (int64_t)es, view = row * H;
    if (e->mode == SC_MODE_CARVE) {
        if (e->pack_ride && e->views_per_launch == 0 && e->pending.empty() && V >= kMinFusedViews &&
            V <= kPackOrderMax && V < e->max_pending && pack16_eligible(masks_dev, W, mask_dtype, row, view)) {
            // the whole batch will be one fused launch: its packing waits for the flush, which knows
            // the order of the views (see flush)
            e->deferred.on = true;
            e->deferred.raw = masks_dev;
            e->deferred.V = V; e->deferred.H = H; e->deferred.W = W; e->deferred.dtype = mask_dtype;
            e->deferred.row_stride = row; e->deferred.view_stride = view;
            for (int q = 0; q < V; ++q) {
                ViewDesc d;
                fill_desc(e, d, K + 4 * q, R + 9 * q, t + 3 * q, nullptr, H, W, nullptr);
                e->pending.push_back(d);
            }
            return SC_OK;
        }
        // one pack launch for the whole batch, then carve launches per views_per_launch
        rc = enqueue_pack(e, V, K, R, t, masks_dev, H, W, mask_dtype, row, view);
        if (rc) return rc;
        return after_enqueue(e);
    }
    if (mask_dtype == SC_MASK_U8_LUT) {
        rc = enqueue_tile8(e, V, K, R, t, masks_dev, H, W, row, view);
        if (rc) return rc;
        return after_enqueue(e);
    }
    if (e->avg_tile_f32) {
        rc = enqueue_tilef32(e, V, K, R, t, masks_dev, H, W, row, view);
        if (rc) return rc;
        return after_enqueue(e);
    }
    for (int q = 0; q < V; ++q) {
        ViewDesc d;
        fill_desc(e, d, K + 4 * q, R + 9 * q, t + 3 * q,
                  static_cast<const char *>(masks_dev) + (int64_t)q * view, H, W);
        e->pending.push_back(d);
        rc = after_enqueue(e);
        if (rc) return rc;
    }
    return SC_OK;
}

// The first nv pending descriptors of an engine into its device ring, by a copy on its stream.
static int stage_descriptors(sc_engine *e, size_t nv, const ViewDesc **out) {
    if (nv > e->views_cap || e->views_head + nv > e->views_cap) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        e->views_head = 0;
    }
    if (nv > e->views_cap) {
        if (e->views_dev) (void)hipFree(e->views_dev);
        if (e->views_pin) (void)hipHostFree(e->views_pin);
        e->views_dev = e->views_pin = nullptr;
        e->views_cap = 0;
        size_t cap = std::max<size_t>(nv * 4, 1024);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->views_dev), cap * sizeof(ViewDesc)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->views_pin), cap * sizeof(ViewDesc), hipHostMallocDefault));
        e->views_cap = cap;
    }
    ViewDesc *pin = e->views_pin + e->views_head, *dev = e->views_dev + e->views_head;
    memcpy(pin, e->pending.data(), nv * sizeof(ViewDesc));
    e->views_head += nv;
    HIP_TRY(hipMemcpyAsync(dev, pin, nv * sizeof(ViewDesc), hipMemcpyHostToDevice, e->stream));
    *out = dev;
    return SC_OK;
}

int sc_average_labels(sc_engine *const *engines, int L, int V, const float *K, const float *R, const float *t,
                      const void *const *masks_dev, int H, int W) {
    if (!engines || !masks_dev || L < 1) return fail(SC_ERR_INVALID, "bad label set");
    for (int l = 0; l < L; ++l) {
        int rc = check_view_args(engines[l], K, R, t, masks_dev[l], H, W);
        if (rc) return rc;
        rc = check_dtype(engines[l], SC_MASK_U8_LUT);
        if (rc) return rc;
    }
    if (V < 0) return fail(SC_ERR_INVALID, "negative view count");
    if (V == 0) return SC_OK;
    sc_engine *e0 = engines[0];
    const int64_t row = W, view = (int64_t)W * H;
    // one launch needs: labels in groups of 2 .. 4 on one device, one grid, the same freshness, nothing pending,
    // the brick form's conditions (flush), whole 16-pixel rows; anything else goes label by label
    // (up to 4 labels: measured on a 6-label segmentation -- groups of 4 + 2, 3 + 3 or 2 + 2 + 2 -- the shared
    // launches took 16.5-17.4 ms where six launches of their own take 14.3: the labels' footprints are mixed in
    // different places, so the union of the (brick, view) pairs to project is nearly their sum, and every label
    // is dragged through every pair.  3 labels: 3.6 ms against 4.5.)
    bool fused = L >= 2 && L <= kMaxLabels && V > 1 && e0->avg_brick && (uint64_t)e0->npitch < 0x80000000ull &&
                 (W % 16) == 0 && V <= 4096;
    const uint32_t abys = (uint32_t)((e0->ny + kBrickY - 1) / kBrickY), abzs = (uint32_t)((e0->nz + kBrickZ - 1) / kBrickZ);
    fused = fused && (uint64_t)e0->planes * abys * abzs < 0x80000000ull;
    for (int l = 0; l < L && fused; ++l) {
        const sc_engine *e = engines[l];
        fused = e->device == e0->device && e->nx == e0->nx && e->ny == e0->ny && e->nz == e0->nz && e->i0 == e0->i0 &&
                e->istride == e0->istride && e->planes == e0->planes && e->vs == e0->vs &&
                memcmp(e->origin, e0->origin, sizeof e->origin) == 0 && e->fresh == e0->fresh && e->pending.empty() &&
                !e->deferred.on && e->avg_brick && (reinterpret_cast<uintptr_t>(masks_dev[l]) % 16) == 0;
        for (int m = 0; m < l && fused; ++m) fused = engines[m] != e;
    }
    if (!fused) {
        for (int l = 0; l < L; ++l) {
            int rc = sc_process_views_device(engines[l], V, K, R, t, masks_dev[l], H, W, SC_MASK_U8_LUT);
            if (rc) return rc;
            rc = sc_flush(engines[l]);
            if (rc) return rc;
        }
        return SC_OK;
    }
    int rc = use_device(e0);
    if (rc) return rc;
    // everything of this call runs on the first engine's stream, behind what the others have on theirs; their
    // streams take up again behind it
    hipStream_t main = e0->stream;
    std::vector<hipStream_t> own((size_t)L);
    for (int l = 0; l < L; ++l) {
        own[(size_t)l] = engines[l]->stream;
        if (l > 0 && own[(size_t)l] != main) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipError_t he = hipEventRecord(ev, own[(size_t)l]);
            if (he == hipSuccess) he = hipStreamWaitEvent(main, ev, 0);
            (void)hipEventDestroy(ev);
            if (he != hipSuccess) return fail(SC_ERR_DEVICE, "stream ordering failed: %s", hipGetErrorString(he));
        }
        engines[l]->stream = main;
    }
    auto restore = [&]() {
        for (int l = 0; l < L; ++l) engines[l]->stream = own[(size_t)l];
    };
    const GridDesc g = grid_desc(e0);
    const uint32_t anb = (uint32_t)((uint64_t)e0->planes * abys * abzs);
    const size_t need = (size_t)anb * (size_t)V;
    const ViewDesc *vd[64];
    rc = SC_OK;
    for (int l = 0; l < L && rc == SC_OK; ++l) {
        sc_engine *e = engines[l];
        rc = enqueue_tile8(e, V, K, R, t, masks_dev[l], H, W, row, view);
        if (rc) break;
        if (e->pending[0].occ == nullptr) { rc = fail(SC_ERR_STATE, "no uniformity flags"); break; }
        rc = stage_descriptors(e, (size_t)V, &vd[l < 64 ? l : 0]);
        if (rc) break;
        if (need > e->verd_cap) {
            hipError_t he = hipStreamSynchronize(main);
            if (e->verd) (void)hipFree(e->verd);
            e->verd = nullptr;
            e->verd_cap = 0;
            if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&e->verd), need);
            if (he != hipSuccess) { rc = fail(SC_ERR_NOMEM, "verdict buffer: %s", hipGetErrorString(he)); break; }
            e->verd_cap = need;
        }
    }
    if (rc == SC_OK && L > 64) rc = fail(SC_ERR_INVALID, "more than 64 labels");
    for (int l0 = 0; l0 < L && rc == SC_OK; l0 += kMaxLabels) {
        const int n = std::min(kMaxLabels, L - l0);
        if (n == 1) {  // a label left over: its own launches
            sc_engine *e = engines[l0];
            hipLaunchKernelGGL(avg_flags_kernel, dim3((anb + kBlock - 1) / kBlock, (uint32_t)V), dim3(kBlock), 0, main, g, vd[l0],
                               V, abys, abzs, anb, e->verd, static_cast<uint32_t *>(nullptr));
            if (e->fresh)
                hipLaunchKernelGGL(average_brick_kernel<true>, dim3(anb), dim3(kBlock), 0, main, static_cast<float *>(e->state), g,
                                   vd[l0], V, e->default_value, e->lut_dev, abys, abzs, e->verd, static_cast<uint32_t *>(nullptr));
            else
                hipLaunchKernelGGL(average_brick_kernel<false>, dim3(anb), dim3(kBlock), 0, main, static_cast<float *>(e->state), g,
                                   vd[l0], V, e->default_value, e->lut_dev, abys, abzs, e->verd, static_cast<uint32_t *>(nullptr));
            continue;
        }
        MultiArgs a;
        memset(&a, 0, sizeof a);
        for (int q = 0; q < n; ++q) {
            sc_engine *e = engines[l0 + q];
            a.values[q] = static_cast<float *>(e->state);
            a.views[q] = vd[l0 + q];
            a.verd[q] = e->verd;
            a.lut[q] = e->lut_dev;
            a.init[q] = e->default_value;
        }
#define LAUNCH_MULTI(N)                                                                                              \
    do {                                                                                                             \
        hipLaunchKernelGGL((avg_flags_multi_kernel<N>), dim3((anb + kBlock - 1) / kBlock, (uint32_t)V), dim3(kBlock), 0, main, a, g, \
                           V, abys, abzs, anb, static_cast<uint8_t *const *>(nullptr));                              \
        if (e0->fresh) hipLaunchKernelGGL((average_multi_kernel<N, true>), dim3(anb), dim3(kBlock), 0, main, a, g, V, abys, abzs); \
        else hipLaunchKernelGGL((average_multi_kernel<N, false>), dim3(anb), dim3(kBlock), 0, main, a, g, V, abys, abzs);          \
    } while (0)
        if (n == 2) LAUNCH_MULTI(2);
#if SC_MAXLABELS >= 3
        else if (n == 3) LAUNCH_MULTI(3);
#endif
#if SC_MAXLABELS >= 4
        else LAUNCH_MULTI(4);
#endif
#undef LAUNCH_MULTI
    }
    if (rc == SC_OK && hipGetLastError() != hipSuccess) rc = fail(SC_ERR_DEVICE, "multi-label launch failed");
    // the other engines' own streams wait for the first one's
    if (rc == SC_OK) {
        hipEvent_t ev;
        hipError_t he = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventRecord(ev, main);
        for (int l = 1; l < L && he == hipSuccess; ++l)
            if (own[(size_t)l] != main) he = hipStreamWaitEvent(own[(size_t)l], ev, 0);
        if (he == hipSuccess) (void)hipEventDestroy(ev);
        if (he != hipSuccess) rc = fail(SC_ERR_DEVICE, "stream ordering failed: %s", hipGetErrorString(he));
    }
    for (int l = 0; l < L; ++l) {
        sc_engine *e = engines[l];
        e->pending.clear();
        if (rc == SC_OK) e->fresh = false;
        arena_reset(e);  // (on the first engine's stream, which every later use of this engine's arena is behind)
    }
    restore();
    return rc;
}

int sc_flush(sc_engine *e) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    return flush(e);
}

int sc_synchronize(sc_engine *e) {
    int rc = sc_flush(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SC_OK;
}

int sc_get_values(sc_engine *e, void *out) {
    if (!e || !out) return fail(SC_ERR_INVALID, "null argument");
    int rc = sc_flush(e);
    if (rc) return rc;
    rc = materialize(e);
    if (rc) return rc;
    void *src = nullptr;
    rc = dense_state(e, &src);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, src, (size_t)e->n * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SC_OK;
}

int sc_get_values_i8(sc_engine *e, int8_t *out) {
    if (!e || !out) return fail(SC_ERR_INVALID, "null argument");
    if (e->mode != SC_MODE_CARVE) return fail(SC_ERR_STATE, "int8 read-back is for carve labels");
    const int32_t init = init_bits_i32(e);
    if (init < -128 || init > 127) return fail(SC_ERR_STATE, "default_value %d does not fit int8", init);
    int rc = sc_flush(e);
    if (rc) return rc;
    rc = materialize(e);
    if (rc) return rc;
    if (!e->narrow) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->narrow), (size_t)e->n));
    const uint64_t n = (uint64_t)e->n;
    const uint64_t blocks = (n + (uint64_t)kBlock * 16 - 1) / ((uint64_t)kBlock * 16);
    if (e->nzp == e->nz) {
        hipLaunchKernelGGL(narrow_i8_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, e->stream,
                           static_cast<const int32_t *>(e->state), e->narrow, n);
    } else {  // rows without their padding, narrowed on the way
        const uint64_t rows = (uint64_t)e->planes * (uint64_t)e->ny;
        hipLaunchKernelGGL(depitch_kernel<int8_t>, dim3((uint32_t)std::min<uint64_t>((rows + 3) / 4, 65536)), dim3(kBlock), 0,
                           e->stream, static_cast<const uint32_t *>(e->state), e->narrow, rows, (uint32_t)e->nz,
                           (uint32_t)e->nzp);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, e->narrow, (size_t)e->n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SC_OK;
}

int sc_values_device_ptr(sc_engine *e, void **ptr) {
    if (!e || !ptr) return fail(SC_ERR_INVALID, "null argument");
    int rc = sc_flush(e);
    if (rc) return rc;
    rc = materialize(e);
    if (rc) return rc;
    return dense_state(e, ptr);  // planes * ny * nz elements, no row padding
}

int64_t sc_packed_bytes(int64_t voxels, int bits) {
    if (voxels < 0 || (bits != 1 && bits != 2)) return -1;
    const int64_t per = 32 / bits;
    return ((voxels + per - 1) / per * 4 + 15) / 16 * 16;  // whole words, whole 16-byte groups
}

int sc_values_packed(sc_engine *e, int bits, void **ptr, int64_t *bytes) {
    if (!e || !ptr || !bytes) return fail(SC_ERR_INVALID, "null argument");
    if (bits != 1 && bits != 2) return fail(SC_ERR_INVALID, "bits must be 1 or 2");
    if (e->mode != SC_MODE_CARVE) return fail(SC_ERR_STATE, "packed labels are carve labels");
    const int32_t init = init_bits_i32(e);
    if (bits == 2 && (init < -1 || init > 1 || (float)init != e->default_value))
        return fail(SC_ERR_STATE, "default_value %g is not one of -1, 0, 1: two bits cannot hold it", (double)e->default_value);
    int rc = sc_flush(e);
    if (rc) return rc;
    rc = materialize(e);
    if (rc) return rc;
    const int64_t nbytes = sc_packed_bytes(e->n, bits);
    if (!e->packed_labels) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->packed_labels), (size_t)sc_packed_bytes(e->n, 2)));
    const uint64_t words = ((uint64_t)e->n + (32 / bits) - 1) / (32 / bits);
    // bricks an earlier launch found empty are all -1 until the next clear: not read (see the kernel)
    const uint32_t bys = (uint32_t)((e->ny + kBrickY - 1) / kBrickY), bzs = (uint32_t)((e->nz + kBrickZ - 1) / kBrickZ);
    const uint8_t *dead = (e->dead && e->dead_clean) ? e->dead : nullptr;
    const dim3 grid((uint32_t)((words + kBlock - 1) / kBlock));
    if ((uint64_t)grid.x * kBlock < words) return fail(SC_ERR_INVALID, "grid too large for one launch");
    if (bits == 2)
        hipLaunchKernelGGL(pack_labels_kernel<2>, grid, dim3(kBlock), 0, e->stream, static_cast<const int32_t *>(e->state),
                           e->packed_labels, (uint64_t)e->n, (uint32_t)e->nz, (uint32_t)e->nzp, (uint32_t)e->ny, dead, bys, bzs);
    else
        hipLaunchKernelGGL(pack_labels_kernel<1>, grid, dim3(kBlock), 0, e->stream, static_cast<const int32_t *>(e->state),
                           e->packed_labels, (uint64_t)e->n, (uint32_t)e->nz, (uint32_t)e->nzp, (uint32_t)e->ny, dead, bys, bzs);
    HIP_TRY(hipGetLastError());
    // (the tail of the last 16-byte group is never read by a consumer that knows the voxel count)
    *ptr = e->packed_labels;
    *bytes = nbytes;
    return SC_OK;
}

namespace {
// words [w0, w1) of 2-bit labels (16 per word, label = the pair sign-extended) into int32
void widen2_range(const uint32_t *src, int32_t *dst, int64_t w0, int64_t w1, int64_t n) {
    for (int64_t w = w0; w < w1; ++w) {
        const uint32_t x = src[w];
        int32_t *o = dst + w * 16;
        if (w * 16 + 16 <= n) {
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = (int32_t)(x << (30 - 2 * i)) >> 30;
        } else {
            for (int i = 0; w * 16 + i < n; ++i) o[i] = (int32_t)(x << (30 - 2 * i)) >> 30;
        }
    }
}
}  // namespace

int sc_get_values_wire2(sc_engine *e, int32_t *out, void *staging, int64_t staging_bytes, int threads) {
    if (!e || !out || !staging) return fail(SC_ERR_INVALID, "null argument");
    void *ptr = nullptr;
    int64_t bytes = 0;
    int rc = sc_values_packed(e, 2, &ptr, &bytes);
    if (rc) return rc;
    const int64_t n = e->n, words = (n + 15) / 16;
    if (staging_bytes < words * 4) return fail(SC_ERR_INVALID, "staging buffer too small: %lld bytes needed", (long long)(words * 4));
    if (threads <= 0) threads = 8;
    threads = std::min(threads, 64);
    // pieces of 1 MiB of packed labels (16 MiB of int32): copied in order, widened as they land
    const int64_t piece = (int64_t)1 << 18;  // words
    const int64_t npieces = (words + piece - 1) / piece;
    std::atomic<int64_t> landed{0};
    std::atomic<int> failed{0};
    uint32_t *stg = static_cast<uint32_t *>(staging);
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads);
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            for (int64_t k = t; k < npieces; k += threads) {
                while (landed.load(std::memory_order_acquire) <= k) {
                    if (failed.load(std::memory_order_relaxed)) return;
                    std::this_thread::yield();
                }
                widen2_range(stg, out, k * piece, std::min(words, (k + 1) * piece), n);
            }
        });
    hipError_t err = hipSuccess;
    for (int64_t k = 0; k < npieces && err == hipSuccess; ++k) {
        const int64_t w0 = k * piece, w1 = std::min(words, (k + 1) * piece);
        err = hipMemcpyAsync(stg + w0, static_cast<const uint32_t *>(ptr) + w0, (size_t)(w1 - w0) * 4, hipMemcpyDeviceToHost, e->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
        if (err == hipSuccess) landed.store(k + 1, std::memory_order_release);
    }
    if (err != hipSuccess) failed.store(1);
    for (auto &th : pool) th.join();
    if (err != hipSuccess) return fail(SC_ERR_DEVICE, "label read-back failed: %s", hipGetErrorString(err));
    return SC_OK;
}

int sc_get_values_packed(sc_engine *e, int bits, void *out) {
    if (!out) return fail(SC_ERR_INVALID, "null argument");
    void *ptr = nullptr;
    int64_t bytes = 0;
    int rc = sc_values_packed(e, bits, &ptr, &bytes);
    if (rc) return rc;
    const int64_t words = (e->n + (32 / bits) - 1) / (32 / bits);
    HIP_TRY(hipMemcpyAsync(out, ptr, (size_t)words * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return SC_OK;
}

int sc_unpack_labels(int device, void *hip_stream, const void *recv_dev, int64_t rank_bytes, int world, int partition,
                     int64_t nx, int64_t ny, int64_t nz, int bits, void *out_dev, int out_bytes) {
    if (!recv_dev || !out_dev) return fail(SC_ERR_INVALID, "null argument");
    if (bits != 1 && bits != 2) return fail(SC_ERR_INVALID, "bits must be 1 or 2");
    if (out_bytes != 1 && out_bytes != 4) return fail(SC_ERR_INVALID, "output elements are int8 (1) or int32 (4)");
    if (partition != 0 && partition != 1) return fail(SC_ERR_INVALID, "partition: 0 plane-cyclic, 1 slabs");
    if (world < 1 || nx < world || ny < 1 || nz < 1 || rank_bytes < 0 || (rank_bytes & 3))
        return fail(SC_ERR_INVALID, "bad shape / world / stride");
    const uint64_t plane = (uint64_t)ny * (uint64_t)nz, n = (uint64_t)nx * plane;
    const uint64_t pmax = (uint64_t)(nx + world - 1) / world;
    if ((uint64_t)rank_bytes * 8 < pmax * plane * (uint64_t)bits) return fail(SC_ERR_INVALID, "rank stride too small for its planes");
    HIP_TRY(hipSetDevice(device));
    const uint64_t lanes = (n + 15) / 16, blocks = (lanes + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffULL) return fail(SC_ERR_INVALID, "grid too large for one launch");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const uint32_t *recv = static_cast<const uint32_t *>(recv_dev);
    const uint64_t rw = (uint64_t)rank_bytes / 4;
#define LAUNCH_UNPACK(B, T)                                                                                        \
    hipLaunchKernelGGL((unpack_labels_kernel<B, T>), dim3((uint32_t)blocks), dim3(kBlock), 0, st, recv,              \
                       static_cast<T *>(out_dev), rw, (uint32_t)world, (uint32_t)nx, plane, partition == 0 ? 1 : 0)
    if (bits == 2 && out_bytes == 1) LAUNCH_UNPACK(2, int8_t);
    else if (bits == 2) LAUNCH_UNPACK(2, int32_t);
    else if (out_bytes == 1) LAUNCH_UNPACK(1, int8_t);
    else LAUNCH_UNPACK(1, int32_t);
#undef LAUNCH_UNPACK
    HIP_TRY(hipGetLastError());
    return SC_OK;
}

int64_t sc_num_voxels(const sc_engine *e) { return e ? e->n : 0; }

int sc_kernel_stats(sc_engine *e, int kernel_id, int64_t *launches, double *total_ms) {
    if (!e || !launches || !total_ms) return fail(SC_ERR_INVALID, "null argument");
    if (kernel_id < 0 || kernel_id >= kNumKernels) return fail(SC_ERR_INVALID, "bad kernel id");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    double sum = 0.0;
    for (auto &tl : e->timed[kernel_id]) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, tl.start, tl.stop));
        sum += ms;
    }
    *launches = (int64_t)e->timed[kernel_id].size();
    *total_ms = sum;
    return SC_OK;
}

int sc_reset_kernel_stats(sc_engine *e) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int k = 0; k < kNumKernels; ++k) {
        for (auto &tl : e->timed[k]) {
            e->event_pool.push_back(tl.start);
            e->event_pool.push_back(tl.stop);
        }
        e->timed[k].clear();
    }
    return SC_OK;
}

int sc_span_begin(sc_engine *e) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    if (e->span_open) return fail(SC_ERR_STATE, "a span is open already");
    rc = get_event(e, &e->span_start);
    if (rc) return rc;
    hipEvent_t stop;
    rc = get_event(e, &stop);  // the second event exists before the span starts
    if (rc) return rc;
    e->event_pool.push_back(stop);
    HIP_TRY(hipEventRecord(e->span_start, e->stream));
    e->span_open = true;
    return SC_OK;
}

int sc_span_end(sc_engine *e, double *ms) {
    if (!e || !ms) return fail(SC_ERR_INVALID, "null argument");
    int rc = use_device(e);
    if (rc) return rc;
    if (!e->span_open) return fail(SC_ERR_STATE, "no span is open");
    hipEvent_t stop;
    rc = get_event(e, &stop);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(stop, e->stream));
    HIP_TRY(hipEventSynchronize(stop));
    float f = 0.0f;
    HIP_TRY(hipEventElapsedTime(&f, e->span_start, stop));
    *ms = (double)f;
    e->event_pool.push_back(e->span_start);
    e->event_pool.push_back(stop);
    e->span_open = false;
    return SC_OK;
}

int sc_fused_counts_ex(sc_engine *e, int64_t out[8]) {
    if (!e || !out) return fail(SC_ERR_INVALID, "bad argument");
    for (int q = 0; q < 8; ++q) out[q] = 0;
    int rc = sc_fused_counts(e, out);
    if (rc || !e->ctl) return rc;
    ListCtl host;
    HIP_TRY(hipMemcpy(&host, e->ctl, sizeof(ListCtl), hipMemcpyDeviceToHost));
    out[4] = host.nlate;
    if (e->last_bulk)
        for (int q = 0; q < kSub; ++q) {
            out[5] += std::min<uint32_t>(host.count[3][q].n, e->bulkcap);
            out[6] += std::min<uint32_t>(host.count[4][q].n, e->itemcap);
        }
    out[7] = e->bulk_hold;
    return SC_OK;
}

int sc_fused_counts(sc_engine *e, int64_t out[4]) {
    if (!e || !out) return fail(SC_ERR_INVALID, "bad argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!e->ctl) return SC_OK;  // no fused carve launched yet
    std::vector<ListCtl> host(1);
    HIP_TRY(hipMemcpy(host.data(), e->ctl, sizeof(ListCtl), hipMemcpyDeviceToHost));
    out[0] = host[0].nlive[e->last_parity];
    for (int s = 0; s < kSub; ++s) {
        out[1] += host[0].count[0][s].n;
        out[2] += host[0].count[1][s].n;
    }
    out[3] = host[0].overflow;

    return SC_OK;
}

int sc_view_certified(const float origin[3], float voxel_size, int64_t nx, int64_t ny, int64_t nz, const float K[4],
                      const float R[9], const float t[3], int *certified) {
    if (!origin || !K || !R || !t || !certified) return fail(SC_ERR_INVALID, "null argument");
    if (nx < 1 || ny < 1 || nz < 1) return fail(SC_ERR_INVALID, "shape must be positive");
    const int64_t first[3] = {0, 0, 0}, last[3] = {nx - 1, ny - 1, nz - 1};
    *certified = certify_view(K, R, t, origin, voxel_size, first, last);
    return SC_OK;
}

int sc_selftest_division(sc_engine *e, int64_t count, uint32_t seed, int mode,
                         uint64_t *mismatches, uint64_t *fast_pairs) {
    if (!e || !mismatches || !fast_pairs || count < 0) return fail(SC_ERR_INVALID, "bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    unsigned long long *out = nullptr, host[2] = {0, 0};
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&out), sizeof host));
    HIP_TRY(hipMemsetAsync(out, 0, sizeof host, e->stream));
    hipLaunchKernelGGL(div_selftest_kernel, dim3(4096), dim3(kBlock), 0, e->stream, (uint64_t)count,
                       seed, mode, out);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpyAsync(host, out, sizeof host, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    (void)hipFree(out);
    if (he != hipSuccess) return fail(SC_ERR_DEVICE, "division self-test failed: %s", hipGetErrorString(he));
    *mismatches = host[0];
    *fast_pairs = host[1];
    return SC_OK;
}

int sc_selftest_project(sc_engine *e, int64_t count, uint32_t seed, int nposes, const float *poses,
                        const int32_t *ijk, const int32_t *pose_idx, uint32_t *words_out,
                        uint64_t *digests_out) {
    if (!e || count < 0 || nposes < 1 || !poses || (!words_out && !digests_out))
        return fail(SC_ERR_INVALID, "bad argument");
    if (!ijk && pose_idx) return fail(SC_ERR_INVALID, "pose_idx goes with explicit voxel indices");
    const PoseRec *hp = reinterpret_cast<const PoseRec *>(poses);
    for (int q = 0; q < nposes; ++q) {
        if (hp[q].W < 1 || hp[q].H < 1 || (int64_t)hp[q].W * hp[q].H >= 0xffffffffLL)
            return fail(SC_ERR_INVALID, "pose %d: bad picture size %d x %d", q, hp[q].W, hp[q].H);
        if (!ijk && (hp[q].nx < 1 || hp[q].ny < 1 || hp[q].nz < 1))
            return fail(SC_ERR_INVALID, "pose %d: hashed samples need a grid shape", q);
    }
    if (ijk && pose_idx)
        for (int64_t i = 0; i < count; ++i)
            if (pose_idx[i] < 0 || pose_idx[i] >= nposes) return fail(SC_ERR_INVALID, "pose index out of range");
    if (count == 0) return SC_OK;
    int rc = use_device(e);
    if (rc) return rc;
    // each pose is certified (or not) for the box its samples come from, as fill_desc does for an engine's grid
    std::vector<PoseRec> cert(hp, hp + nposes);
    {
        int64_t ilo[3] = {0, 0, 0}, ihi[3] = {0, 0, 0};
        if (ijk) {
            for (int a = 0; a < 3; ++a) ilo[a] = ihi[a] = ijk[a];
            for (int64_t i = 0; i < count; ++i)
                for (int a = 0; a < 3; ++a) {
                    ilo[a] = std::min<int64_t>(ilo[a], ijk[3 * i + a]);
                    ihi[a] = std::max<int64_t>(ihi[a], ijk[3 * i + a]);
                }
        }
        for (int q = 0; q < nposes; ++q) {
            PoseRec &r = cert[q];
            if (!ijk) { ihi[0] = r.nx - 1; ihi[1] = r.ny - 1; ihi[2] = r.nz - 1; }
            const float o[3] = {r.ox, r.oy, r.oz};
            r.pad[0] = certify_view(r.K, r.R, r.t, o, r.vs, ilo, ihi);
        }
    }
    poses = reinterpret_cast<const float *>(cert.data());
    const size_t ndig = (size_t)((count + 65535) >> 16);
    PoseRec *dp = nullptr;
    int32_t *dijk = nullptr, *didx = nullptr;
    uint32_t *dw = nullptr;
    unsigned long long *dd = nullptr;
    hipError_t he = hipMalloc(reinterpret_cast<void **>(&dp), (size_t)nposes * sizeof(PoseRec));
    if (he == hipSuccess) he = hipMemcpy(dp, poses, (size_t)nposes * sizeof(PoseRec), hipMemcpyHostToDevice);
    if (he == hipSuccess && ijk) {
        he = hipMalloc(reinterpret_cast<void **>(&dijk), (size_t)count * 12);
        if (he == hipSuccess) he = hipMemcpy(dijk, ijk, (size_t)count * 12, hipMemcpyHostToDevice);
    }
    if (he == hipSuccess && pose_idx) {
        he = hipMalloc(reinterpret_cast<void **>(&didx), (size_t)count * 4);
        if (he == hipSuccess) he = hipMemcpy(didx, pose_idx, (size_t)count * 4, hipMemcpyHostToDevice);
    }
    if (he == hipSuccess && words_out) he = hipMalloc(reinterpret_cast<void **>(&dw), (size_t)count * 4);
    if (he == hipSuccess && digests_out) {
        he = hipMalloc(reinterpret_cast<void **>(&dd), ndig * 8);
        if (he == hipSuccess) he = hipMemsetAsync(dd, 0, ndig * 8, e->stream);
    }
    if (he == hipSuccess) {
        const uint64_t nwaves = ((uint64_t)count + 63) >> 6;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((nwaves + 3) / 4, 16384);
        hipLaunchKernelGGL(project_selftest_kernel, dim3(blocks), dim3(kBlock), 0, e->stream, (uint64_t)count,
                           seed, (uint32_t)nposes, dp, dijk, didx, dw, dd);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess && words_out) he = hipMemcpy(words_out, dw, (size_t)count * 4, hipMemcpyDeviceToHost);
    if (he == hipSuccess && digests_out) he = hipMemcpy(digests_out, dd, ndig * 8, hipMemcpyDeviceToHost);
    (void)hipFree(dp); (void)hipFree(dijk); (void)hipFree(didx); (void)hipFree(dw); (void)hipFree(dd);
    if (he != hipSuccess) return fail(SC_ERR_DEVICE, "projection self-test failed: %s", hipGetErrorString(he));
    return SC_OK;
}

int sc_host_alloc(int device, int64_t bytes, void **ptr) {
    if (!ptr || bytes <= 0) return fail(SC_ERR_INVALID, "bad argument");
    *ptr = nullptr;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return SC_OK;
}

void sc_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

int sc_dev_alloc(sc_engine *e, int64_t bytes, void **ptr) {
    if (!e || !ptr || bytes <= 0) return fail(SC_ERR_INVALID, "bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipMalloc(ptr, (size_t)bytes));
    return SC_OK;
}

int sc_dev_free(sc_engine *e, void *ptr) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipFree(ptr));
    return SC_OK;
}

int sc_dev_upload(sc_engine *e, void *dst_dev, const void *src_host, int64_t bytes) {
    if (!e || !dst_dev || !src_host || bytes < 0) return fail(SC_ERR_INVALID, "bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
    return SC_OK;
}

int sc_dev_download(sc_engine *e, void *dst_host, const void *src_dev, int64_t bytes) {
    if (!e || !dst_host || !src_dev || bytes < 0) return fail(SC_ERR_INVALID, "bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return SC_OK;
}

// ---- several devices from one process (sc_create_sharded) ----------------------------
// One engine per device, the x-planes dealt round-robin (or in contiguous slabs) exactly as the
// one-process-per-GPU path deals them to ranks; every view goes to every engine; the read-back lands
// each engine's planes at their global x positions with one strided copy per device.

}  // extern "C"

struct sc_group {
    std::vector<sc_engine *> eng;
    int64_t nx = 0, ny = 0, nz = 0;
    int partition = 0;
    // a call that failed on engine k after engines 0 .. k-1 took it leaves the x-planes in different states
    // (different view sets, tables or options): the group then refuses everything but a clear, which puts
    // every engine back to default_value, and its destruction
    bool mixed = false;
};

extern "C" {

int sc_create_sharded(sc_group **out, int64_t nx, int64_t ny, int64_t nz, const float origin[3],
                      float voxel_size, int mode, float default_value, const int *devices, int ndev,
                      int partition) {
    if (!out) return fail(SC_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > nx) return fail(SC_ERR_INVALID, "need 1..nx devices");
    if (partition != 0 && partition != 1) return fail(SC_ERR_INVALID, "partition: 0 plane-cyclic, 1 slabs");
    sc_group *g = new (std::nothrow) sc_group();
    if (!g) return fail(SC_ERR_NOMEM, "host allocation failed");
    g->nx = nx; g->ny = ny; g->nz = nz; g->partition = partition;
    for (int r = 0; r < ndev; ++r) {
        sc_engine *e = nullptr;
        int rc = partition == 0
                     ? sc_create_cyclic(&e, nx, ny, nz, r, ndev, origin, voxel_size, mode, default_value, devices[r])
                     : sc_create_slab(&e, nx, ny, nz, nx * r / ndev, nx * (r + 1) / ndev, origin, voxel_size, mode,
                                      default_value, devices[r]);
        if (rc) {
            for (auto *q : g->eng) sc_destroy(q);
            delete g;
            return rc;
        }
        g->eng.push_back(e);
    }
    *out = g;
    return SC_OK;
}

void sc_group_destroy(sc_group *g) {
    if (!g) return;
    for (auto *e : g->eng) sc_destroy(e);
    delete g;
}

int sc_group_size(const sc_group *g) { return g ? (int)g->eng.size() : 0; }

sc_engine *sc_group_engine(sc_group *g, int i) {
    return (g && i >= 0 && i < (int)g->eng.size()) ? g->eng[(size_t)i] : nullptr;
}

#define SC_GROUP_CHECK(g)                                                                                   \
    do {                                                                                                    \
        if (!(g)) return fail(SC_ERR_INVALID, "null group");                                                \
        if ((g)->mixed)                                                                                     \
            return fail(SC_ERR_STATE, "an earlier call failed on some engines of the group only: its planes are in " \
                                      "different states; sc_group_clear it (or destroy it)");               \
    } while (0)

#define SC_GROUP_EACH(call)                                   \
    do {                                                      \
        SC_GROUP_CHECK(g);                                    \
        size_t done_ = 0;                                     \
        for (auto *e : g->eng) {                              \
            int rc_ = (call);                                 \
            if (rc_) {                                        \
                if (done_ > 0) g->mixed = true;               \
                return rc_;                                   \
            }                                                 \
            ++done_;                                          \
        }                                                     \
        return SC_OK;                                         \
    } while (0)

int sc_group_clear(sc_group *g) {
    if (!g) return fail(SC_ERR_INVALID, "null group");
    int first = SC_OK;
    for (auto *e : g->eng) {  // every engine, whatever the others say
        int rc = sc_clear(e);
        if (rc && !first) first = rc;
    }
    g->mixed = first != SC_OK;
    return first;
}
int sc_group_flush(sc_group *g) { SC_GROUP_EACH(sc_flush(e)); }
int sc_group_set_option(sc_group *g, int key, int64_t value) { SC_GROUP_EACH(sc_set_option(e, key, value)); }
int sc_group_set_lut(sc_group *g, const float *lut256) {
    if (g && !lut256) return fail(SC_ERR_INVALID, "null argument");
    SC_GROUP_EACH(sc_set_lut(e, lut256));
}
int sc_group_process_view(sc_group *g, const float K[4], const float R[9], const float t[3], const void *mask,
                          int H, int W, int mask_dtype, int64_t row_stride_bytes) {
    SC_GROUP_CHECK(g);
    if (!g->eng.empty()) {  // the arguments are judged once, before any engine takes the view
        int rc = check_view_args(g->eng[0], K, R, t, mask, H, W);
        if (rc) return rc;
        for (auto *e : g->eng) {
            rc = check_dtype(e, mask_dtype);
            if (rc) return rc;
        }
        const int64_t row = (int64_t)W * (int64_t)elem_size(mask_dtype);
        if (row_stride_bytes != 0 && row_stride_bytes < row) return fail(SC_ERR_INVALID, "row stride smaller than a row");
    }
    SC_GROUP_EACH(sc_process_view(e, K, R, t, mask, H, W, mask_dtype, row_stride_bytes));
}

int sc_group_synchronize(sc_group *g) {
    SC_GROUP_CHECK(g);
    for (auto *e : g->eng) {  // every device launches before any is waited for
        int rc = sc_flush(e);
        if (rc) return rc;
    }
    for (auto *e : g->eng) {
        int rc = sc_synchronize(e);
        if (rc) return rc;
    }
    return SC_OK;
}

int sc_group_get_values(sc_group *g, void *out) {
    if (!g || !out) return fail(SC_ERR_INVALID, "null argument");
    SC_GROUP_CHECK(g);
    const size_t plane = (size_t)g->ny * (size_t)g->nz * 4;
    const int ndev = (int)g->eng.size();
    for (auto *e : g->eng) {  // launch everywhere first: the devices work side by side
        int rc = sc_flush(e);
        if (rc) return rc;
        rc = use_device(e);
        if (rc) return rc;
        rc = materialize(e);
        if (rc) return rc;
    }
    for (int r = 0; r < ndev; ++r) {
        sc_engine *e = g->eng[(size_t)r];
        int rc = use_device(e);
        if (rc) return rc;
        char *dst = static_cast<char *>(out) + (size_t)e->i0 * plane;
        void *src = nullptr;
        rc = dense_state(e, &src);  // without the row padding
        if (rc) return rc;
        // the engine's planes are contiguous on the device and istride planes apart in the grid
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)e->istride * plane, src, plane, plane, (size_t)e->planes,
                                 hipMemcpyDeviceToHost, e->stream));
    }
    for (auto *e : g->eng) {
        int rc = use_device(e);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return SC_OK;
}

}  // extern "C"
