// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- block placement, the exact shared-reciprocal division, project() (backprojection.c:3-34), mask words, unravel_index (common.h:1-12).

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
// hipcc expands an IEEE f32 division into  div_scale x2, rcp, 2 fma (Newton step on the reciprocal), mul, 4 fma (two
// corrections of the quotient), div_fmas, div_fixup: 22 instructions and two quarter-rate v_rcp_f32 for the two image
// coordinates.  Here: the reciprocal, refined by one Newton step, is SHARED between the two numerators, and each
// quotient takes ONE correction:  q = n r;  q += r (n - d q)  -- 9 instructions in all.
// That this is the correctly rounded quotient was established exhaustively (round 4, tools/probes/div_exhaustive.hip,
// profiles/r04_div_exhaustive.json): all 2^23 x 2^23 = 7.04e13 pairs of significands against hipcc's division on an
// MI355X, 0 differences (the same sequence from the UNREFINED reciprocal differs on 47 045 pairs, so the probe can see
// a failure).  Every operation is a multiplication or an FMA, hence a power-of-two scale of n or d scales every
// intermediate exactly as long as none of them leaves the normal range, and the rounding decisions depend on the
// significands alone: with d and |n| within 2^+-40 the smallest intermediate (the residual n - d q, a multiple of
// 2^-47 |n|) is above 2^-88.  Any lane outside that range sends its whole wavefront through the compiler's division.
// sc_selftest_division() compares the two bit-for-bit on 2^32 sampled operand pairs with the exponents spread.
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
    return __builtin_fmaf(e, r, q);
}

// backproject_point (backprojection.c:3-34) with the x/y partial sums hoisted.
// a{x,y,z} = R[0]*x + R[1]*y etc. (rounded as the reference rounds them).
//
// What the instructions cost on gfx950 (tools/probes/valu_probe.hip, cycles of a SIMD per wavefront
// instruction, independent instructions, 8 wavefronts per SIMD): v_mul_f32 / v_add_f32 / v_sub_f32 /
// v_and / v_lshrrev / v_add_u32 / v_mov 2.6; every three-operand or VOP3-only form (v_fma_f32, v_cmp_*,
// v_cvt_*, v_min/max, v_bfi, v_mad_*) 4.3-4.7; v_rcp_f32 8.3.
//
// CERTIFIED views (d.safe, set by certify_view on the host: every voxel of the grid has 2^-10 < pz and |px|, |py|,
// pz < 2^30 under this pose, intrinsics finite and below 2^30) take the short division with NO test of the operands:
//  * 2^-90 <= |n| < 2^30 (n = px or py): every rounded intermediate is normal (q >= 2^-120) and the residual, a
//    multiple of 2^-47 |n| >= 2^-137, is exactly representable (denormals are not flushed: Makefile), so the scaling
//    argument above holds and q is the correctly rounded quotient;
//  * |n| < 2^-90 (zero and denormals included): the exact and the short quotient are both below 2^-78 in magnitude
//    (r < 2^10 (1 + 2^-22), and the residual of a quotient rounded on the denormal grid is below 2^-119), so
//    |q K| < 2^-48 for K < 2^30, and  uf = q K + c  is  c  itself in both when |c| >= 2^-23 (the addend is below a
//    quarter of c's last place) and below 1 in magnitude in both otherwise: (int)uf and the picture test agree.
//    sc_selftest_division mode 2 samples that regime (the pixel and the picture test, not the quotient's bits).
// The picture test there is two unsigned comparisons of the truncated coordinates -- uf, vf are finite, v_cvt_i32_f32
// truncates toward zero ((-1, 0) -> 0, accepted like the reference's cast) and saturates, so (unsigned)u < W is
// exactly  uf > -1 && uf < W.
// ALL_SAFE: the host has certified EVERY view this kernel instance will see, so the test of d.safe and the general
// path behind it are compiled out: the projection is straight-line code.
// okm: the wavefront's mask of the lanes that return true (for the kernels that keep their per-lane state as masks
// in scalar registers: as the AND of the two comparisons' own masks it costs no vector instruction).
template <bool ALL_SAFE = false>
__device__ __forceinline__ bool project(float ax, float ay, float az, float z,
                                        const ViewDesc &d, int &u, int &v, unsigned long long &okm) {
    float pz = (az + d.R[8] * z) + d.t[2];  // :11
    float px = (ax + d.R[2] * z) + d.t[0];  // :17
    float py = (ay + d.R[5] * z) + d.t[1];  // :18
    if (ALL_SAFE || d.safe != 0) {  // wave-uniform
        const float r = refined_rcp(pz);
        // (the packed forms v_pk_mul/fma_f32 were tried for the two chains: no faster in these kernels)
        const float uf = div_by_rcp(px, pz, r) * d.K[0] + d.K[2];  // :20
        const float vf = div_by_rcp(py, pz, r) * d.K[1] + d.K[3];  // :21
        u = (int)uf;
        v = (int)vf;
        const bool in_u = (uint32_t)u < (uint32_t)d.W, in_v = (uint32_t)v < (uint32_t)d.H;
        okm = __builtin_amdgcn_ballot_w64(in_u) & __builtin_amdgcn_ballot_w64(in_v);
        return in_u & in_v;
    }
    // lanes whose operands the short division does not cover
    unsigned long long outside = __builtin_amdgcn_ballot_w64(!div_fast_range(px, py, pz));
    asm volatile("" : "+s"(outside));
    if (outside == 0) {
        const float r = refined_rcp(pz);
        const float uf = div_by_rcp(px, pz, r) * d.K[0] + d.K[2];  // :20
        const float vf = div_by_rcp(py, pz, r) * d.K[1] + d.K[3];  // :21
        u = (int)uf;
        v = (int)vf;
        const bool ok = (uf > -1.0f) & (uf < d.Wf) & (vf > -1.0f) & (vf < d.Hf);  // pz > 0 here
        okm = __builtin_amdgcn_ballot_w64(ok);
        return ok;
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
    okm = __builtin_amdgcn_ballot_w64(ok);
    return ok;
}
template <bool ALL_SAFE = false>
__device__ __forceinline__ bool project(float ax, float ay, float az, float z, const ViewDesc &d, int &u, int &v) {
    unsigned long long okm;
    return project<ALL_SAFE>(ax, ay, az, z, d, u, v, okm);
}

// The bit tiles of a carve mask lie STRIP BY STRIP (round 5's end; row by row of tiles until then): a strip is 32
// pixels wide and as tall as the picture rounded up to whole tiles, its rows one word each, one after the other -- a
// 128-byte line is the same 32 x 32-pixel tile as before, and pixel (u, v) is bit u & 31 of word
// (u >> 5) * strip + v: a shift, a 24-bit multiply-add and the scaling, where the tile order took four
// instructions with a v_bfi among them.  strip = 32 x tile rows < 2^24 (check_view_args), u >> 5 < 2^19.
__device__ __forceinline__ uint32_t mask_word_index(int u, int v, int strip) {
    return __umul24((uint32_t)u >> 5, (uint32_t)strip) + (uint32_t)v;
}

// The mask pointer comes out of a descriptor, so the compiler cannot tell its address space and
// would emit flat loads; it is always global memory.
typedef const __attribute__((address_space(1))) uint32_t *gmask_t;
__device__ __forceinline__ uint32_t load_mask_word(const void *mask, uint32_t word) {
    return ((gmask_t)(uintptr_t)mask)[word];
}

// The same as a BYTE offset, for loads of the form  scalar base + 32-bit lane offset  (no 64-bit address arithmetic per
// lane); a view's bits are below 2^32 bytes (H W <= 2^34).
__device__ __forceinline__ uint32_t mask_byte_offset(int u, int v, uint32_t strip) {
    return (__umul24((uint32_t)u >> 5, strip) + (uint32_t)v) << 2;
}
typedef const __attribute__((address_space(1))) char *gbytes_t;
__device__ __forceinline__ uint32_t load_mask_at(const void *mask, uint32_t byte_offset) {
    return *(gmask_t)((gbytes_t)(uintptr_t)mask + byte_offset);
}

// The occupancy bytes and cell maps hang off a descriptor too (d.occ, d.cmask): the same cast, or every one of their
// loads is a FLAT load -- the slow path that looks the address up in both apertures and holds up the scalar-memory
// and LDS counter as well as the vector one (round 4: 228 of them in the special kernel, 57 in the dense kernel).
typedef const __attribute__((address_space(1))) uint8_t *gbyte_t;
__device__ __forceinline__ uint32_t load_occ(const uint8_t *occ, uint32_t i) { return ((gbyte_t)(uintptr_t)occ)[i]; }
__device__ __forceinline__ uint32_t load_cells(const uint32_t *cmask, uint32_t i) { return ((gmask_t)(uintptr_t)cmask)[i]; }

typedef const __attribute__((address_space(1))) float *gfloat_t;
__device__ __forceinline__ uint32_t load_mask_byte(const void *mask, uint32_t i) { return ((gbyte_t)(uintptr_t)mask)[i]; }
__device__ __forceinline__ float load_mask_float(const void *mask, int64_t i) { return ((gfloat_t)(uintptr_t)mask)[i]; }

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
