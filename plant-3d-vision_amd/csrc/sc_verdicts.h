// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- bricks, the footprint bound of DESIGN.md 4b, tile- and cell-level verdicts, brick_flags_kernel.

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
    // The four corners (j0 | j1) x (k0 | k1) share their products: per row of R one with x, two with y, two with z
    // -- 15 multiplications where corner by corner there are 36 -- and every corner's sums are the reference's own,
    // in its order, ((R0 x + R1 y) + R2 z) + t (backprojection.c:11,17,18).  The rounding-error bound of a row,
    // 8 x its worst case, is (|R0 x| + |R1 y| + |R2 z| + |t|) 2^-19 at the corner where that is largest: float
    // addition is monotone in each operand, so that is the sum of the larger |R1 y| and the larger |R2 z| -- the
    // same number the four sums' maximum gives, for a quarter of the additions.
    const float y0 = g.oy + (float)j0 * g.vs, y1 = g.oy + (float)j1 * g.vs;  // :72
    const float z0 = g.oz + (float)k0 * g.vs, z1 = g.oz + (float)k1 * g.vs;  // :73
    float p[3][4], e[3];  // rows: depth, x, y; corners: (y0,z0) (y0,z1) (y1,z0) (y1,z1)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int q = r == 0 ? 6 : (r == 1 ? 0 : 3);
        const float tt = d.t[r == 0 ? 2 : (r == 1 ? 0 : 1)];
        const float a = d.R[q] * x, b0 = d.R[q + 1] * y0, b1 = d.R[q + 1] * y1, c0 = d.R[q + 2] * z0, c1 = d.R[q + 2] * z1;
        const float s0 = a + b0, s1 = a + b1;
        p[r][0] = (s0 + c0) + tt;
        p[r][1] = (s0 + c1) + tt;
        p[r][2] = (s1 + c0) + tt;
        p[r][3] = (s1 + c1) + tt;
        e[r] = (((fabsf(a) + fmaxf(fabsf(b0), fabsf(b1))) + fmaxf(fabsf(c0), fabsf(c1))) + fabsf(tt)) * 0x1p-19f;
    }
    const float ez = e[0], ex = e[1], ey = e[2];
    float qxm = 0.0f, qym = 0.0f;
    float pzmin = INFINITY, pzmax = -INFINITY, umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    bool nan = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float pz = p[0][c], px = p[1][c], py = p[2][c];
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
    bool inside;   // see PixelBox: every voxel in front of the camera and inside the picture (whatever the tiles hold)
};

__device__ __forceinline__ Footprint brick_footprint(const ViewDesc &d, const GridDesc &g, float x, int j0, int k0) {
    Footprint fpr{0, 0, 0, 0, false, false, false};
    const PixelBox bx = rect_box(d, g, x, j0, j0 + kBrickY - 1, k0, k0 + kBrickZ - 1);
    fpr.outside = bx.outside;
    fpr.inside = bx.inside;
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
            if (a < nyw && b < nxw) w[a][b] = load_cells(d.cmask, (uint32_t)((ty0 + a) * d.tiles_x + tx0 + b));
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
                                                  int k0, int occ_tx, bool *all_inside = nullptr) {
    const Footprint fpr = brick_footprint(d, g, x, j0, k0);
    if (all_inside != nullptr) *all_inside = fpr.inside;  // (for the averaging kernel: no voxel needs its picture test)
    if (fpr.outside) return 4u;  // OUTSIDE: the view does nothing to the brick
    if (!fpr.ok) return 0u;
    uint32_t any = 0, all = 3;
    if (occ_tx >= 8) {
        // A tile row's bytes lie side by side: EIGHT of them in one (unaligned) load, four tile rows per turn -- the
        // footprint of a brick is 2-4 tiles wide and 4-7 tall, and asked for byte by byte its 12-18 loads are what
        // the flags and confirm kernels spend their time on (each lane's byte in a cache line of its own).  The
        // eight start at the footprint's first column, or further left when that would run past the row's end; the
        // bytes outside the footprint are masked out.  A row looked at twice changes neither the OR nor the AND.
        typedef const __attribute__((address_space(1), aligned(1))) unsigned long long *grow_t;
        unsigned long long any8 = 0ull, all8 = ~0ull;
        for (int cx = fpr.tx0; cx <= fpr.tx1; cx += 8) {
            const int start = min(cx, occ_tx - 8), off = cx - start, ncols = min(fpr.tx1 - cx + 1, 8);
            const unsigned long long m = (ncols >= 8 ? ~0ull : ((1ull << (8 * ncols)) - 1ull)) << (8 * off);
            for (int ty = fpr.ty0; ty <= fpr.ty1; ty += 4) {
                unsigned long long w[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) w[a] = *(grow_t)(uintptr_t)(d.occ + (uint32_t)(min(ty + a, fpr.ty1) * occ_tx + start));
                any8 |= ((w[0] | w[1]) | (w[2] | w[3])) & m;
                all8 &= ((w[0] & w[1]) & (w[2] & w[3])) | ~m;
            }
        }
        any = (any8 & 0x0101010101010101ull) != 0ull ? 1u : 0u;
        all = (all8 & 0x0202020202020202ull) == 0x0202020202020202ull ? 2u : 0u;
    } else {
        for (int ty = fpr.ty0; ty <= fpr.ty1; ++ty)
            for (int tx = fpr.tx0; tx <= fpr.tx1; ++tx) {
                uint32_t o = load_occ(d.occ, (uint32_t)(ty * occ_tx + tx));
                any |= o;
                all &= o;
            }
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
                                                   uint32_t &bits, bool *all_inside = nullptr) {
    const Footprint fpr = brick_footprint(d, g, x, j0, k0);
    if (all_inside != nullptr) *all_inside = fpr.inside;
    bits = 0u;
    if (fpr.outside) return 4u;  // the view adds nothing to the brick
    if (!fpr.ok) return 0u;
    const int otx = (d.W + 31) >> 5, oty = (d.H + 31) >> 5;
    const uint32_t *val = reinterpret_cast<const uint32_t *>(d.occ + (((size_t)otx * oty + 3) & ~(size_t)3));
    const uint32_t first = load_cells(val, (uint32_t)(fpr.ty0 * otx + fpr.tx0));
    bool flat = true;
    for (int ty = fpr.ty0; ty <= fpr.ty1; ++ty)
        for (int tx = fpr.tx0; tx <= fpr.tx1; ++tx)
            flat &= load_occ(d.occ, (uint32_t)(ty * otx + tx)) != 0 && load_cells(val, (uint32_t)(ty * otx + tx)) == first;
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
// The fill that waits for nobody (round 4).  Every label of the batch is written exactly once, three quarters of
// them the -1 of bricks some view finds empty -- and that fill, 537 MB at 512^3, is what bounds the final survivor
// stage, while HBM idles under the kernels in front (packing, these verdicts).  In a FRESH volume (labels exist as
// `init` in name only since the last clear) a brick's labels are whatever its verdict makes the later kernels
// write -- the dense stage writes every voxel of a live brick, the store blocks `kept` / `init` over FULL /
// UNTOUCHED ones, late_unit every voxel of a late one -- so the first `bytes` of the volume can be set to -1 HERE,
// before any verdict exists: the EMPTY bricks among them are then done, and the others are overwritten by kernels
// that come later on the stream.  `nblocks` persistent blocks in front of the verdict blocks write them in address
// order, 8 KB per block and turn (the order a plain fill kernel uses: 6.4-7.0 TB/s, tools/probes/fill_probe.hip).
struct SpecFill { int32_t *labels; uint64_t bytes; uint32_t nblocks; };

__global__ __launch_bounds__(64 * kFlagWaves) void brick_flags_kernel(
    GridDesc g, const ViewDesc *__restrict__ views, int nviews, uint32_t bricks_y, uint32_t bricks_z,
    uint32_t nbricks, uint8_t *__restrict__ flags, uint32_t *__restrict__ live, ListCtl *ctl,
    FlagViews own, DescCopy dc, const ViewDesc *__restrict__ allviews, int nall, int nbatch,
    uint8_t *__restrict__ dead, int dead_stale, uint32_t parity, uint32_t *__restrict__ fill_list, SpecFill sf,
    uint32_t *__restrict__ cands, uint32_t cand_per) {
    __shared__ unsigned long long s_empty[kFlagWaves], s_full[kFlagWaves], s_seen[kFlagWaves];
    if (blockIdx.x < sf.nblocks) {  // block-uniform
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i minus = {-1, -1, -1, -1};
        char *base = reinterpret_cast<char *>(sf.labels);
        for (uint64_t off = ((uint64_t)blockIdx.x * (64 * kFlagWaves) + threadIdx.x) * 16u; off < sf.bytes;
             off += (uint64_t)sf.nblocks * (64 * kFlagWaves * 16))
            __builtin_nontemporal_store(minus, reinterpret_cast<v4i *>(base + off));
        return;
    }
    const uint32_t bid = blockIdx.x - sf.nblocks;
    if (bid == 0) {
        for (uint32_t i = threadIdx.x; i < dc.words; i += 64 * kFlagWaves) dc.dst[i] = dc.src[i];
        if (threadIdx.x == 0) ctl->nlive[parity ^ 1u] = ctl->nfill[parity ^ 1u] = 0u;  // the next launch's counters
        if (threadIdx.x < (uint32_t)kCandSub) ctl->ncand[parity ^ 1u][threadIdx.x].n = 0u;
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t lb = bid * 64u + lane;
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
    if (valid && dead != nullptr && (gone || dead_stale)) dead[lb] = gone ? 1 : 0;
    // the bricks left go on the live list, one atomic per block
    const bool alive = valid && !gone && !kept;
    const unsigned long long m = __ballot(alive);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (m != 0) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ctl->nlive[parity], (uint32_t)__popcll(m));
        base = __shfl(base, 0);
        if (alive) live[base + (uint32_t)__popcll(m & below)] = lb;
    }
    // ... and, while later views are not packed yet (grid-uniform), the kept ones on the candidate list, whose 64-entry
    // groups the confirm kernel asks those views about with every lane at work (asked where they lie, a block of 64
    // bricks with five candidates cost what one with 64 costs); sub-list = this block's eighth of the grid
    const bool open = cands != nullptr && nall < nbatch && inb && !isdead && !gone && kept;
    const unsigned long long mc = __ballot(open);
    if (mc != 0) {
        const uint32_t sub = bid / cand_per;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ctl->ncand[parity][sub].n, (uint32_t)__popcll(mc));
        base = __shfl(base, 0);
        if (open) cands[(size_t)sub * cand_per * 64u + base + (uint32_t)__popcll(mc & below)] = lb;
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
