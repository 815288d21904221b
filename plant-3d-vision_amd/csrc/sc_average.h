// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- average (backprojection.c:36-55): mask forms, flat-footprint verdicts, one label or several per launch.

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
// (round 5: strips as for the bytes below -- 8 pixels wide, a strip's rows one after the other, 32 bytes a row, so a
// 128-byte line is still an 8x4 tile; `strip` = the floats of a strip = 8 x its rows, ViewDesc::tiles_x holds it)
__device__ __forceinline__ uint32_t ftile_offset(int u, int v, int strip) {
    return __umul24((uint32_t)u >> 3, (uint32_t)strip) + (((uint32_t)v << 3) | ((uint32_t)u & 7u));
}

// Where pixel (u, v) of a uint8 averaging mask lies.  The picture is cut into STRIPS 16 pixels wide and as tall as
// the picture (rounded up to 8 rows); a strip is stored row after row, 16 bytes a row, so that a 128-byte line is
// still a 16x8-pixel tile -- what a wavefront's gather touches a handful of -- and the offset is
//     (u >> 4) * strip + v * 16 + (u & 15)          (strip = 16 bytes x the strip's rows; ViewDesc::tiles_x holds it)
// four vector instructions where the row-major order of the tiles ((v >> 3) * tiles_x + (u >> 4)) * 128 + (v & 7) * 16
// + (u & 15) took nine (round 5: the averaging kernel is bound by what it issues, 37 instructions per voxel.view).
// Both factors of the product are below 2^24 and the offset below 2^31: enqueue_tile8 refuses pictures beyond that
// (a million rows, or 2 GiB of pixels).
__device__ __forceinline__ uint32_t u8strip_offset(int u, int v, uint32_t strip) {
    return __umul24((uint32_t)u >> 4, strip) + (((uint32_t)v << 4) | ((uint32_t)u & 15u));
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
                uint32_t off = u8strip_offset(u, v, (uint32_t)d.tiles_x);
                uint32_t b = 0;
                if (ok[e]) b = load_mask_byte(m, off);
                add[e] = lut_s[b];
            }
        } else if (d.pad == 2) {  // float32 in 8x4 tiles
            const float *m = static_cast<const float *>(d.mask);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int u, v;
                ok[e] = project(ax, ay, az, z[e], d, u, v) & (e < (int)vx.nvalid);
                add[e] = 0.0f;
                if (ok[e]) add[e] = load_mask_float(m, (int64_t)ftile_offset(u, v, d.tiles_x));
            }
        } else {
            const float *m = static_cast<const float *>(d.mask);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int u, v;
                ok[e] = project(ax, ay, az, z[e], d, u, v) & (e < (int)vx.nvalid);
                add[e] = 0.0f;
                if (ok[e]) add[e] = load_mask_float(m, (int64_t)v * d.W + u);  // nearest texel
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

// uint8 [V][H][W] row-major -> strips of 16x8-pixel tiles (128 B each; u8strip_offset) for the averaging gather.
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
    uint8_t *dst = out + (int64_t)view * tiles_y * tiles_x * 128 + ((int64_t)c * tiles_y * 8 + v) * 16;  // strip c, row v
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
        const uint8_t *src = tiled + (int64_t)view * tiles_y * tiles_x * 128 + ((int64_t)c * tiles_y * 8 + v) * 16;  // strip c, row v
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
    float *dst = out + (int64_t)view * tiles_y * tiles_x * 32 + ftile_offset(c * 4, v, tiles_y * 32);
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
    const uint32_t first = base[ftile_offset(rx * 32, ry * 32, tiles_y * 32)];
    bool same = true;
    // lane l: row ry*32 + l/2, half l & 1 of the 32 columns
    const int v = ry * 32 + (lane >> 1);
    if (v < H) {
        for (int q = 0; q < 16; ++q) {
            const int u = rx * 32 + (lane & 1) * 16 + q;
            if (u < W) same &= base[ftile_offset(u, v, tiles_y * 32)] == first;
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
    // Verdict 5 (round 5): the footprint is MIXED -- the voxels have to be projected -- but lies wholly inside the picture
    // (rect_box: every voxel of the brick in front of the camera, its pixel inside the picture, rounding included):
    // the averaging kernel then projects without the picture test (backprojection.c:13, :23-31 hold for every voxel).
    bool all_inside = false;
    if (d.pad == 2) {  // tiled float32 mask: flat when every region under the brick holds one value
        uint32_t bits;
        uint32_t v = brick_flat_f32(d, g, x, (int)(by * kBrickY), (int)(bz * kBrickZ), bits, &all_inside);
        if (v == 0u && all_inside && d.safe != 0) v = 5u;
        verd[(size_t)lb * (uint32_t)nviews + vi] = (uint8_t)v;
        if (verdf != nullptr) verdf[(size_t)lb * (uint32_t)nviews + vi] = bits;
        return;
    }
    uint32_t v = brick_verdict(d, g, x, (int)(by * kBrickY), (int)(bz * kBrickZ), (d.W + 31) >> 5, &all_inside);
    if (v == 0u && all_inside && d.safe != 0) v = 5u;  // (d.safe: the short division needs no operand test, project())
    verd[(size_t)lb * (uint32_t)nviews + vi] = (uint8_t)v;
}

// The pixel of a voxel that is KNOWN to lie in front of a certified camera and inside the picture (verdict 5): the
// arithmetic of project()'s short path (backprojection.c:11-21, the correctly rounded shared-reciprocal division) without
// the tests of :13 and :23-31, which hold.
__device__ __forceinline__ void project_inside(float ax, float ay, float az, float z, const ViewDesc &d, int &u, int &v) {
    const float pz = (az + d.R[8] * z) + d.t[2];  // :11
    const float px = (ax + d.R[2] * z) + d.t[0];  // :17
    const float py = (ay + d.R[5] * z) + d.t[1];  // :18
    const float r = refined_rcp(pz);
    u = (int)(div_by_rcp(px, pz, r) * d.K[0] + d.K[2]);  // :20
    v = (int)(div_by_rcp(py, pz, r) * d.K[1] + d.K[3]);  // :21
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
            if (c != 0u && c != 5u) {
                const float add = c == 1u ? add0 : (c == 2u ? add255 : __uint_as_float(__builtin_amdgcn_readlane(minef, q)));
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = val[e] + add;  // :54, every voxel is in-image
                continue;
            }
            const ViewDesc d = views[v0 + q];
            const float ax = d.R[0] * x + d.R[1] * y, ay = d.R[3] * x + d.R[4] * y, az = d.R[6] * x + d.R[7] * y;
            if (c == 5u) {  // mixed, and every voxel of the brick inside the picture: no test, no branch around the gather
                if (d.pad == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int u, v;
                        project_inside(ax, ay, az, z[e], d, u, v);
                        val[e] = val[e] + load_mask_float(d.mask, (int64_t)ftile_offset(u, v, d.tiles_x));  // :54
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int u, v;
                        project_inside(ax, ay, az, z[e], d, u, v);
                        const uint32_t off = u8strip_offset(u, v, (uint32_t)d.tiles_x);
                        val[e] = val[e] + lut_s[load_mask_byte(d.mask, off)];  // :54
                    }
                }
                continue;
            }
            if (d.pad == 2) {  // tiled float32 mask (wave-uniform)
                const float *mf = static_cast<const float *>(d.mask);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int u, v;
                    const bool ok = project(ax, ay, az, z[e], d, u, v);
                    float add = 0.0f;
                    if (ok) add = load_mask_float(mf, (int64_t)ftile_offset(u, v, d.tiles_x));
                    if (ok) val[e] = val[e] + add;  // :54
                }
                continue;
            }
            const uint8_t *m = static_cast<const uint8_t *>(d.mask);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int u, v;
                const bool ok = project(ax, ay, az, z[e], d, u, v);
                const uint32_t off = u8strip_offset(u, v, (uint32_t)d.tiles_x);
                uint32_t b = 0;
                if (ok) b = load_mask_byte(m, off);
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
                    const uint32_t o = load_occ(a.views[l][vi].occ, (uint32_t)(ty * occ_tx + tx));
                    any[l] |= o;
                    all[l] &= o;
                }
            }
        // (5: mixed, but every voxel of the brick inside the picture -- see avg_flags_kernel)
        const uint32_t mixed = (fpr.inside && d.safe != 0) ? 5u : 0u;
#pragma unroll
        for (int l = 0; l < L; ++l) v[l] = (any[l] & 1u) == 0 ? 1u : ((all[l] & 2u) != 0 ? 2u : mixed);
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
            bool mixed = false, inside5 = false;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                c[l] = __builtin_amdgcn_readlane(mine[l], q);  // wave-uniform (brick-uniform)
                mixed |= c[l] == 0u || c[l] == 5u;
                inside5 |= c[l] == 5u;  // (a matter of the pose: the same for every label whose footprint is mixed)
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
                if (inside5) {  // wave-uniform: no picture test
                    project_inside(ax, ay, az, z[e], d, u, v);
                    ok[e] = true;
                } else {
                    ok[e] = project(ax, ay, az, z[e], d, u, v);
                }
                off[e] = u8strip_offset(u, v, (uint32_t)d.tiles_x);
            }
#pragma unroll
            for (int l = 0; l < L; ++l) {
                if (c[l] != 0u && c[l] != 5u) {  // flat for this label: every voxel is in-image and adds the one value
                    const float add = c[l] == 1u ? lut_s[l][0] : lut_s[l][255];
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[l][e] = val[l][e] + add;
                    continue;
                }
                const uint8_t *m = static_cast<const uint8_t *>(a.views[l][v0 + q].mask);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t b = 0;
                    if (ok[e]) b = load_mask_byte(m, off[e]);
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
