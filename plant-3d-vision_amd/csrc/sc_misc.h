// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- self-tests, narrowing / packing / unpacking of labels, de-pitching, fills, the generic mask packer.

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
        if (mode == 2) {  // certified views: numerators the short division is not proved for (|n| <= 2^-40, zero and
                          // denormals included) -- the PIXEL and the picture test must be those of the exact quotient
            n1 = (e & 0x80000000u) ? __uint_as_float(a & 0x80000000u) : __uint_as_float((a & 0x807fffffu) | ((e & 127u) % 88u) << 23);
            dd = __uint_as_float((c & 0x007fffffu) | ((117u + ((e >> 10) & 63u) % 40u) << 23));
            const float k = __uint_as_float((b & 0x807fffffu) | (((e >> 16) & 255u) % 157u) << 23);
            const uint32_t h = mix32(e ^ 0x51ed270bu);
            const float cc = __uint_as_float((h & 0x807fffffu) | (((h >> 23) & 255u) % 157u) << 23);
            const int W = 1 + (int)(mix32(h) & 0xffffffu);
            ++fast;
            const float r = refined_rcp(dd);
            const float uf = div_by_rcp(n1, dd, r) * k + cc, wf = (n1 / dd) * k + cc;
            const int u = (int)uf, w = (int)wf;
            const bool oku = (uint32_t)u < (uint32_t)W, okw = (wf > -1.0f) & (wf < (float)W);
            bad += (oku != okw) || (okw && u != w);
            continue;
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
            d.safe = r.pad[0]; d.strip = 0;  // certified by the host for the box the samples come from
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
    const size_t strip = (size_t)tiles_y * 32u;  // tile (tx, ty) at word tx * strip + 32 ty: row v of strip tx at tx * strip + v
    uint8_t *oc = occ + (int64_t)view * tiles_x * tiles_y;  // zeroed by the host; racing stores all write 1
    if (lane == 0) {
        o[(size_t)(seg * 2) * strip + (uint32_t)v] = (uint32_t)vote;
        if ((uint32_t)vote) oc[base + seg * 2] = 1;
    } else if (lane == 32 && seg * 2 + 1 < tiles_x) {
        o[(size_t)(seg * 2 + 1) * strip + (uint32_t)v] = (uint32_t)(vote >> 32);
        if ((uint32_t)(vote >> 32)) oc[base + seg * 2 + 1] = 1;
    }
}
