// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- 1-byte masks -> bit tiles, tile occupancy and cell maps: the panel and the band form.

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

// 16 pixels (bytes) -> 16 bits, bit j = (pixel j != 0).  Round 5: bit 7 of every non-zero byte by the carry trick as
// above (and, add, one three-input v_bitop3), then the four flags of a dword are gathered -- and put at their place
// in the 16 bits -- by ONE v_dot4_u32_u8 with the weights 1 2 4 8 (or 16 .. 128), accumulating: 0x80 w = 128 w per
// flag, so the sum of a pair of dwords is 128 x their byte.  18 vector instructions for 16 pixels; the form above is
// 27, four of them v_mul_lo_u32, which issue at a quarter of the rate (tools/probes/valu_probe.hip).  The packers are
// half of the dense stage's vector instructions (the riders: 62 views of a batch of 72), and that stage is bound by
// what it issues.
__device__ __forceinline__ uint32_t nonzero_bits16(uint4 q) {
    const uint32_t t0 = (q.x | ((q.x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    const uint32_t t1 = (q.y | ((q.y & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    const uint32_t t2 = (q.z | ((q.z & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    const uint32_t t3 = (q.w | ((q.w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    const uint32_t lo = __builtin_amdgcn_udot4(t1, 0x80402010u, __builtin_amdgcn_udot4(t0, 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(t3, 0x80402010u, __builtin_amdgcn_udot4(t2, 0x08040201u, 0u, false), false);
    return ((hi << 8) + lo) >> 7;  // lo, hi: 128 x a byte
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
        uint32_t half = nonzero_bits16(make_uint4(q[k].x ^ flip, q[k].y ^ flip, q[k].z ^ flip, q[k].w ^ flip));
        uint32_t other = __shfl_xor(half, 1);
        uint32_t word = half | (other << 16);
        if ((c & 1) == 0 && tx < tiles_x && ty < tiles_y)
            pj.out[(int64_t)slot * pj.out_view_words + ((int64_t)tx * tiles_y + ty) * 32 + row] = word;  // strip tx, tile ty of it
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
    const int W = pj.W, H = pj.H, tiles_x = pj.tiles_x, tiles_y = pj.tiles_y;
    const uint32_t flip = pj.flip;
    const uint32_t tid = threadIdx.x;
    const int ty = (int)(b % (uint32_t)tiles_y);
    int slot = (int)(b / (uint32_t)tiles_y);
    if (slot >= pj.nslots) return;  // block-uniform
    slot += pj.slot0;
    const int64_t view = pj.use_order ? (int64_t)pj.order[slot] : (int64_t)slot;
    // the band's first byte (block-uniform: a scalar base) and 32-bit offsets from it: 32 rows of a picture are below
    // 2^31 bytes (check_pack_job: row_stride < 2^26)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(1))) uint4 *gq_t;  // (global loads: scalar base + 32-bit lane offset)
#else
    typedef const uint4 *gq_t;
#endif
    const char *raw = reinterpret_cast<const char *>(pj.raw) + view * pj.view_stride + (int64_t)ty * 32 * pj.row_stride;
    const uint32_t stride = (uint32_t)pj.row_stride;
    const uint32_t cpr = (uint32_t)(W >> 4);       // 16-pixel chunks per row (W % 16 == 0)
    const uint32_t cprp = 2u * (uint32_t)tiles_x;  // ... rounded up to an even number
    const uint32_t ntasks = 32u * cprp;
    const uint32_t rows_here = (uint32_t)min(32, H - ty * 32);
    // task q = tid + 256 i: row q / cprp, chunk q % cprp, stepped without a division
    uint32_t row = tid / cprp, c = tid - row * cprp;
    const uint32_t drow = (uint32_t)kBlock / cprp, dc = (uint32_t)kBlock - drow * cprp;
    for (uint32_t base = 0; base < ntasks; base += (uint32_t)kBlock * kBandPhase) {
        uint4 q[kBandPhase];
        uint32_t dst[kBandPhase];  // where the task's word goes in band_s (0xffffffff: nowhere)
#pragma unroll
        for (int i = 0; i < kBandPhase; ++i) {
            // (tasks past the band's end belong to nobody; the odd chunk of a pair hands its half to the even one)
            dst[i] = (row < 32u && (c & 1u) == 0u) ? (c >> 1) * 32u + row : 0xffffffffu;
            q[i] = make_uint4(flip, flip, flip, flip);  // padding stays background after the flip
            if (row < rows_here && c < cpr) q[i] = *(gq_t)(uintptr_t)(raw + (row * stride + c * 16u));
            row += drow; c += dc;
            if (c >= cprp) { c -= cprp; ++row; }
        }
#pragma unroll
        for (int i = 0; i < kBandPhase; ++i) {
            if (flip) q[i] = make_uint4(q[i].x ^ flip, q[i].y ^ flip, q[i].z ^ flip, q[i].w ^ flip);  // (block-uniform)
            const uint32_t half = nonzero_bits16(q[i]);
            const uint32_t other = __shfl_xor(half, 1);  // the task next door: same row, the tile's other half
            if (dst[i] != 0xffffffffu) band_s[dst[i]] = half | (other << 16);
        }
    }
    __syncthreads();
    // the band's tiles: tile tx of it is tile ty of strip tx in the packed arena (8 pieces of 16 bytes each)
    uint4 *out = reinterpret_cast<uint4 *>(pj.out + (int64_t)slot * pj.out_view_words + (int64_t)ty * 32);
    const uint4 *src = reinterpret_cast<const uint4 *>(band_s);
    const uint32_t strip4 = (uint32_t)pj.tiles_y * 8u;  // a strip in 16-byte pieces
    for (uint32_t i = tid; i < (uint32_t)tiles_x * 8u; i += kBlock) out[(size_t)(i >> 3) * strip4 + (i & 7u)] = src[i];
    // The 8x8-pixel cells from the finished words (round 5; until then every task worked out its two cells' flags and
    // sent them to LDS atomics: a third of the packer's instructions): thread 4 t + cy takes cell row cy of tile t --
    // the OR and the AND of its eight words, a byte of them per cell -- and the four threads of a tile join their
    // nibbles.  Padding is background (zero bits): a tile over the picture's edge is never FULL.
    for (uint32_t t4 = tid; t4 < (uint32_t)tiles_x * 4u; t4 += kBlock) {  // (kBandTiles * 4 == kBlock: one turn)
        const uint32_t t = t4 >> 2, cy = t4 & 3u;
        const uint4 a = src[t * 8u + cy * 2u], bq = src[t * 8u + cy * 2u + 1u];
        const uint32_t any = (a.x | a.y) | (a.z | a.w) | (bq.x | bq.y) | (bq.z | bq.w);
        const uint32_t all = (a.x & a.y) & (a.z & a.w) & (bq.x & bq.y) & (bq.z & bq.w);
        uint32_t f = 0u, g = 0u;
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) {
            f |= (((any >> (8 * cx)) & 0xffu) != 0u ? 1u : 0u) << cx;
            g |= (((all >> (8 * cx)) & 0xffu) != 0xffu ? 1u : 0u) << cx;
        }
        uint32_t cm = (f << (4u * cy)) | (g << (16u + 4u * cy));
        cm |= __shfl_xor(cm, 1);
        cm |= __shfl_xor(cm, 2);
        if (cy == 0u) {
            const int64_t tile = (int64_t)slot * tiles_x * tiles_y + (int64_t)ty * tiles_x + t;
            pj.occ[tile] = ((cm & 0xffffu) ? 1 : 0) | ((cm >> 16) ? 0 : 2);
            if (pj.cmask != nullptr) pj.cmask[tile] = cm;
        }
    }
}

__global__ __launch_bounds__(kBlock) void pack_band_kernel(PackJob pj) { pack_band_block(pj, blockIdx.x); }

// Masks that arrive from the HOST cross PCIe as bits already (hostpack.h: `pixel != 0` after the optional invert, on
// host threads, row-major, one word per 32 pixels, 0 beyond the picture).  What is left for the device is a pass over
// those bits: the words into the 32x32-tile order the carve kernels gather from, and per tile its occupancy byte and
// the 4x4 map of its 8x8-pixel cells (see ViewDesc) -- the same three products the byte packers above make.
// One record per view, in a table that travels with the bits; a wavefront takes two tiles (lane = tile half * 32 + row).
struct BitsRec {
    uint64_t src_off;  // of the view's bit rows in the arena
    uint32_t *tiles;
    uint8_t *occ;
    uint32_t *cmask;   // may be null
    int32_t W, H, tiles_x, tiles_y;
    uint64_t pad;
};
static_assert(sizeof(BitsRec) == 56, "BitsRec layout");

__global__ __launch_bounds__(kBlock) void bits_tiles_kernel(const char *__restrict__ arena, uint64_t table_off) {
    const BitsRec rec = reinterpret_cast<const BitsRec *>(arena + table_off)[blockIdx.y];  // block-uniform: scalar loads
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pair = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t ntiles = (uint32_t)rec.tiles_x * (uint32_t)rec.tiles_y;
    if (pair * 2u >= ntiles) return;  // wave-uniform
    const uint32_t t = pair * 2u + (lane >> 5), r = lane & 31u;
    const bool have = t < ntiles;
    const uint32_t ty = t / (uint32_t)rec.tiles_x, tx = t - ty * (uint32_t)rec.tiles_x;
    const uint32_t v = ty * 32u + r;
    const uint32_t *bits = reinterpret_cast<const uint32_t *>(arena + rec.src_off);
    uint32_t word = 0u;
    if (have && v < (uint32_t)rec.H) word = bits[(size_t)v * (uint32_t)rec.tiles_x + tx];
    if (have) rec.tiles[((size_t)tx * (uint32_t)rec.tiles_y + ty) * 32u + r] = word;  // strip tx, tile ty of it
    // cells: row of cells r >> 3, column of cells = the byte of the word
    unsigned long long anyf[4], anyb[4];
#pragma unroll
    for (int cx = 0; cx < 4; ++cx) {
        const uint32_t byte = (word >> (8 * cx)) & 0xffu;
        anyf[cx] = __ballot(byte != 0u);
        anyb[cx] = __ballot(byte != 0xffu);  // padding is background: a tile over the picture's edge is never FULL
    }
    if (r == 0u && have) {
        const uint32_t sh = lane;  // 0 or 32: this tile's 32 rows in the ballots
        uint32_t cm = 0u;
#pragma unroll
        for (int cy = 0; cy < 4; ++cy)
#pragma unroll
            for (int cx = 0; cx < 4; ++cx) {
                const uint32_t f = (uint32_t)((anyf[cx] >> (sh + 8u * cy)) & 0xffull), b = (uint32_t)((anyb[cx] >> (sh + 8u * cy)) & 0xffull);
                cm |= (f ? 1u : 0u) << (cy * 4 + cx);
                cm |= (b ? 1u : 0u) << (16 + cy * 4 + cx);
            }
        rec.occ[t] = ((cm & 0xffffu) ? 1 : 0) | ((cm >> 16) ? 0 : 2);
        if (rec.cmask != nullptr) rec.cmask[t] = cm;
    }
}
