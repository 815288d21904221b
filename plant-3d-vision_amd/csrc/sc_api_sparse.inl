// Part of spacecarve.hip (included at its end): the host side of the brick-sparse label transport (sc_sparse.h) --
// sc_values_sparse and friends.  No reference counterpart (the reference is single-device, cl.py:29-30); what the
// consumer on the other end needs is what cl.py:229-232 returns: the labels.

namespace {

int sparse_geometry(int64_t planes, int64_t ny, int64_t nz, uint32_t *bys, uint32_t *bzs, uint64_t *nbricks) {
    if (planes < 1 || ny < 1 || nz < 1) return fail(SC_ERR_INVALID, "bad shape");
    const uint64_t by = (uint64_t)(ny + kBrickY - 1) / kBrickY, bz = (uint64_t)(nz + kBrickZ - 1) / kBrickZ;
    const uint64_t nb = (uint64_t)planes * by * bz;
    if (nb >= 0x40000000ull) return fail(SC_ERR_INVALID, "too many bricks for the sparse form");
    *bys = (uint32_t)by;
    *bzs = (uint32_t)bz;
    *nbricks = nb;
    return SC_OK;
}

uint32_t sparse_round_cap(uint64_t cap, uint64_t nbricks) {
    cap = std::min<uint64_t>(std::max<uint64_t>(cap, 16), nbricks);
    return (uint32_t)((cap + 15u) & ~(uint64_t)15u);  // (may exceed nbricks by < 16: whole 64-byte groups of ids)
}

// header r of a gathered buffer, checked against the buffer it lies in
int sparse_check_header(const SparseHeader &h, int64_t rank_bytes, int r) {
    if (h.magic != kSparseMagic || h.version != 1u || h.bits != 2u)
        return fail(SC_ERR_INVALID, "rank %d: not a sparse label buffer", r);
    if (h.nbricks == 0 || (uint64_t)h.planes * h.bricks_y * h.bricks_z != h.nbricks || h.stride == 0 ||
        h.bricks_y != (h.ny + kBrickY - 1) / kBrickY || h.bricks_z != (h.nz + kBrickZ - 1) / kBrickZ)
        return fail(SC_ERR_INVALID, "rank %d: inconsistent sparse header", r);
    if (sparse_layout(h.nbricks, h.cap).total > (uint64_t)rank_bytes)
        return fail(SC_ERR_INVALID, "rank %d: sparse buffer larger than the rank stride", r);
    return SC_OK;
}

}  // namespace

extern "C" {

int64_t sc_sparse_bricks(int64_t planes, int64_t ny, int64_t nz) {
    uint32_t bys, bzs;
    uint64_t nb;
    if (sparse_geometry(planes, ny, nz, &bys, &bzs, &nb)) return -1;
    return (int64_t)nb;
}

int64_t sc_sparse_rank_bytes(int64_t nbricks, int64_t cap) {
    if (nbricks < 1 || nbricks >= 0x40000000ll || cap < 0) return -1;
    return (int64_t)sparse_layout((uint32_t)nbricks, sparse_round_cap((uint64_t)cap, (uint64_t)nbricks)).total;
}

}  // extern "C"

namespace {

// min_bytes: the send buffer holds at least that much (an all-gather sends the stride of the rank with the most bricks)
int values_sparse(sc_engine *e, int64_t cap, int64_t min_bytes, void **ptr, int64_t *bytes) {
    if (!e || !ptr || !bytes) return fail(SC_ERR_INVALID, "null argument");
    if (e->mode != SC_MODE_CARVE) return fail(SC_ERR_STATE, "packed labels are carve labels");
    const int32_t init = init_bits_i32(e);
    if (init < -1 || init > 1 || (float)init != e->default_value)
        return fail(SC_ERR_STATE, "default_value %g is not one of -1, 0, 1: two bits cannot hold it", (double)e->default_value);
    if (cap < 0) return fail(SC_ERR_INVALID, "negative capacity");
    int rc = sc_flush(e);
    if (rc) return rc;
    rc = materialize(e);  // (a volume no view has touched: written out, then read like any other)
    if (rc) return rc;
    uint32_t bys, bzs;
    uint64_t nb64;
    rc = sparse_geometry(e->planes, e->ny, e->nz, &bys, &bzs, &nb64);
    if (rc) return rc;
    const uint32_t nbricks = (uint32_t)nb64;
    if (cap == 0) cap = e->sparse_cap > 0 ? e->sparse_cap : (int64_t)std::max<uint64_t>(1024, nb64 / 8);
    const uint32_t capb = sparse_round_cap((uint64_t)cap, nb64);
    e->sparse_cap = capb;
    const SparseLayout lay = sparse_layout(nbricks, capb);
    const int q = e->sparse_idx;
    const size_t need = std::max<size_t>(lay.total, (size_t)std::max<int64_t>(min_bytes, 0));
    if (e->sparse_bytes[q] < need) {
        // (the other buffer may still be read by a collective; this one was last read two calls ago, and the caller has
        // ordered that collective before this call: sc_all_gather_sparse, sc_order_after)
        HIP_TRY(schost::wait_stream(e->stream));
        if (e->sparse_buf[q]) (void)hipFree(e->sparse_buf[q]);
        e->sparse_buf[q] = nullptr;
        e->sparse_bytes[q] = 0;
        HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->sparse_buf[q]), need));
        e->sparse_bytes[q] = need;
        // what the pack kernel does not write (the padding of the codes, slots nobody took) travels too: zeroes the first
        // time (later: whatever an earlier call with another capacity left there; a reader goes by the header)
        HIP_TRY(hipMemsetAsync(e->sparse_buf[q], 0, need, e->stream));
    }
    if (!e->sparse_cnt) {
        HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->sparse_cnt), 2 * sizeof(SparseCounters)));
        HIP_TRY(hipMemsetAsync(e->sparse_cnt, 0, 2 * sizeof(SparseCounters), e->stream));
    }
    char *wire = e->sparse_buf[q];
    SparseCounters *cnt = e->sparse_cnt + (e->sparse_calls & 1u), *cnt_next = e->sparse_cnt + ((e->sparse_calls + 1) & 1u);
    SparseHeader hdr{};
    hdr.magic = kSparseMagic;
    hdr.version = 1;
    hdr.bits = 2;
    hdr.nbricks = nbricks;
    hdr.cap = capb;
    hdr.nmixed = 0;
    hdr.planes = (uint32_t)e->planes;
    hdr.ny = (uint32_t)e->ny;
    hdr.nz = (uint32_t)e->nz;
    hdr.bricks_y = bys;
    hdr.bricks_z = bzs;
    hdr.first = (uint32_t)e->i0;
    hdr.stride = (uint32_t)e->istride;
    const GridDesc g = grid_desc(e);
    const uint32_t nscan = (nbricks + kBlock - 1) / kBlock;
    const uint32_t code_full = (uint32_t)(init == 0 ? 1 : init) & 3u, code_untouched = (uint32_t)init & 3u;
    const int32_t *st = static_cast<const int32_t *>(e->state);
    const bool exact = e->sparse_exact && e->flags != nullptr && e->ctl != nullptr && e->live != nullptr;
    if (exact) {
        // ONE launch: the verdict bytes of the batch that made these labels settle most bricks, its live list (and the
        // two ends of its late list: candidates a late view rejected) name the others
        SparseScan sc{e->flags, nbricks, 2, code_full, code_untouched, nullptr};
        SparseLists sl{};
        sl.list0 = e->live;
        sl.count0 = &e->ctl->nlive[e->last_parity];
        sl.step0 = 1;
        if (e->sparse_late && e->late != nullptr) {
            sl.list1 = e->late;
            sl.count1 = &e->ctl->nlate;
            sl.step1 = 1;
            sl.list2 = e->late + (nbricks - 1u);
            sl.count2 = &e->ctl->nlate_units;
            sl.step2 = -1;
        }
        // 8 listed bricks per wavefront and turn, 32 per block: 512 blocks take 16 384 bricks in one turn (a plant's 7 577)
        const uint32_t npack = std::min<uint32_t>(512u, std::max<uint32_t>(16u, (nbricks + 31u) / 32u));
        hipLaunchKernelGGL(sparse_pack_kernel, dim3(nscan + npack), dim3(kBlock), 0, e->stream, st, g, bys, bzs, sc, nscan,
                           sl, wire, hdr, cnt, cnt_next);
    } else {
        // the labels' history is not one fused batch: bricks an earlier launch found empty are all -1 until the next
        // clear (the dead bytes), every other brick is read
        if (!e->sparse_work) HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->sparse_work), (size_t)nbricks * 4));
        const bool dead = e->dead != nullptr && e->dead_clean;
        SparseScan sc{dead ? e->dead : nullptr, nbricks, dead ? 1 : 0, code_full, code_untouched, e->sparse_work};
        SparseLists none{};
        hipLaunchKernelGGL(sparse_pack_kernel, dim3(nscan), dim3(kBlock), 0, e->stream, st, g, bys, bzs, sc, nscan, none,
                           wire, hdr, cnt, static_cast<SparseCounters *>(nullptr));
        SparseLists sl{};
        sl.list0 = e->sparse_work;
        sl.count0 = &cnt->nwork;
        sl.step0 = 1;
        SparseScan noscan{nullptr, nbricks, 0, 0u, 0u, nullptr};
        const uint32_t npack = std::min<uint32_t>(2048u, std::max<uint32_t>(16u, (nbricks + 31u) / 32u));
        hipLaunchKernelGGL(sparse_pack_kernel, dim3(npack), dim3(kBlock), 0, e->stream, st, g, bys, bzs, noscan, 0u, sl, wire,
                           hdr, cnt, cnt_next);
    }
    HIP_TRY(hipGetLastError());
    ++e->sparse_calls;
    e->sparse_idx ^= 1;
    *ptr = wire;
    *bytes = (int64_t)lay.total;
    return SC_OK;
}

}  // namespace

extern "C" {

int sc_values_sparse(sc_engine *e, int64_t cap, void **ptr, int64_t *bytes) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    return values_sparse(e, cap, 0, ptr, bytes);
}

int sc_get_values_sparse(sc_engine *e, int64_t cap, void *out, int64_t out_bytes) {
    if (!out) return fail(SC_ERR_INVALID, "null argument");
    void *ptr = nullptr;
    int64_t bytes = 0;
    int rc = sc_values_sparse(e, cap, &ptr, &bytes);
    if (rc) return rc;
    if (out_bytes < bytes) return fail(SC_ERR_INVALID, "output smaller than the sparse buffer (%lld bytes)", (long long)bytes);
    HIP_TRY(hipMemcpyAsync(out, ptr, (size_t)bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(schost::wait_stream(e->stream));
    return SC_OK;
}

int sc_sparse_headers(int device, void *hip_stream, void *done_event, const void *recv_dev, int64_t rank_bytes, int world,
                      uint32_t *nmixed, uint32_t *cap) {
    if (!recv_dev || !nmixed || !cap) return fail(SC_ERR_INVALID, "null argument");
    if (world < 1 || world > 4096 || rank_bytes < 64) return fail(SC_ERR_INVALID, "bad world / stride");
    HIP_TRY(hipSetDevice(device));
    // A page-locked landing place and a stream of the library's own, per device: with `done_event` (what
    // sc_all_gather_sparse hands out) the host waits for THAT collective and copies beside whatever the collectives'
    // stream has been given since -- the next batch's collective, which waits for the next batch's carve
    static std::mutex mu;
    static SparseHeader *pin[64] = {nullptr};
    static hipStream_t copy_stream[64] = {nullptr};
    if (device < 0 || device >= 64) return fail(SC_ERR_INVALID, "device %d", device);
    // (the wait for the collective comes first and outside the lock: another thread's headers need not queue behind it)
    if (done_event) HIP_TRY(schost::wait_event(static_cast<hipEvent_t>(done_event)));
    std::lock_guard<std::mutex> lock(mu);
    if (!pin[device]) {
        HIP_TRY(sc_pin_malloc(reinterpret_cast<void **>(&pin[device]), 4096 * sizeof(SparseHeader), hipHostMallocDefault));
        HIP_TRY(hipStreamCreateWithFlags(&copy_stream[device], hipStreamNonBlocking));
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (done_event) st = copy_stream[device];
    // one strided copy of the W headers
    HIP_TRY(hipMemcpy2DAsync(pin[device], sizeof(SparseHeader), recv_dev, (size_t)rank_bytes, sizeof(SparseHeader), (size_t)world,
                             hipMemcpyDeviceToHost, st));
    HIP_TRY(schost::wait_stream(st));
    for (int r = 0; r < world; ++r) {
        int rc = sparse_check_header(pin[device][r], rank_bytes, r);
        if (rc) return rc;
        nmixed[r] = pin[device][r].nmixed;
        cap[r] = pin[device][r].cap;
    }
    return SC_OK;
}

int sc_unpack_sparse(int device, void *hip_stream, const void *recv_dev, int64_t rank_bytes, int world, int64_t nx, int64_t ny,
                     int64_t nz, void *out_dev, int out_kind) {
    if (!recv_dev || !out_dev) return fail(SC_ERR_INVALID, "null argument");
    if (out_kind != 0 && out_kind != 1 && out_kind != 4)
        return fail(SC_ERR_INVALID, "output: 0 uint8 occupancy (label == 1), 1 int8 labels, 4 int32 labels");
    if (world < 1 || nx < world || ny < 1 || nz < 1 || rank_bytes < 64) return fail(SC_ERR_INVALID, "bad shape / world / stride");
    uint32_t bys, bzs;
    uint64_t nbmax;
    int rc = sparse_geometry((nx + world - 1) / world, ny, nz, &bys, &bzs, &nbmax);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    // (the kernel checks every header against rank_bytes before it follows it: a buffer that is not what it should be
    // leaves its bricks unwritten instead of reading out of bounds; sc_sparse_headers is the call that reports it)
    const uint64_t cap_max = std::min<uint64_t>(nbmax + 15, (uint64_t)rank_bytes / (kSparseBrickBytes + 4));
    SparseIn in{static_cast<const char *>(recv_dev), (uint64_t)rank_bytes, (uint32_t)world, (uint32_t)nx, 0, (uint32_t)ny,
                (uint32_t)nz, bys, bzs, (uint32_t)nbmax, (uint32_t)std::max<uint64_t>(cap_max, 1)};
    const uint64_t wpb = kBlock / 64;
    const uint64_t nfill = ((uint64_t)world * nbmax + wpb - 1) / wpb, nslot = ((uint64_t)world * in.cap_max + wpb - 1) / wpb;
    if (nfill + nslot > 0x7fffffffULL) return fail(SC_ERR_INVALID, "grid too large for one launch");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const dim3 grid((uint32_t)(nfill + nslot));
    if (out_kind == 0)
        hipLaunchKernelGGL((sparse_unpack_kernel<uint8_t, true>), grid, dim3(kBlock), 0, st, in, static_cast<uint8_t *>(out_dev), (uint32_t)nfill);
    else if (out_kind == 1)
        hipLaunchKernelGGL((sparse_unpack_kernel<int8_t, false>), grid, dim3(kBlock), 0, st, in, static_cast<int8_t *>(out_dev), (uint32_t)nfill);
    else
        hipLaunchKernelGGL((sparse_unpack_kernel<int32_t, false>), grid, dim3(kBlock), 0, st, in, static_cast<int32_t *>(out_dev), (uint32_t)nfill);
    HIP_TRY(hipGetLastError());
    return SC_OK;
}

// Host code only (no device): `world` ranks' sparse buffers in host memory into ONE int32 grid out[nx][ny][nz] in global
// order -- the host end of ShardedBackprojection.gather_to_host (cl.py:229-232 of a sharded run).  On the host pool.
int sc_widen_sparse_ranks(const void *packed, int64_t rank_bytes, int world, int64_t nx, int64_t ny, int64_t nz, int32_t *out) {
    if (!packed || !out) return fail(SC_ERR_INVALID, "null argument");
    if (world < 1 || nx < 1 || ny < 1 || nz < 1 || rank_bytes < 64) return fail(SC_ERR_INVALID, "bad shape / world / stride");
    const char *base0 = static_cast<const char *>(packed);
    std::vector<std::vector<uint32_t>> slot((size_t)world);
    int64_t planes_seen = 0;
    for (int r = 0; r < world; ++r) {
        SparseHeader h;
        memcpy(&h, base0 + (int64_t)r * rank_bytes, sizeof h);
        int rc = sparse_check_header(h, rank_bytes, r);
        if (rc) return rc;
        if (h.ny != (uint32_t)ny || h.nz != (uint32_t)nz || (uint64_t)h.first + (uint64_t)(h.planes - 1) * h.stride >= (uint64_t)nx)
            return fail(SC_ERR_INVALID, "rank %d: its planes do not fit the grid", r);
        if (h.nmixed > h.cap) return fail(SC_ERR_STATE, "rank %d: %u mixed bricks for a capacity of %u", r, h.nmixed, h.cap);
        planes_seen += h.planes;
        const SparseLayout lay = sparse_layout(h.nbricks, h.cap);
        const uint32_t *ids = reinterpret_cast<const uint32_t *>(base0 + (int64_t)r * rank_bytes + lay.ids);
        slot[(size_t)r].assign(h.nbricks, 0xffffffffu);
        for (uint32_t s = 0; s < h.nmixed; ++s) {
            if (ids[s] >= h.nbricks) return fail(SC_ERR_INVALID, "rank %d: slot %u names brick %u", r, s, ids[s]);
            slot[(size_t)r][ids[s]] = s;
        }
    }
    if (planes_seen != nx) return fail(SC_ERR_INVALID, "the ranks hold %lld planes of %lld", (long long)planes_seen, (long long)nx);
    std::atomic<int> bad{0};
    for (int r = 0; r < world; ++r) {
        const char *base = base0 + (int64_t)r * rank_bytes;
        SparseHeader h;
        memcpy(&h, base, sizeof h);
        const SparseLayout lay = sparse_layout(h.nbricks, h.cap);
        const uint8_t *codes = reinterpret_cast<const uint8_t *>(base + lay.codes);
        const uint32_t *payload = reinterpret_cast<const uint32_t *>(base + lay.payload);
        const std::vector<uint32_t> &sl = slot[(size_t)r];
        // a task = one column-strip of bricks (a plane's bricks_z bricks at one by): rows of the output in order
        schost::parallel_for((int)(h.planes * h.bricks_y), [&](int strip) {
            const uint32_t il = (uint32_t)strip / h.bricks_y, by = (uint32_t)strip % h.bricks_y;
            const int64_t i = (int64_t)h.first + (int64_t)il * h.stride;
            for (uint32_t bz = 0; bz < h.bricks_z; ++bz) {
                const uint32_t b = ((uint32_t)il * h.bricks_y + by) * h.bricks_z + bz;
                const uint32_t code = codes[b];
                const uint32_t *words = nullptr;
                if (code == kSparseMixed) {
                    if (sl[b] == 0xffffffffu) { bad.store(1); continue; }
                    words = payload + (size_t)sl[b] * 64u;
                } else if (code > 3u) { bad.store(1); continue; }
                const int32_t uni = code == 3u ? -1 : (int32_t)code;
                for (uint32_t jl = 0; jl < (uint32_t)kBrickY; ++jl) {
                    const int64_t j = (int64_t)by * kBrickY + jl;
                    if (j >= ny) break;
                    const int64_t k0 = (int64_t)bz * kBrickZ, kn = std::min<int64_t>(kBrickZ, nz - k0);
                    int32_t *dst = out + (i * ny + j) * nz + k0;
                    if (words == nullptr) {
                        for (int64_t k = 0; k < kn; ++k) dst[k] = uni;
                    } else {
                        for (int64_t k = 0; k < kn; ++k) {
                            const uint32_t v = jl * 64u + (uint32_t)k;
                            dst[k] = (int32_t)(words[v >> 4] << (30 - 2 * (int)(v & 15u))) >> 30;
                        }
                    }
                }
            }
        });
    }
    if (bad.load()) return fail(SC_ERR_INVALID, "a mixed brick without a slot, or an unknown code");
    return SC_OK;
}

}  // extern "C"
