// Part of spacecarve.hip (included there): the C ABI, second half -- sc_flush .. sc_get_values*, the packed label forms,
// statistics, self-tests, host and device memory helpers.

extern "C" {

int sc_flush(sc_engine *e) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    return flush(e);
}

int sc_synchronize(sc_engine *e) {
    int rc = sc_flush(e);
    if (rc) return rc;
    HIP_TRY(schost::wait_stream(e->stream));
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
    HIP_TRY(schost::wait_stream(e->stream));
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
    if (!e->narrow) HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->narrow), (size_t)e->n));
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
    HIP_TRY(schost::wait_stream(e->stream));
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
    if (!e->packed_labels) {
        // room for one more plane than the engine owns: in an all-gather every rank sends the stride of the rank with
        // the most planes (sc_all_gather_packed)
        const size_t cap = (size_t)sc_packed_bytes(e->n + e->ny * e->nz, 2), own = (size_t)sc_packed_bytes(e->n, 2);
        HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->packed_labels), cap));
        // the tail of the last 16-byte group lies behind the last word the pack kernel writes and travels with the
        // buffer (all-gather, read-back), and so does the slack: zero once, never garbage
        HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(e->packed_labels) + (own - 16), 0, cap - (own - 16), e->stream));
        e->packed_cap = cap;
    }
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

int sc_hostpack_bits(const void *mask, int H, int W, int mask_dtype, int64_t row_stride_bytes, uint32_t *out) {
    if (!mask || !out || H <= 0 || W <= 0) return fail(SC_ERR_INVALID, "bad argument");
    if (mask_dtype != SC_MASK_U8 && mask_dtype != SC_MASK_I32 && mask_dtype != SC_MASK_U8_INV && mask_dtype != SC_MASK_BOOL_INV)
        return fail(SC_ERR_INVALID, "mask dtype %d has no bit form", mask_dtype);
    const int elem = mask_dtype == SC_MASK_I32 ? 4 : 1;
    if (row_stride_bytes == 0) row_stride_bytes = (int64_t)W * elem;
    if (row_stride_bytes < (int64_t)W * elem) return fail(SC_ERR_INVALID, "row stride smaller than a row");
    const uint8_t flip = mask_dtype == SC_MASK_U8_INV ? 255 : mask_dtype == SC_MASK_BOOL_INV ? 1 : 0;
    const int wpr = (W + 31) / 32, band = 64, nparts = (H + band - 1) / band;
    schost::parallel_for(nparts, [&](int part) {
        schost::pack_rows(mask, row_stride_bytes, W, part * band, std::min(H, part * band + band), out, wpr, elem, flip);
    });
    return SC_OK;
}

int sc_widen_labels2(const uint32_t *packed, int64_t voxels, int32_t *out, int threads) {
    if (!packed || !out || voxels < 0) return fail(SC_ERR_INVALID, "bad argument");
    (void)threads;  // the library's host pool does it (SC_OPT_HOST_THREADS)
    const int64_t words = (voxels + 15) / 16, piece = (int64_t)1 << 16;
    const int nparts = (int)std::min<int64_t>((words + piece - 1) / piece, 1 << 20);
    schost::parallel_for(nparts, [&](int part) {
        schost::widen2(packed, out, part * piece, std::min(words, (part + 1) * piece), voxels);
    });
    return SC_OK;
}

int sc_widen_labels2_ranks(const uint32_t *packed, int64_t rank_bytes, int world, int partition, int64_t nx, int64_t ny,
                           int64_t nz, int32_t *out) {
    if (!packed || !out) return fail(SC_ERR_INVALID, "null argument");
    if (partition != 0 && partition != 1) return fail(SC_ERR_INVALID, "partition: 0 plane-cyclic, 1 slabs");
    if (world < 1 || nx < world || ny < 1 || nz < 1 || rank_bytes < 0 || (rank_bytes & 3)) return fail(SC_ERR_INVALID, "bad shape / world / stride");
    const int64_t plane = ny * nz, pmax = (nx + world - 1) / world;
    if (rank_bytes * 4 < pmax * plane) return fail(SC_ERR_INVALID, "rank stride too small for its planes");
    const int64_t rw = rank_bytes / 4;
    if (nx > (1 << 30)) return fail(SC_ERR_INVALID, "too many planes");
    schost::parallel_for((int)nx, [&](int i) {
        int64_t r, p;
        if (partition == 0) {
            r = i % world;
            p = i / world;
        } else {
            r = ((int64_t)i * world + world - 1) / nx;
            while (nx * r / world > i) --r;
            while (nx * (r + 1) / world <= i) ++r;
            p = i - nx * r / world;
        }
        const uint32_t *src = packed + r * rw;
        const int64_t l0 = p * plane;  // first label of the plane in the rank's stream
        int32_t *dst = out + (int64_t)i * plane;
        if ((l0 & 15) == 0) {
            // whole words from a word boundary: the fast loop, with dst shifted so that label l lands at dst[l - l0]
            schost::widen2(src, dst - l0, l0 / 16, (l0 + plane + 15) / 16, l0 + plane);
        } else {
            for (int64_t q = 0; q < plane; ++q) {
                const int64_t l = l0 + q;
                dst[q] = (int32_t)(src[l >> 4] << (30 - 2 * (int)(l & 15))) >> 30;
            }
        }
    });
    return SC_OK;
}

int sc_get_values_wire2(sc_engine *e, int32_t *out, void *staging, int64_t staging_bytes, int threads) {
    if (!e || !out) return fail(SC_ERR_INVALID, "null argument");
    (void)threads;
    (void)staging;        // (rounds 3: the caller's pageable buffer; a copy into pageable memory runs at ~13 GB/s, a
    (void)staging_bytes;  //  quarter of what the page-locked buffer the engine now keeps gets)
    void *ptr = nullptr;
    int64_t bytes = 0;
    int rc = sc_values_packed(e, 2, &ptr, &bytes);
    if (rc) return rc;
    const int64_t n = e->n, words = (n + 15) / 16;
    if (e->wire_stage_words < (size_t)words) {
        if (e->wire_stage) (void)hipHostFree(e->wire_stage);
        e->wire_stage = nullptr;
        e->wire_stage_words = 0;
        HIP_TRY(sc_pin_malloc(reinterpret_cast<void **>(&e->wire_stage), (size_t)words * 4, hipHostMallocDefault));
        e->wire_stage_words = (size_t)words;
    }
    // Pieces of 1 MiB of packed labels (16 MiB of int32): every copy is put on the stream at once, an event behind
    // each; this thread waits for the events in turn and hands each landed piece to the host pool, whose workers
    // widen it while the next ones are on their way.  Nobody spins.
    const int64_t piece = (int64_t)1 << 18;  // words
    const int64_t npieces = (words + piece - 1) / piece;
    uint32_t *stg = e->wire_stage;
    std::vector<hipEvent_t> evs((size_t)npieces, nullptr);
    hipError_t err = hipSuccess;
    int64_t queued = 0;
    for (int64_t k = 0; k < npieces && err == hipSuccess; ++k) {
        const int64_t w0 = k * piece, w1 = std::min(words, (k + 1) * piece);
        if (get_event(e, &evs[(size_t)k]) != SC_OK) { err = hipErrorOutOfMemory; break; }
        ++queued;
        err = hipMemcpyAsync(stg + w0, static_cast<const uint32_t *>(ptr) + w0, (size_t)(w1 - w0) * 4, hipMemcpyDeviceToHost, e->stream);
        if (err == hipSuccess) err = hipEventRecord(evs[(size_t)k], e->stream);
    }
    {
        schost::TaskGroup tg;
        for (int64_t k = 0; k < queued && err == hipSuccess; ++k) {
            err = schost::wait_event(evs[(size_t)k]);
            if (err != hipSuccess) break;
            const int64_t w0 = k * piece, w1 = std::min(words, (k + 1) * piece);
            // two halves per piece: a finer grain for the pool at the transfer's end
            const int64_t mid = w0 + (w1 - w0) / 2;
            tg.submit([=]() { schost::widen2(stg, out, w0, mid, n); });
            tg.submit([=]() { schost::widen2(stg, out, mid, w1, n); });
        }
        tg.wait();
    }
    (void)schost::wait_stream(e->stream);  // (every copy has landed or failed before the events go back)
    for (int64_t k = 0; k < queued; ++k) e->event_pool.push_back(evs[(size_t)k]);
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
    HIP_TRY(schost::wait_stream(e->stream));
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
    HIP_TRY(schost::wait_stream(e->stream));
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
    HIP_TRY(schost::wait_stream(e->stream));
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
    HIP_TRY(schost::wait_event(stop));
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
    out[4] = (int64_t)host.nlate + (int64_t)host.nlate_units;  // failed candidates, whichever road they took
    if (e->last_bulk)
        for (int q = 0; q < kSub; ++q) {
            out[5] += std::min<uint32_t>(host.count[3][q].n, e->bulkcap);
            out[6] += std::min<uint32_t>(host.count[4][q].n, e->itemcap);
        }
    out[7] = 0;  // (was: batches the host kept the bulk list off; the decision is the device's now)
    return SC_OK;
}

int sc_fused_counts(sc_engine *e, int64_t out[4]) {
    if (!e || !out) return fail(SC_ERR_INVALID, "bad argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(schost::wait_stream(e->stream));
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
    HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&out), sizeof host));
    HIP_TRY(hipMemsetAsync(out, 0, sizeof host, e->stream));
    hipLaunchKernelGGL(div_selftest_kernel, dim3(4096), dim3(kBlock), 0, e->stream, (uint64_t)count,
                       seed, mode, out);
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpyAsync(host, out, sizeof host, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = schost::wait_stream(e->stream);
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
    hipError_t he = sc_dev_malloc(reinterpret_cast<void **>(&dp), (size_t)nposes * sizeof(PoseRec));
    if (he == hipSuccess) he = hipMemcpy(dp, poses, (size_t)nposes * sizeof(PoseRec), hipMemcpyHostToDevice);
    if (he == hipSuccess && ijk) {
        he = sc_dev_malloc(reinterpret_cast<void **>(&dijk), (size_t)count * 12);
        if (he == hipSuccess) he = hipMemcpy(dijk, ijk, (size_t)count * 12, hipMemcpyHostToDevice);
    }
    if (he == hipSuccess && pose_idx) {
        he = sc_dev_malloc(reinterpret_cast<void **>(&didx), (size_t)count * 4);
        if (he == hipSuccess) he = hipMemcpy(didx, pose_idx, (size_t)count * 4, hipMemcpyHostToDevice);
    }
    if (he == hipSuccess && words_out) he = sc_dev_malloc(reinterpret_cast<void **>(&dw), (size_t)count * 4);
    if (he == hipSuccess && digests_out) {
        he = sc_dev_malloc(reinterpret_cast<void **>(&dd), ndig * 8);
        if (he == hipSuccess) he = hipMemsetAsync(dd, 0, ndig * 8, e->stream);
    }
    if (he == hipSuccess) {
        const uint64_t nwaves = ((uint64_t)count + 63) >> 6;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((nwaves + 3) / 4, 16384);
        hipLaunchKernelGGL(project_selftest_kernel, dim3(blocks), dim3(kBlock), 0, e->stream, (uint64_t)count,
                           seed, (uint32_t)nposes, dp, dijk, didx, dw, dd);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = schost::wait_stream(e->stream);
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
    HIP_TRY(sc_pin_malloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return SC_OK;
}

void sc_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

int sc_dev_alloc(sc_engine *e, int64_t bytes, void **ptr) {
    if (!e || !ptr || bytes <= 0) return fail(SC_ERR_INVALID, "bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(sc_dev_malloc(ptr, (size_t)bytes));
    return SC_OK;
}

int sc_dev_free(sc_engine *e, void *ptr) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(schost::wait_stream(e->stream));
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
    HIP_TRY(schost::wait_stream(e->stream));
    HIP_TRY(hipMemcpy(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return SC_OK;
}

}  // extern "C"
