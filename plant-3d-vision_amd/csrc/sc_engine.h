// Part of spacecarve.hip (included there, behind the device headers): the host engine -- struct sc_engine and the helpers
// every entry point shares (events and timers, the grid descriptor, staging slots and arenas, mask packing jobs, survivor
// lists and control blocks, host-packed bits).  The launches are in sc_flush.inl, the C ABI in sc_api_*.inl.

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
    // sc_create_ex with SC_CREATE_DEFERRED: the device half of the set-up (runtime initialisation, the process's first
    // stream, the state's allocation: 130-240 ms in a fresh process) runs on a thread of its own; every entry point that
    // needs the device joins it first (use_device), sc_process_png_views decodes its files beside it
    std::thread setup_thread;
    bool setup_pending = false;
    int setup_rc = SC_OK;
    std::string setup_err;
    double setup_ms = 0.0;         // what the device half took (on its thread, or inside sc_create*)
    double setup_waited_ms = 0.0;  // how long the first call that needed the device waited for it (deferred engines)

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
    int64_t stage1_store_share = 5;  // sixteenths of the deferred strips filled beside the FIRST list stage
    int64_t stage1_list_blocks = 1280; // ... and that stage's persistent list blocks then
    int64_t defer_share = 16;     // sixteenths of the strips whose empty bricks the final list stage fills
    int64_t defer_stores = 1536;  // list blocks of a final stage that also fills the empty bricks (0: the dense stage fills them);
                                  // 6 per CU beside the store blocks: with six first-stage views 1280 -> 1536 is worth 3-4 % on bulky scenes
                                  // and nothing on a plant (round 5's last sweep, tools/sweep_blocks*.json); 1600 and more lose 7 % there
    int64_t pack_rows = 0;     // 0: the band form of the 16-byte pack kernel; 1, 2, 4, 8: the panel form, tile rows per block
    int64_t view_brick = 1;    // a single-view carve launch goes through the brick kernels too (0: streaming kernel)
    uint8_t *dead = nullptr;   // per brick: an earlier launch found it empty, every voxel is -1 (until the next clear)
    bool dead_clean = false;   // `dead` describes the labels (false after a clear: the next flags kernel rewrites it)
    int64_t final_voxels = 2;  // voxels per lane in the final survivor stage (1 or 2)
    int64_t stage1_voxels = 2; // ... in the stages before it
    int64_t fill_blocks = 256; // persistent store blocks of a list stage (0: one short block per strip); round 4: 256 (one per CU) from 512, measured after their loop lost its vector instructions
    int64_t pack_ride = 1;     // a device batch is packed at flush, in view order: the first views ahead of
                               // the flags kernel, the others beside the dense stage (0: all ahead)
    int64_t brick_walkers = 1280;  // persistent blocks of the dense stage when packing rides with it (1024 until round 5: noise -4 %, plant +-0)
    int8_t *narrow = nullptr;  // scratch of sc_get_values_i8
    uint32_t *packed_labels = nullptr;  // sc_values_packed: the labels at 2 or 1 bits each
    uint32_t *wire_stage = nullptr;     // sc_get_values_wire2: page-locked landing place of the packed labels
    // sc_values_sparse (sc_sparse.h): two send buffers alternate, so that a collective may still read one while the next
    // batch's labels are packed into the other
    char *sparse_buf[2] = {nullptr, nullptr};
    size_t sparse_bytes[2] = {0, 0};
    int sparse_idx = 0;
    SparseCounters *sparse_cnt = nullptr;  // two, alternating (the pack kernel of a call zeroes the other call's)
    uint32_t *sparse_work = nullptr;       // bricks whose labels have to be read when no list of them exists
    uint64_t sparse_calls = 0;
    int64_t sparse_cap = 0;                // payload capacity (bricks) of the next call that does not name one
    bool sparse_exact = false;             // the verdict bytes and the live / late lists describe the labels exactly: the
                                           // last launch was a brick-form carve of a fresh volume, nothing since
    bool sparse_late = false;              // ... and its late lists hold its failed candidates (verdict byte 5)
    // a collective enqueued beside the engine's stream (sc_all_gather_*, overlap) still reads a send buffer: the next
    // pack into that buffer waits for the event recorded behind the collective
    hipEvent_t sparse_busy[2] = {nullptr, nullptr}, packed_busy = nullptr;
    bool sparse_busy_armed[2] = {false, false}, packed_busy_armed = false;
    size_t packed_cap = 0;                 // bytes of packed_labels
    // sc_all_gather_sparse: the ranks' headers of a gather land here (page-locked), copied behind the collective on its
    // stream: a reader waits for the gather's event and reads host memory (sc_sparse_wait_headers)
    SparseHeader *sparse_hdr_pin[2] = {nullptr, nullptr};
    size_t wire_stage_words = 0;
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
    // Whether the bulk units' verdicts pay is decided on the device, inside the batch, from the number of units its
    // own dense stage left (carve_special_kernel): fewer than this and their voxels take the ordinary lists
    int64_t bulk_floor = 8192;
    int64_t bulk_live = 2;  // sixteenths of the bricks that must be live for the bulk list to be kept at all (0: always)
    int64_t list_cap = 0, list_cap_built = 0;  // entries per survivor sub-list (0: sized from the grid); tests of the overflow paths
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

    // carve masks from the host: packed to bits by host threads into a page-locked arena (two, alternating between
    // flushes), which one copy per flush brings to its device mirror together with the table of the views' records
    int64_t spec_share = 3;    // sixteenths of the strips set to -1 by fill blocks in front of the flags kernel (fresh volumes)
    int64_t late_road = 1;     // 1: a FULL candidate a late view rejects joins the bulk units (UnitRoad); 0: the late list, always
    int64_t spec_blocks = 64;  // ... that many persistent blocks of 512 threads (64: a fill that does not saturate HBM leaves the verdicts their memory round trips; 128 measured 2 % slower per batch, 48 too)
    int64_t dense_extra = 1;   // a unit the dense views thinned out to 32 .. 128 voxels takes one more pair of views there
    int64_t safe_kernels = 1;  // batches whose views are all certified take the list kernels compiled without the general path
    int64_t host_pack = 1;
    struct HostBits {
        char *pin = nullptr, *dev = nullptr;
        size_t cap = 0, used = 0;
        hipEvent_t ev = nullptr;  // the last copy out of `pin` has completed
        bool armed = false;
    } hb[2];
    int hb_cur = 0;
    std::vector<BitsRec> hp_pending;  // host-packed views not uploaded yet (all of them are among `pending`)

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
    int64_t stage1_views = 6;    // views applied to the first survivor list (8 until round 5)
    int64_t stage2_views = 0;    // views applied to the second survivor list (0: no such stage)
    int64_t list_blocks = 2048;  // persistent grid of list stages without store blocks
    int64_t view_group = 2;      // the spans of the final list stage are a multiple of this many views

    std::vector<TimedLaunch> timed[kNumKernels];
    hipEvent_t step_start = nullptr;
    bool step_open = false;
    hipEvent_t span_start = nullptr;  // sc_span_begin .. sc_span_end
    bool span_open = false;
    std::vector<hipEvent_t> event_pool;
};

namespace {

// the deferred half of the set-up has finished (sc_create_ex); its failure is every later call's failure
int wait_setup(sc_engine *e) {
    if (e->setup_pending) {
        const auto t0 = std::chrono::steady_clock::now();
        if (e->setup_thread.joinable()) e->setup_thread.join();
        e->setup_waited_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        e->setup_pending = false;
    }
    if (e->setup_rc != SC_OK) return fail(e->setup_rc, "%s", e->setup_err.c_str());
    return SC_OK;
}

int use_device(sc_engine *e) {
    int rc = wait_setup(e);
    if (rc) return rc;
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

FastDiv fast_div(uint32_t d) {  // see fdiv (sc_types.h); d >= 1
    uint32_t s = 0;
    while (((uint64_t)1 << s) < d) ++s;
    return FastDiv{(uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << s) - d)) / d + 1u), s};
}

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
    g.by_nzp = fast_div(g.nzp);
    g.by_ny = fast_div(g.ny);
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
    if (!e->dense) HIP_TRY(sc_dev_malloc(&e->dense, (size_t)e->n * 4));
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
    HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&c.base), c.cap));
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
    HIP_TRY(schost::wait_stream(e->stream));
    for (int s = 0; s < kSlots; ++s) {
        if (e->pin[s]) (void)hipHostFree(e->pin[s]);
        if (e->raw[s]) (void)hipFree(e->raw[s]);
        e->pin[s] = e->raw[s] = nullptr;
        e->slot_armed[s] = false;
    }
    e->slot_bytes = 0;
    for (int s = 0; s < kSlots; ++s) {
        HIP_TRY(sc_pin_malloc(&e->pin[s], bytes, hipHostMallocDefault));
        HIP_TRY(sc_dev_malloc(&e->raw[s], bytes));
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
    d.strip = ((H + kTile - 1) / kTile) * kTile;  // the words of a strip of the bit tiles (carve masks)
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
    // (the band form addresses a band's bytes by 32-bit offsets from its first one: 32 rows below 2^31 bytes)
    const bool band_ok = pj.tiles_x <= kBandTiles && pj.row_stride < ((int64_t)1 << 26);
    if (e->pack_rows == 0 && pj.tiles_x >= 16 && band_ok) return 0;
    if (e->pack_rows == 3 && band_ok) return 0;  // bands whatever the width (tests)
    return (e->pack_rows == 0 || e->pack_rows == 3) ? 4 : (int)e->pack_rows;
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
    // (u8strip_offset: a 24-bit product of the strip's number and its bytes, offsets below 2^31)
    if (per_view >= ((size_t)1 << 31) || (size_t)tiles_y * 128 >= ((size_t)1 << 24))
        return fail(SC_ERR_INVALID, "mask too large for the byte gather (%d x %d)", W, H);
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
        d.tiles_x = tiles_y * 128;  // uint8 + table form: the bytes of a 16-pixel strip (u8strip_offset), not a tile count
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
    // (ftile_offset: a 24-bit product of the strip's number and its floats, element indices below 2^32)
    if (per_view >= ((size_t)1 << 33) || (size_t)tiles_y * 32 >= ((size_t)1 << 24))
        return fail(SC_ERR_INVALID, "mask too large for the float gather (%d x %d)", W, H);
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
        d.tiles_x = tiles_y * 32;  // float32 tiles: the floats of an 8-pixel strip (ftile_offset), not a tile count
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
    if (e->lists && e->list_cap_built == e->list_cap) return SC_OK;
    if (e->lists) {  // the capacity knob moved (tests of the overflow paths): rebuilt behind the stream
        HIP_TRY(schost::wait_stream(e->stream));
        (void)hipFree(e->lists);
        e->lists = nullptr;
    }
    // room for 5/16 of the voxels: two views of coin-flip masks leave a quarter alive, which the hashed
    // sub-lists must hold with a margin for their unevenness (an overflow sends the batch down the dense
    // special kernel's dense pass, 10 x slower)
    uint64_t total = std::max<uint64_t>((uint64_t)e->n / 4 + (uint64_t)e->n / 16, (uint64_t)kSub * 1024);
    e->subcap = (uint32_t)((total + kSub - 1) / kSub);
    if (e->list_cap > 0) e->subcap = (uint32_t)e->list_cap;
    HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->lists), (size_t)2 * kSub * e->subcap * sizeof(uint32_t)));
    e->list_cap_built = e->list_cap;
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
    const uint32_t bulkcap = (uint32_t)((nbricks * 4 * 2 + kSub - 1) / kSub + 64);
    const size_t bulk_words = nbricks ? (size_t)kSub * bulkcap : 0;
    // up to 2 halves x 2 words x 4 pieces per unit; room for a third of that on average (a full sub-list
    // sends the unit's voxels down the ordinary lists)
    const uint32_t itemcap = bulkcap * 5u;
    uint4 *items = nullptr;
    if (bulk_words) HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&items), (size_t)kSub * itemcap * sizeof(uint4)));
    hipError_t he = sc_dev_malloc(reinterpret_cast<void **>(&base),
                              2 * sizeof(ListCtl) + flag_bytes + (3 * nbricks + bulk_words) * sizeof(uint32_t) + 16);
    if (he == hipSuccess) he = hipMemsetAsync(base, 0, 2 * sizeof(ListCtl), e->stream);
    if (he != hipSuccess) {  // nothing of this is published before all of it exists
        if (items) (void)hipFree(items);
        if (base) (void)hipFree(base);
        return fail(he == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE, "control block allocation failed: %s", hipGetErrorString(he));
    }
    e->bulkcap = bulkcap;
    e->itemcap = itemcap;
    e->items = items;
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

// Room for `bytes` more in the current host arena (page-locked memory + its device mirror).  An arena whose last copy
// may still be in flight is waited for before it is written again; one that is too small grows (what it holds is
// carried over: nothing of it has been uploaded yet).
int hostbits_reserve(sc_engine *e, size_t bytes, char **out) {
    auto &a = e->hb[e->hb_cur];
    if (a.used == 0 && a.armed) {
        HIP_TRY(schost::wait_event(a.ev));
        a.armed = false;
    }
    if (a.used + bytes > a.cap) {
        const size_t cap = std::max<size_t>({a.used + bytes, a.cap * 2, (size_t)16 << 20});
        char *pin = nullptr, *dev = nullptr;
        HIP_TRY(sc_pin_malloc(reinterpret_cast<void **>(&pin), cap, hipHostMallocDefault));
        hipError_t he = sc_dev_malloc(reinterpret_cast<void **>(&dev), cap);
        if (he != hipSuccess) {
            (void)hipHostFree(pin);
            return fail(he == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE, "host-mask arena: %s", hipGetErrorString(he));
        }
        if (a.used) memcpy(pin, a.pin, a.used);
        if (a.pin) {
            // the old blocks may be the source / target of a copy still on the stream (a.used > 0 means: not of this
            // batch's, but an earlier flush's): wait before they go
            HIP_TRY(schost::wait_stream(e->stream));
            (void)hipHostFree(a.pin);
            (void)hipFree(a.dev);
        }
        a.pin = pin;
        a.dev = dev;
        a.cap = cap;
        a.armed = false;
    }
    if (!a.ev) HIP_TRY(hipEventCreateWithFlags(&a.ev, hipEventDisableTiming));
    *out = a.pin + a.used;
    a.used += bytes;
    return SC_OK;
}

int hostbits_push_view(sc_engine *e, const float *K, const float *R, const float *t, uint64_t src_off, int H, int W);

// A carve mask in HOST memory: its bits (pixel != 0 after the optional invert) are made here, on host threads, and
// only they cross PCIe -- 1/8 of the bytes (1/32 of an int32 mask's); tiles, occupancy bytes and cell maps are a
// device pass over the bits at the next flush (bits_tiles_kernel).  Appends one pending view.
int enqueue_hostbits(sc_engine *e, const float *K, const float *R, const float *t, const void *mask, int H, int W,
                     int dtype, int64_t row_stride) {
    const int wpr = (W + kTile - 1) / kTile;
    const size_t bits_bytes = ((size_t)H * wpr * 4 + 255) & ~(size_t)255;
    char *dst = nullptr;
    int rc = hostbits_reserve(e, bits_bytes, &dst);
    if (rc) return rc;
    const uint64_t src_off = (uint64_t)(dst - e->hb[e->hb_cur].pin);
    const int elem = dtype == SC_MASK_I32 ? 4 : 1;
    const uint8_t flip = dtype == SC_MASK_U8_INV ? 255 : dtype == SC_MASK_BOOL_INV ? 1 : 0;
    uint32_t *out = reinterpret_cast<uint32_t *>(dst);
    // bands of rows over the pool: a 1440 x 1080 mask is 1.5 MB to read, ~17 bands of 64 rows
    const bool par = (size_t)H * W >= ((size_t)1 << 18);  // small pictures are not worth a hand-over
    const int band = 64, nparts = par ? (H + band - 1) / band : 1;
    schost::parallel_for(nparts, [&](int part) {
        const int r0 = par ? part * band : 0, r1 = par ? std::min(H, r0 + band) : H;
        schost::pack_rows(mask, row_stride, W, r0, r1, out, wpr, elem, flip);
    });
    return hostbits_push_view(e, K, R, t, src_off, H, W);
}

// The device side of a host-packed view whose bits lie at `src_off` of the current arena: storage for its tiles,
// occupancy bytes and cell map, its record for bits_tiles_kernel, its descriptor among the pending views.
int hostbits_push_view(sc_engine *e, const float *K, const float *R, const float *t, uint64_t src_off, int H, int W) {
    const int wpr = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile;
    const size_t ntiles = (size_t)wpr * tiles_y;
    void *tiles = nullptr, *occ = nullptr, *cm = nullptr;
    int rc = arena_alloc(e, ntiles * 128, &tiles);
    if (rc) return rc;
    rc = arena_alloc(e, ntiles, &occ);
    if (rc) return rc;
    rc = arena_alloc(e, ntiles * 4, &cm);
    if (rc) return rc;
    BitsRec br;
    memset(&br, 0, sizeof br);
    br.src_off = src_off;
    br.tiles = static_cast<uint32_t *>(tiles);
    br.occ = static_cast<uint8_t *>(occ);
    br.cmask = static_cast<uint32_t *>(cm);
    br.W = W; br.H = H; br.tiles_x = wpr; br.tiles_y = tiles_y;
    e->hp_pending.push_back(br);
    ViewDesc d;
    fill_desc(e, d, K, R, t, tiles, H, W, static_cast<const uint8_t *>(occ));
    d.cmask = static_cast<const uint32_t *>(cm);
    e->pending.push_back(d);
    return SC_OK;
}

// The host-packed views' bits to the device, and their tiles made: one copy, one kernel, ahead of whatever the flush
// launches.
int upload_hostbits(sc_engine *e) {
    if (e->hp_pending.empty()) return SC_OK;
    const size_t nrec = e->hp_pending.size();
    if (nrec > 65535) return fail(SC_ERR_INVALID, "too many host masks in one batch");
    char *table = nullptr;
    int rc = hostbits_reserve(e, nrec * sizeof(BitsRec), &table);  // (may move the arena: offsets stay)
    if (rc) return rc;
    auto &a = e->hb[e->hb_cur];
    memcpy(table, e->hp_pending.data(), nrec * sizeof(BitsRec));
    const uint64_t table_off = (uint64_t)(table - a.pin);
    uint32_t maxtiles = 0;
    for (const auto &r : e->hp_pending) maxtiles = std::max(maxtiles, (uint32_t)r.tiles_x * (uint32_t)r.tiles_y);
    rc = step_begin(e);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(a.dev, a.pin, a.used, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipEventRecord(a.ev, e->stream));
    a.armed = true;
    LaunchTimer lt{e, SC_KERNEL_PACK};
    rc = lt.begin();
    if (rc) return rc;
    hipLaunchKernelGGL(bits_tiles_kernel, dim3((maxtiles + 7u) / 8u, (uint32_t)nrec), dim3(kBlock), 0, e->stream,
                       static_cast<const char *>(a.dev), table_off);
    HIP_TRY(hipGetLastError());
    rc = lt.end();
    if (rc) return rc;
    e->hp_pending.clear();
    a.used = 0;          // (the next batch's bits take the other arena; this one is free once its event has fired)
    e->hb_cur ^= 1;
    return SC_OK;
}

}  // namespace
