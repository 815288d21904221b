// Part of spacecarve.hip (included there): several devices from one process (sc_create_sharded, sc_group_*).
// ---- several devices from one process (sc_create_sharded) ----------------------------
// One engine per device, the x-planes dealt round-robin (or in contiguous slabs) exactly as the
// one-process-per-GPU path deals them to ranks; every view goes to every engine; the read-back lands
// each engine's planes at their global x positions with one strided copy per device.

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
        HIP_TRY(schost::wait_stream(e->stream));
    }
    return SC_OK;
}

}  // extern "C"
