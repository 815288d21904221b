// Part of spacecarve.hip (included at its end): RCCL bound directly behind the C ABI (round 6, VERDICT r05 missing 3) --
// a carve rank assembles the grid without torch: the collective is enqueued by the library, on the engine's own
// stream or on the communicator's beside the next batch's carve.  No reference counterpart (single device,
// cl.py:29-30); the contract is SURVEY 8e.  librccl is opened at first use (dlopen: a process that never gathers
// does not load its 570 MB), the copy already in the process if there is one (torch brings its own).

#include <dlfcn.h>

namespace {

struct NcclId { char internal[128]; };  // ncclUniqueId
typedef int (*nccl_get_unique_id_t)(NcclId *);
typedef int (*nccl_comm_init_rank_t)(void **, int, NcclId, int);
typedef int (*nccl_comm_destroy_t)(void *);
typedef int (*nccl_all_gather_t)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*nccl_get_error_string_t)(int);

struct Rccl {
    void *lib = nullptr;
    nccl_get_unique_id_t get_unique_id = nullptr;
    nccl_comm_init_rank_t comm_init_rank = nullptr;
    nccl_comm_destroy_t comm_destroy = nullptr;
    nccl_all_gather_t all_gather = nullptr;
    nccl_get_error_string_t get_error_string = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char *env = getenv("SC_RCCL_LIB");
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (env && *env) r.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        for (int i = 0; i < 2 && !r.lib; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);  // already here?
        for (int i = 0; i < 3 && !r.lib; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) {
            const char *de = dlerror();
            r.error = std::string("librccl not found: ") + (de ? de : "?");
            return;
        }
        r.get_unique_id = reinterpret_cast<nccl_get_unique_id_t>(dlsym(r.lib, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<nccl_comm_init_rank_t>(dlsym(r.lib, "ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<nccl_comm_destroy_t>(dlsym(r.lib, "ncclCommDestroy"));
        r.all_gather = reinterpret_cast<nccl_all_gather_t>(dlsym(r.lib, "ncclAllGather"));
        r.get_error_string = reinterpret_cast<nccl_get_error_string_t>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.get_error_string) {
            r.error = "librccl lacks an entry point";
            r.lib = nullptr;
        }
    });
    return r;
}

int rccl_ready() {
    Rccl &r = rccl();
    if (!r.lib) return fail(SC_ERR_DEVICE, "%s", r.error.c_str());
    return SC_OK;
}

#define RCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        int _r = (expr);                                                                                   \
        if (_r != 0) return fail(SC_ERR_DEVICE, "%s failed: %s", #expr, rccl().get_error_string(_r));      \
    } while (0)

}  // namespace

struct sc_comm {
    void *comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    hipStream_t stream = nullptr;  // collectives that run beside the engine's stream
    hipEvent_t ev_ready = nullptr; // the engine's pack has been enqueued: the collective waits for it
    char *scratch = nullptr;       // (nranks + 1) x 16 bytes: sc_comm_barrier's all-gather
};

extern "C" {

int sc_comm_available(void) { return rccl_ready(); }  // SC_OK when librccl could be opened (no device call)

int sc_comm_unique_id(void *id, int64_t id_bytes) {
    if (!id || id_bytes < (int64_t)sizeof(NcclId)) return fail(SC_ERR_INVALID, "the id buffer holds %d bytes", (int)sizeof(NcclId));
    int rc = rccl_ready();
    if (rc) return rc;
    NcclId nid;
    RCCL_TRY(rccl().get_unique_id(&nid));
    memcpy(id, &nid, sizeof nid);
    return SC_OK;
}

int sc_comm_create(sc_comm **out, const void *id, int nranks, int rank, int device) {
    if (!out || !id) return fail(SC_ERR_INVALID, "null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(SC_ERR_INVALID, "rank %d of %d", rank, nranks);
    int rc = rccl_ready();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    sc_comm *c = new (std::nothrow) sc_comm;
    if (!c) return fail(SC_ERR_NOMEM, "out of host memory");
    c->nranks = nranks;
    c->rank = rank;
    c->device = device;
    NcclId nid;
    memcpy(&nid, id, sizeof nid);
    int r = rccl().comm_init_rank(&c->comm, nranks, nid, rank);
    if (r != 0) {
        delete c;
        return fail(SC_ERR_DEVICE, "ncclCommInitRank failed: %s", rccl().get_error_string(r));
    }
    hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&c->scratch), (size_t)(nranks + 1) * 16);
    if (he == hipSuccess) he = hipMemset(c->scratch, 0, (size_t)(nranks + 1) * 16);
    if (he != hipSuccess) {
        if (c->scratch) (void)hipFree(c->scratch);
        if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        (void)rccl().comm_destroy(c->comm);
        delete c;
        return fail(SC_ERR_DEVICE, "communicator stream: %s", hipGetErrorString(he));
    }
    *out = c;
    return SC_OK;
}

void sc_comm_destroy(sc_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)schost::wait_stream(c->stream);
    if (c->comm) (void)rccl().comm_destroy(c->comm);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sc_comm_size(const sc_comm *c) { return c ? c->nranks : 0; }
int sc_comm_rank(const sc_comm *c) { return c ? c->rank : -1; }

int sc_comm_stream(sc_comm *c, void **hip_stream) {
    if (!c || !hip_stream) return fail(SC_ERR_INVALID, "null argument");
    *hip_stream = c->stream;
    return SC_OK;
}

int sc_comm_synchronize(sc_comm *c) {
    if (!c) return fail(SC_ERR_INVALID, "null communicator");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(schost::wait_stream(c->stream));
    return SC_OK;
}

// Every rank has reached this call when it returns anywhere: a 16-byte all-gather on the communicator's stream, waited
// for on the host (what a bench brackets its timed steps with; torch.distributed is not needed for it).
int sc_comm_barrier(sc_comm *c) {
    if (!c) return fail(SC_ERR_INVALID, "null communicator");
    HIP_TRY(hipSetDevice(c->device));
    RCCL_TRY(rccl().all_gather(c->scratch, c->scratch + 16, 16, /* ncclInt8 */ 0, c->comm, c->stream));
    HIP_TRY(schost::wait_stream(c->stream));
    return SC_OK;
}

int sc_comm_all_gather(sc_comm *c, const void *send_dev, void *recv_dev, int64_t bytes_per_rank, void *hip_stream) {
    if (!c || !send_dev || !recv_dev || bytes_per_rank < 0) return fail(SC_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    RCCL_TRY(rccl().all_gather(send_dev, recv_dev, (size_t)bytes_per_rank, /* ncclInt8 */ 0, c->comm, st));
    return SC_OK;
}

int sc_engine_stream(sc_engine *e, void **hip_stream) {
    if (!e || !hip_stream) return fail(SC_ERR_INVALID, "null argument");
    int rcw = wait_setup(e);
    if (rcw) return rcw;
    *hip_stream = e->stream;
    return SC_OK;
}

// pack on the engine's stream, gather on `st` (the engine's own, or the communicator's behind an event); the send buffer
// is busy until the collective has read it: an event the engine's next pack into that buffer waits for
constexpr int kMaxHeaderRanks = 256;

static int gather_behind_pack(sc_engine *e, sc_comm *c, const void *send, void *recv_dev, int64_t bytes, int overlap,
                              hipEvent_t *busy, SparseHeader *hdr_pin = nullptr) {
    if (overlap) {
        HIP_TRY(hipEventRecord(c->ev_ready, e->stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    }
    hipStream_t st = overlap ? c->stream : e->stream;
    RCCL_TRY(rccl().all_gather(send, recv_dev, (size_t)bytes, /* ncclInt8 */ 0, c->comm, st));
    // the ranks' headers to page-locked host memory, behind the collective on its stream (sparse form)
    if (hdr_pin != nullptr && c->nranks <= kMaxHeaderRanks)
        HIP_TRY(hipMemcpy2DAsync(hdr_pin, sizeof(SparseHeader), recv_dev, (size_t)bytes, sizeof(SparseHeader), (size_t)c->nranks,
                                 hipMemcpyDeviceToHost, st));
    // an event behind the collective (and that copy), whichever stream it is on: what a reader of the receive buffer
    // waits for without queueing behind the NEXT batch's collective, and -- overlap -- what the engine's next pack into
    // this send buffer waits for
    if (!*busy) HIP_TRY(hipEventCreateWithFlags(busy, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(*busy, st));
    return SC_OK;
}

int sc_all_gather_sparse(sc_engine *e, sc_comm *c, int64_t cap, void *recv_dev, int64_t rank_stride, int overlap,
                         void **done_event, const void **headers_host) {
    if (!e || !c || !recv_dev) return fail(SC_ERR_INVALID, "null argument");
    if (c->device != e->device) return fail(SC_ERR_INVALID, "communicator on device %d, engine on %d", c->device, e->device);
    if (cap <= 0) return fail(SC_ERR_INVALID, "every rank names the same capacity (bricks)");
    int rc = use_device(e);
    if (rc) return rc;
    rc = sc_flush(e);  // the batch's kernels first: the wait below belongs in front of the pack, not of the carve
    if (rc) return rc;
    // the send buffer of this call was last read by the collective of two calls ago
    const int q = e->sparse_idx;
    if (e->sparse_busy[q] && e->sparse_busy_armed[q]) {
        HIP_TRY(hipStreamWaitEvent(e->stream, e->sparse_busy[q], 0));
        e->sparse_busy_armed[q] = false;
    }
    void *ptr = nullptr;
    int64_t bytes = 0;
    rc = values_sparse(e, cap, rank_stride, &ptr, &bytes);
    if (rc) return rc;
    if (rank_stride < bytes)
        return fail(SC_ERR_INVALID, "rank stride of %lld bytes for a buffer of %lld", (long long)rank_stride, (long long)bytes);
    if (!e->sparse_hdr_pin[q])
        HIP_TRY(sc_pin_malloc(reinterpret_cast<void **>(&e->sparse_hdr_pin[q]), kMaxHeaderRanks * sizeof(SparseHeader), hipHostMallocDefault));
    rc = gather_behind_pack(e, c, ptr, recv_dev, rank_stride, overlap, &e->sparse_busy[q], e->sparse_hdr_pin[q]);
    if (rc) return rc;
    e->sparse_busy_armed[q] = overlap != 0;
    if (done_event) *done_event = e->sparse_busy[q];
    if (headers_host) *headers_host = c->nranks <= kMaxHeaderRanks ? e->sparse_hdr_pin[q] : nullptr;
    return SC_OK;
}

int sc_sparse_wait_headers(void *done_event, const void *headers_host, int64_t rank_bytes, int world, uint32_t *nmixed,
                           uint32_t *cap) {
    if (!done_event || !headers_host || !nmixed || !cap) return fail(SC_ERR_INVALID, "null argument");
    if (world < 1 || world > kMaxHeaderRanks) return fail(SC_ERR_INVALID, "bad world");
    HIP_TRY(schost::wait_event(static_cast<hipEvent_t>(done_event)));
    const SparseHeader *h = static_cast<const SparseHeader *>(headers_host);
    for (int r = 0; r < world; ++r) {
        int rc = sparse_check_header(h[r], rank_bytes, r);
        if (rc) return rc;
        nmixed[r] = h[r].nmixed;
        cap[r] = h[r].cap;
    }
    return SC_OK;
}

int sc_all_gather_packed(sc_engine *e, sc_comm *c, int bits, void *recv_dev, int64_t rank_stride, int overlap) {
    if (!e || !c || !recv_dev) return fail(SC_ERR_INVALID, "null argument");
    if (c->device != e->device) return fail(SC_ERR_INVALID, "communicator on device %d, engine on %d", c->device, e->device);
    int rc = use_device(e);
    if (rc) return rc;
    rc = sc_flush(e);
    if (rc) return rc;
    if (e->packed_busy && e->packed_busy_armed) {  // ONE packed buffer: the previous collective must have read it
        HIP_TRY(hipStreamWaitEvent(e->stream, e->packed_busy, 0));
        e->packed_busy_armed = false;
    }
    void *ptr = nullptr;
    int64_t bytes = 0;
    rc = sc_values_packed(e, bits, &ptr, &bytes);
    if (rc) return rc;
    if (rank_stride < bytes || (size_t)rank_stride > e->packed_cap)
        return fail(SC_ERR_INVALID, "rank stride of %lld bytes for a buffer of %lld (at most one plane more)", (long long)rank_stride, (long long)bytes);
    rc = gather_behind_pack(e, c, ptr, recv_dev, rank_stride, overlap, &e->packed_busy);
    if (rc) return rc;
    e->packed_busy_armed = overlap != 0;
    return SC_OK;
}

}  // extern "C"
