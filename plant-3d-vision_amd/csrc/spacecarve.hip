// spacecarve.hip -- MI355X (gfx950 / CDNA4) voxel back-projection engine + its C ABI.
//
// Replaces, behind include/spacecarve.h, the device layer of the reference:
//   plant3dvision/kernels/backprojection.c  carve :57-84, average :36-55,
//                                           backproject_point :3-34
//   plant3dvision/kernels/common.h          unravel_index :1-12
//   plant3dvision/cl.py                     Backprojection buffer lifecycle :118-311
//
// Design:
//   * State stays in the reference's layout: C-order [nx][ny][nz], int32 / float32.
//   * One lane owns 4 consecutive z-voxels of one (i,j) column: one 16-byte load and one
//     16-byte store per lane, 1 KiB per wavefront instruction; one owner per voxel, so
//     there are no atomics and no races (as in the reference, one work-item per voxel).
//   * A launch applies a CHUNK of views to the state it holds in registers (1 view per
//     launch = the reference's schedule).  The carve update is order-independent, so
//     dead lanes drop out and a wavefront leaves the view loop as soon as a ballot says
//     every one of its voxels is carved.
//   * A fused carve (many views) first settles whole 16x64-voxel BRICKS from four corner
//     projections each: bricks some view sees entirely over background are EMPTY (-1, filled
//     by store blocks beside the final stage), bricks every view sees entirely over
//     foreground are FULL (0 -> 1).  Only the remaining LIVE bricks are projected voxel by
//     voxel, for two views; the voxels still alive are compacted into survivor lists
//     (wave-aggregated atomics on 256 sharded counters) and finished by persistent kernels
//     with one lane per survivor.
//   * Carve masks live in HBM as 1 bit per pixel in 32x32-pixel tiles (one 128-byte line
//     per tile): the 64..256 z-neighbours a wavefront projects land on a short image
//     segment of arbitrary orientation, i.e. on a handful of lines, whatever the camera roll.
//   * The x/y partial sums of every dot product are hoisted per column WITHOUT changing
//     the reference's left-to-right rounding: ((R0*x + R1*y) + R2*z) + t0.
//   * Arithmetic contract: IEEE binary32, no FMA contraction (built with
//     -ffp-contract=off), correctly rounded division, and the (int) cast guarded so
//     that NaN / inf / out-of-range are rejected exactly like x86 cvttss2si -> INT_MIN.
//
// gfx950 only.  No fallback path: every entry point fails with SC_ERR_DEVICE when HIP
// cannot run the kernels.

#include <hip/hip_runtime.h>

#include "hostwait.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "hostpack.h"
#include "spacecarve.h"
#include "spacecarve_tuning.h"

// SC_TRACE_ALLOC=1 in the environment: every allocation of this file reports "sc_alloc <call>:<line> <bytes> <ms>"
// on stderr (bench.py's cold-process leg reads them: what a first batch pays the driver for memory).  Off: a flag test.
namespace sctrace {
inline bool on() {
    static const bool v = [] { const char *s = getenv("SC_TRACE_ALLOC"); return s && *s && *s != '0'; }();
    return v;
}
template <class F>
inline hipError_t timed(const char *what, int line, size_t bytes, F &&f) {
    if (!on()) return f();
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t r = f();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "sc_alloc %s:%d %zu %.3f\n", what, line, bytes, ms);
    return r;
}
template <class T>
inline hipError_t dev(T **p, size_t bytes, int line) {
    return timed("hipMalloc", line, bytes, [&] { return (hipMalloc)(reinterpret_cast<void **>(p), bytes); });
}
template <class T>
inline hipError_t host(T **p, size_t bytes, unsigned flags, int line) {
    return timed("hipHostMalloc", line, bytes, [&] { return (hipHostMalloc)(reinterpret_cast<void **>(p), bytes, flags); });
}
}  // namespace sctrace
#define hipMalloc(p, n) sctrace::dev((p), (n), __LINE__)
#define hipHostMalloc(p, n, f) sctrace::host((p), (n), (f), __LINE__)

namespace {

// ------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------
#include "sc_types.h"
#include "sc_project.h"
#include "sc_stream.h"
#include "sc_pack.h"
#include "sc_verdicts.h"
#include "sc_bricks.h"
#include "sc_lists.h"
#include "sc_average.h"
#include "sc_misc.h"
#include "sc_sparse.h"


// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

thread_local std::string g_err;
std::atomic<int64_t> g_avg_labels_fused{0};  // sc_average_labels calls that took the shared-launch form

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(_e == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE,            \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                        __LINE__);                                                           \
    } while (0)

struct Chunk {
    char *base = nullptr;
    size_t cap = 0, used = 0;
};

struct TimedLaunch {
    hipEvent_t start, stop;
};

constexpr int kSlots = 4;
constexpr int kNumKernels = 7;

}  // namespace

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
        if (e->setup_thread.joinable()) e->setup_thread.join();
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
    if (!e->dense) HIP_TRY(hipMalloc(&e->dense, (size_t)e->n * 4));
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
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c.base), c.cap));
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
        HIP_TRY(hipHostMalloc(&e->pin[s], bytes, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&e->raw[s], bytes));
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
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->lists), (size_t)2 * kSub * e->subcap * sizeof(uint32_t)));
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
    if (bulk_words) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&items), (size_t)kSub * itemcap * sizeof(uint4)));
    hipError_t he = hipMalloc(reinterpret_cast<void **>(&base),
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
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pin), cap, hipHostMallocDefault));
        hipError_t he = hipMalloc(reinterpret_cast<void **>(&dev), cap);
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

// What a fused carve of `nv` views will look like (see flush): decided before anything is launched,
// because a deferred batch is packed according to it.
struct FusedPlan {
    int ndense, nstage1, s1, flag_views;
    bool compact, brick, defer_stores;
    uint32_t bys, bzs, nbricks, nstrips, dense_store_strips;
};

FusedPlan fused_plan(const sc_engine *e, size_t nv, bool has_occ) {
    FusedPlan p{};
    p.ndense = (int)e->dense_views;
    p.nstage1 = (int)e->stage1_views;
    p.compact = e->compact && nv >= (size_t)kMinFusedViews && nv > (size_t)p.ndense &&
                (uint64_t)e->npitch < 0x80000000ull;
    p.bys = (uint32_t)((e->ny + kBrickY - 1) / kBrickY);
    p.bzs = (uint32_t)((e->nz + kBrickZ - 1) / kBrickZ);
    p.brick = (nv > 1 || e->view_brick) && e->brick && p.bzs <= 64 && (uint64_t)e->npitch < 0x80000000ull &&
              (uint64_t)e->planes * p.bys * p.bzs < 0x40000000ull && has_occ;  // brick ids carry two flag bits in the fill list
    p.nbricks = p.brick ? (uint32_t)((uint64_t)e->planes * p.bys * p.bzs) : 0u;
    p.flag_views = (int)nv;  // every view of the batch may veto a brick, not only the dense stage's
    if (e->flag_views > 0 && e->flag_views < (int64_t)p.flag_views) p.flag_views = (int)e->flag_views;
    p.s1 = (int)std::min<size_t>(nv, (size_t)p.ndense + (size_t)p.nstage1);
    // the -1 fill of empty bricks rides along with the list stages when there are any: strips
    // [0, dense_store_strips) are filled by the dense kernel's store blocks, the others by the list
    // stages' (defer_share sixteenths of them)
    p.nstrips = p.brick ? (uint32_t)((uint64_t)e->planes * p.bys) : 0u;
    p.dense_store_strips = p.nstrips;
    if (p.brick && p.compact && e->defer_stores > 0 && e->defer_share > 0)
        p.dense_store_strips = (uint32_t)((uint64_t)p.nstrips * (uint64_t)(16 - e->defer_share) / 16u);
    p.defer_stores = p.dense_store_strips < p.nstrips;
    return p;
}

// Launch the first `count` pending views (count == 0: all of them).
int flush(sc_engine *e, size_t count = 0) {
    if (e->pending.empty()) return SC_OK;
    {
        int rch = upload_hostbits(e);
        if (rch) return rch;
    }
    size_t nv = count ? std::min(count, e->pending.size()) : e->pending.size();
    // A device batch whose packing was deferred is packed here, in the order its views will be
    // applied: the views the flags kernel, the dense stage and the first survivor stage need go
    // ahead, the others ride beside the dense stage (brick form).  Any other shape of launch packs
    // the whole batch first, in the order given.
    bool ordered = false;
    PackJob ride;
    memset(&ride, 0, sizeof ride);
    uint32_t ride_blocks = 0;
    int packed_ahead = (int)nv;
    if (e->deferred.on) {
        const bool whole = nv == e->pending.size() && nv == (size_t)e->deferred.V && e->mode == SC_MODE_CARVE && nv > 1;
        if (!whole) {
            int rcd = materialize_deferred(e);
            if (rcd) return rcd;
        } else {
            int rcd = step_begin(e);
            if (rcd) return rcd;
            std::vector<uint32_t> perm;
            if (e->view_order == 1) order_views(e->pending, &perm);
            else { perm.resize(nv); for (size_t q = 0; q < nv; ++q) perm[q] = (uint32_t)q; }
            ordered = true;
            PackJob pj;
            rcd = deferred_job(e, &pj);
            if (rcd) return rcd;
            e->deferred.on = false;
            pj.use_order = 1;
            for (size_t q = 0; q < nv; ++q) pj.order[q] = (uint16_t)perm[q];
            const FusedPlan fp = fused_plan(e, nv, true);
            int ahead = (int)nv;
            // (not when the dense kernel fills a share of the strips itself: with riders the flags kernel leaves FULL
            // candidates open until the confirm kernel -- behind the dense kernel -- and only the list stages' store
            // blocks come after that.  Round 4 tried both together for the fill's sake: 4 of 2 600 fuzz cases, all with
            // SC_OPT_DEFER_SHARE 5, kept bricks that nobody filled.)
            if (e->pack_ride && fp.brick && fp.compact && fp.defer_stores && fp.dense_store_strips == 0)
                ahead = std::min<int>((int)nv, std::max(fp.flag_views, fp.s1));
            pj.slot0 = 0;
            pj.nslots = ahead;
            LaunchTimer ltp{e, SC_KERNEL_PACK};
            rcd = ltp.begin();
            if (rcd) return rcd;
            rcd = launch_pack16(e, pj);
            if (rcd) return rcd;
            rcd = ltp.end();
            if (rcd) return rcd;
            if (ahead < (int)nv) {
                ride = pj;
                ride.slot0 = ahead;
                ride.nslots = (int)nv - ahead;
                int64_t rb = pack16_blocks(e, ride);
                if (rb > 0x3fffffffLL) return fail(SC_ERR_INVALID, "mask batch too large");
                ride_blocks = (uint32_t)rb;
                packed_ahead = ahead;
            }
        }
    }
    GridDesc g = grid_desc(e);
    uint64_t blocks = (g.ngroups + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffULL) return fail(SC_ERR_INVALID, "grid too large for one launch");
    // (rows are whole 16-byte groups -- the pitch is a multiple of 64 voxels -- so every kernel takes its
    // vector form, VEC = true; the element-wise forms remain in the templates for a layout without padding)
    dim3 grid((uint32_t)blocks), block(kBlock);
    const ViewDesc *vd = nullptr, *vpin = nullptr;
    if (nv > 1) {
        int rcs = step_begin(e);
        if (rcs) return rcs;
    }
    // a single view in brick form goes through the same kernels as a batch: it needs its descriptor
    // in the device array too
    const bool single_brick = nv == 1 && e->mode == SC_MODE_CARVE &&
                              fused_plan(e, nv, e->pending[0].occ != nullptr).brick;
    if (nv > 1 || single_brick) {
        if (!ordered && e->mode == SC_MODE_CARVE && e->view_order == 1 && nv == e->pending.size())
            order_views(e->pending);
        // descriptor ring: slots are reused only after a wrap, which waits for the stream
        if (nv > e->views_cap || e->views_head + nv > e->views_cap) {
            HIP_TRY(schost::wait_stream(e->stream));
            e->views_head = 0;
        }
        if (nv > e->views_cap) {
            if (e->views_dev) (void)hipFree(e->views_dev);
            if (e->views_pin) (void)hipHostFree(e->views_pin);
            e->views_dev = e->views_pin = nullptr;
            e->views_cap = 0;
            size_t cap = std::max<size_t>(nv * 8, 4096);  // (a wrap every 56 batches of 72 views; 1024 until round 4)
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->views_dev), cap * sizeof(ViewDesc)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->views_pin), cap * sizeof(ViewDesc),
                                  hipHostMallocDefault));
            e->views_cap = cap;
        }
        ViewDesc *pin = e->views_pin + e->views_head, *dev = e->views_dev + e->views_head;
        memcpy(pin, e->pending.data(), nv * sizeof(ViewDesc));
        e->views_head += nv;
        vd = dev;
        vpin = pin;
    }
    const ViewDesc &one = e->pending[0];
    int rc;
    // the descriptors reach the device array either by a copy on the stream, or -- brick form of
    // the fused carve -- through the flags kernel, which gets its own in its arguments
    bool desc_uploaded = false;
    auto upload_desc = [&]() -> int {
        if (desc_uploaded || vd == nullptr) return SC_OK;
        desc_uploaded = true;
        HIP_TRY(hipMemcpyAsync(const_cast<ViewDesc *>(vd), vpin, nv * sizeof(ViewDesc), hipMemcpyHostToDevice, e->stream));
        return SC_OK;
    };
    if (e->mode == SC_MODE_CARVE) {
        int32_t *st = static_cast<int32_t *>(e->state);
        int32_t init = init_bits_i32(e);
        // fused carve with survivor compaction: dense for the first `ndense` views, then lists
        const FusedPlan fp = fused_plan(e, nv, one.occ != nullptr);
        const int ndense = fp.ndense, nstage1 = fp.nstage1, flag_views = fp.flag_views;
        const uint32_t list_blocks = (uint32_t)e->list_blocks;
        const bool compact = fp.compact, brick = fp.brick, defer_stores = fp.defer_stores;
        // what sc_values_sparse may take from this launch's verdict bytes and lists (sc_sparse.h)
        e->sparse_exact = brick && e->fresh;
        e->sparse_late = false;
        Append ap{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
        int dense_views = (int)nv;
        const uint32_t bys = fp.bys, bzs = fp.bzs, nbricks = fp.nbricks, nstrips = fp.nstrips;
        const uint32_t dense_store_strips = fp.dense_store_strips;
        // strips set to -1 ahead of the verdicts, by fill blocks in front of the flags kernel's own (SpecFill): a fresh
        // volume whose fill is all the list stages' (so that everything behind the flags kernel that writes labels
        // comes later on the stream)
        uint32_t spec_strips = 0;
        if (fp.brick && fp.compact && fp.defer_stores && dense_store_strips == 0 && e->fresh && e->spec_share > 0)
            spec_strips = (uint32_t)((uint64_t)fp.nstrips * (uint64_t)e->spec_share / 16u);
        const bool desc_by_flags = brick && flag_views <= kFlagWaves;
        if (!desc_by_flags) {
            rc = upload_desc();
            if (rc) return rc;
        }
        if (compact || brick) {
            rc = ensure_ctl(e);
            if (rc) return rc;
            // list counters, overflow flag, live-brick count: this batch takes the block the previous
            // batch's final stage left zeroed (a memset only if there was no such stage)
            if (compact) {
                e->ctl_idx ^= 1;
                e->ctl = e->ctl2[e->ctl_idx];
                if (!e->ctl_clean[e->ctl_idx]) HIP_TRY(hipMemsetAsync(e->ctl, 0, sizeof(ListCtl), e->stream));
            }
            // (a launch without survivor stages keeps the block: its two counters alternate, see ListCtl)
            e->ctl_clean[e->ctl_idx] = false;
        }
        const uint32_t parity = (uint32_t)(e->flag_launches & 1u);
        // blocks of the flags kernel per sub-list of the candidate list (ListCtl::ncand): sub-list s holds the candidates of
        // blocks [s per, (s + 1) per) at cands + s per 64 -- at most the bricks of those blocks, so the lists fit in nbricks words
        const uint32_t cand_per = std::max<uint32_t>(1u, (uint32_t)(((nbricks + 63u) / 64u + kCandSub - 1) / kCandSub));
        if (brick) ++e->flag_launches;
        if (compact) {
            rc = ensure_lists(e);
            if (rc) return rc;
            ap.list = e->lists;
            ap.ctl = e->ctl;
            ap.subcap = e->subcap;
            dense_views = ndense;
        }
        // bulk units: brick form with survivor stages, every view with its cell level
        bool bulk_on = compact && brick && e->bulk_min > 0 && e->bulk != nullptr && e->items != nullptr;
        if (nv > 128) bulk_on = false;  // the units' verdict masks cover 128 views
        for (size_t q = 0; q < nv && bulk_on; ++q) bulk_on = e->pending[q].cmask != nullptr;
        if (bulk_on) {
            ap.bulk = e->bulk;
            ap.bulkcap = e->bulkcap;
            ap.bulk_min = (uint32_t)e->bulk_min;
        }
        e->last_bulk = bulk_on;
        LaunchTimer lt{e, SC_KERNEL_CARVE};
        if (!brick) {  // the brick form starts the timer after its flags kernel
            rc = lt.begin();
            if (rc) return rc;
        }
        if (nv == 1 && !brick) {
            // kStreamGroups groups per lane when the state is streamed through (see kernel)
            uint32_t per_block = !e->fresh ? kBlock * kStreamGroups : kBlock;
            dim3 grid1((uint32_t)((g.ngroups + per_block - 1) / per_block));
#define LAUNCH_CARVE1(F, V) \
    hipLaunchKernelGGL((carve_kernel_1<F, V>), grid1, block, 0, e->stream, st, g, one, init)
            if (e->fresh) {
                LAUNCH_CARVE1(true, true);
            } else {
                LAUNCH_CARVE1(false, true);
            }
#undef LAUNCH_CARVE1
        } else {
            if (brick) {
                // live-list walkers (whole groups of 8 XCDs), then store blocks, then packing riders; with
                // riders the walkers leave wavefront slots free for them
                const uint32_t nwalkers = ((uint32_t)(ride_blocks ? e->brick_walkers : e->list_blocks) + 7u) & ~7u;
                dim3 bgrid(nwalkers + dense_store_strips + ride_blocks);
                if (!e->dead) {
                    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->dead), (size_t)nbricks));
                    e->dead_clean = false;
                }
                const int dead_stale = e->dead_clean ? 0 : 1;  // the flags kernel rewrites them all
                e->dead_clean = true;
                // unit verdicts (cell level) by the views packed ahead, inside the dense stage
                int nverd = 0;
                const uint32_t verd_max_live = e->unit_cull == 2 ? 0xffffffffu : (uint32_t)(nbricks / 2);
                const uint32_t bulk_min_live = (uint32_t)((uint64_t)nbricks * (uint64_t)e->bulk_live / 16u);
                if (compact && e->unit_cull) {
                    nverd = std::min(packed_ahead, 16);
                    for (int q = 0; q < nverd; ++q)
                        if (e->pending[(size_t)q].cmask == nullptr) nverd = 0;
                }
                LaunchTimer ltf{e, SC_KERNEL_FLAGS};
                rc = ltf.begin();
                if (rc) return rc;
                FlagViews own{};
                DescCopy dc{nullptr, nullptr, 0u};
                if (desc_by_flags) {
                    for (int q = 0; q < flag_views; ++q) own.v[q] = e->pending[(size_t)q];
                    dc = DescCopy{reinterpret_cast<const uint32_t *>(vpin),
                                  reinterpret_cast<uint32_t *>(const_cast<ViewDesc *>(vd)),
                                  (uint32_t)(nv * sizeof(ViewDesc) / 4)};
                    desc_uploaded = true;
                }
                SpecFill sf{nullptr, 0u, 0u};
                if (spec_strips > 0) {
                    // strip s starts at column (s / bys) * ny + (s % bys) * 16; the columns are contiguous rows of nzp labels
                    const uint64_t cols = (uint64_t)(spec_strips / bys) * (uint64_t)e->ny + (uint64_t)(spec_strips % bys) * kBrickY;
                    sf = SpecFill{st, cols * (uint64_t)e->nzp * 4u, (uint32_t)e->spec_blocks};
                }
                hipLaunchKernelGGL(brick_flags_kernel, dim3(sf.nblocks + (nbricks + 63u) / 64u), dim3(64 * kFlagWaves), 0,
                                   e->stream, g, desc_by_flags ? static_cast<const ViewDesc *>(nullptr) : vd,
                                   flag_views, bys, bzs, nbricks, e->flags, e->live, e->ctl, own, dc,
                                   desc_by_flags ? vpin : vd, e->full_bricks ? packed_ahead : 0, (int)nv, e->dead,
                                   dead_stale, parity, compact ? static_cast<uint32_t *>(nullptr) : e->fill_list, sf,
                                   compact ? e->fill_list : static_cast<uint32_t *>(nullptr), cand_per);  // (the room of the fill list holds the candidate list when nothing fills from a list)
                e->last_parity = parity;
                rc = ltf.end();
                if (rc) return rc;
                rc = lt.begin();  // SC_KERNEL_CARVE times the dense kernel alone
                if (rc) return rc;
                if (!compact) {
                    // no survivor stages: walkers on the live list, fillers on the fill list
                    const dim3 lgrid(nwalkers + (uint32_t)std::max<int64_t>(e->fill_blocks, 64));
                    if (e->fresh)
                        hipLaunchKernelGGL((carve_brick_light_kernel<true>), lgrid, block, 0, e->stream, st, g, vd,
                                           dense_views, init, bys, bzs, e->live, e->fill_list, e->ctl, nwalkers, parity);
                    else
                        hipLaunchKernelGGL((carve_brick_light_kernel<false>), lgrid, block, 0, e->stream, st, g, vd,
                                           dense_views, init, bys, bzs, e->live, e->fill_list, e->ctl, nwalkers, parity);
                } else {
                    // (every dense view certified by the host: the instance without the general projection path)
                    // (a thinned-out unit may take one more pair of the views packed ahead: brick_voxels)
                    const int nextra = e->dense_extra ? std::max(0, std::min(2, packed_ahead - dense_views)) : 0;
                    bool dense_safe = e->safe_kernels != 0;
                    for (int q = 0; q < dense_views + nextra && dense_safe; ++q) dense_safe = e->pending[(size_t)q].safe != 0;
#define LAUNCH_BRICK(F, S)                                                                                          \
    hipLaunchKernelGGL((carve_brick_kernel<F, S>), bgrid, block, 0, e->stream, st, g, vd, dense_views, init, ap, bys, bzs, \
                       e->flags, e->live, e->ctl, nwalkers, dense_store_strips, ride, pack_form(e, ride), parity, nverd,    \
                       verd_max_live, bulk_min_live, nextra)
                    if (e->fresh && dense_safe) LAUNCH_BRICK(true, true);
                    else if (e->fresh) LAUNCH_BRICK(true, false);
                    else if (dense_safe) LAUNCH_BRICK(false, true);
                    else LAUNCH_BRICK(false, false);
#undef LAUNCH_BRICK
                }
            } else {
#define LAUNCH_CARVE(F, V)                                                                    \
    hipLaunchKernelGGL((carve_kernel<F, V>), grid, block, 0, e->stream, st, g, vd, dense_views, \
                       init, ap)
                if (e->fresh) {
                    LAUNCH_CARVE(true, true);
                } else {
                    LAUNCH_CARVE(false, true);
                }
#undef LAUNCH_CARVE
            }
        }
        HIP_TRY(hipGetLastError());
        rc = lt.end();
        if (rc) return rc;
        if (compact) {
            int s1 = (int)std::min<size_t>(nv, (size_t)ndense + nstage1);
            uint32_t *l0 = e->lists, *l1 = e->lists + (size_t)kSub * e->subcap;
            LaunchTimer lt2{e, SC_KERNEL_LIST};
            rc = lt2.begin();
            if (rc) return rc;
            int vg = (int)e->view_group;
            // open FULL candidates exist only when packing rode beside the dense stage
            CullStores none{nullptr, 0u, 0u, 0u, 0u, 0, 0, 0u, 0, 0u}, cs = none;
            if (ride_blocks) {
                e->sparse_late = true;
                // the riders have packed the rest of the masks: open FULL candidates get their answer
                // (a block per 64 entries of the candidate list, a persistent grid of at most 4096; without candidates
                // every block leaves after eight scalar loads)
                const uint32_t nconfirm = std::min<uint32_t>((nbricks + 63u) / 64u, 4096u);
                // (a candidate that fails takes the bulk units' road when the batch has a bulk list: UnitRoad)
                const UnitRoad road{(bulk_on && e->late_road) ? st : nullptr, init, e->fresh ? 1 : 0, nbricks};
                hipLaunchKernelGGL(brick_confirm_kernel, dim3(nconfirm), dim3(64 * kConfirmWaves), 0, e->stream, g, vd,
                                   packed_ahead, (int)nv, bys, bzs, e->flags, e->fill_list, cand_per, e->late, e->ctl, parity, road);
            }
            // Too few bulk units for their verdicts are taken by the first survivor stage as they are (UnitSpill); a
            // batch with a single (final) list stage has no such stage: its units are always asked
            const uint32_t unit_floor = (size_t)ndense + (size_t)nstage1 >= nv ? 0u : (uint32_t)e->bulk_floor;
            {
                // bulk units, late bricks, the dense fallback, the next batch's counters: one launch, always there
                // (what it finds to do is decided on the device)
                SpecialJob sj;
                memset(&sj, 0, sizeof sj);
                if (bulk_on)
                    sj.uj = UnitJob{e->bulk, e->bulkcap, e->items, e->itemcap, vd, (int32_t)nv, ndense, bys, bzs, st,
                                    e->lists, e->subcap, (uint32_t)e->item_bias, unit_floor};
                sj.lb = LateBricks{ride_blocks ? e->late : nullptr, nbricks, vd, e->flags, (int32_t)nv, init, e->fresh ? 1 : 0, bys, bzs};
                sj.next = e->ctl2[e->ctl_idx ^ 1];
                sj.rest = vd + ndense;
                sj.nrest = (int32_t)nv - ndense;
                sj.flags = brick ? e->flags : nullptr;
                sj.bricks_y = bys;
                sj.bricks_z = bzs;
                hipLaunchKernelGGL(carve_special_kernel, dim3((uint32_t)e->unit_blocks), dim3(64 * kFlagWaves), 0,
                                   e->stream, st, g, e->ctl, sj);
                e->ctl_clean[e->ctl_idx ^ 1] = true;
            }
            // final stage with deferred stores: e->defer_stores persistent list blocks (they leave
            // wavefront slots free) and one short store block per strip behind them
            dim3 fgrid(list_blocks);
            CullStores cs1 = none;
            dim3 grid1(list_blocks);
            if (defer_stores) {
                // the first list stage may take a share of the fill as well (it waits on memory)
                const uint32_t first = std::max(dense_store_strips, spec_strips);  // (one of the two is 0)
                uint32_t mid = first;
                if ((size_t)s1 < nv && e->stage1_store_share > 0) {
                    mid += (uint32_t)((uint64_t)(nstrips - first) * (uint64_t)e->stage1_store_share / 16u);
                    const uint32_t n1 = mid - first, f1 = std::min<uint32_t>((uint32_t)e->fill_blocks, n1);
                    cs1 = CullStores{e->flags, bys, bzs, mid, first, init == 0 ? 1 : init, e->fresh ? 1 : 0, f1, init, 0u};
                    grid1 = dim3((uint32_t)e->stage1_list_blocks + (f1 ? f1 : n1));
                }
                // (the final stage also walks the strips filled ahead, for their FULL / UNTOUCHED bricks)
                const uint32_t nf = nstrips - mid + spec_strips, ff = std::min<uint32_t>((uint32_t)e->fill_blocks, nf);
                cs = CullStores{e->flags, bys, bzs, nstrips, mid, init == 0 ? 1 : init, e->fresh ? 1 : 0, ff, init, spec_strips};
                fgrid = dim3((uint32_t)e->defer_stores + (ff ? ff : nf));
            }
            // every view of the batch certified by the host (certify_view: any real rig): the instances without the general path
            bool all_safe = e->safe_kernels != 0;
            for (size_t q = 0; q < nv && all_safe; ++q) all_safe = e->pending[q].safe != 0;
#define LAUNCH_LIST(FIN, GRID, ...)                                                                      \
    do {                                                                                                 \
        if ((FIN ? e->final_voxels : e->stage1_voxels) == 4) hipLaunchKernelGGL((carve_list_kernel<FIN, 4>), GRID, block, 0, e->stream, __VA_ARGS__); \
        else if ((FIN ? e->final_voxels : e->stage1_voxels) == 2 && all_safe) hipLaunchKernelGGL((carve_list_kernel<FIN, 2, true>), GRID, block, 0, e->stream, __VA_ARGS__); \
        else if ((FIN ? e->final_voxels : e->stage1_voxels) == 2) hipLaunchKernelGGL((carve_list_kernel<FIN, 2>), GRID, block, 0, e->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((carve_list_kernel<FIN, 1>), GRID, block, 0, e->stream, __VA_ARGS__);     \
    } while (0)
            // stage 1 (l0 -> l1), optional stage 2 (l1 -> l0), final stage on what is left
            int s2 = (int)std::min<size_t>(nv, (size_t)s1 + (size_t)e->stage2_views);
            uint32_t *nolist = nullptr;
            // the final stage also takes the work items of the bulk units
            const UnitItems noitems{nullptr, 0u, nullptr, 0u, 0u}, ui{bulk_on ? e->items : nullptr, e->itemcap, vd, bys, bzs};
            // ... the first one the bulk units of a batch that has too few for their verdicts (decided on the device)
            const UnitSpill nospill{nullptr, 0u, 0u, 0u, 0u}, us{bulk_on ? e->bulk : nullptr, e->bulkcap, unit_floor, bys, bzs};
            if ((size_t)s1 >= nv) {
                LAUNCH_LIST(true, fgrid, st, g, vd + ndense, s1 - ndense, l0, nolist, e->ctl, 0, 0, e->subcap, vg, cs, ui, nospill, s1 - ndense);
            } else {
                LAUNCH_LIST(false, grid1, st, g, vd + ndense, s1 - ndense, l0, l1, e->ctl, 0, 1, e->subcap, vg, cs1, noitems, us, (int)nv - ndense);
                if (s2 > s1 && (size_t)s2 < nv) {
                    LAUNCH_LIST(false, dim3(list_blocks), st, g, vd + s1, s2 - s1, l1, l0, e->ctl, 1, 2, e->subcap, vg, none, noitems, nospill, (int)nv - s1);
                    LAUNCH_LIST(true, fgrid, st, g, vd + s2, (int)nv - s2, l0, nolist, e->ctl, 2, 2, e->subcap, vg, cs, ui, nospill, (int)nv - s2);
                } else {
                    LAUNCH_LIST(true, fgrid, st, g, vd + s1, (int)nv - s1, l1, nolist, e->ctl, 1, 1, e->subcap, vg, cs, ui, nospill, (int)nv - s1);
                }
            }
#undef LAUNCH_LIST
            HIP_TRY(hipGetLastError());
            rc = lt2.end();
            if (rc) return rc;
        }
    } else {
        float *st = static_cast<float *>(e->state);
        rc = upload_desc();
        if (rc) return rc;
        // brick form: uint8 masks with uniformity flags on every view of the batch, a table, a grid it fits
        const uint32_t abys = (uint32_t)((e->ny + kBrickY - 1) / kBrickY), abzs = (uint32_t)((e->nz + kBrickZ - 1) / kBrickZ);
        bool abrick = nv > 1 && e->avg_brick && (uint64_t)e->npitch < 0x80000000ull &&
                      (uint64_t)e->planes * abys * abzs < 0x80000000ull;
        bool any_f32 = false;
        for (size_t q = 0; q < nv && abrick; ++q) {
            const ViewDesc &pd = e->pending[q];
            abrick = (pd.pad == 1 && e->lut_dev != nullptr && pd.occ != nullptr) || (pd.pad == 2 && pd.occ != nullptr);
            any_f32 |= pd.pad == 2;
        }
        if (abrick) {
            const uint32_t anb = (uint32_t)((uint64_t)e->planes * abys * abzs);
            const size_t need = (size_t)anb * nv;
            if (need > e->verd_cap) {
                HIP_TRY(schost::wait_stream(e->stream));
                if (e->verd) (void)hipFree(e->verd);
                e->verd = nullptr;
                e->verd_cap = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->verd), need));
                e->verd_cap = need;
            }
            if (any_f32 && need > e->verdf_cap) {  // the flat values of float32 views
                HIP_TRY(schost::wait_stream(e->stream));
                if (e->verdf) (void)hipFree(e->verdf);
                e->verdf = nullptr;
                e->verdf_cap = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->verdf), need * 4));
                e->verdf_cap = need;
            }
            uint32_t *verdf = any_f32 ? e->verdf : nullptr;
            LaunchTimer ltf{e, SC_KERNEL_FLAGS};
            rc = ltf.begin();
            if (rc) return rc;
            hipLaunchKernelGGL(avg_flags_kernel, dim3((anb + kBlock - 1) / kBlock, (uint32_t)nv), block, 0, e->stream,
                               g, vd, (int)nv, abys, abzs, anb, e->verd, verdf);
            rc = ltf.end();
            if (rc) return rc;
            LaunchTimer lta{e, SC_KERNEL_AVERAGE};
            rc = lta.begin();
            if (rc) return rc;
            if (e->fresh)
                hipLaunchKernelGGL(average_brick_kernel<true>, dim3(anb), block, 0, e->stream, st, g, vd, (int)nv,
                                   e->default_value, e->lut_dev, abys, abzs, e->verd, verdf);
            else
                hipLaunchKernelGGL(average_brick_kernel<false>, dim3(anb), block, 0, e->stream, st, g, vd, (int)nv,
                                   e->default_value, e->lut_dev, abys, abzs, e->verd, verdf);
            HIP_TRY(hipGetLastError());
            rc = lta.end();
            if (rc) return rc;
            rc = step_end(e, nv > 1);
            if (rc) return rc;
            e->fresh = false;
            e->pending.erase(e->pending.begin(), e->pending.begin() + (ptrdiff_t)nv);
            if (e->pending.empty()) arena_reset(e);
            return SC_OK;
        }
        LaunchTimer lt{e, SC_KERNEL_AVERAGE};
        rc = lt.begin();
        if (rc) return rc;
#define LAUNCH_AVG(F, V)                                                                         \
    do {                                                                                         \
        if (nv == 1)                                                                             \
            hipLaunchKernelGGL((average_kernel_1<F, V>), grid, block, 0, e->stream, st, g, one,  \
                               e->default_value, e->lut_dev);                                    \
        else                                                                                     \
            hipLaunchKernelGGL((average_kernel<F, V>), grid, block, 0, e->stream, st, g, vd,     \
                               (int)nv, e->default_value, e->lut_dev);                           \
    } while (0)
        if (e->fresh) {
            LAUNCH_AVG(true, true);
        } else {
            LAUNCH_AVG(false, true);
        }
#undef LAUNCH_AVG
        HIP_TRY(hipGetLastError());
        rc = lt.end();
        if (rc) return rc;
    }
    rc = step_end(e, nv > 1);
    if (rc) return rc;
    e->fresh = false;
    e->pending.erase(e->pending.begin(), e->pending.begin() + (ptrdiff_t)nv);
    if (e->pending.empty()) arena_reset(e);  // masks of launched views are dead in stream order
    return SC_OK;
}

int after_enqueue(sc_engine *e) {
    if (e->views_per_launch > 0) {
        while ((int64_t)e->pending.size() >= e->views_per_launch) {
            int rc = flush(e, (size_t)e->views_per_launch);
            if (rc) return rc;
        }
        return SC_OK;
    }
    if ((int64_t)e->pending.size() >= e->max_pending) return flush(e);
    return SC_OK;
}

int check_view_args(const sc_engine *e, const float *K, const float *R, const float *t,
                    const void *mask, int H, int W) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    if (!K || !R || !t || !mask) return fail(SC_ERR_INVALID, "null view argument");
    // (a view's bits: below 2^32 bytes; the words of a strip of its bit tiles, H rounded up to 32: below 2^24)
    if (H <= 0 || W <= 0 || H > (1 << 24) - 32 || W > (1 << 24) || (int64_t)H * W > ((int64_t)1 << 34))
        return fail(SC_ERR_INVALID, "bad mask shape %d x %d", H, W);
    return SC_OK;
}

// Engine streams are kept between engines (round 5).  Creating a non-blocking stream is a hardware queue's worth of
// set-up in the runtime: 84-139 ms for the first one of a process and, now and then, 6-40 ms for a later one -- the
// "37 ms first batch" of a fresh engine that round 4's bench line showed on the driver's box (SC_TRACE_ALLOC=1 names the
// call).  A destroyed engine's stream (idle: sc_destroy has waited for it) goes on a short per-device list and the next
// engine on that device takes it from there; a process's first engine still pays the first creation, once.
std::mutex g_stream_mu;
std::vector<std::pair<int, hipStream_t>> g_stream_pool;  // (device, idle stream)
constexpr size_t kStreamPoolMax = 16;

// sc_prewarm: a thread that is bringing the runtime up and making the device's first stream; whoever wants a stream of
// that device waits for it (one creation, not two side by side) and finds the stream on the list
std::condition_variable g_prewarm_cv;
int g_prewarm_running[64] = {0};

hipError_t take_stream(int device, hipStream_t *out) {
    {
        std::unique_lock<std::mutex> lk(g_stream_mu);
        if (device >= 0 && device < 64) g_prewarm_cv.wait(lk, [&] { return g_prewarm_running[device] == 0; });
        for (size_t i = 0; i < g_stream_pool.size(); ++i)
            if (g_stream_pool[i].first == device) {
                *out = g_stream_pool[i].second;
                g_stream_pool.erase(g_stream_pool.begin() + (ptrdiff_t)i);
                return hipSuccess;
            }
    }
    return sctrace::timed("hipStreamCreate", __LINE__, 0, [&] { return hipStreamCreateWithFlags(out, hipStreamNonBlocking); });
}

void give_stream_back(int device, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        if (g_stream_pool.size() < kStreamPoolMax) {
            g_stream_pool.emplace_back(device, s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}

// The device half of an engine's set-up: the device is there and is a gfx950, a stream, the state.
int device_setup(sc_engine *e) {
    const int device = e->device;
    // `device` is a HIP ordinal; only that device has to be a gfx950
    int ndev = 0;
    hipError_t hq = hipGetDeviceCount(&ndev);
    if (hq != hipSuccess) return fail(SC_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(hq));
    if (device < 0 || device >= ndev)
        return fail(SC_ERR_DEVICE, "device %d not available (%d HIP device(s) visible)", device, ndev);
    {
        hipDeviceProp_t prop;
        hq = sctrace::timed("hipGetDeviceProperties", __LINE__, 0, [&] { return hipGetDeviceProperties(&prop, device); });
        if (hq != hipSuccess) return fail(SC_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(hq));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SC_ERR_DEVICE, "device %d is %s; this engine is built for gfx950 only", device,
                        prop.gcnArchName);
    }
    hipError_t he = sctrace::timed("hipSetDevice", __LINE__, 0, [&] { return hipSetDevice(device); });
    if (he == hipSuccess) he = take_stream(device, &e->own_stream);
    if (he == hipSuccess) he = hipMalloc(&e->state, (size_t)e->npitch * 4);
    if (he != hipSuccess)
        return fail(he == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE, "engine setup failed: %s", hipGetErrorString(he));
    e->stream = e->own_stream;
    return SC_OK;
}

// The engine owns the x-planes  i0, i0 + istride, ...  (`planes` of them) of the global grid.
int create(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t i0, int64_t istride,
           int64_t planes, const float *origin, float vs, int mode, float default_value, int device, bool deferred = false) {
    if (!out) return fail(SC_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!origin) return fail(SC_ERR_INVALID, "null origin");
    if (nx <= 0 || ny <= 0 || nz <= 0) return fail(SC_ERR_INVALID, "shape must be positive");
    // int -> float of an index must be exact
    if (nx > (1 << 24) || ny > (1 << 24) || nz > (1 << 24))
        return fail(SC_ERR_INVALID, "axis longer than 2^24 voxels");
    if (i0 < 0 || istride < 1 || planes < 1 || i0 + (planes - 1) * istride >= nx)
        return fail(SC_ERR_INVALID, "bad slab / plane set (first %lld, stride %lld, planes %lld of %lld)",
                    (long long)i0, (long long)istride, (long long)planes, (long long)nx);
    if (mode != SC_MODE_CARVE && mode != SC_MODE_AVERAGE)
        return fail(SC_ERR_INVALID, "unknown mode %d", mode);
    sc_engine *e = new (std::nothrow) sc_engine();
    if (!e) return fail(SC_ERR_NOMEM, "host allocation failed");
    e->device = device;
    e->mode = mode;
    e->nx = nx; e->ny = ny; e->nz = nz; e->i0 = i0; e->istride = istride; e->planes = planes;
    e->n = planes * ny * nz;
    e->nzp = (nz + 63) / 64 * 64;
    e->npitch = planes * ny * e->nzp;
    memcpy(e->origin, origin, sizeof e->origin);
    e->vs = vs;
    e->default_value = default_value;
    e->fresh = true;
    if (deferred) {
        // the device half on a thread of its own: the caller goes on (reads its files, decodes them) and the first
        // call that needs the device joins -- and takes the failure, if there is one
        e->setup_pending = true;
        try {
            e->setup_thread = std::thread([e]() {
                e->setup_rc = device_setup(e);
                if (e->setup_rc != SC_OK) e->setup_err = g_err;  // (the thread's own message)
            });
        } catch (...) {
            e->setup_pending = false;
            int rc = device_setup(e);
            if (rc) {
                sc_destroy(e);
                return rc;
            }
        }
        *out = e;
        return SC_OK;
    }
    int rc = device_setup(e);
    if (rc) {
        sc_destroy(e);
        return rc;
    }
    *out = e;
    return SC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

int sc_abi_version(void) { return SC_ABI_VERSION; }

const char *sc_last_error(void) { return g_err.c_str(); }

int sc_device_count(int *count) {
    if (!count) return fail(SC_ERR_INVALID, "null count");
    *count = 0;
    int n = 0;
    hipError_t he = hipGetDeviceCount(&n);
    if (he != hipSuccess) return fail(SC_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(he));
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
        if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;  // other devices are simply not ours
    }
    *count = ok;
    return SC_OK;
}

int sc_create(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, const float origin[3],
              float voxel_size, int mode, float default_value, int device) {
    return create(out, nx, ny, nz, 0, 1, nx, origin, voxel_size, mode, default_value, device);
}

int sc_create_slab(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t i0, int64_t i1,
                   const float origin[3], float voxel_size, int mode, float default_value,
                   int device) {
    if (i0 < 0 || i1 > nx || i0 >= i1) return fail(SC_ERR_INVALID, "bad slab [%lld, %lld)", (long long)i0, (long long)i1);
    return create(out, nx, ny, nz, i0, 1, i1 - i0, origin, voxel_size, mode, default_value, device);
}

int sc_create_cyclic(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t first, int64_t stride,
                     const float origin[3], float voxel_size, int mode, float default_value,
                     int device) {
    if (stride < 1 || first < 0 || first >= stride || first >= nx)
        return fail(SC_ERR_INVALID, "bad plane set (first %lld, stride %lld)", (long long)first, (long long)stride);
    int64_t planes = (nx - first + stride - 1) / stride;
    return create(out, nx, ny, nz, first, stride, planes, origin, voxel_size, mode, default_value, device);
}

// What cl.py:29-30 does at import -- the context and the queue -- without blocking the importer: a thread of the library's
// initialises the runtime on `device` and creates the device's first non-blocking stream (84-147 ms in a fresh process),
// which the first engine on that device then takes from the idle-stream list.  Returns at once; errors are the first
// engine's to report (it runs the same calls).
int sc_prewarm(int device) {
    if (device < 0 || device >= 64) return fail(SC_ERR_INVALID, "device %d", device);
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        if (g_prewarm_running[device]) return SC_OK;
        for (auto &p : g_stream_pool)
            if (p.first == device) return SC_OK;  // a stream is waiting already
        g_prewarm_running[device] = 1;
    }
    try {
        std::thread([device]() {
            hipStream_t s = nullptr;
            int n = 0;
            bool ok = hipGetDeviceCount(&n) == hipSuccess && device < n && hipSetDevice(device) == hipSuccess &&
                      hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
            std::lock_guard<std::mutex> lk(g_stream_mu);
            if (ok) g_stream_pool.emplace_back(device, s);
            g_prewarm_running[device] = 0;
            g_prewarm_cv.notify_all();
        }).detach();
    } catch (...) {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        g_prewarm_running[device] = 0;
        g_prewarm_cv.notify_all();
    }
    return SC_OK;
}

int sc_create_ex(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t first, int64_t stride, int64_t planes,
                 const float origin[3], float voxel_size, int mode, float default_value, int device, int flags) {
    if (flags & ~SC_CREATE_DEFERRED) return fail(SC_ERR_INVALID, "unknown creation flags %d", flags);
    return create(out, nx, ny, nz, first, stride, planes, origin, voxel_size, mode, default_value, device,
                  (flags & SC_CREATE_DEFERRED) != 0);
}

void sc_destroy(sc_engine *e) {
    if (!e) return;
    if (e->setup_pending && e->setup_thread.joinable()) e->setup_thread.join();
    e->setup_pending = false;
    if (e->setup_rc != SC_OK) {  // the device half never came up: nothing but host memory to give back
        if (e->own_stream) give_stream_back(e->device, e->own_stream);
        delete e;
        return;
    }
    (void)hipSetDevice(e->device);
    if (e->stream) (void)schost::wait_stream(e->stream);
    for (int k = 0; k < kNumKernels; ++k)
        for (auto &tl : e->timed[k]) {
            (void)hipEventDestroy(tl.start);
            (void)hipEventDestroy(tl.stop);
        }
    if (e->step_open) (void)hipEventDestroy(e->step_start);
    if (e->span_open) (void)hipEventDestroy(e->span_start);
    for (auto ev : e->event_pool) (void)hipEventDestroy(ev);
    for (auto &c : e->chunks) (void)hipFree(c.base);
    for (int s = 0; s < kSlots; ++s) {
        if (e->pin[s]) (void)hipHostFree(e->pin[s]);
        if (e->raw[s]) (void)hipFree(e->raw[s]);
        if (e->slot_ev[s]) (void)hipEventDestroy(e->slot_ev[s]);
    }
    for (auto &a : e->hb) {
        if (a.pin) (void)hipHostFree(a.pin);
        if (a.dev) (void)hipFree(a.dev);
        if (a.ev) (void)hipEventDestroy(a.ev);
    }
    if (e->views_dev) (void)hipFree(e->views_dev);
    if (e->views_pin) (void)hipHostFree(e->views_pin);
    if (e->narrow) (void)hipFree(e->narrow);
    if (e->packed_labels) (void)hipFree(e->packed_labels);
    for (int q = 0; q < 2; ++q)
        if (e->sparse_buf[q]) (void)hipFree(e->sparse_buf[q]);
    for (int q = 0; q < 2; ++q)
        if (e->sparse_busy[q]) (void)hipEventDestroy(e->sparse_busy[q]);
    if (e->packed_busy) (void)hipEventDestroy(e->packed_busy);
    for (int q = 0; q < 2; ++q)
        if (e->sparse_hdr_pin[q]) (void)hipHostFree(e->sparse_hdr_pin[q]);
    if (e->sparse_cnt) (void)hipFree(e->sparse_cnt);
    if (e->sparse_work) (void)hipFree(e->sparse_work);
    if (e->wire_stage) (void)hipHostFree(e->wire_stage);
    if (e->dense) (void)hipFree(e->dense);
    if (e->verd) (void)hipFree(e->verd);
    if (e->verdf) (void)hipFree(e->verdf);
    if (e->dead) (void)hipFree(e->dead);
    if (e->lut_dev) (void)hipFree(e->lut_dev);
    if (e->lists) (void)hipFree(e->lists);
    if (e->ctl2[0]) (void)hipFree(e->ctl2[0]);
    if (e->items) (void)hipFree(e->items);
    if (e->state) (void)hipFree(e->state);
    if (e->own_stream) {
        (void)schost::wait_stream(e->own_stream);  // (the engine may have worked on a caller's stream: its own is idle now)
        give_stream_back(e->device, e->own_stream);
    }
    delete e;
}

int sc_clear(sc_engine *e) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    e->pending.clear();
    e->hp_pending.clear();
    e->hb[e->hb_cur].used = 0;  // (nothing of it was uploaded)
    e->deferred.on = false;
    e->dead_clean = false;  // the labels go back to default_value: no brick is known to be all -1
    e->sparse_exact = false;
    arena_reset(e);
    if (e->step_open) {  // the views of an open SC_KERNEL_STEP window are gone: no sample for them
        e->event_pool.push_back(e->step_start);
        e->step_open = false;
    }
    e->fresh = true;  // materialised lazily: a fused launch never needs to read it
    return SC_OK;
}

int sc_set_option(sc_engine *e, int key, int64_t value) {
    if (e) {
        int rcw = wait_setup(e);  // (some keys create events or move buffers)
        if (rcw) return rcw;
    }
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    switch (key) {
        case SC_OPT_VIEWS_PER_LAUNCH:
            if (value < 0) return fail(SC_ERR_INVALID, "views_per_launch must be >= 0");
            e->views_per_launch = value;
            return SC_OK;
        case SC_OPT_VIEW_ORDER:
            if (value != 0 && value != 1) return fail(SC_ERR_INVALID, "view_order must be 0 or 1");
            e->view_order = value;
            return SC_OK;
        case SC_OPT_TIME_KERNELS:
            e->time_kernels = value == 2 ? 2 : (value ? 1 : 0);
            return SC_OK;
        case SC_OPT_COMPACT:
            e->compact = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_DENSE_VIEWS:
            if (value < 1 || value > 64) return fail(SC_ERR_INVALID, "dense_views must be in [1, 64]");
            e->dense_views = value;
            return SC_OK;
        case SC_OPT_STAGE1_VIEWS:
            if (value < 1 || value > 4096) return fail(SC_ERR_INVALID, "stage1_views must be in [1, 4096]");
            e->stage1_views = value;
            return SC_OK;
        case SC_OPT_LIST_BLOCKS:
            if (value < 1 || value > 65536) return fail(SC_ERR_INVALID, "list_blocks must be in [1, 65536]");
            e->list_blocks = value;
            return SC_OK;
        case SC_OPT_BRICK:
            e->brick = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_STAGE2_VIEWS:
            if (value < 0 || value > 4096) return fail(SC_ERR_INVALID, "stage2_views must be in [0, 4096]");
            e->stage2_views = value;
            return SC_OK;
        case SC_OPT_FULL_BRICKS:
            e->full_bricks = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_AVG_TILE_F32:
            e->avg_tile_f32 = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_AVG_BRICK:
            e->avg_brick = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_STAGE1_STORE_SHARE:
            if (value < 0 || value > 16) return fail(SC_ERR_INVALID, "stage1_store_share must be in [0, 16]");
            e->stage1_store_share = value;
            return SC_OK;
        case SC_OPT_STAGE1_LIST_BLOCKS:
            if (value < 1 || value > 65536) return fail(SC_ERR_INVALID, "stage1_list_blocks must be in [1, 65536]");
            e->stage1_list_blocks = value;
            return SC_OK;
        case SC_OPT_DEFER_SHARE:
            if (value < 0 || value > 16) return fail(SC_ERR_INVALID, "defer_share must be in [0, 16]");
            e->defer_share = value;
            return SC_OK;
        case SC_OPT_DEFER_STORES:
            if (value < 0 || value > 65536) return fail(SC_ERR_INVALID, "defer_stores must be in [0, 65536]");
            e->defer_stores = value;
            return SC_OK;
        case SC_OPT_PACK_ROWS:
            if (value != 0 && value != 1 && value != 2 && value != 3 && value != 4 && value != 8)
                return fail(SC_ERR_INVALID, "pack_rows must be 0, 1, 2, 3, 4 or 8");
            e->pack_rows = value;
            return SC_OK;
        case SC_OPT_VIEW_BRICK:
            e->view_brick = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_RESERVE_EVENTS: {
            if (value < 0 || value > 65536) return fail(SC_ERR_INVALID, "reserve_events must be in [0, 65536]");
            int rc = use_device(e);
            if (rc) return rc;
            while ((int64_t)e->event_pool.size() < value) {
                hipEvent_t ev;
                HIP_TRY(hipEventCreate(&ev));
                e->event_pool.push_back(ev);
            }
            return SC_OK;
        }
        case SC_OPT_STAGE1_VOXELS:
            if (value != 1 && value != 2 && value != 4) return fail(SC_ERR_INVALID, "stage1_voxels must be 1, 2 or 4");
            e->stage1_voxels = value;
            return SC_OK;
        case SC_OPT_FINAL_VOXELS:
            if (value != 1 && value != 2 && value != 4) return fail(SC_ERR_INVALID, "final_voxels must be 1, 2 or 4");
            e->final_voxels = value;
            return SC_OK;
        case SC_OPT_FILL_BLOCKS:
            if (value < 0 || value > 65536) return fail(SC_ERR_INVALID, "fill_blocks must be in [0, 65536]");
            e->fill_blocks = value;
            return SC_OK;
        case SC_OPT_PACK_RIDE:
            e->pack_ride = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_BRICK_WALKERS:
            if (value < 8 || value > 65536) return fail(SC_ERR_INVALID, "brick_walkers must be in [8, 65536]");
            e->brick_walkers = value;
            return SC_OK;
        case SC_OPT_FLAG_VIEWS:
            if (value < 0) return fail(SC_ERR_INVALID, "flag_views must be >= 0");
            e->flag_views = value;
            return SC_OK;
        case SC_OPT_VIEW_GROUP:
            if (value < 1 || value > 4096) return fail(SC_ERR_INVALID, "view_group must be in [1, 4096]");
            e->view_group = value;
            return SC_OK;
        case SC_OPT_BULK_MIN:
            if (value < 0 || value > 256) return fail(SC_ERR_INVALID, "bulk_min must be in [0, 256]");
            e->bulk_min = value;
            return SC_OK;
        case SC_OPT_ITEM_BIAS:
            if (value < 0 || value > 64) return fail(SC_ERR_INVALID, "item_bias must be in [0, 64]");
            e->item_bias = value;
            return SC_OK;
        case SC_OPT_UNIT_CULL:
            if (value < 0 || value > 2) return fail(SC_ERR_INVALID, "unit_cull must be 0, 1 or 2");
            e->unit_cull = value;
            return SC_OK;
        case SC_OPT_BULK_LIVE:
            if (value < 0 || value > 16) return fail(SC_ERR_INVALID, "bulk_live must be in [0, 16]");
            e->bulk_live = value;
            return SC_OK;
        case SC_OPT_BULK_FLOOR:
            if (value < 0 || value > 0x7fffffffLL) return fail(SC_ERR_INVALID, "bulk_floor must be in [0, 2^31)");
            e->bulk_floor = value;
            return SC_OK;
        case SC_OPT_SPEC_SHARE:
            if (value < 0 || value > 16) return fail(SC_ERR_INVALID, "spec_share must be in [0, 16]");
            e->spec_share = value;
            return SC_OK;
        case SC_OPT_SPEC_BLOCKS:
            if (value < 1 || value > 4096) return fail(SC_ERR_INVALID, "spec_blocks must be in [1, 4096]");
            e->spec_blocks = value;
            return SC_OK;
        case SC_OPT_LATE_ROAD:
            e->late_road = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_DENSE_EXTRA:
            e->dense_extra = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_SAFE_KERNELS:
            e->safe_kernels = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_LDS_TILES:  // the experiment was measured and removed (DESIGN_APPENDIX.md 12): accepted, no effect
            return SC_OK;
        case SC_OPT_HOST_PACK:
            if (!e->hp_pending.empty()) return fail(SC_ERR_STATE, "host-packed views are pending: flush first");
            e->host_pack = value ? 1 : 0;
            return SC_OK;
        case SC_OPT_HOST_THREADS:
            if (value < 0 || value > 256) return fail(SC_ERR_INVALID, "host_threads must be in [0, 256]");
            schost::pool_set_threads((int)value);  // process-wide; takes effect before the pool's first use
            return SC_OK;
        case SC_OPT_LIST_CAP:
            if (value < 0 || value > 0x7fffffffLL) return fail(SC_ERR_INVALID, "list_cap must be in [0, 2^31)");
            e->list_cap = value;
            return SC_OK;
        case SC_OPT_UNIT_BLOCKS:
            if (value < 1 || value > 65536) return fail(SC_ERR_INVALID, "unit_blocks must be in [1, 65536]");
            e->unit_blocks = value;
            return SC_OK;
        case SC_OPT_MAX_PENDING:
            if (value < 1) return fail(SC_ERR_INVALID, "max_pending must be >= 1");
            e->max_pending = value;
            return SC_OK;
        default:
            return fail(SC_ERR_INVALID, "unknown option %d", key);
    }
}

int sc_set_lut(sc_engine *e, const float *lut256) {
    if (!e || !lut256) return fail(SC_ERR_INVALID, "null argument");
    if (e->mode != SC_MODE_AVERAGE) return fail(SC_ERR_STATE, "the table belongs to averaging engines");
    int rc = use_device(e);
    if (rc) return rc;
    rc = flush(e);  // views already enqueued keep the old table
    if (rc) return rc;
    HIP_TRY(schost::wait_stream(e->stream));
    if (!e->lut_dev) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->lut_dev), 256 * sizeof(float)));
    HIP_TRY(hipMemcpy(e->lut_dev, lut256, 256 * sizeof(float), hipMemcpyHostToDevice));
    return SC_OK;
}

int sc_set_stream(sc_engine *e, void *hip_stream) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    rc = flush(e);
    if (rc) return rc;
    HIP_TRY(schost::wait_stream(e->stream));
    e->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->own_stream;
    return SC_OK;
}

int sc_order_after(sc_engine *e, void *producer_stream) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    // NULL is the legacy default stream here (torch's default stream has handle 0); the engine's own
    // stream is non-blocking, so it does NOT synchronise with that stream by itself
    // (the handle 0 itself: hipStreamLegacy is not understood by every runtime this library meets --
    // torch's bundled one crashed on it)
    hipStream_t prod = static_cast<hipStream_t>(producer_stream);
    if (prod != nullptr && prod == e->stream) return SC_OK;  // same stream: already in order
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t he = hipEventRecord(ev, prod);
    if (he == hipSuccess) he = hipStreamWaitEvent(e->stream, ev, 0);
    (void)hipEventDestroy(ev);  // released once the wait has been satisfied
    if (he != hipSuccess) return fail(SC_ERR_DEVICE, "ordering after the producer stream failed: %s", hipGetErrorString(he));
    return SC_OK;
}

int sc_order_before(sc_engine *e, void *consumer_stream) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    int rc = use_device(e);
    if (rc) return rc;
    hipStream_t cons = static_cast<hipStream_t>(consumer_stream);
    if (cons != nullptr && cons == e->stream) return SC_OK;  // same stream: already in order
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t he = hipEventRecord(ev, e->stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(cons, ev, 0);
    (void)hipEventDestroy(ev);  // released once the wait has been satisfied
    if (he != hipSuccess) return fail(SC_ERR_DEVICE, "ordering the consumer stream behind the engine failed: %s", hipGetErrorString(he));
    return SC_OK;
}

int sc_process_view(sc_engine *e, const float K[4], const float R[9], const float t[3],
                    const void *mask, int H, int W, int mask_dtype, int64_t row_stride_bytes) {
    int rc = check_view_args(e, K, R, t, mask, H, W);
    if (rc) return rc;
    rc = check_dtype(e, mask_dtype);
    if (rc) return rc;
    rc = use_device(e);
    if (rc) return rc;
    rc = materialize_deferred(e);  // a device batch waiting for its flush: packed now, order as given
    if (rc) return rc;
    size_t es = elem_size(mask_dtype);
    size_t row = (size_t)W * es;
    if (row_stride_bytes == 0) row_stride_bytes = (int64_t)row;
    if (row_stride_bytes < (int64_t)row) return fail(SC_ERR_INVALID, "row stride smaller than a row");
    if (e->mode == SC_MODE_CARVE && e->host_pack) {
        // the caller's buffer is consumed here (its bits are in the arena when this returns)
        rc = enqueue_hostbits(e, K, R, t, mask, H, W, mask_dtype, row_stride_bytes);
        if (rc) return rc;
        return after_enqueue(e);
    }
    size_t bytes = row * (size_t)H;
    rc = ensure_slots(e, bytes);
    if (rc) return rc;
    int s = e->next_slot;
    e->next_slot = (s + 1) % kSlots;
    if (e->slot_armed[s]) {
        HIP_TRY(schost::wait_event(e->slot_ev[s]));
        e->slot_armed[s] = false;
    }
    // consume the caller's buffer now (tight rows in the pinned slot)
    if (row_stride_bytes == (int64_t)row) {
        memcpy(e->pin[s], mask, bytes);
    } else {
        for (int r = 0; r < H; ++r)
            memcpy(static_cast<char *>(e->pin[s]) + (size_t)r * row,
                   static_cast<const char *>(mask) + (size_t)r * row_stride_bytes, row);
    }
    if (e->mode == SC_MODE_CARVE) {
        HIP_TRY(hipMemcpyAsync(e->raw[s], e->pin[s], bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->slot_ev[s], e->stream));
        e->slot_armed[s] = true;
        rc = enqueue_pack(e, 1, K, R, t, e->raw[s], H, W, mask_dtype, (int64_t)row, (int64_t)bytes);
        if (rc) return rc;
    } else if (mask_dtype == SC_MASK_U8_LUT) {
        HIP_TRY(hipMemcpyAsync(e->raw[s], e->pin[s], bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->slot_ev[s], e->stream));
        e->slot_armed[s] = true;
        rc = enqueue_tile8(e, 1, K, R, t, e->raw[s], H, W, (int64_t)row, (int64_t)bytes);
        if (rc) return rc;
    } else if (e->avg_tile_f32) {
        HIP_TRY(hipMemcpyAsync(e->raw[s], e->pin[s], bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->slot_ev[s], e->stream));
        e->slot_armed[s] = true;
        rc = enqueue_tilef32(e, 1, K, R, t, e->raw[s], H, W, (int64_t)row, (int64_t)bytes);
        if (rc) return rc;
    } else {
        void *dst = nullptr;
        rc = arena_alloc(e, bytes, &dst);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(dst, e->pin[s], bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->slot_ev[s], e->stream));
        e->slot_armed[s] = true;
        ViewDesc d;
        fill_desc(e, d, K, R, t, dst, H, W);
        e->pending.push_back(d);
    }
    return after_enqueue(e);
}

int sc_process_views(sc_engine *e, int V, const float *K, const float *R, const float *t,
                     const void *const *masks, int H, int W, int mask_dtype,
                     int64_t row_stride_bytes) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    if (V < 0 || (V > 0 && (!K || !R || !t || !masks))) return fail(SC_ERR_INVALID, "bad view batch");
    for (int q = 0; q < V; ++q) {
        int rc = sc_process_view(e, K + 4 * q, R + 9 * q, t + 3 * q, masks[q], H, W, mask_dtype,
                                 row_stride_bytes);
        if (rc) return rc;
    }
    return SC_OK;
}

int sc_process_png_views(sc_engine *e, int V, const float *K, const float *R, const float *t, const void *const *png,
                         const int64_t *sizes, int invert, int threads) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    if (V < 0 || (V > 0 && (!K || !R || !t || !png || !sizes))) return fail(SC_ERR_INVALID, "bad view batch");
    if (V == 0) return SC_OK;
    if (e->mode != SC_MODE_CARVE) return fail(SC_ERR_STATE, "encoded masks are carve masks (the averaging path converts pixels on the host)");
    std::vector<int> Ws((size_t)V), Hs((size_t)V);
    std::vector<size_t> offs((size_t)V);
    size_t total = 0;
    for (int q = 0; q < V; ++q) {
        if (!png[q]) return fail(SC_ERR_INVALID, "null file %d", q);
        if (sc_png_info(png[q], sizes[q], &Ws[(size_t)q], &Hs[(size_t)q]) != SC_OK)
            return fail(SC_ERR_INVALID, "file %d: %s", q, sc_png_last_error());
        // the limits check_view_args puts on a mask, before a pixel buffer of that size is asked for
        if (Ws[(size_t)q] <= 0 || Hs[(size_t)q] <= 0 || Ws[(size_t)q] > (1 << 24) || Hs[(size_t)q] > (1 << 24) ||
            (int64_t)Ws[(size_t)q] * Hs[(size_t)q] > (int64_t)1 << 31)
            return fail(SC_ERR_INVALID, "file %d: a %d x %d mask is beyond the limits of a view", q, Ws[(size_t)q], Hs[(size_t)q]);
        offs[(size_t)q] = total;
        total += ((size_t)Hs[(size_t)q] * (size_t)((Ws[(size_t)q] + kTile - 1) / kTile) * 4 + 255) & ~(size_t)255;
    }
    // An engine whose device half is still coming up (sc_create_ex, SC_CREATE_DEFERRED): the files are decoded into
    // plain host memory beside it and copied to the page-locked arena once the device is there -- 14 MB, 2 ms, against
    // the 15 ms of decoding that would otherwise wait for 130-240 ms of runtime set-up
    const bool beside_setup = e->setup_pending;
    std::vector<char> heap;
    int rc = SC_OK;
    char *base = nullptr;
    if (beside_setup) {
        try {
            heap.resize(total);
        } catch (...) {
            return fail(SC_ERR_NOMEM, "out of host memory while decoding the masks");
        }
        base = heap.data();
    } else {
        rc = use_device(e);
        if (rc) return rc;
        rc = materialize_deferred(e);
        if (rc) return rc;
        rc = hostbits_reserve(e, total, &base);  // one reservation: the arena does not move while the threads write
        if (rc) return rc;
    }
    // decode + pack, a file per thread at a time.  Threads of this call's own (16 by default: inflate is the floor
    // of the files -> volume time, ~1.3 ms per mask and thread, and the pool's 8 are sized for the per-mask hand-overs)
    int nth = threads > 0 ? threads : 16;
    nth = std::min(std::min(nth, V), 64);
    std::atomic<int> next{0}, bad{-1};
    std::atomic<bool> nomem{false};
    const uint8_t flip = invert ? 255 : 0;
    // (nothing may leave a thread function or this C entry point as an exception -- std::terminate: the pixel buffer's
    // allocation and the threads' creation are caught and reported as SC_ERR_NOMEM)
    auto work = [&]() {
        try {
            std::vector<uint8_t> pix;
            for (;;) {
                const int q = next.fetch_add(1, std::memory_order_relaxed);
                if (q >= V || bad.load(std::memory_order_relaxed) >= 0) return;
                const int W = Ws[(size_t)q], H = Hs[(size_t)q];
                pix.resize((size_t)W * H);
                if (sc_png_decode_gray8(png[q], sizes[q], pix.data(), W, H) != SC_OK) {
                    int expect = -1;
                    bad.compare_exchange_strong(expect, q);
                    return;
                }
                schost::pack_rows(pix.data(), W, W, 0, H, reinterpret_cast<uint32_t *>(base + offs[(size_t)q]), (W + kTile - 1) / kTile, 1, flip);
            }
        } catch (...) {
            nomem.store(true);
            int expect = -1;
            bad.compare_exchange_strong(expect, V);  // stops the others
        }
    };
    {
        std::vector<std::thread> pool;
        try {
            pool.reserve((size_t)nth);
            for (int i = 1; i < nth; ++i) pool.emplace_back(work);
        } catch (...) {
            // (fewer threads than asked for: the ones that exist and this one do the work)
        }
        work();
        for (auto &th : pool) th.join();
    }
    if (nomem.load()) {
        if (!beside_setup) e->hb[e->hb_cur].used -= total;
        return fail(SC_ERR_NOMEM, "out of host memory while decoding the masks");
    }
    if (bad.load() >= 0) {
        if (!beside_setup) e->hb[e->hb_cur].used -= total;  // nothing of this call stays
        return fail(SC_ERR_INVALID, "file %d could not be decoded", bad.load());
    }
    if (beside_setup) {
        rc = use_device(e);  // joins the set-up
        if (rc) return rc;
        rc = materialize_deferred(e);
        if (rc) return rc;
        char *pinned = nullptr;
        rc = hostbits_reserve(e, total, &pinned);
        if (rc) return rc;
        memcpy(pinned, heap.data(), total);
        base = pinned;
    }
    const uint64_t base_off = (uint64_t)(base - e->hb[e->hb_cur].pin);
    for (int q = 0; q < V; ++q) {
        rc = hostbits_push_view(e, K + 4 * q, R + 9 * q, t + 3 * q, base_off + offs[(size_t)q], Hs[(size_t)q], Ws[(size_t)q]);
        if (rc) return rc;
    }
    return after_enqueue(e);
}

int sc_process_views_device(sc_engine *e, int V, const float *K, const float *R, const float *t,
                            const void *masks_dev, int H, int W, int mask_dtype) {
    if (V == 0) return e ? SC_OK : fail(SC_ERR_INVALID, "null engine");
    int rc = check_view_args(e, K, R, t, masks_dev, H, W);
    if (rc) return rc;
    if (V < 0) return fail(SC_ERR_INVALID, "negative view count");
    rc = check_dtype(e, mask_dtype);
    if (rc) return rc;
    rc = use_device(e);
    if (rc) return rc;
    rc = materialize_deferred(e);
    if (rc) return rc;
    size_t es = elem_size(mask_dtype);
    int64_t row = (int64_t)W * (int64_t)es, view = row * H;
    if (e->mode == SC_MODE_CARVE) {
        if (e->pack_ride && e->views_per_launch == 0 && e->pending.empty() && V >= kMinFusedViews &&
            V <= kPackOrderMax && V < e->max_pending && pack16_eligible(masks_dev, W, mask_dtype, row, view)) {
            // the whole batch will be one fused launch: its packing waits for the flush, which knows
            // the order of the views (see flush)
            e->deferred.on = true;
            e->deferred.raw = masks_dev;
            e->deferred.V = V; e->deferred.H = H; e->deferred.W = W; e->deferred.dtype = mask_dtype;
            e->deferred.row_stride = row; e->deferred.view_stride = view;
            for (int q = 0; q < V; ++q) {
                ViewDesc d;
                fill_desc(e, d, K + 4 * q, R + 9 * q, t + 3 * q, nullptr, H, W, nullptr);
                e->pending.push_back(d);
            }
            return SC_OK;
        }
        // one pack launch for the whole batch, then carve launches per views_per_launch
        rc = enqueue_pack(e, V, K, R, t, masks_dev, H, W, mask_dtype, row, view);
        if (rc) return rc;
        return after_enqueue(e);
    }
    if (mask_dtype == SC_MASK_U8_LUT) {
        rc = enqueue_tile8(e, V, K, R, t, masks_dev, H, W, row, view);
        if (rc) return rc;
        return after_enqueue(e);
    }
    if (e->avg_tile_f32) {
        rc = enqueue_tilef32(e, V, K, R, t, masks_dev, H, W, row, view);
        if (rc) return rc;
        return after_enqueue(e);
    }
    for (int q = 0; q < V; ++q) {
        ViewDesc d;
        fill_desc(e, d, K + 4 * q, R + 9 * q, t + 3 * q,
                  static_cast<const char *>(masks_dev) + (int64_t)q * view, H, W);
        e->pending.push_back(d);
        rc = after_enqueue(e);
        if (rc) return rc;
    }
    return SC_OK;
}

// The first nv pending descriptors of an engine into its device ring, by a copy on its stream.
static int stage_descriptors(sc_engine *e, size_t nv, const ViewDesc **out) {
    if (nv > e->views_cap || e->views_head + nv > e->views_cap) {
        HIP_TRY(schost::wait_stream(e->stream));
        e->views_head = 0;
    }
    if (nv > e->views_cap) {
        if (e->views_dev) (void)hipFree(e->views_dev);
        if (e->views_pin) (void)hipHostFree(e->views_pin);
        e->views_dev = e->views_pin = nullptr;
        e->views_cap = 0;
        size_t cap = std::max<size_t>(nv * 4, 1024);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->views_dev), cap * sizeof(ViewDesc)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->views_pin), cap * sizeof(ViewDesc), hipHostMallocDefault));
        e->views_cap = cap;
    }
    ViewDesc *pin = e->views_pin + e->views_head, *dev = e->views_dev + e->views_head;
    memcpy(pin, e->pending.data(), nv * sizeof(ViewDesc));
    e->views_head += nv;
    HIP_TRY(hipMemcpyAsync(dev, pin, nv * sizeof(ViewDesc), hipMemcpyHostToDevice, e->stream));
    *out = dev;
    return SC_OK;
}

int sc_average_labels(sc_engine *const *engines, int L, int V, const float *K, const float *R, const float *t,
                      const void *const *masks_dev, int H, int W) {
    if (!engines || !masks_dev || L < 1) return fail(SC_ERR_INVALID, "bad label set");
    for (int l = 0; l < L; ++l) {
        int rc = check_view_args(engines[l], K, R, t, masks_dev[l], H, W);
        if (rc) return rc;
        rc = check_dtype(engines[l], SC_MASK_U8_LUT);
        if (rc) return rc;
    }
    if (V < 0) return fail(SC_ERR_INVALID, "negative view count");
    if (V == 0) return SC_OK;
    sc_engine *e0 = engines[0];
    const int64_t row = W, view = (int64_t)W * H;
    // one launch needs: labels in groups of 2 .. 4 on one device, one grid, the same freshness, nothing pending,
    // the brick form's conditions (flush), whole 16-pixel rows; anything else goes label by label
    // (up to 4 labels: measured on a 6-label segmentation -- groups of 4 + 2, 3 + 3 or 2 + 2 + 2 -- the shared
    // launches took 16.5-17.4 ms where six launches of their own take 14.3: the labels' footprints are mixed in
    // different places, so the union of the (brick, view) pairs to project is nearly their sum, and every label
    // is dragged through every pair.  3 labels: 3.6 ms against 4.5.)
    bool fused = L >= 2 && L <= kMaxLabels && V > 1 && e0->avg_brick && (uint64_t)e0->npitch < 0x80000000ull &&
                 (W % 16) == 0 && V <= 4096;
    const uint32_t abys = (uint32_t)((e0->ny + kBrickY - 1) / kBrickY), abzs = (uint32_t)((e0->nz + kBrickZ - 1) / kBrickZ);
    fused = fused && (uint64_t)e0->planes * abys * abzs < 0x80000000ull;
    for (int l = 0; l < L && fused; ++l) {
        const sc_engine *e = engines[l];
        fused = e->device == e0->device && e->nx == e0->nx && e->ny == e0->ny && e->nz == e0->nz && e->i0 == e0->i0 &&
                e->istride == e0->istride && e->planes == e0->planes && e->vs == e0->vs &&
                memcmp(e->origin, e0->origin, sizeof e->origin) == 0 && e->fresh == e0->fresh && e->pending.empty() &&
                !e->deferred.on && e->avg_brick && (reinterpret_cast<uintptr_t>(masks_dev[l]) % 16) == 0;
        for (int m = 0; m < l && fused; ++m) fused = engines[m] != e;
    }
    if (!fused) {
        for (int l = 0; l < L; ++l) {
            int rc = sc_process_views_device(engines[l], V, K, R, t, masks_dev[l], H, W, SC_MASK_U8_LUT);
            if (rc) return rc;
            rc = sc_flush(engines[l]);
            if (rc) return rc;
        }
        return SC_OK;
    }
    int rc = use_device(e0);
    if (rc) return rc;
    // everything of this call runs on the first engine's stream, behind what the others have on theirs; their
    // streams take up again behind it
    hipStream_t main = e0->stream;
    std::vector<hipStream_t> own((size_t)L);
    for (int l = 0; l < L; ++l) own[(size_t)l] = engines[l]->stream;
    // the ordering first, for every engine; the `stream` fields are switched only once all of it has succeeded, and
    // whatever way this function is left they go back to the engines' own (an engine must never keep another's)
    for (int l = 1; l < L; ++l) {
        if (own[(size_t)l] == main) continue;
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t he = hipEventRecord(ev, own[(size_t)l]);
        if (he == hipSuccess) he = hipStreamWaitEvent(main, ev, 0);
        (void)hipEventDestroy(ev);
        if (he != hipSuccess) return fail(SC_ERR_DEVICE, "stream ordering failed: %s", hipGetErrorString(he));
    }
    struct StreamGuard {
        sc_engine *const *eng;
        const std::vector<hipStream_t> &own;
        int n;
        ~StreamGuard() { for (int l = 0; l < n; ++l) eng[l]->stream = own[(size_t)l]; }
    } guard{engines, own, L};
    for (int l = 0; l < L; ++l) engines[l]->stream = main;
    const GridDesc g = grid_desc(e0);
    const uint32_t anb = (uint32_t)((uint64_t)e0->planes * abys * abzs);
    const size_t need = (size_t)anb * (size_t)V;
    const ViewDesc *vd[64];
    rc = SC_OK;
    for (int l = 0; l < L && rc == SC_OK; ++l) {
        sc_engine *e = engines[l];
        rc = enqueue_tile8(e, V, K, R, t, masks_dev[l], H, W, row, view);
        if (rc) break;
        if (e->pending[0].occ == nullptr) { rc = fail(SC_ERR_STATE, "no uniformity flags"); break; }
        rc = stage_descriptors(e, (size_t)V, &vd[l < 64 ? l : 0]);
        if (rc) break;
        if (need > e->verd_cap) {
            hipError_t he = schost::wait_stream(main);
            if (e->verd) (void)hipFree(e->verd);
            e->verd = nullptr;
            e->verd_cap = 0;
            if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&e->verd), need);
            if (he != hipSuccess) { rc = fail(SC_ERR_NOMEM, "verdict buffer: %s", hipGetErrorString(he)); break; }
            e->verd_cap = need;
        }
    }
    if (rc == SC_OK && L > 64) rc = fail(SC_ERR_INVALID, "more than 64 labels");
    for (int l0 = 0; l0 < L && rc == SC_OK; l0 += kMaxLabels) {
        const int n = std::min(kMaxLabels, L - l0);
        if (n == 1) {  // a label left over: its own launches
            sc_engine *e = engines[l0];
            hipLaunchKernelGGL(avg_flags_kernel, dim3((anb + kBlock - 1) / kBlock, (uint32_t)V), dim3(kBlock), 0, main, g, vd[l0],
                               V, abys, abzs, anb, e->verd, static_cast<uint32_t *>(nullptr));
            if (e->fresh)
                hipLaunchKernelGGL(average_brick_kernel<true>, dim3(anb), dim3(kBlock), 0, main, static_cast<float *>(e->state), g,
                                   vd[l0], V, e->default_value, e->lut_dev, abys, abzs, e->verd, static_cast<uint32_t *>(nullptr));
            else
                hipLaunchKernelGGL(average_brick_kernel<false>, dim3(anb), dim3(kBlock), 0, main, static_cast<float *>(e->state), g,
                                   vd[l0], V, e->default_value, e->lut_dev, abys, abzs, e->verd, static_cast<uint32_t *>(nullptr));
            continue;
        }
        MultiArgs a;
        memset(&a, 0, sizeof a);
        for (int q = 0; q < n; ++q) {
            sc_engine *e = engines[l0 + q];
            a.values[q] = static_cast<float *>(e->state);
            a.views[q] = vd[l0 + q];
            a.verd[q] = e->verd;
            a.lut[q] = e->lut_dev;
            a.init[q] = e->default_value;
        }
#define LAUNCH_MULTI(N)                                                                                              \
    do {                                                                                                             \
        hipLaunchKernelGGL((avg_flags_multi_kernel<N>), dim3((anb + kBlock - 1) / kBlock, (uint32_t)V), dim3(kBlock), 0, main, a, g, \
                           V, abys, abzs, anb, static_cast<uint8_t *const *>(nullptr));                              \
        if (e0->fresh) hipLaunchKernelGGL((average_multi_kernel<N, true>), dim3(anb), dim3(kBlock), 0, main, a, g, V, abys, abzs); \
        else hipLaunchKernelGGL((average_multi_kernel<N, false>), dim3(anb), dim3(kBlock), 0, main, a, g, V, abys, abzs);          \
    } while (0)
        if (n == 2) LAUNCH_MULTI(2);
#if SC_MAXLABELS >= 3
        else if (n == 3) LAUNCH_MULTI(3);
#endif
#if SC_MAXLABELS >= 4
        else LAUNCH_MULTI(4);
#endif
#undef LAUNCH_MULTI
    }
    if (rc == SC_OK && hipGetLastError() != hipSuccess) rc = fail(SC_ERR_DEVICE, "multi-label launch failed");
    // the other engines' own streams wait for the first one's
    if (rc == SC_OK) {
        hipEvent_t ev;
        hipError_t he = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventRecord(ev, main);
        for (int l = 1; l < L && he == hipSuccess; ++l)
            if (own[(size_t)l] != main) he = hipStreamWaitEvent(own[(size_t)l], ev, 0);
        if (he == hipSuccess) (void)hipEventDestroy(ev);
        if (he != hipSuccess) rc = fail(SC_ERR_DEVICE, "stream ordering failed: %s", hipGetErrorString(he));
    }
    for (int l = 0; l < L; ++l) {
        sc_engine *e = engines[l];
        e->pending.clear();
        if (rc == SC_OK) e->fresh = false;
        arena_reset(e);  // (on the first engine's stream, which every later use of this engine's arena is behind)
    }
    if (rc == SC_OK) g_avg_labels_fused.fetch_add(1, std::memory_order_relaxed);
    return rc;  // (the guard hands the engines their own streams back)
}

int64_t sc_average_labels_fused_count(void) { return g_avg_labels_fused.load(std::memory_order_relaxed); }

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
    if (!e->narrow) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->narrow), (size_t)e->n));
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
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->packed_labels), cap));
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
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->wire_stage), (size_t)words * 4, hipHostMallocDefault));
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
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&out), sizeof host));
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
    hipError_t he = hipMalloc(reinterpret_cast<void **>(&dp), (size_t)nposes * sizeof(PoseRec));
    if (he == hipSuccess) he = hipMemcpy(dp, poses, (size_t)nposes * sizeof(PoseRec), hipMemcpyHostToDevice);
    if (he == hipSuccess && ijk) {
        he = hipMalloc(reinterpret_cast<void **>(&dijk), (size_t)count * 12);
        if (he == hipSuccess) he = hipMemcpy(dijk, ijk, (size_t)count * 12, hipMemcpyHostToDevice);
    }
    if (he == hipSuccess && pose_idx) {
        he = hipMalloc(reinterpret_cast<void **>(&didx), (size_t)count * 4);
        if (he == hipSuccess) he = hipMemcpy(didx, pose_idx, (size_t)count * 4, hipMemcpyHostToDevice);
    }
    if (he == hipSuccess && words_out) he = hipMalloc(reinterpret_cast<void **>(&dw), (size_t)count * 4);
    if (he == hipSuccess && digests_out) {
        he = hipMalloc(reinterpret_cast<void **>(&dd), ndig * 8);
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
    HIP_TRY(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return SC_OK;
}

void sc_host_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

int sc_dev_alloc(sc_engine *e, int64_t bytes, void **ptr) {
    if (!e || !ptr || bytes <= 0) return fail(SC_ERR_INVALID, "bad argument");
    int rc = use_device(e);
    if (rc) return rc;
    HIP_TRY(hipMalloc(ptr, (size_t)bytes));
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

// ---- several devices from one process (sc_create_sharded) ----------------------------
// One engine per device, the x-planes dealt round-robin (or in contiguous slabs) exactly as the
// one-process-per-GPU path deals them to ranks; every view goes to every engine; the read-back lands
// each engine's planes at their global x positions with one strided copy per device.

}  // extern "C"

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

#ifdef SC_TRACE_DENSE  // diagnostic builds only (tools/probes/dense_trace.py)
extern "C" int sc_debug_dense_trace(uint32_t *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dense_trace), sizeof(uint32_t) * 8192 * 8);
}
#endif

#include "sc_api_sparse.inl"
#include "sc_comm.inl"
