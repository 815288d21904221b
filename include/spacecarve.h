/*
 * spacecarve.h -- C ABI of the MI355X (gfx950) voxel back-projection engine.
 *
 * This is the drop-in boundary for the reference's device layer: everything
 * plant3dvision/cl.py::Backprojection does with PyOpenCL (cl.py:118-311) maps onto
 * the entry points below, and nothing else of plant-3d-vision is replaced.  Plain
 * pointers and sizes only; no torch / HIP types in any signature.  The host-side
 * mirror (plant-3d-vision_amd/cl.py) binds these with cffi (ABI mode) or ctypes;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions (all from the reference):
 *   - grid state is C-order [nx][ny][nz], z fastest   (kernels/common.h:1-12)
 *   - carve state int32 in {-1 carved, 0 unseen, 1 kept} (kernels/backprojection.c:67-83)
 *   - average state float32 running sum in view order    (kernels/backprojection.c:54)
 *   - K = [fx, fy, cx, cy], R row-major 3x3, t: float32   (cl.py:293-296)
 *   - masks are row-major [H][W]; u = column, v = row     (cl.py:217)
 *   - arithmetic: IEEE binary32, no contraction, correctly rounded divide,
 *     (int) cast rejecting NaN/inf/out-of-range
 *
 * Threading: an engine is driven by one host thread at a time (calls on it may come from different
 * threads one after the other -- every entry selects the engine's device first); different engines
 * may be driven from different threads at once.  Every call
 * returns SC_OK (0) or a negative code; sc_last_error() gives the thread-local
 * message of the most recent failure.
 */
#ifndef SPACECARVE_H
#define SPACECARVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 1

/* error codes */
#define SC_OK 0
#define SC_ERR_INVALID (-1)   /* bad argument                        */
#define SC_ERR_DEVICE (-2)    /* HIP runtime error / no gfx950 device */
#define SC_ERR_NOMEM (-3)     /* host or device allocation failed     */
#define SC_ERR_STATE (-4)     /* call not valid in the engine's state */

/* sc_create mode -- replaces the kernel choice at cl.py:145-152 */
#define SC_MODE_CARVE 0   /* `carve`,   int32 state   (backprojection.c:57-84) */
#define SC_MODE_AVERAGE 1 /* `average`, float32 state (backprojection.c:36-55) */

/* mask element types accepted by sc_process_view* */
#define SC_MASK_U8 0  /* carve: foreground = (value != 0), the test at backprojection.c:79 */
#define SC_MASK_I32 1 /* carve: what cl.py:215 casts to                                    */
#define SC_MASK_F32 2 /* average: the value added at backprojection.c:54                   */
#define SC_MASK_U8_INV 3   /* carve: uint8 mask to be inverted first, i.e. `np.invert(mask)` of
                              cl.py:300-301 done on the device: foreground = (value != 255)   */
#define SC_MASK_BOOL_INV 4 /* carve: bool mask (bytes 0/1) to be inverted first (logical not):
                              foreground = (value == 0)                                       */
#define SC_MASK_U8_LUT 5   /* average: the ORIGINAL uint8 mask; the value added is table[byte],
                              the table (sc_set_lut) being what the host conversion of
                              cl.py:205-208 (img_as_float32, log) makes of each byte value      */

/* sc_set_option keys */
#define SC_OPT_VIEWS_PER_LAUNCH 1 /* 0 (default): defer views, fuse all pending views into one
                                     launch at flush; n>0: launch every n views (1 = one launch
                                     per view, the reference's schedule cl.py:223-226)        */
#define SC_OPT_VIEW_ORDER 2       /* carve only. 0: as given; 1 (default): inside a fused launch,
                                     most-perpendicular-first order (legal: the carve state is
                                     order-independent). average always keeps
                                     the given order (float sum).                             */
#define SC_OPT_TIME_KERNELS 3     /* 1: bracket every kernel launch with HIP events on the
                                     engine's stream; read with sc_kernel_stats.  2: only the
                                     carve / average kernel of a per-view launch and one pair
                                     around each fused batch, SC_KERNEL_STEP (an event pair costs
                                     a few microseconds of stream time)                       */
#define SC_OPT_MAX_PENDING 4      /* deferred views that force a flush (default 256)          */
#define SC_OPT_RESERVE_EVENTS 31  /* with SC_OPT_TIME_KERNELS: create n HIP events now (0..65536) so that the timed
                                     launches that follow find them in the engine's pool instead of creating them */
/* The fused carve's tuning knobs (SC_OPT_COMPACT .. SC_OPT_UNIT_CULL: where work moves between its kernels; results
 * never depend on them) are sc_set_option keys too, declared in spacecarve_tuning.h (included at the end of this
 * header) -- not part of what a caller of the reference's interface needs. */

/* kernel ids for sc_kernel_stats */
#define SC_KERNEL_CARVE 0
#define SC_KERNEL_AVERAGE 1
#define SC_KERNEL_PACK 2
#define SC_KERNEL_FILL 3
#define SC_KERNEL_LIST 4 /* confirm + special kernel + survivor-list stages of a fused carve */
#define SC_KERNEL_FLAGS 5 /* brick emptiness verdicts ahead of the dense stage (brick form) */
#define SC_KERNEL_STEP 6  /* a whole fused batch: mask packing + every kernel of its launch     */

typedef struct sc_engine sc_engine;

int sc_abi_version(void);
const char *sc_last_error(void);
/* number of visible HIP devices whose architecture is gfx950 (others are skipped); sc_create*
 * take a HIP ordinal and check that device alone */
int sc_device_count(int *count);

/*
 * Create an engine owning the whole grid.  Replaces Backprojection.__init__ +
 * init_buffers (cl.py:118-188): state = default_value everywhere (cl.py:173-175),
 * volinfo = {origin, voxel_size} as float32 (cl.py:181-183), shape (cl.py:185-187).
 * device: HIP device ordinal.
 */
int sc_create(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, const float origin[3],
              float voxel_size, int mode, float default_value, int device);

/*
 * Same, but the engine owns only the X-slab i in [i0, i1) of the global grid
 * (multi-GPU sharding).  Voxel coordinates are still computed from
 * the GLOBAL index (origin_x + (float)i * voxel_size), never from a shifted origin.
 * State / sc_get_values cover (i1-i0)*ny*nz elements.
 */
int sc_create_slab(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t i0, int64_t i1,
                   const float origin[3], float voxel_size, int mode, float default_value,
                   int device);

/*
 * Same, but the engine owns the x-planes  first, first + stride, first + 2*stride, ...  of the
 * global grid (plane-cyclic sharding: rank r of W takes first = r, stride = W).  An object in
 * the middle of the grid then loads every rank alike, which contiguous slabs do not.  State /
 * sc_get_values are [planes][ny][nz] in that order.
 */
int sc_create_cyclic(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t first, int64_t stride,
                     const float origin[3], float voxel_size, int mode, float default_value,
                     int device);

/*
 * The three above in one call, with flags: the engine owns `planes` x-planes first, first + stride, ... of the grid
 * (sc_create: 0, 1, nx; sc_create_slab: i0, 1, i1 - i0; sc_create_cyclic: first, stride, ceil((nx - first) / stride)).
 * SC_CREATE_DEFERRED (round 6): the arguments are judged now, the DEVICE half of the set-up -- runtime initialisation,
 * the process's first stream, the state's allocation: 130-240 ms in a fresh process, where cl.py:29-30 pays its context
 * and queue at import -- runs on a thread of the library's; the call returns at once, every later call that needs the
 * device joins that thread first and reports ITS failure (no device, not a gfx950, out of memory) as its own, and
 * sc_process_png_views decodes its files beside it: a cold `Voxels.run` (tasks/cl.py:100, 162-165) hides the decode
 * and whatever else the host does between the two calls behind the set-up.
 */
/* What cl.py:29-30 does at import (the context and the queue), without blocking the importer: a thread of the library's
 * initialises the runtime on `device` and creates its first stream; the first engine there takes that stream.  Returns
 * at once; a failure is the first engine's to report. */
int sc_prewarm(int device);
void sc_prewarm_wait(void); /* returns when no such thread is running any more: call it before the process exits */
#define SC_CREATE_DEFERRED 1
int sc_create_ex(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t first, int64_t stride, int64_t planes,
                 const float origin[3], float voxel_size, int mode, float default_value, int device, int flags);

/* Diagnostic of a (deferred) engine's set-up: out[0] = milliseconds its device half took, out[1] = milliseconds the first
 * call that needed the device waited for it. */
int sc_setup_times(sc_engine *e, double out[2]);

void sc_destroy(sc_engine *e);

/* Backprojection.clear (cl.py:307-311): reset state to default_value, drop pending views. */
int sc_clear(sc_engine *e);

int sc_set_option(sc_engine *e, int key, int64_t value);

/* Averaging engines: the 256-entry float32 table used by SC_MASK_U8_LUT views enqueued
 * afterwards (views already enqueued are flushed with the previous table). */
int sc_set_lut(sc_engine *e, const float *lut256);

/* Run the engine's work on an existing hipStream_t (a non-default stream of the caller);
 * NULL restores the engine's own stream.  The engine's own stream is non-blocking: it does not
 * synchronise with the legacy default stream (handle 0, which is what torch's default stream
 * is), so a caller whose masks are produced there uses sc_order_after instead. */
int sc_set_stream(sc_engine *e, void *hip_stream);

/* Order everything the engine enqueues from now on AFTER the work enqueued so far on
 * `producer_stream` (a hipStream_t; NULL = the legacy default stream): an event recorded there
 * and waited for on the engine's stream.  For device-resident masks another stream is still
 * writing (sc_process_views_device); replaces nothing in the reference, whose single in-order
 * queue (cl.py:29-30) has no such hazard. */
int sc_order_after(sc_engine *e, void *producer_stream);

/* The other direction: everything enqueued from now on on `consumer_stream` (NULL = the legacy default stream)
 * runs AFTER the work the engine has enqueued so far -- for a consumer on another stream of what the engine
 * leaves in device memory (sc_values_device_ptr, sc_values_packed), and for a producer who wants to reuse a
 * buffer handed to sc_process_views_device, without waiting on the host. */
int sc_order_before(sc_engine *e, void *consumer_stream);

/*
 * Backprojection.process_view (cl.py:190-227): one view from a HOST mask.
 * The mask is consumed before the call returns (the caller may free it); the kernel
 * launch itself may be deferred (SC_OPT_VIEWS_PER_LAUNCH).  row_stride_bytes = 0 means
 * tightly packed rows.  mask_dtype must fit the mode (U8/I32 carve, F32 average).
 * Carve masks (cl.py:215 + backprojection.c:79: a pixel counts when it is != 0) are reduced to 1 bit per pixel on
 * the library's host threads during the call and cross PCIe as bits, one copy per flush (sc_hostpack_bits is that
 * bit form by itself; SC_OPT_HOST_PACK 0: the bytes travel and are packed on the device).
 * Picture sizes: 1 <= H <= 2^24 - 32, 1 <= W <= 2^24, H W <= 2^34; else SC_ERR_INVALID ("bad mask shape").
 * Averaging masks are re-laid in strips of 128-byte tiles on the device, whose addressing is narrower:
 * uint8 masks with a table (SC_MASK_U8_LUT: tiles of 16 x 8 pixels) need ceil(H / 8) * 128 < 2^24 (H < 1 048 576) and
 * ceil(W / 16) * ceil(H / 8) * 128 < 2^31 bytes per view; float32 masks (SC_MASK_F32: tiles of 8 x 4 pixels) need
 * ceil(H / 4) * 32 < 2^24 (H < 2 097 152) and ceil(W / 8) * ceil(H / 4) * 128 < 2^33 bytes per view; beyond that
 * SC_ERR_INVALID.  sc_process_png_views takes files of at most 2^24 x 2^24 pixels and H W <= 2^31.
 */
int sc_process_view(sc_engine *e, const float K[4], const float R[9], const float t[3],
                    const void *mask, int H, int W, int mask_dtype, int64_t row_stride_bytes);

/* The file loop of Backprojection.process_label in one call (cl.py:282-303, for a carve engine and masks stored as
 * 8-bit greyscale PNG -- what `io.write_image(f, im, 'png')` makes of tasks/proc2d.py's uint8 masks and what
 * plantdb.io.read_image decodes for cl.py:298): V ENCODED files (png[q], sizes[q] bytes) with their poses, decoded on
 * `threads` host threads of this call's own (<= 0: 16), each mask reduced to bits as it comes out of the decoder
 * (invert != 0: np.invert on the uint8 pixels first, cl.py:300-301), the views enqueued in the order given.
 * SC_ERR_INVALID, and nothing enqueued, if a file is not a PNG this decoder takes (sc_png_info) or is damaged. */
int sc_process_png_views(sc_engine *e, int V, const float *K, const float *R, const float *t, const void *const *png,
                         const int64_t *sizes, int invert, int threads);

/* V views at once: K[V*4], R[V*9], t[V*3], masks[V] host pointers, common H, W, dtype. */
int sc_process_views(sc_engine *e, int V, const float *K, const float *R, const float *t,
                     const void *const *masks, int H, int W, int mask_dtype,
                     int64_t row_stride_bytes);

/*
 * V views whose masks are already resident in device memory on the engine's device,
 * contiguous [V][H][W] (the Masks2D / bench path: no host round trip).  The masks are read on the
 * engine's stream when the batch is launched (a fused batch packs them at the flush, part of them beside its
 * dense stage), and sc_flush only enqueues: the buffer must stay valid AND UNCHANGED until sc_synchronize or
 * sc_get_values* returns -- or until the caller has ordered its own stream behind the engine's work.
 */
int sc_process_views_device(sc_engine *e, int V, const float *K, const float *R, const float *t,
                            const void *masks_dev, int H, int W, int mask_dtype);

/*
 * The labels of one scan at once: L averaging engines of one grid on one device (one per label, each with its
 * table, sc_set_lut), V views with ONE set of poses, masks_dev[l] = label l's uint8 masks [V][H][W] in device
 * memory (SC_MASK_U8_LUT).  The reference runs the view loop once per label over the same cameras
 * (Backprojection.process_fileset, cl.py:248-255); here a voxel is projected once per view and every label's
 * mask is read at that pixel -- each label's sum is the same additions in the same order as its own
 * sc_process_views_device + sc_flush would make, bit for bit.  Launches at once (asynchronous); every engine's
 * stream is ordered behind the work.  Engines or masks that do not fit the one-launch form (different grids or
 * freshness, views pending, rows that are not whole 16-byte groups) are processed label by label instead.
 */
int sc_average_labels(sc_engine *const *engines, int L, int V, const float *K, const float *R, const float *t,
                      const void *const *masks_dev, int H, int W);
/* How many sc_average_labels calls of this process took the shared-launch form (the others went label by label:
 * more than 4 labels left over, masks or widths that are not 16-byte multiples, engines that differ): tests assert
 * the form they mean to exercise. */
int64_t sc_average_labels_fused_count(void);

/* Launch everything still deferred (asynchronous on the engine's stream). */
int sc_flush(sc_engine *e);
/* sc_flush + wait for the stream: the role of queue.finish() (cl.py:226). */
int sc_synchronize(sc_engine *e);

/*
 * Backprojection.get_values (cl.py:229-232): flush, wait, copy the state to `out`
 * (int32 or float32, C-order, (i1-i0)*ny*nz elements).
 */
int sc_get_values(sc_engine *e, void *out);

/*
 * The same read-back for carve labels as int8 (labels are -1 / 0 / 1, or default_value where no view
 * reached; SC_ERR_STATE if default_value does not fit): a quarter of the bytes cross PCIe; the host
 * mirror widens them back to the int32 array cl.py:229-232 returns.  (i1-i0)*ny*nz bytes.
 */
int sc_get_values_i8(sc_engine *e, int8_t *out);

/* Flush and return a device pointer to the engine's values as planes * ny * nz contiguous elements; work may
 * still be running on the engine's stream.  When nz is a multiple of 64 this IS the state (valid until
 * sc_destroy, and it changes as views are applied, like the reference's buffer); otherwise the state's rows are
 * padded and the pointer is a SNAPSHOT without the padding, made on the engine's stream by this call: valid
 * until the next call that changes the state or takes another snapshot. */
int sc_values_device_ptr(sc_engine *e, void **ptr);

/*
 * Carve labels packed for the wire (multi-GPU assembly; no reference counterpart -- the reference
 * is single-device).  bits = 2: the three states, label & 3 (-1 -> 3, 0 -> 0, 1 -> 1; SC_ERR_STATE unless
 * default_value is one of them); bits = 1: the occupancy the consumer binarises to (label == 1, proc3d.py:515).
 * Voxel v of the engine's voxels (its planes in its own order, no padding) is at bit  bits * (v % (32 / bits))
 * of 32-bit word  v / (32 / bits).  sc_values_packed flushes, packs on the engine's stream into an engine-owned
 * device buffer (valid until the next sc_values_packed / sc_destroy; work may still be running) and returns it
 * with its size, sc_packed_bytes(voxels, bits): whole 16-byte groups.  sc_get_values_packed copies the words to
 * the host.  sc_unpack_labels is the other end of an all-gather of such buffers: `world` ranks' buffers
 * rank_bytes apart (rank r's planes are r, r + world, ... of the grid: partition 0; or the slab
 * [nx r / world, nx (r + 1) / world): partition 1) into ONE [nx][ny][nz] grid in global order on `device`,
 * as int8 (out_bytes 1) or int32 (4), on hip_stream (NULL: the legacy default stream).
 */
int64_t sc_packed_bytes(int64_t voxels, int bits);
int sc_values_packed(sc_engine *e, int bits, void **ptr, int64_t *bytes);
int sc_get_values_packed(sc_engine *e, int bits, void *out);
/* cl.py:229-232 get_values of a carve volume, the fast way round: the labels cross PCIe at 2 bits each, in pieces,
 * and the library's host pool (SC_OPT_HOST_THREADS; `threads` is ignored since round 4) widens the pieces that
 * have arrived into out[voxels] (int32, the array the reference returns) while the next ones are on their way --
 * 512 MiB of int32 in 3-4 ms on 8 threads.  The packed labels land in a page-locked buffer the engine keeps
 * (32 MiB at 512^3, allocated at the first call); `staging` / `staging_bytes` are ignored since round 4 and may be
 * NULL / 0 (round 3: a caller's pageable buffer -- a copy into pageable memory runs at a quarter of the rate).
 * SC_ERR_STATE unless default_value is one of -1, 0, 1 (use sc_get_values / sc_get_values_i8 then). */
int sc_get_values_wire2(sc_engine *e, int32_t *out, void *staging, int64_t staging_bytes, int threads);
/* Host code only (no device, no engine): the widening sc_get_values_wire2 does, by itself -- `voxels` labels at
 * 2 bits each in packed[(voxels + 15) / 16] (the layout above) into out[voxels] on the host pool (`threads` ignored). */
int sc_widen_labels2(const uint32_t *packed, int64_t voxels, int32_t *out, int threads);
/* Host code only: `world` ranks' packed planes (2 bits per label, rank-major, `rank_bytes` apart: what a gather of
 * the ranks' sc_values_packed buffers gives) into ONE int32 grid out[nx][ny][nz] in global plane order -- the host
 * end of ShardedBackprojection.gather_to_host (cl.py:229-232 get_values of a sharded run).  partition: 0 plane-cyclic
 * (plane i is plane i / world of rank i % world), 1 slabs.  On the library's host pool. */
int sc_widen_labels2_ranks(const uint32_t *packed, int64_t rank_bytes, int world, int partition, int64_t nx, int64_t ny,
                           int64_t nz, int32_t *out);
/* Host code only: the bit form in which sc_process_view sends a carve mask over PCIe (cl.py:215 + backprojection.c:79:
 * a pixel counts when it is != 0; SC_MASK_U8_INV / SC_MASK_BOOL_INV: after np.invert, cl.py:300-301) -- row-major,
 * out[H][(W + 31) / 32] words, pixel u of a row at bit u & 31 of word u >> 5, bits beyond W zero.  mask_dtype: the
 * carve dtypes (SC_MASK_U8, SC_MASK_I32, SC_MASK_U8_INV, SC_MASK_BOOL_INV); row_stride_bytes 0 = tight rows. */
int sc_hostpack_bits(const void *mask, int H, int W, int mask_dtype, int64_t row_stride_bytes, uint32_t *out);
int sc_unpack_labels(int device, void *hip_stream, const void *recv_dev, int64_t rank_bytes, int world, int partition,
                     int64_t nx, int64_t ny, int64_t nz, int bits, void *out_dev, int out_bytes);

/*
 * The BRICK-SPARSE transport form of carve labels (round 6; multi-GPU assembly; no reference counterpart:
 * the reference is single-device, cl.py:29-30 -- what the far end needs is what cl.py:229-232 returns, the labels).
 * A carved volume is uniform almost everywhere, and the engine already holds a 1-byte verdict per 16 x 64-voxel brick
 * of one x-plane (brick b = (plane * bricks_y + j / 16) * bricks_z + k / 64).  One rank's buffer:
 *     header   64 bytes: uint32 magic "SCSP", version 1, bits 2, nbricks, cap, nmixed, planes, ny, nz, bricks_y,
 *              bricks_z, first, stride (its planes are first, first + stride, ... of the grid), nread (bricks the sender had to read: diagnostic), 2 spare
 *     codes    [nbricks] bytes: 0 / 1 / 3 = every voxel of the brick is 0 / 1 / -1 (label & 3); 2 = MIXED   (to 64 bytes)
 *     ids      [cap] uint32: the brick of payload slot s                                       (cap is a multiple of 16)
 *     payload  [cap][256] bytes: slot s = the brick's labels at 2 bits each, voxel (column jl, depth kl) of the brick
 *              at bits 2 (v % 16) of word v / 16, v = jl * 64 + kl; voxels beyond ny / nz are 0
 * sc_sparse_bricks / sc_sparse_rank_bytes give the sizes (cap is clamped to [16, nbricks] and rounded up to 16).
 * sc_values_sparse flushes and packs on the engine's stream into an engine-owned device buffer (two alternate: the
 * buffer of a call stays untouched until the call after next) and returns it (work may still be running).  It reads
 * only bricks nothing settles: behind ONE fused batch on a cleared volume the batch's verdict bytes settle most
 * bricks and its live list names the others (one launch, a plant's 7 577 live bricks of 131 072: 31 MB read, 2 MB
 * written, where the dense 2-bit form writes 32 MiB); otherwise bricks an earlier launch found empty are skipped and
 * the rest is read.  A brick that is read and turns out uniform gets its code, not a slot.  cap = 0: the engine's
 * last capacity (first call: an eighth of the bricks).  nmixed > cap: slots were refused -- the codes are complete,
 * the payload is not; ask again with cap >= nmixed.  SC_ERR_STATE unless default_value is -1, 0 or 1.
 * sc_sparse_headers returns nmixed[r], cap[r] of the `world` buffers rank_bytes apart in device memory (every rank of
 * an all-gather sees the same numbers, so all ranks take the same decision about a retry without talking); it waits
 * for done_event -- the event sc_all_gather_sparse hands out, recorded behind ITS collective -- and copies on a stream
 * of the library's own, or (done_event NULL) copies on hip_stream behind whatever that stream still has to do.  sc_unpack_sparse writes ONE [nx][ny][nz] grid in global order from them on hip_stream:
 * out_kind 4 int32 labels, 1 int8 labels, 0 the uint8 occupancy label == 1 (what proc3d.py:515 binarises to).
 * sc_widen_sparse_ranks is the same on the host (no device) into int32, on the library's host pool.
 */
int64_t sc_sparse_bricks(int64_t planes, int64_t ny, int64_t nz);
int64_t sc_sparse_rank_bytes(int64_t nbricks, int64_t cap);
int sc_values_sparse(sc_engine *e, int64_t cap, void **ptr, int64_t *bytes);
int sc_get_values_sparse(sc_engine *e, int64_t cap, void *out, int64_t out_bytes);
int sc_sparse_headers(int device, void *hip_stream, void *done_event, const void *recv_dev, int64_t rank_bytes, int world,
                      uint32_t *nmixed, uint32_t *cap);
int sc_unpack_sparse(int device, void *hip_stream, const void *recv_dev, int64_t rank_bytes, int world, int64_t nx, int64_t ny,
                     int64_t nz, void *out_dev, int out_kind);
int sc_widen_sparse_ranks(const void *packed, int64_t rank_bytes, int world, int64_t nx, int64_t ny, int64_t nz, int32_t *out);

/*
 * RCCL behind the C ABI (round 6): one communicator per process and GPU, so that a carve rank assembles the grid without
 * torch.  librccl is opened at the first call (the copy already in the process, if any); rank 0 makes the 128-byte id
 * (sc_comm_unique_id) and hands it to the others by whatever means the host has (plant-3d-vision_amd/sharded.py: a TCP
 * socket on MASTER_ADDR / MASTER_PORT); sc_comm_create is collective (ncclCommInitRank).  The communicator owns a
 * non-blocking stream (sc_comm_stream) for collectives that run beside an engine's work.
 * sc_all_gather_sparse / sc_all_gather_packed: flush, pack the engine's labels (sc_values_sparse with the capacity EVERY
 * rank names alike / sc_values_packed) and all-gather `rank_stride` bytes per rank -- the size of the rank with the most
 * planes, the same on every rank -- into recv_dev[nranks][rank_stride].  overlap = 0: the collective is enqueued on the
 * engine's own stream.  overlap = 1: on the communicator's stream behind an event, and the engine's stream does not
 * wait -- the next batch's carve runs beside the collective; the engine waits by itself before it packs into a send
 * buffer a collective may still be reading.  The caller keeps recv_dev untouched until it has waited
 * (sc_comm_synchronize, sc_sparse_headers on the stream the collective ran on).
 */
typedef struct sc_comm sc_comm;
int sc_comm_available(void); /* SC_OK when librccl could be opened (nothing else is done) */
int sc_comm_unique_id(void *id, int64_t id_bytes /* >= 128 */);
int sc_comm_create(sc_comm **out, const void *id, int nranks, int rank, int device);
void sc_comm_destroy(sc_comm *c);
int sc_comm_size(const sc_comm *c);
int sc_comm_rank(const sc_comm *c);
int sc_comm_stream(sc_comm *c, void **hip_stream);
int sc_comm_synchronize(sc_comm *c);
int sc_comm_barrier(sc_comm *c); /* a 16-byte all-gather, waited for: every rank has called it when it returns */
int sc_comm_all_gather(sc_comm *c, const void *send_dev, void *recv_dev, int64_t bytes_per_rank, void *hip_stream /* NULL: its own */);
int sc_engine_stream(sc_engine *e, void **hip_stream);
int sc_all_gather_sparse(sc_engine *e, sc_comm *c, int64_t cap, void *recv_dev, int64_t rank_stride, int overlap,
                         void **done_event /* may be NULL; the engine's, valid until its next sc_all_gather_sparse but one */,
                         const void **headers_host /* may be NULL; likewise */);
/* The headers of THAT gather without a copy at the time of asking: sc_all_gather_sparse has put a copy of the ranks'
 * headers to page-locked host memory on the collective's stream, in front of done_event; this waits for the event
 * (host code: a poll) and reads them.  nmixed[r], cap[r] as sc_sparse_headers. */
int sc_sparse_wait_headers(void *done_event, const void *headers_host, int64_t rank_bytes, int world, uint32_t *nmixed,
                           uint32_t *cap);
int sc_all_gather_packed(sc_engine *e, sc_comm *c, int bits, void *recv_dev, int64_t rank_stride, int overlap);

/* number of voxels this engine owns */
int64_t sc_num_voxels(const sc_engine *e);

/* With SC_OPT_TIME_KERNELS: launches and summed HIP-event milliseconds per kernel id
 * since the last sc_reset_kernel_stats (waits for the stream). */
int sc_kernel_stats(sc_engine *e, int kernel_id, int64_t *launches, double *total_ms);
int sc_reset_kernel_stats(sc_engine *e);

/* One HIP event pair on the engine's stream around whatever the caller enqueues in between (any number
 * of batches): sc_span_begin records the first event where the stream stands, sc_span_end records the
 * second, waits for it and returns the milliseconds between them.  Unlike SC_OPT_TIME_KERNELS this puts
 * nothing between the batches (an event pair per batch is a barrier packet each, 3-5 us of stream time). */
int sc_span_begin(sc_engine *e);
int sc_span_end(sc_engine *e, double *ms);

/* Diagnostics of the last fused carve launch (waits for the stream): out[0] bricks no view found
 * empty (brick form, else 0), out[1] voxels alive after the dense stage, out[2] after the first
 * survivor stage, out[3] 1 if a survivor list overflowed (the special kernel's dense pass took over). */
int sc_fused_counts(sc_engine *e, int64_t out[4]);
/* ... the same four, then out[4]: candidate bricks (kept as they are by the views packed ahead) that a later
 * view did not keep -- carved unit by unit by the special kernel; out[5]: units on the bulk list
 * (SC_OPT_BULK_MIN); out[6]: their work items (0 when the batch had fewer units than SC_OPT_BULK_FLOOR: their
 * voxels took the ordinary lists); out[7]: 0 (round 3: batches the host kept the bulk list off). */
int sc_fused_counts_ex(sc_engine *e, int64_t out[8]);

/* Self-test: runs the kernels' shared-reciprocal division and the compiler's IEEE division on
 * `count` pseudo-random operand triples (mode 0: raw bit patterns, 1: projection-like
 * magnitudes) and reports how many quotients differ bit-for-bit (must be 0) and how many
 * triples took the fast path.  Mode 2: numerators of at most 2^-40 in magnitude (zero and denormals included) over
 * denominators and intrinsics of a certified view -- there the PIXEL and the picture test are compared, not the
 * quotient (csrc/sc_project.h says why that is what matters); every sample counts as fast. */
int sc_selftest_division(sc_engine *e, int64_t count, uint32_t seed, int mode,
                         uint64_t *mismatches, uint64_t *fast_pairs);

/*
 * Self-test of the projection the voxel kernels share (backproject_point, backprojection.c:3-34,
 * with the coordinates of :71-73): `count` samples, each a pose record and a voxel index, give
 * one result word each:  v * W + u + 1  when the reference would touch mask[v][u], 0 when it
 * rejects the point.  `poses`: nposes records of 28 32-bit words
 *   float K[4], R[9], t[3], origin[3], voxel_size;  int32 W, H, nx, ny, nz, 0, 0, 0.
 * Explicit samples: ijk[count][3] voxel indices (+ pose_idx[count], else record 0).
 * Hashed samples (ijk == NULL): record and voxel are drawn from (seed, sample index) by an
 * integer hash the caller can reproduce (oracle/spacecarve_oracle.c restates it), the record per
 * 64 consecutive samples, the voxel inside that record's nx x ny x nz grid.
 * words_out[count] and/or digests_out[ceil(count / 65536)] (sum over a run of 65536 samples of
 * mix32(word ^ (uint32)index)) receive the results.  The caller compares them with the reference
 * arithmetic; no reference counterpart (the reference has no tests on this function).
 */
int sc_selftest_project(sc_engine *e, int64_t count, uint32_t seed, int nposes, const float *poses,
                        const int32_t *ijk, const int32_t *pose_idx, uint32_t *words_out,
                        uint64_t *digests_out);

/*
 * Diagnostic (host code only, no device call): would a view with this pose take the kernels' certified
 * path on a grid of nx x ny x nz voxels with this origin and voxel size?  The host checks, in double precision and with wide margins, that every voxel centre of the grid
 * projects with 2^-10 < p_z and |p_x|, |p_y|, p_z < 2^30 and that the intrinsics are finite and below
 * 2^30; for such a view the projection (backprojection.c:11-31) spends two comparisons on the range test
 * of its shared-reciprocal division and two on the picture test instead of nine.  Results are the same
 * either way (tests/test_project_selftest.py runs both); *certified = 1 or 0.  No reference counterpart.
 */
int sc_view_certified(const float origin[3], float voxel_size, int64_t nx, int64_t ny, int64_t nz, const float K[4],
                      const float R[9], const float t[3], int *certified);

/*
 * The carve's immediate consumer: plant3dvision/proc3d.py::vol2pcd (:490-570, called by
 * tasks/proc3d.py:134) on the GPU.  volume: host pointer, or device pointer on `device` when
 * on_device != 0 (e.g. sc_values_device_ptr: the volume then never crosses PCIe); dtype 0 int32,
 * 1 float32, 2 float64, 3 uint8; C-order [nx][ny][nz].  gauss_w: the 5 distinct weights of
 * scipy's radius-4 Gaussian kernel (centre first), computed by the host as scipy does.
 * Returns malloc'ed float64 arrays [count][3] (release with sc_free_host): points in world
 * coordinates and unit normals, in C-order of the shell voxels; voxels whose gradient is zero
 * carry NaNs (the reference drops them afterwards, proc3d.py:559-561).
 */
int sc_vol2pcd(const void *volume, int on_device, int dtype, int64_t nx, int64_t ny, int64_t nz,
               const double origin[3], double voxel_size, double level_set_value,
               const double gauss_w[5], int device, double **points_out, double **normals_out,
               int64_t *count);
/* The same from carve labels in their PACKED, rank-major form -- what an all-gather of the ranks' sc_values_packed
 * buffers leaves on a device (`world` ranks x `rank_bytes`, `bits` 2 or 1 per label, `partition` 0 plane-cyclic /
 * 1 slabs: the arguments of sc_unpack_labels).  Labels are -1 / 0 / 1, so the reference's `volume > 0.5`
 * (proc3d.py:515) is `label == 1` and is read off the packed words: the full-size grid (1 GiB of int8 at 1024^3)
 * is never written.  Runs on the device's default stream: the caller has waited for the collective. */
int sc_vol2pcd_packed(const void *recv_dev, int64_t rank_bytes, int world, int partition, int bits, int64_t nx,
                      int64_t ny, int64_t nz, const double origin[3], double voxel_size, double level_set_value,
                      const double gauss_w[5], int device, double **points_out, double **normals_out, int64_t *count);
const char *sc_vol2pcd_last_error(void);
/* sc_vol2pcd keeps its device work buffers (49 bytes per voxel, per device) between calls while they are at most
 * 1 GiB (larger ones are freed when the call ends); this gives back what is kept.  The caller's current HIP
 * device is left as it was. */
void sc_vol2pcd_release(void);
/* Largest device work buffers a sc_vol2pcd call may take, in bytes (default 8 GiB; 0 = no limit).  A volume that
 * needs more (49 bytes per voxel) goes through in x-slabs with a halo of the pipeline's reach on either side:
 * same points, same order. */
void sc_vol2pcd_set_scratch_limit(int64_t bytes);
void sc_free_host(void *p);

/*
 * SegmentedPointCloud's scoring loop (plant3dvision/tasks/proc3d.py:203-232 over
 * proc3d.py::backproject_points :655-659) on the GPU: P float64 points, V views (K[V][4] =
 * fx,fy,cx,cy, R[V][9], t[V][3], float64), L label images per view, masks uint8 [L][V][H][W]
 * (host, or device when masks_on_device != 0).  Writes scores_out [L][P] (float64 sums of mask
 * values at int(pixel + 0.5)) and labels_out [P] = arg-max over labels (first maximum).
 */
int sc_label_points(const double *points, int64_t P, int L, int V, const double *K, const double *R,
                    const double *t, const void *masks, int masks_on_device, int H, int W, int device,
                    double *scores_out, int32_t *labels_out);
const char *sc_label_points_last_error(void);

/* Page-locked host memory for the read-back of sc_get_values (no reference counterpart: the
 * reference's values_h is a pageable NumPy array, cl.py:173).  A 512 MiB volume reads back in
 * ~10 ms into such a buffer against ~50 ms into pageable memory, but allocating it takes ~0.1 s
 * during which other threads' HIP calls stall (measured: doing it on a thread beside the mask
 * ingest made a one-shot run slower), so it only pays for callers that keep the buffer across
 * many read-backs.  sc_host_free needs no engine: the buffer may outlive it. */
int sc_host_alloc(int device, int64_t bytes, void **ptr);
void sc_host_free(void *ptr);

/* Device-memory helpers so that hosts without a HIP binding can stage inputs in HBM
 * (bench.py, tests): plain hipMalloc / hipMemcpy / hipFree on the engine's device. */
int sc_dev_alloc(sc_engine *e, int64_t bytes, void **ptr);
int sc_dev_free(sc_engine *e, void *ptr);
int sc_dev_upload(sc_engine *e, void *dst_dev, const void *src_host, int64_t bytes);
int sc_dev_download(sc_engine *e, void *dst_host, const void *src_dev, int64_t bytes);

/*
 * Mask ingest (cl.py:298, `io.read_image`): decoder for 8-bit greyscale, non-interlaced PNG -- the files a
 * `Masks` / `Segmentation2D` fileset holds -- callable from the decode-ahead threads with the interpreter
 * lock released (the Python decoders stop scaling at ~3 threads).  sc_png_info returns SC_OK and the
 * size for a file this decoder takes, SC_ERR_INVALID (reason: sc_png_last_error) for anything else
 * (other colour types, bit depths, interlacing, palette or transparency chunks): the caller then uses
 * its usual reader.  sc_png_decode_gray8 writes H*W bytes, row-major.  Host code only (zlib).
 */
int sc_png_info(const void *data, int64_t len, int *W, int *H);
int sc_png_decode_gray8(const void *data, int64_t len, uint8_t *out, int W, int H);
const char *sc_png_last_error(void);

/*
 * Several GPUs from ONE process (SURVEY.md 8b `sc_create_sharded`; the reference drives a single
 * device, cl.py:29-30): one engine per entry of `devices` (HIP ordinals; an ordinal may repeat), the
 * x-planes of the grid dealt round-robin over them (partition 0; every device then holds the same share
 * of the object) or in contiguous slabs (partition 1), voxel coordinates from the GLOBAL plane index as
 * in sc_create_cyclic / sc_create_slab.  Every view goes to every engine; sc_group_get_values writes
 * the whole [nx][ny][nz] grid in global order (one strided device-to-host copy per device).  The
 * one-process-per-GPU form of the same sharding is plant-3d-vision_amd/sharded.py over RCCL.
 * A view's arguments are judged once, before any engine takes it.  Should a call still fail on some engine
 * after others took it (a device error), the planes are in different states: the group then answers
 * SC_ERR_STATE to everything but sc_group_clear (every engine back to default_value) and sc_group_destroy.
 */
typedef struct sc_group sc_group;
int sc_create_sharded(sc_group **out, int64_t nx, int64_t ny, int64_t nz, const float origin[3],
                      float voxel_size, int mode, float default_value, const int *devices, int ndev,
                      int partition);
void sc_group_destroy(sc_group *g);
int sc_group_size(const sc_group *g);
sc_engine *sc_group_engine(sc_group *g, int i);  /* for per-engine calls (options, statistics, device batches) */
int sc_group_clear(sc_group *g);
int sc_group_set_option(sc_group *g, int key, int64_t value);
int sc_group_set_lut(sc_group *g, const float *lut256);
int sc_group_process_view(sc_group *g, const float K[4], const float R[9], const float t[3], const void *mask,
                          int H, int W, int mask_dtype, int64_t row_stride_bytes);
int sc_group_flush(sc_group *g);
int sc_group_synchronize(sc_group *g);
int sc_group_get_values(sc_group *g, void *out);

#ifdef __cplusplus
}
#endif
#include "spacecarve_tuning.h" /* the SC_OPT_* tuning keys: same key space, results never depend on them */
#endif /* SPACECARVE_H */
