// hostpack.h -- the host half of mask ingest and label read-back (replaces the host work of
// plant3dvision/cl.py:205-221, 298-301 and 229-232 of the reference): carve masks are reduced to 1 bit per pixel
// (`pixel != 0` after the optional np.invert: the test of backprojection.c:79 on the cast of cl.py:215) on host
// threads before they cross PCIe, and 2-bit labels are widened to int32 as their pieces land.  Plain C++, no HIP.
#ifndef SC_HOSTPACK_H
#define SC_HOSTPACK_H

#include <cstddef>
#include <cstdint>
#include <functional>

namespace schost {

// Rows [row0, row1) of a mask -> row-major bits: word (v, tx) = pixels 32 tx .. 32 tx + 31 of row v, pixel u at bit
// u & 31; bits beyond W are 0 (background).  elem: 1 (bytes; a pixel is foreground when byte != flip_byte: flip_byte
// 0 plain, 255 = np.invert of uint8, 1 = np.invert of bool bytes) or 4 (int32, foreground when != 0).
void pack_rows(const void *mask, int64_t row_stride_bytes, int W, int row0, int row1, uint32_t *out, int wpr,
               int elem, uint8_t flip_byte);

// words [w0, w1) of 2-bit labels (16 per word, label = the pair sign-extended: 3 -> -1) into int32; `n` labels in all
void widen2(const uint32_t *src, int32_t *dst, int64_t w0, int64_t w1, int64_t n);

// A small process-wide pool for these loops.  Workers sleep on a condition variable (after a short spin, so that a
// burst of sc_process_view calls does not pay a wake-up each); nothing spins while there is no work.
int pool_threads();                 // workers + the caller
void pool_set_threads(int n);       // 0: default (min(8, hardware threads / 2)); takes effect before the first use
// fn(part) for part in [0, nparts), on the caller and the workers; returns when all are done
void parallel_for(int nparts, const std::function<void(int)> &fn);

// A latch-style group of tasks run by the pool's workers (the caller does not take part: it is busy feeding them)
struct TaskGroup {
    void *impl;
    TaskGroup();
    ~TaskGroup();
    void submit(std::function<void()> fn);
    void wait();  // every submitted task has run
};

}  // namespace schost

#endif
