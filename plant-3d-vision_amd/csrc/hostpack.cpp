// hostpack.cpp -- see hostpack.h.  SSE2 is the baseline of x86-64; the AVX2 forms are chosen at run time.
#include "hostpack.h"

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

namespace schost {

namespace {

bool has_avx2() {
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}

// ---- 1-byte pixels -> bits ---------------------------------------------------------------------------------
inline uint32_t bits32_sse2(const uint8_t *p, __m128i flip) {  // 32 pixels -> one word, bit i = pixel i is foreground
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p + 16));
    const uint32_t ea = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(a, flip));  // bit i = pixel i == flip byte (background)
    const uint32_t eb = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(b, flip));
    return ~(ea | (eb << 16));
}

__attribute__((target("avx2"))) inline uint32_t bits32_avx2(const uint8_t *p, __m256i flip) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
    return ~(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, flip));
}

inline uint32_t bits_tail(const uint8_t *p, int n, uint8_t flip) {  // n < 32 pixels
    uint32_t w = 0;
    for (int i = 0; i < n; ++i) w |= (uint32_t)(p[i] != flip) << i;
    return w;
}

void pack_bytes_sse2(const uint8_t *src, int64_t stride, int W, int row0, int row1, uint32_t *out, int wpr, uint8_t flip) {
    const __m128i f = _mm_set1_epi8((char)flip);
    const int full = W >> 5, tail = W & 31;
    for (int v = row0; v < row1; ++v) {
        const uint8_t *p = src + (int64_t)v * stride;
        uint32_t *o = out + (int64_t)v * wpr;
        for (int t = 0; t < full; ++t) o[t] = bits32_sse2(p + 32 * t, f);
        if (tail) o[full] = bits_tail(p + 32 * full, tail, flip);
    }
}

__attribute__((target("avx2"))) void pack_bytes_avx2(const uint8_t *src, int64_t stride, int W, int row0, int row1,
                                                     uint32_t *out, int wpr, uint8_t flip) {
    const __m256i f = _mm256_set1_epi8((char)flip);
    const int full = W >> 5, tail = W & 31;
    for (int v = row0; v < row1; ++v) {
        const uint8_t *p = src + (int64_t)v * stride;
        uint32_t *o = out + (int64_t)v * wpr;
        int t = 0;
        for (; t + 4 <= full; t += 4) {  // 128 pixels per turn: four independent loads in flight
            const uint32_t a = bits32_avx2(p + 32 * t, f), b = bits32_avx2(p + 32 * t + 32, f);
            const uint32_t c = bits32_avx2(p + 32 * t + 64, f), d = bits32_avx2(p + 32 * t + 96, f);
            o[t] = a; o[t + 1] = b; o[t + 2] = c; o[t + 3] = d;
        }
        for (; t < full; ++t) o[t] = bits32_avx2(p + 32 * t, f);
        if (tail) o[full] = bits_tail(p + 32 * full, tail, flip);
    }
}

// ---- int32 pixels -> bits (cl.py:215 casts every carve mask to int32; a caller may hand that over) -------------
void pack_i32(const int32_t *src, int64_t stride_bytes, int W, int row0, int row1, uint32_t *out, int wpr) {
    const __m128i zero = _mm_setzero_si128();
    for (int v = row0; v < row1; ++v) {
        const int32_t *p = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(src) + (int64_t)v * stride_bytes);
        uint32_t *o = out + (int64_t)v * wpr;
        int u = 0;
        for (int t = 0; t < wpr; ++t) {
            uint32_t w = 0;
            const int n = std::min(32, W - u);
            int i = 0;
            for (; i + 4 <= n; i += 4) {
                const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p + u + i));
                const uint32_t e = (uint32_t)_mm_movemask_ps(_mm_castsi128_ps(_mm_cmpeq_epi32(a, zero)));
                w |= (~e & 0xfu) << i;
            }
            for (; i < n; ++i) w |= (uint32_t)(p[u + i] != 0) << i;
            o[t] = w;
            u += 32;
        }
    }
}

// ---- 2-bit labels -> int32 ------------------------------------------------------------------------------------
void widen2_scalar(const uint32_t *src, int32_t *dst, int64_t w0, int64_t w1, int64_t n) {
    for (int64_t w = w0; w < w1; ++w) {
        const uint32_t x = src[w];
        int32_t *o = dst + w * 16;
        const int m = (int)std::min<int64_t>(16, n - w * 16);
        for (int i = 0; i < m; ++i) o[i] = (int32_t)(x << (30 - 2 * i)) >> 30;
    }
}

__attribute__((target("avx2"))) void widen2_avx2(const uint32_t *src, int32_t *dst, int64_t w0, int64_t w1, int64_t n) {
    // label i of a word: (x << (30 - 2 i)) >> 30 (arithmetic).  Plain stores: streaming ones measured SLOWER here
    // (512^3 labels on 8 threads of an EPYC 9575F: 4.5 ms against 3.7; 16 threads: 2.9 either way)
    const __m256i sh_lo = _mm256_setr_epi32(30, 28, 26, 24, 22, 20, 18, 16);
    const __m256i sh_hi = _mm256_setr_epi32(14, 12, 10, 8, 6, 4, 2, 0);
    int64_t w = w0;
    const int64_t wfull = std::min(w1, n / 16);  // words whose 16 labels all exist
    for (; w < wfull; ++w) {
        const __m256i x = _mm256_set1_epi32((int)src[w]);
        const __m256i a = _mm256_srai_epi32(_mm256_sllv_epi32(x, sh_lo), 30);
        const __m256i b = _mm256_srai_epi32(_mm256_sllv_epi32(x, sh_hi), 30);
        __m256i *o = reinterpret_cast<__m256i *>(dst + w * 16);
        _mm256_storeu_si256(o, a);
        _mm256_storeu_si256(o + 1, b);
    }
    if (w < w1) widen2_scalar(src, dst, w, w1, n);
}

// ---- the pool --------------------------------------------------------------------------------------------------
struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::vector<std::thread> workers;
    std::atomic<int> spinners{0};
    std::atomic<uint64_t> pushed{0};
    bool stop = false;
    int wanted = 0;
    bool started = false;

    void start_locked() {
        if (started) return;
        started = true;
        int n = wanted;
        if (n <= 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            // (measured on the GPU box, 16 CPUs of 256 for one GPU: 72 masks packed + 512^3 labels read back take 6.4-7.2 ms
            // with 8 threads, 9 with 12, 10 with 16 -- the pack hand-overs collide -- and 8-9 with 4)
            n = (int)std::min<unsigned>(8u, std::max<unsigned>(2u, hw / 2u));
        }
        for (int i = 0; i < n - 1; ++i) workers.emplace_back([this]() { run(); });
    }

    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            while (q.empty() && !stop) {
                // a short spin (work usually comes in bursts: one sc_process_view after the other), then sleep
                const uint64_t seen = pushed.load(std::memory_order_relaxed);
                lk.unlock();
                bool woke = false;
                for (int i = 0; i < 2000 && !woke; ++i) {
                    _mm_pause();
                    woke = pushed.load(std::memory_order_relaxed) != seen;
                }
                lk.lock();
                if (!woke && q.empty() && !stop) cv.wait(lk);
            }
            if (stop && q.empty()) return;
            std::function<void()> fn = std::move(q.front());
            q.pop_front();
            lk.unlock();
            fn();
            lk.lock();
        }
    }

    void push(std::function<void()> fn) {
        {
            std::lock_guard<std::mutex> lk(mu);
            start_locked();
            if (!workers.empty()) {
                q.push_back(std::move(fn));
                pushed.fetch_add(1, std::memory_order_relaxed);
                fn = nullptr;
            }
        }
        if (fn) fn();  // a pool of one: the caller is the worker
        else cv.notify_one();
    }

    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
};

// A fork()ed child inherits the pool object but none of its threads (`started` true, `workers` holding the parent's
// thread objects, perhaps a locked mutex): work queued there would wait for nobody (ADVICE r04 -- multiprocessing's
// fork start method, luigi workers).  The child gets a fresh pool, which starts its own threads at its first push;
// the old object is leaked on purpose (its mutex may be held by a thread that does not exist here).
std::atomic<Pool *> g_pool{nullptr};
void pool_after_fork_child() { g_pool.store(new Pool(), std::memory_order_release); }

Pool &pool() {
    static const bool once = [] {
        g_pool.store(new Pool(), std::memory_order_release);  // never destroyed: worker threads may outlive static destruction order otherwise
        pthread_atfork(nullptr, nullptr, pool_after_fork_child);
        return true;
    }();
    (void)once;
    return *g_pool.load(std::memory_order_acquire);
}

struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    int pending = 0;
    void add() {
        std::lock_guard<std::mutex> lk(mu);
        ++pending;
    }
    void done() {
        std::lock_guard<std::mutex> lk(mu);
        if (--pending == 0) cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this]() { return pending == 0; });
    }
};

}  // namespace

void pack_rows(const void *mask, int64_t stride, int W, int row0, int row1, uint32_t *out, int wpr, int elem,
               uint8_t flip) {
    if (elem == 4) {
        pack_i32(static_cast<const int32_t *>(mask), stride, W, row0, row1, out, wpr);
    } else if (has_avx2()) {
        pack_bytes_avx2(static_cast<const uint8_t *>(mask), stride, W, row0, row1, out, wpr, flip);
    } else {
        pack_bytes_sse2(static_cast<const uint8_t *>(mask), stride, W, row0, row1, out, wpr, flip);
    }
}

void widen2(const uint32_t *src, int32_t *dst, int64_t w0, int64_t w1, int64_t n) {
    if (has_avx2()) widen2_avx2(src, dst, w0, w1, n);
    else widen2_scalar(src, dst, w0, w1, n);
}

int pool_threads() {
    Pool &p = pool();
    std::lock_guard<std::mutex> lk(p.mu);
    p.start_locked();
    return (int)p.workers.size() + 1;
}

void pool_set_threads(int n) {
    Pool &p = pool();
    std::lock_guard<std::mutex> lk(p.mu);
    if (!p.started) p.wanted = n;
}

void parallel_for(int nparts, const std::function<void(int)> &fn) {
    if (nparts <= 0) return;
    if (nparts == 1) {
        fn(0);
        return;
    }
    Pool &p = pool();
    const int helpers = std::min(pool_threads() - 1, nparts - 1);
    auto next = std::make_shared<std::atomic<int>>(0);
    auto latch = std::make_shared<Latch>();
    auto body = [next, nparts, &fn]() {
        for (;;) {
            const int i = next->fetch_add(1, std::memory_order_relaxed);
            if (i >= nparts) return;
            fn(i);
        }
    };
    for (int h = 0; h < helpers; ++h) {
        latch->add();
        p.push([body, latch]() {
            body();
            latch->done();
        });
    }
    body();
    latch->wait();  // fn is referenced by the helpers until here
}

TaskGroup::TaskGroup() : impl(new Latch()) {}
TaskGroup::~TaskGroup() {
    wait();
    delete static_cast<Latch *>(impl);
}
void TaskGroup::submit(std::function<void()> fn) {
    Latch *l = static_cast<Latch *>(impl);
    l->add();
    pool().push([fn, l]() {
        fn();
        l->done();
    });
}
void TaskGroup::wait() { static_cast<Latch *>(impl)->wait(); }

}  // namespace schost
