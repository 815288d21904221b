// hostwait.h -- how the library's host threads wait for the device.
// hipStreamSynchronize / hipEventSynchronize hand a wait of more than a few microseconds to the runtime's blocking
// path, and on this stack that path now and then returns tens of milliseconds late: round 4 caught a 3 ms wait for
// the stream taking 47-54 ms in one process out of seven (tools/dbg/literal_outlier.sh; the device was idle long
// before), which turned a 0.25 ms batch into "1.8 ms" when averaged over thirty.  The waits here poll the stream or
// the event instead (a status read, no system call): spinning for the first microseconds, yielding the core for the
// next 2 ms (every wait of the carve path is shorter), then sleeping 50 us a turn -- a long wait (an averaging batch,
// 8 ranks of a node each waiting beside the host pools that decode and pack) must not burn a core per engine thread
//  -- and only after 200 ms -- a wait that long is a long kernel, not a race -- the blocking call.
// SC_WAIT_MODE (environment): 0 block at once, 1 (default) as above, 2 spin then sleep 20 us a turn from the start.
#ifndef SC_HOSTWAIT_H
#define SC_HOSTWAIT_H

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <thread>

namespace schost {

inline int wait_mode() {  // EXPERIMENT (SC_WAIT_MODE): 0 block, 1 spin then yield, 2 spin then sleep 20 us a turn
    static const int mode = [] { const char *e = std::getenv("SC_WAIT_MODE"); return e ? std::atoi(e) : 1; }();
    return mode;
}
template <typename Query, typename Block>
inline hipError_t poll_then_block(Query query, Block block) {
    const int mode = wait_mode();
    if (mode == 0) return block();
    const auto t0 = std::chrono::steady_clock::now();
    bool napping = false;
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = query();
        if (q != hipErrorNotReady) return q;  // done, or a real error
        if (spins < 256) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        } else {
            if (mode == 2) std::this_thread::sleep_for(std::chrono::microseconds(20));
            else if (napping) std::this_thread::sleep_for(std::chrono::microseconds(50));
            else std::this_thread::yield();
            if (napping || (spins & 63u) == 0) {
                const auto waited = std::chrono::steady_clock::now() - t0;
                if (waited > std::chrono::milliseconds(200)) return block();
                napping = waited > std::chrono::milliseconds(2);
            }
        }
    }
}

inline hipError_t wait_stream(hipStream_t s) {
    return poll_then_block([s] { return hipStreamQuery(s); }, [s] { return hipStreamSynchronize(s); });
}
inline hipError_t wait_event(hipEvent_t ev) {
    return poll_then_block([ev] { return hipEventQuery(ev); }, [ev] { return hipEventSynchronize(ev); });
}

}  // namespace schost

#endif
