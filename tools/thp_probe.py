#!/usr/bin/env python3
"""How fast can 512 MiB of fresh host memory be made ready (diagnostic)?  On the GPU box: 2 MiB pages
(what NumPy asks for) 20 / 6 / 3.5 ms on 1 / 4 / 16 threads, 4 KiB pages 36-40 ms whatever the thread count;
returning such an array to the OS costs 25 ms -- keep that outside a timed region."""
import mmap, threading, time
import numpy as np
for f in ("enabled", "defrag", "shmem_enabled"):
    try:
        print(f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except OSError as e:
        print(f, e)
N = 512 << 20
import os, sys
if len(sys.argv) > 1 and sys.argv[1] == "hip":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from plant3dvision_amd import _native as nat
    eng = nat.Engine([512, 512, 512], [0, 0, 0], 1.0, nat.SC_MODE_CARVE)
    print("HIP engine alive in this process")
T = time.perf_counter

def touch(flat, threads, step=4096):
    n = flat.size
    b = [(n * q // threads) // step * step for q in range(threads)] + [n]
    def work(a, e):
        flat[a:e:step] = 0
    th = [threading.Thread(target=work, args=(b[q], b[q + 1])) for q in range(threads)]
    t0 = T()
    for t in th: t.start()
    for t in th: t.join()
    return 1e3 * (T() - t0)

for threads in (1, 4, 8, 16):
    mss = []
    for rep in range(3):
        a = np.empty(N, np.uint8)
        mss.append(round(touch(a, threads), 1))
        del a
    print(f"np.empty, {threads} threads: {mss} ms")
for threads in (1, 4, 8, 16):
    for rep in range(2):
        mm = mmap.mmap(-1, N, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        mm.madvise(mmap.MADV_HUGEPAGE)
        a = np.frombuffer(mm, np.uint8)
        ms = touch(a, threads)
        del a; mm.close()
    print(f"mmap + MADV_HUGEPAGE, {threads} threads: {ms:.1f} ms")
MADV_POPULATE_WRITE = 23
for threads in (1, 4, 8):
    for huge in (0, 1):
        for rep in range(2):
            mm = mmap.mmap(-1, N, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
            if huge: mm.madvise(mmap.MADV_HUGEPAGE)
            part = N // threads
            err = []
            def work(q):
                try:
                    mm.madvise(MADV_POPULATE_WRITE, q * part, part)
                except OSError as e:
                    err.append(e)
            th = [threading.Thread(target=work, args=(q,)) for q in range(threads)]
            t0 = T()
            for t in th: t.start()
            for t in th: t.join()
            ms = 1e3 * (T() - t0)
            mm.close()
        print(f"mmap huge={huge} + MADV_POPULATE_WRITE, {threads} threads: {ms:.1f} ms {err[:1]}")

# allocations that stay alive (what a caller holding earlier volumes looks like)
def private(advice):
    mm = mmap.mmap(-1, N, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    if advice is not None:
        mm.madvise(advice)
    return np.frombuffer(mm, np.uint8)

for name, make in (("np.empty", lambda: np.empty(N, np.uint8)),
                   ("mmap MADV_HUGEPAGE", lambda: private(mmap.MADV_HUGEPAGE)),
                   ("mmap MADV_NOHUGEPAGE", lambda: private(mmap.MADV_NOHUGEPAGE)),
                   ("mmap no advice", lambda: private(None))):
    for threads in (4, 16):
        keep, mss = [], []
        for rep in range(8):
            a = make()
            mss.append(round(touch(a, threads), 1))
            keep.append(a)
        del keep
        print(f"kept alive, {name}, {threads} threads: {mss} ms")
print(open("/proc/meminfo").read().split("Hugepagesize")[0][-400:])
if len(sys.argv) > 1 and sys.argv[1] == "hip":
    keep = []
    for rep in range(4):
        t0 = T(); te = nat.TouchedEmpty((512, 512, 512), np.int32); a = te.result(); ms = 1e3 * (T() - t0)
        keep.append(a)
        print(f"TouchedEmpty int32 512^3: {ms:.1f} ms")
    for rep in range(4):
        t0 = T(); te = nat.TouchedEmpty((512, 512, 512), np.int32); t8 = nat.TouchedEmpty((512, 512, 512), np.int8, threads=2)
        a = te.result(); b = t8.result(); ms = 1e3 * (T() - t0)
        keep.append(a); keep.append(b)
        print(f"TouchedEmpty int32 + int8 together: {ms:.1f} ms")
