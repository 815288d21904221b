#!/usr/bin/env python3
"""Host-side costs of the read-back at 512^3 (diagnostic): page touching, widening int8 -> int32 into fresh
and into touched arrays, device-to-host copies into fresh and touched arrays."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat
T = time.perf_counter
shape = (512, 512, 512)
e = nat.Engine(shape, [0, 0, 0], 1.0, nat.SC_MODE_CARVE)
e.get_values(np.zeros(shape, np.int32))  # warm
src8 = np.zeros(shape, np.int8)
for rep in range(3):
    t0 = T(); a = nat.TouchedEmpty(shape, np.int32, threads=4).result(); t1 = T()
    b = nat.TouchedEmpty(shape, np.int32, threads=16).result(); t2 = T()
    c = nat.TouchedEmpty(shape, np.int8, threads=4).result(); t3 = T()
    f = np.empty(shape, np.int32); t4 = T(); nat.widen_i8(f, src8, workers=8); t5 = T()
    f2 = np.empty(shape, np.int32); t6 = T(); nat.widen_i8(f2, src8, workers=16); t7 = T()
    nat.widen_i8(a, src8, workers=8); t8 = T()
    g = np.empty(shape, np.int32); t9 = T(); e.get_values(g); t10 = T()
    e.get_values(a); t11 = T()
    e.get_values_i8(c); t12 = T()
    h = np.empty(shape, np.int8); t13 = T(); e.get_values_i8(h); t14 = T()
    del a, b, c, f, f2, g, h
print(f"touch int32 4thr {1e3*(t1-t0):.1f}  16thr {1e3*(t2-t1):.1f}  touch int8 4thr {1e3*(t3-t2):.1f} | widen->fresh 8thr {1e3*(t5-t4):.1f}  16thr {1e3*(t7-t6):.1f}  "
      f"widen->touched {1e3*(t8-t7):.1f} | D2H int32 fresh {1e3*(t10-t9):.1f}  touched {1e3*(t11-t10):.1f}  D2H int8 touched {1e3*(t12-t11):.1f}  fresh {1e3*(t14-t13):.1f} ms")
e.close()
