#!/usr/bin/env python3
"""Fused averaging, 512^3 x 72 uint8 masks + table, for rocprofv3: binary plant masks and random
grey masks, brick form and linear form (SC_OPT_AVG_BRICK)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench_avg as m
m.run(512, 72, 1440, 1080, 0, reps=3, u8=True, binary=True, brick=1)
m.run(512, 72, 1440, 1080, 0, reps=3, u8=True, binary=False, brick=1)
m.run(512, 72, 1440, 1080, 0, reps=3, u8=True, binary=True, brick=0)
