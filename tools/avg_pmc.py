#!/usr/bin/env python3
"""One fused averaging launch (512^3 x 72 uint8 masks + table) for PMC collection."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench_avg as m
m.run(512, 72, 1440, 1080, 0, reps=2, u8=True)
