#!/usr/bin/env python3
"""The assembly kernel alone at the size the north star names (GPU box): `world` ranks' planes of a 1024^3 grid,
packed at 2 bits (or 1) per label as an all-gather leaves them, into ONE volume in global plane order on this
device -- sc_unpack_labels, int8 and int32 out, cyclic and slab partitions.  Device time by a torch event pair
around 20 launches; nothing crosses xGMI here, the collective's own time is the driver's SCALE run's to show.

    python tools/bench_unpack.py [--n 1024] [--world 8]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plant3dvision_amd import _native as nat  # noqa: E402
from plant3dvision_amd.sharded import rank_planes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--world", type=int, default=8)
    a = ap.parse_args()
    n, W = a.n, a.world
    dev = torch.device("cuda:0")
    out = {"grid": [n, n, n], "world": W}
    for bits in (2, 1):
        for partition in ("cyclic", "slab"):
            planes_max = max(len(rank_planes(n, W, r, partition)) for r in range(W))
            rank_bytes = nat.packed_bytes(planes_max * n * n, bits)
            recv = torch.randint(0, 256, (W * rank_bytes,), dtype=torch.uint8, device=dev)
            if bits == 2:  # no pair may read 2 (not a label)
                recv &= 0x55
            for out_bytes, dt in ((1, torch.int8), (4, torch.int32)):
                vol = torch.empty((n, n, n), dtype=dt, device=dev)
                s = torch.cuda.current_stream().cuda_stream
                for _ in range(3):
                    nat.unpack_labels(0, s, recv.data_ptr(), rank_bytes, W, partition, (n, n, n), bits, vol.data_ptr(), out_bytes)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    nat.unpack_labels(0, s, recv.data_ptr(), rank_bytes, W, partition, (n, n, n), bits, vol.data_ptr(), out_bytes)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                moved = W * rank_bytes + n ** 3 * out_bytes
                out[f"{bits}bit_{partition}_int{8 * out_bytes}"] = {"ms": round(ms, 4), "GB_per_s": round(moved / ms / 1e6, 1),
                                                                     "wire_MiB_per_rank": round(rank_bytes / 2 ** 20, 1)}
                del vol
            del recv
    print(json.dumps(out))


if __name__ == "__main__":
    main()
