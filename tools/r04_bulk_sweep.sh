#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/r04
for bm in 128 192 256; do for fl in 0 8192 1000000; do
  python3 tools/bench_scenes.py --steps 30 --scenes plant,dense,literal --opt SC_OPT_BULK_MIN=$bm --opt SC_OPT_BULK_FLOOR=$fl --tag bm${bm}_fl${fl} 2>/dev/null | tail -1 >> gpurun_out/r04/bulk_sweep.jsonl
done; done
python3 tools/bench_scenes.py --steps 30 --scenes plant,dense,literal --opt SC_OPT_BULK_MIN=0 --tag bm0 2>/dev/null | tail -1 >> gpurun_out/r04/bulk_sweep.jsonl
echo done
