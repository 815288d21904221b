#!/bin/bash
# Runs on the GPU box: the host-mask paths -- bench e2e leg (host masks -> host labels), files -> volume, breakdown
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out/r04
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --skip-other-path --traffic-passes off --cold-reps 0 --e2e-reps 7 > gpurun_out/r04/e2e_bench.json 2> gpurun_out/r04/e2e_bench.err || exit 1
python3 tools/bench_e2e.py --reps 5 > gpurun_out/r04/e2e_files.json 2> gpurun_out/r04/e2e_files.err || exit 2
python3 tools/e2e_breakdown.py > gpurun_out/r04/e2e_breakdown.log 2>&1 || exit 3
python3 tools/bench_host_masks.py > gpurun_out/r04/host_masks.log 2>&1 || true
echo ok
