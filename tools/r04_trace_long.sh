#!/bin/bash
# Runs on the GPU box: per-dispatch kernel trace of a long run (do kernel durations drift with time / clocks?)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04/cadence
mkdir -p "$O"
LIGHT="--cpu-seconds 0 --extra-steps 0 --e2e-reps 0 --skip-other-path"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace400" -- \
    python3 "$R/bench.py" --gpus 1 --steps 400 --warmup 20 $LIGHT > "$O/b400_trace.json" 2> "$O/b400_trace.err" || exit 4
echo done
