#!/bin/bash
# Runs on the GPU box: the driver's cadence (20 steps / 5 warm-up) against the long run, and a per-dispatch
# kernel trace of the driver's cadence (does the first millisecond differ from the steady state?).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04/cadence
mkdir -p "$O"
cd "$R"
LIGHT="--cpu-seconds 0 --extra-steps 0 --e2e-reps 0 --skip-other-path"
python3 bench.py --gpus 1 --steps 20 --warmup 5 $LIGHT > "$O/b20_a.json" 2> "$O/b20_a.err" || exit 1
python3 bench.py --gpus 1 --steps 200 --warmup 20 $LIGHT > "$O/b200.json" 2> "$O/b200.err" || exit 2
python3 bench.py --gpus 1 --steps 20 --warmup 5 $LIGHT > "$O/b20_b.json" 2> "$O/b20_b.err" || exit 3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace20" -- \
    python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 $LIGHT > "$O/b20_trace.json" 2> "$O/b20_trace.err" || exit 4
cd "$R"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/b20_full.json" 2> "$O/b20_full.err" || exit 5
echo cadence done
