#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + the two PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>_*
# PMC passes are separate runs, never combined with --stats/sys-trace (pool rule), and the
# program itself follows `--` (no env/bash hop: the profiler initialises the GPU first).
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_stats" -- \
    python3 "$R/bench.py" --steps 10 --warmup 2 --cpu-seconds 0 --extra-steps 0 "$@" \
    > "$OUT/bench_${TAG}_stats.json" 2> "$OUT/bench_${TAG}_stats.err" || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_fetch" -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 --extra-steps 0 "$@" \
    > "$OUT/bench_${TAG}_fetch.json" 2> "$OUT/bench_${TAG}_fetch.err" || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_write" -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 --extra-steps 0 "$@" \
    > "$OUT/bench_${TAG}_write.json" 2> "$OUT/bench_${TAG}_write.err" || exit 3
echo "profile ${TAG} done"
