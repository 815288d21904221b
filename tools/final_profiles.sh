#!/bin/bash
# The measurement set kept under profiles/ for a round, in one go on the GPU box (gpurun, <= 20 min):
#   bash tools/final_profiles.sh r04
# then, back in the build container:  python tools/pmc_traffic.py r04s && python tools/pmc_traffic.py r04
# (that order: the fused entry of profiles/pmc_traffic.json must come from the fused run) and copy the files
# listed in profiles/README.md out of gpurun_out/<tag>/.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
# the driver's own command first: THE headline
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_cadence.json" 2> "$O/bench_driver_cadence.err" || exit 1
# the same cadence under rocprofv3 (kernel means that must agree with the line's roofline) + per-dispatch trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_cadence" -- \
    python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off \
    > "$O/bench_under_rocprof.json" 2> "$O/bench_under_rocprof.err" || exit 2
rocprofv3 --kernel-trace --output-format csv -d "$O/prof_ramp" -- \
    python3 "$R/bench.py" --gpus 1 --steps 400 --warmup 20 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off --skip-other-path \
    > "$O/bench_ramp.json" 2> "$O/bench_ramp.err" || exit 3
cd "$R"
bash tools/profile_gpu.sh "$TAG" --skip-other-path --e2e-reps 0 --cold-reps 0 --traffic-passes off || exit 4
bash tools/profile_gpu.sh "${TAG}s" --path stream --e2e-reps 0 --cold-reps 0 --traffic-passes off || exit 5
bash tools/pmc_sq.sh --e2e-reps 0 --cold-reps 0 --traffic-passes off || exit 6
python3 tools/pmc_sq_report.py "$O/sq_counters.json" > "$O/sq_report.log" 2>&1 || exit 7
python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 8
python3 bench.py --rccl-rehearsal --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 \
    > "$O/bench_rccl1.json" 2> "$O/bench_rccl1.err" || exit 9
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --share-device --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 \
    > "$O/bench_gloo2.json" 2> "$O/bench_gloo2.err" || exit 10
python3 tools/bench_e2e.py --reps 4 > "$O/e2e.json" 2> "$O/e2e.err" || exit 11
for cfg in "8 1" "8 0"; do python3 tools/r04_e2e_ab.py $cfg 12 2>/dev/null | tail -1 >> "$O/e2e_ab.jsonl"; done
python3 tools/bench_ml.py > "$O/ml_averaging.json" 2> "$O/ml_averaging.err" || true
python3 tools/bench_ml.py --type carving > "$O/ml_carving.json" 2> "$O/ml_carving.err" || true
python3 tools/bench_vol2pcd.py > "$O/vol2pcd.json" 2> "$O/vol2pcd.err" || true
python3 tools/e2e_breakdown.py > "$O/e2e_breakdown.log" 2>&1 || true
echo "final profiles $TAG done"
