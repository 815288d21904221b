#!/bin/bash
# The measurement set kept under profiles/ for a round, in one go on the GPU box (gpurun, <= 20 min):
#   bash tools/final_profiles.sh r03
# then, back in the build container:  python tools/pmc_traffic.py r03s && python tools/pmc_traffic.py r03
# (that order: the fused entry of profiles/pmc_traffic.json must come from the fused run) and copy the files
# listed in profiles/README.md out of gpurun_out/<tag>/.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
bash tools/profile_gpu.sh "$TAG" --skip-other-path --e2e-reps 0 || exit 1
bash tools/profile_gpu.sh "${TAG}s" --path stream --e2e-reps 0 || exit 2
bash tools/pmc_sq.sh --e2e-reps 0 || exit 3
python3 tools/pmc_sq_report.py "$O/sq_counters.json" > "$O/sq_report.log" 2>&1 || exit 4
python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 5
python3 bench.py --rccl-rehearsal --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 \
    > "$O/bench_rccl1.json" 2> "$O/bench_rccl1.err" || exit 6
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --share-device --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 \
    > "$O/bench_gloo2.json" 2> "$O/bench_gloo2.err" || exit 7
python3 tools/bench_e2e.py --reps 4 > "$O/e2e.json" 2> "$O/e2e.err" || exit 8
python3 tools/bench_ml.py > "$O/ml_averaging.json" 2> "$O/ml_averaging.err" || exit 9
python3 tools/bench_ml.py --type carving > "$O/ml_carving.json" 2> "$O/ml_carving.err" || exit 10
python3 tools/bench_vol2pcd.py > "$O/vol2pcd.json" 2> "$O/vol2pcd.err" || true
python3 tools/e2e_breakdown.py > "$O/e2e_breakdown.log" 2>&1 || true
echo "final profiles $TAG done"
