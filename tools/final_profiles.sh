#!/bin/bash
# The measurement set kept under profiles/ for a round, in one go on the GPU box (gpurun, <= 20 min):
#   bash tools/final_profiles.sh r06
# then, back in the build container:  python tools/pmc_traffic.py r05s && python tools/pmc_traffic.py r05
# (that order: the fused entry of profiles/pmc_traffic.json must come from the fused run) and copy the files
# listed in profiles/README.md out of gpurun_out/<tag>/.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
# under the profiler the line's own child legs (traffic passes, cold process) and the parity check stay out
QUIET="--cpu-seconds 0 --extra-steps 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off --parity-check off --cold-process off"
# the driver's own command first: THE headline
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_cadence.json" 2> "$O/bench_driver_cadence.err" || exit 1
# the same cadence under rocprofv3 (kernel means that must agree with the line's roofline) + per-dispatch trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_cadence" -- \
    python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 $QUIET \
    > "$O/bench_under_rocprof.json" 2> "$O/bench_under_rocprof.err" || exit 2
rocprofv3 --kernel-trace --output-format csv -d "$O/prof_ramp" -- \
    python3 "$R/bench.py" --gpus 1 --steps 400 --warmup 20 $QUIET --skip-other-path \
    > "$O/bench_ramp.json" 2> "$O/bench_ramp.err" || exit 3
cd "$R"
bash tools/profile_gpu.sh "$TAG" --skip-other-path --e2e-reps 0 --cold-reps 0 --traffic-passes off --parity-check off --cold-process off || exit 4
bash tools/profile_gpu.sh "${TAG}s" --path stream --e2e-reps 0 --cold-reps 0 --traffic-passes off --parity-check off --cold-process off || exit 5
# counters per kernel: the headline scene, the reference's literal configuration, the dense scene, the averaging launch
bash tools/kernel_counters.sh ${TAG}_sq_plant tools/bench_scenes.py --steps 6 --scenes plant > "$O/kc_plant.log" 2>&1 || exit 6
bash tools/kernel_counters.sh ${TAG}_sq_literal tools/bench_scenes.py --steps 6 --scenes literal > "$O/kc_literal.log" 2>&1 || exit 6
bash tools/kernel_counters.sh ${TAG}_sq_dense tools/bench_scenes.py --steps 6 --scenes dense > "$O/kc_dense.log" 2>&1 || exit 6
bash tools/kernel_counters.sh ${TAG}_avg_u8_grey tools/bench_avg.py --reps 2 --forms u8_grey > "$O/kc_avg_u8.log" 2>&1 || exit 6
bash tools/kernel_counters.sh ${TAG}_avg_f32_grey tools/bench_avg.py --reps 2 --forms f32_grey > "$O/kc_avg_f32.log" 2>&1 || exit 6
for sc in noise solid; do bash tools/kstats_any.sh ${TAG}_$sc tools/bench_scenes.py --steps 12 --scenes $sc; done > "$O/scene_kernel_means.txt" 2>&1
python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 8
python3 bench.py --rccl-rehearsal --steps 20 --warmup 5 --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 \
    > "$O/bench_rccl1.json" 2> "$O/bench_rccl1.err" || exit 9
# N = 2 as a plain command: bench.py starts its own ranks (two of them sharing this box's one GPU, gloo as the transport)
python3 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --share-device --cpu-seconds 0 --extra-steps 0 --e2e-reps 0 \
    > "$O/bench_gloo2.json" 2> "$O/bench_gloo2.err" || exit 10
python3 tools/bench_e2e.py --reps 4 > "$O/e2e.json" 2> "$O/e2e.err" || exit 11
python3 tools/bench_ml.py > "$O/ml_averaging.json" 2> "$O/ml_averaging.err" || true
python3 tools/bench_ml.py --type carving > "$O/ml_carving.json" 2> "$O/ml_carving.err" || true
python3 tools/bench_vol2pcd.py > "$O/vol2pcd.json" 2> "$O/vol2pcd.err" || true
python3 tools/e2e_breakdown.py > "$O/e2e_breakdown.log" 2>&1 || true
for rep in 1 2; do
  SPACECARVE_LIB=$R/build/prev/plant-3d-vision_amd/libspacecarve.so python3 tools/bench_avg.py --tag prev 2>/dev/null | tail -1
  python3 tools/bench_avg.py --tag new 2>/dev/null | tail -1
done > "$O/avg_ab.jsonl"
bash tools/ab.sh "" plant,dense,literal,noise,solid 3 40 > "$O/scenes_ab.jsonl" 2>&1
echo "final profiles $TAG done"
