#!/bin/bash
# (CPU) After `gpurun -- bash tools/final_profiles.sh <tag>`: the files kept for the round, from gpurun_out/ into profiles/.
#     bash tools/collect_profiles.sh r05
set -eu
TAG=${1:-r05}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
P=$R/profiles
cd "$R"
python3 tools/pmc_traffic.py ${TAG}s > /dev/null
python3 tools/pmc_traffic.py $TAG > /dev/null   # (that order: the fused entry of pmc_traffic.json comes from the fused run)
python3 tools/ramp_summary.py $TAG > $P/${TAG}_clock_ramp.txt
cp $O/bench_driver_cadence.json $P/${TAG}_bench_driver_cadence.json
cp $O/bench_default.json $P/${TAG}_bench_default.json
cp $O/bench_under_rocprof.json $P/${TAG}_bench_under_rocprof.json
cp "$(ls -t $(find $O/prof_cadence -name '*kernel_stats.csv') | head -1)" $P/${TAG}_kernel_stats_driver_cadence.csv
cp "$(ls -t $(find $R/gpurun_out/prof_${TAG}s_stats -name '*kernel_stats.csv') | head -1)" $P/${TAG}s_kernel_stats.csv
cp $O/scenes_ab.jsonl $P/${TAG}_scenes_ab.jsonl
cp $O/avg_ab.jsonl $P/${TAG}_avg_ab.jsonl
cp $O/scene_kernel_means.txt $P/${TAG}_scene_kernel_means_noise_solid.txt
cp $O/bench_rccl1.json $P/${TAG}_bench_rccl_group_of_one.json
cp $O/bench_gloo2.json $P/${TAG}_bench_gloo_2ranks_self_launched.json
cp $O/e2e.json $P/${TAG}_e2e.json
cp $O/e2e_breakdown.log $P/${TAG}_e2e_breakdown.txt
cp $O/ml_averaging.json $P/${TAG}_ml_averaging.json
cp $O/ml_carving.json $P/${TAG}_ml_carving.json
cp $O/vol2pcd.json $P/${TAG}_vol2pcd.json
for s in sq_plant sq_literal sq_dense; do cp $R/gpurun_out/${TAG}_${s}_counters.json $P/${TAG}_${s}.json; done
for s in avg_u8_grey avg_f32_grey; do cp $R/gpurun_out/${TAG}_${s}_counters.json $P/${TAG}_${s}_counters.json; done
echo "collected $TAG"
