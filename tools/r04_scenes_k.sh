#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out/r04
for sc in dense literal noise solid plant; do
  bash tools/kstats_any.sh sc_$sc tools/bench_scenes.py --scenes $sc --steps 10 2>&1 | tail -1
done
