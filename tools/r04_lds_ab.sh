#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  python3 tools/bench_scenes.py --steps 40 --scenes plant,noise,dense,literal --tag lds0 2>/dev/null | tail -1
  python3 tools/bench_scenes.py --steps 40 --scenes plant,noise,dense,literal --tag lds1 --opt SC_OPT_LDS_TILES=1 2>/dev/null | tail -1
done
bash tools/kstats_any.sh lds1_noise tools/bench_scenes.py --scenes noise --steps 10 --opt SC_OPT_LDS_TILES=1
bash tools/kstats_any.sh lds1_plant tools/bench_scenes.py --scenes plant --steps 10 --opt SC_OPT_LDS_TILES=1
