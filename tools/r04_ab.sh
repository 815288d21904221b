#!/bin/bash
# Same-box A/B of two builds of the library over the bench scenes: bash tools/r04_ab.sh "<opts for new>" [scenes]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
SC=${2:-plant,noise,dense,literal}
for rep in 1 2; do
  SPACECARVE_LIB=$R/build/prev/plant-3d-vision_amd/libspacecarve.so python3 tools/bench_scenes.py --steps 40 --scenes $SC --tag prev 2>/dev/null | tail -1
  python3 tools/bench_scenes.py --steps 40 --scenes $SC --tag new $1 2>/dev/null | tail -1
done
