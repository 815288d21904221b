#!/bin/bash
# Same-box A/B of two builds of the library over the bench scenes (GPU box):
#     bash tools/ab.sh "<bench_scenes.py options for the new build>" [scenes] [reps] [steps]
# `prev` = build/prev/plant-3d-vision_amd/libspacecarve.so (the build the change is measured against -- built in the
# container from the commit named in build/prev/COMMIT), `new` = the tree's own library.  The two alternate inside
# one call, so both see the same box, clock state and neighbours.  Summarise with tools/ab_show.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
SC=${2:-plant,noise,dense,literal,solid}
REPS=${3:-2}
STEPS=${4:-40}
for rep in $(seq $REPS); do
  SPACECARVE_LIB=$R/build/prev/plant-3d-vision_amd/libspacecarve.so python3 tools/bench_scenes.py --steps $STEPS --scenes $SC --tag prev 2>/dev/null | tail -1
  python3 tools/bench_scenes.py --steps $STEPS --scenes $SC --tag new $1 2>/dev/null | tail -1
  echo "rep $rep done" >&2
done
