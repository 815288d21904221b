#!/bin/bash
# Same-box comparison of several builds of the library (GPU box): bash tools/ab_variants.sh "libA.so libB.so ..." [scenes] [reps] [steps] ["bench_scenes options"]
# The builds alternate inside one call; `-` stands for the tree's own library.  Summarise with tools/ab_show.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
SC=${2:-plant,dense,literal,noise,solid}
REPS=${3:-2}
STEPS=${4:-40}
for rep in $(seq $REPS); do
  for lib in $1; do
    if [ "$lib" = "-" ]; then
      python3 tools/bench_scenes.py --steps $STEPS --scenes $SC --tag tree $5 2>/dev/null | tail -1
    else
      SPACECARVE_LIB=$R/$lib python3 tools/bench_scenes.py --steps $STEPS --scenes $SC --tag $(basename $lib .so) $5 2>/dev/null | tail -1
    fi
    echo "rep $rep $lib done" >&2
  done
done
