#!/bin/bash
# Same-box A/B of several builds: bash tools/r04_libs_ab.sh "scenes" lib1 lib2 ...   (paths relative to the repo)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
SC=$1; shift
for rep in 1 2; do for lib in "$@"; do
  SPACECARVE_LIB=$R/$lib python3 tools/bench_scenes.py --steps 30 --scenes $SC --tag $lib $UNITB 2>/dev/null | tail -1
done; done
