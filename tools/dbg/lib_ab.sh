# same-box A/B of build/libspacecarve_x.so (an experiment build of the working tree) against the tree's committed library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06
X=$R/build/libspacecarve_x.so
SC=${1:-plant,dense,solid,literal,noise}; REPS=${2:-3}; STEPS=${3:-40}
O=gpurun_out/r06/lib_ab.txt; : > $O
for rep in $(seq $REPS); do
  python3 tools/bench_scenes.py --steps $STEPS --scenes $SC --tag base 2>/dev/null | tail -1 >> $O
  SPACECARVE_LIB=$X python3 tools/bench_scenes.py --steps $STEPS --scenes $SC --tag new 2>/dev/null | tail -1 >> $O
done
python3 - "$SC" <<'PY'
import json, sys
sc = sys.argv[1].split(",")
for l in open("gpurun_out/r06/lib_ab.txt"):
    d = json.loads(l); print(d["tag"], {k: d[k]["ms"] for k in sc})
PY
