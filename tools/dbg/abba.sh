# plant only, prev (build/prev) against the tree's library in ABBA order: is a difference the build's or the order's?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06
P=$R/build/prev/plant-3d-vision_amd/libspacecarve.so
O=gpurun_out/r06/abba.txt; : > $O
run() { if [ "$1" = prev ]; then SPACECARVE_LIB=$P python3 tools/bench_scenes.py --steps 60 --scenes plant --tag prev 2>/dev/null | tail -1 >> $O; else python3 tools/bench_scenes.py --steps 60 --scenes plant --tag new 2>/dev/null | tail -1 >> $O; fi; }
for rep in 1 2 3 4 5 6; do run prev; run new; run new; run prev; done
python3 - <<'PY'
import json
v = {"prev": [], "new": []}
for l in open("gpurun_out/r06/abba.txt"):
    d = json.loads(l); v[d["tag"]].append(d["plant"]["ms"])
for k, x in v.items():
    x = sorted(x); print(k, "n", len(x), "mean %.4f median %.4f min %.4f max %.4f" % (sum(x) / len(x), x[len(x) // 2], x[0], x[-1]))
PY
