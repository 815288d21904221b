# A/B of an experiment build (build/libspacecarve_x.so, -DSC_X_CONFIRM_STREAM): the confirm kernel on a second stream
# beside the special kernel (plant: no candidates, so the two are independent there)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06
X=$R/build/libspacecarve_x.so
O=gpurun_out/r06/xstream_ab.txt; : > $O
for rep in 1 2 3; do
  SPACECARVE_LIB=$X python3 tools/bench_scenes.py --steps 40 --scenes plant,solid --tag x_off 2>/dev/null | tail -1 >> $O
  SC_X_CONFIRM_STREAM=1 SPACECARVE_LIB=$X python3 tools/bench_scenes.py --steps 40 --scenes plant,solid --tag x_on 2>/dev/null | tail -1 >> $O
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r06/xstream_ab.txt"):
    d = json.loads(l); print(d["tag"], {k: d[k]["ms"] for k in ("plant", "solid")})
PY
