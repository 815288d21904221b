# A/B of an experiment build (build/libspacecarve_x.so, -DSC_X_FILL_STREAM): the -1 fill of the EMPTY bricks on a side
# stream from the flags kernel's end on, through the main stream's kernel boundaries
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06
X=$R/build/libspacecarve_x.so
O=gpurun_out/r06/xfill_ab.txt; : > $O
for rep in 1 2; do
  SPACECARVE_LIB=$X python3 tools/bench_scenes.py --steps 40 --scenes plant,dense,solid,literal --tag off 2>/dev/null | tail -1 >> $O
  SC_X_FILL_STREAM=1 SPACECARVE_LIB=$X python3 tools/bench_scenes.py --steps 40 --scenes plant,dense,solid,literal --tag on 2>/dev/null | tail -1 >> $O
  SC_X_FILL_STREAM=1 SC_X_FILL_BLOCKS=128 SPACECARVE_LIB=$X python3 tools/bench_scenes.py --steps 40 --scenes plant,dense,solid,literal --tag on128 2>/dev/null | tail -1 >> $O
  SC_X_FILL_STREAM=1 SC_X_FILL_BLOCKS=512 SPACECARVE_LIB=$X python3 tools/bench_scenes.py --steps 40 --scenes plant,dense,solid,literal --tag on512 2>/dev/null | tail -1 >> $O
done
SC_X_FILL_STREAM=1 SPACECARVE_LIB=$X python3 bench.py --steps 20 --warmup 5 --skip-other-path --extra-steps 0 --e2e-reps 0 --cold-reps 0 --traffic-passes off 2>/dev/null | tail -1 > gpurun_out/r06/xfill_bench.json
python3 - <<'PY'
import json
for l in open("gpurun_out/r06/xfill_ab.txt"):
    d = json.loads(l); print(d["tag"], {k: d[k]["ms"] for k in ("plant", "dense", "solid", "literal")})
d = json.loads(open("gpurun_out/r06/xfill_bench.json").read())
print("bench", d["ms_per_step"], d["roofline"]["frac"], {k: v for k, v in d["parity_check"].items() if k != "note"})
PY
