#!/bin/bash
# One scene of tools/bench_scenes.py under rocprofv3 (GPU box): kernel stats, SQ instruction counters, HBM bytes
# (FETCH_SIZE / WRITE_SIZE, separate passes) and write-request counts -- per kernel, means per dispatch.
# usage: bash tools/pmc_scene_full.sh <scene> <out.json>
S=${1:-noise}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
J=${2:-$R/gpurun_out/r04/pmc_$S.json}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_full_$S
rm -rf "$OUT"; mkdir -p "$OUT"
run() { rocprofv3 "$@" --kernel-trace --output-format csv -d "$OUT/$TAG" -- python3 "$R/tools/bench_scenes.py" --scenes "$S" --steps 3 > "$OUT/$TAG.json" 2> "$OUT/$TAG.err"; }
TAG=stats; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$TAG" -- python3 "$R/tools/bench_scenes.py" --scenes "$S" --steps 10 > "$OUT/$TAG.json" 2> "$OUT/$TAG.err" || exit 1
TAG=sq; run --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 2
TAG=fetch; run --pmc FETCH_SIZE || exit 3
TAG=write; run --pmc WRITE_SIZE || exit 4
TAG=wrreq; run --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum || echo "no wrreq counters"
python3 - "$OUT" "$J" "$S" <<'PY'
import csv, glob, collections, re, sys, json
out = collections.defaultdict(dict)
def short(n): return re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        out[short(r["Name"])]["mean_us"] = float(r["AverageNs"]) / 1e3
        out[short(r["Name"])]["calls"] = int(r["Calls"])
for tag in ("sq", "fetch", "write", "wrreq"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[1] + f"/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, c in acc.items():
        for k, v in c.items():
            out[n][k] = sum(v) / len(v)
res = {"scene": sys.argv[3], "kernels": {k: v for k, v in sorted(out.items()) if "rocclr" not in k and "fill_kernel" not in k},
       "note": "means per dispatch; FETCH_SIZE / WRITE_SIZE in KiB (gfx950: wide streaming reads count at half their bytes), "
               "SQ_* raw wavefront-instruction counts; tools/pmc_scene_full.sh"}
json.dump(res, open(sys.argv[2], "w"), indent=1)
for k, v in res["kernels"].items():
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
PY
