#!/bin/bash
# SQ instruction counters per kernel of one scene of tools/bench_scenes.py (GPU box; one PMC pass, kernel-trace only).
# usage: bash tools/pmc_scene.sh <scene>      -> gpurun_out/pmc_scene_<scene>/, prints per-kernel means
S=${1:-dense}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_scene_$S
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT" -- \
    python3 "$R/tools/bench_scenes.py" --scenes "$S" --steps 3 > "$OUT/out.json" 2> "$OUT/err.txt" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(acc.items()):
    print(n, {k: round(sum(v) / len(v) / 1e6, 3) for k, v in sorted(c.items())}, "(millions per dispatch)")
PY
