#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass (kernel-trace only: pool rule) over any python tool; prints per-kernel means.
# usage: bash tools/pmc_any.sh <tag> "<counters>" <script.py> [args...]
set -u
TAG=$1; CNT=$2; S=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 150 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/$S" "$@" > "$OUT/out.json" 2> "$OUT/err.txt" || { echo "$TAG failed"; tail -3 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(acc.items()):
    if "list_kernel" in n or "brick_kernel" in n:
        print(sys.argv[2], n, {k: round(sum(v) / len(v) / 1e6, 3) for k, v in sorted(c.items())}, "(millions per dispatch)")
PY
