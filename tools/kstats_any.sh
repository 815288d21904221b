#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of any python tool; prints mean µs per kernel.
# usage: bash tools/kstats_any.sh <tag> <script.py> [args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/ks_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
S=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/$S" "$@" > "$OUT/out.json" 2> "$OUT/err.txt" || { echo "$TAG failed"; exit 1; }
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    out.append(f"{name} x{r['Calls']} {float(r['AverageNs']) / 1e3:.1f}")
print(sys.argv[2], "|", "; ".join(out))
PY
