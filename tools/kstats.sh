#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of a short bench.py run; prints mean µs per kernel.
# usage: bash tools/kstats.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/ks_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
    python3 "$R/bench.py" --steps 10 --warmup 2 --cpu-seconds 0 --extra-steps 0 --skip-other-path "$@" \
    > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "$TAG failed"; exit 1; }
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
out = []
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "fill_kernel" in name or "selftest" in name: continue
    us = float(r["AverageNs"]) / 1e3
    out.append(f"{name} {us:.1f}")
    tot += us
print(sys.argv[2], "|", "; ".join(out), "| sum", round(tot, 1))
PY
