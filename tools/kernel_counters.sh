#!/bin/bash
# Runs on the GPU box: the SQ / cache counters of every kernel a python tool launches, in SEPARATE rocprofv3 --pmc passes
# (kernel trace only beside them: the pool rule), and a --kernel-trace --stats pass for the durations.
#   bash tools/kernel_counters.sh <tag> <script.py> [args...]     ->  gpurun_out/<tag>_counters.json
# The program itself stands directly behind `--` (python3 <script>): no env / bash -c hop under the profiler.
set -u
TAG=$1; shift
S=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/kc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES"
 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH"
 "TCC_HIT_sum TCC_MISS_sum"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/$S" "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $P" >&2
  i=$((i+1))
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/$S" "$@" > "$OUT/stats.log" 2>&1 || echo "stats pass failed" >&2
python3 - "$OUT" "$R/gpurun_out/${TAG}_counters.json" "$S $*" <<'PY'
import csv, glob, collections, re, sys, json
def clean(n):
    return re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[clean(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
ks = {n: {k: sum(v) / len(v) for k, v in sorted(c.items())} for n, c in acc.items() if "rocclr" not in n}
for n, c in acc.items():
    if n in ks:
        ks[n]["dispatches_per_pass"] = len(next(iter(c.values())))
for f in glob.glob(f"{sys.argv[1]}/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = clean(r["Name"])
        if n in ks:
            ks[n]["mean_us"] = float(r["AverageNs"]) / 1e3
            ks[n]["calls"] = int(r["Calls"])
for f in glob.glob(f"{sys.argv[1]}/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = clean(r["Kernel_Name"])
        if n in ks and "VGPR_Count" in r:
            ks[n]["VGPR_Count"] = int(r.get("VGPR_Count") or 0)
            ks[n]["SGPR_Count"] = int(r.get("SGPR_Count") or 0)
            ks[n]["LDS_Block_Size"] = int(r.get("LDS_Block_Size") or 0)
json.dump({"command": "python3 " + sys.argv[3], "kernels": ks,
           "note": "mean per dispatch; separate rocprofv3 --pmc passes (kernel trace only beside them), durations from a "
                   "--kernel-trace --stats pass of the same command; SQ cycle counters are quad-cycles summed over wavefronts "
                   "(SQ_BUSY_CYCLES: summed over shader engines); tools/kernel_counters.sh"}, open(sys.argv[2], "w"), indent=1)
for n, c in sorted(ks.items(), key=lambda kv: -kv[1].get("mean_us", 0)):
    print(n, round(c.get("mean_us", 0), 1), "us", {k: round(v) for k, v in c.items() if k.startswith("SQ_INSTS_VALU") or k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")})
PY
