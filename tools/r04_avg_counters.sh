#!/bin/bash
# Runs on the GPU box: SQ counters of the fused averaging launch on grey masks (every (brick, view) pair projected),
# uint8 + table and float32 forms, 512^3 x 72 -> gpurun_out/r04/avg_sq_counters.json (lane-ops per voxel.view)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r04/avg_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cat > /tmp/avg_drv.py <<PY
import sys
sys.path.insert(0, "$R/tools")
import microbench_avg as m
form = sys.argv[1]
m.run(512, 72, 1440, 1080, 0, reps=2, u8=(form == "u8"), binary=False, brick=1)
PY
for form in u8 f32; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/$form" -- python3 /tmp/avg_drv.py $form > "$OUT/$form.log" 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${form}_stats" -- python3 /tmp/avg_drv.py $form > "$OUT/${form}_stats.log" 2>&1 || exit 2
done
python3 - "$OUT" "$R/gpurun_out/r04/avg_sq_counters.json" <<'PY'
import csv, glob, collections, re, sys, json
res = {}
for form in ("u8", "f32"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/{form}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ks = {n: {k: sum(v) / len(v) for k, v in c.items()} for n, c in acc.items() if "rocclr" not in n}
    for f in glob.glob(f"{sys.argv[1]}/{form}_stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*$", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            if n in ks:
                ks[n]["mean_us"] = float(r["AverageNs"]) / 1e3
    nvv = 512 ** 3 * 72
    main = [n for n in ks if n.startswith("average_brick_kernel")]
    lane_ops = ks[main[0]]["SQ_INSTS_VALU"] * 64 / nvv if main else None
    res[form] = {"kernels": ks, "lane_ops_per_voxel_view": lane_ops}
    print(form, "lane-ops per voxel.view", lane_ops, {n: round(v.get("mean_us", 0), 1) for n, v in ks.items()})
res["source"] = ("rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace (and a separate "
                 "--kernel-trace --stats pass) of one fused averaging launch, 512^3 x 72 random GREY masks of 1440 x 1080 "
                 "(uint8 + table / float32 in 8x4 tiles), brick form: every (brick, view) pair is projected; tools/r04_avg_counters.sh")
json.dump(res, open(sys.argv[2], "w"), indent=1)
PY
