"""Per-kernel means of the SQ counters collected by tools/pmc_sq.sh; optional JSON summary for profiles/.
    python tools/pmc_sq_report.py [out.json]"""
import csv, glob, collections, json, re, sys
out = collections.defaultdict(dict)
for tag in ("pmc_sqA", "pmc_sqB"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            n = re.sub(r"\(.*$", "", n)
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, c in acc.items():
        means = {k: sum(v) / len(v) for k, v in sorted(c.items())}
        out[n].update(means)
        print(tag, n, {k: round(v) for k, v in means.items()})
if len(sys.argv) > 1:
    json.dump({"kernels": {k: v for k, v in sorted(out.items()) if "rocclr" not in k},
               "note": "mean per dispatch of rocprofv3 --pmc SQ_* counters over the fused steps of bench.py --skip-other-path "
                       "(tools/pmc_sq.sh, two passes); raw values, cycle counters are summed over wavefronts (quad-cycles)"},
              open(sys.argv[1], "w"), indent=1)
