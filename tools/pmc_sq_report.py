"""Per-kernel means of the SQ counters collected by tools/pmc_sq.sh."""
import csv, glob, collections, sys
for tag in ("pmc_sqA", "pmc_sqB"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            n = n[n.find("::") + 2:][:28] if "::" in n else n[:28]
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, c in acc.items():
        print(tag, n, {k: round(sum(v) / len(v)) for k, v in sorted(c.items())})
