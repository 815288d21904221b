#!/usr/bin/env python3
"""Prints the figures of a bench line that the round's work is judged on."""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("ms_per_step %.4f  span %.4f  frac %.4f  steps %d" % (d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["steps"]))
print("traffic", r.get("traffic"), (r.get("traffic_source") or "")[:70])
if "cold_first_batch" in d:
    print("cold", {k: round(v, 3) for k, v in d["cold_first_batch"]["breakdown"].items()}, d["cold_first_batch"]["all_total_ms"])
if d.get("e2e"):
    print("e2e", d["e2e"]["ms"], d["e2e"]["ms_all"])
print("lead", d.get("lead_in"))
for k, v in (d.get("scenes") or {}).items():
    print(k, round(v["ms_per_step"], 4), v["fused_counts"])
for k, v in (d.get("average") or {}).items():
    print(k, round(v["ms_per_step"], 3))
if d.get("stream"):
    print("stream", d["stream"]["ms_per_step"], d["stream"]["roofline"]["frac"], "per_view", d["per_view"]["ms_per_step"])
kb = d.get("kernels_breakdown_pass")
if kb:
    print({k: round(v["avg_ms"] * 1e3, 1) for k, v in kb.items() if isinstance(v, dict) and "avg_ms" in v})
if "traffic_per_kernel" in r:
    print({n: round((x.get("FETCH_SIZE_bytes_per_batch", 0) + x.get("WRITE_SIZE_bytes_per_batch", 0)) / 1e6, 1)
           for n, x in r["traffic_per_kernel"].items()})
