"""files_to_volume of a fresh process, deferred set-up against all-in-constructor, alternating (the parent never touches the GPU)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from plant3dvision_amd import scenes
a = bench.parse()
g, o, v, views = bench.cached_scene(a, scenes, bench.global_shape(a.n, 1))
rows = []
for rep in range(int(os.environ.get("REPS", "5"))):
    r = bench.cold_files(a, views, g, o, v)
    rows.append({k: {x: round(y, 1) for x, y in r[k].items() if isinstance(y, float)} for k in r})
    print(json.dumps(rows[-1]), flush=True)
import statistics
for k in ("deferred", "all_in_constructor"):
    print(k, "median files_to_volume_ms", statistics.median(r[k]["files_to_volume_ms"] for r in rows))
