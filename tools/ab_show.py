"""One line per run of tools/ab.sh / ab_variants.sh output: python tools/ab_show.py [file.jsonl] (or on stdin)."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for l in src:
    l = l.strip()
    if not l.startswith('{'):
        continue
    d = json.loads(l)
    print(d['tag'], d.get('opts'), ' '.join(f"{k}={d[k]['ms']:.4f}" for k in d if isinstance(d[k], dict) and 'ms' in d[k]))
