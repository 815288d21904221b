#!/bin/bash
# Runs on the GPU box: where the -1 fill should ride after the survivor stages lost a fifth of their vector instructions.
# bash tools/r04_share_sweep.sh "<scenes>" "<opt set>" ...   (an opt set: "15=12 17=4", "-" = defaults)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
SC=$1; shift
for spec in "$@"; do
  args=""
  if [ "$spec" != "-" ]; then for kv in $spec; do args="$args --opt $kv"; done; fi
  python3 tools/bench_scenes.py --steps 40 --scenes $SC --tag "$spec" $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['tag'].ljust(28), ' '.join(f\"{k} {v['ms']:.4f}\" for k,v in d.items() if isinstance(v,dict)))"
done
