#!/bin/bash
# usage (on the GPU box): tools/sweep_opts.sh <outdir> "<label>:<bench args>" ...   -> one JSON per label
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  label=${spec%%:*}; args=${spec#*:}
  python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --extra-steps 0 --skip-other-path $args > "$out/$label.json" 2> "$out/$label.err" || echo "$label failed"
done
