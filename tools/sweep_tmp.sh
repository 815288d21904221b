R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for cfg in "--scene solid --steps 5 --warmup 1" "--steps 20 --warmup 3"; do
  echo "== $cfg"; python $R/bench.py --cpu-seconds 0 $cfg | python -c "import json,sys; d=json.load(sys.stdin); print('fused', round(d['ms_per_step'],4), '%.4g'%d['value'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()}, '| stream', round(d['stream']['ms_per_step'],3), '%.4g'%d['stream']['value'], round(d['stream']['roofline']['frac'],3))"
done
