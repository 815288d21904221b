R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python $R/bench.py --steps 20 --warmup 3 --cpu-seconds 0 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],4), d['value'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()}, 'stream', round(d['stream']['ms_per_step'],3), round(d['stream']['roofline']['frac'],3))"
