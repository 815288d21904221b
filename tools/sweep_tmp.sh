R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "" "--opt 9=4" "--opt 9=16" "--opt 9=62" "--opt 7=4" "--opt 7=16" "--opt 7=70" "--opt 7=2"; do
  echo "== $cfg"; python $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --skip-other-path $cfg | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
