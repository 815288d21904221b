R=$GRAFT_REPO_ROOT
for v in 9 1 10 0 9 1 10; do
  echo "== var $v"; python $R/bench.py --path stream --steps 5 --warmup 1 --cpu-seconds 0 --skip-other-path --opt 99=$v | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['kernels']['carve']['avg_ms'],5))"
done
