R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/microbench_avg.py
