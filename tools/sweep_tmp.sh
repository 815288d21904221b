R=$GRAFT_REPO_ROOT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --share-device > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err; echo rc=$?; tail -3 gpurun_out/bench_2rank.err; cat gpurun_out/bench_2rank.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['global_grid'], d['config']['slab_per_gpu'], d['scaling'])"
