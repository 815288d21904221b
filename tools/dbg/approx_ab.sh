set -e
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06
A=$R/build/libspacecarve_approx.so
O=gpurun_out/r06/approx_ab.txt; : > $O
for rep in 1 2; do
  python3 tools/bench_scenes.py --steps 30 --scenes plant,noise,dense,literal,solid --tag base 2>/dev/null | tail -1 >> $O
  SPACECARVE_LIB=$A python3 tools/bench_scenes.py --steps 30 --scenes plant,noise,dense,literal,solid --tag approx 2>/dev/null | tail -1 >> $O
done
python3 tools/bench_avg.py --reps 3 --tag base 2>/dev/null | tail -1 >> $O
SPACECARVE_LIB=$A python3 tools/bench_avg.py --reps 3 --tag approx 2>/dev/null | tail -1 >> $O
SPACECARVE_LIB=$A python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O
cat $O
