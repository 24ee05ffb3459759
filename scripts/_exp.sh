cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_r2a.json 2> gpurun_out/r2/bench_r2a.err
tail -2 gpurun_out/r2/bench_r2a.err
