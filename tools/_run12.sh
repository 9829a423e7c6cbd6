cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --graph 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
cd $GRAFT_REPO_ROOT
timeout 200 python tools/diag_noise.py hr32_k5_full eval > gpurun_out/diag_noise_eval.log 2>&1
ls -la gpurun_out/prof/*/ | head; cut -c1-300 gpurun_out/bench_prof.json
