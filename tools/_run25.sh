cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "distance or rank" > gpurun_out/t25.log 2>&1
tail -4 gpurun_out/t25.log
timeout 300 python tools/eval_bench.py > gpurun_out/eval_bench.json 2> gpurun_out/eval_bench.err
cat gpurun_out/eval_bench.json; tail -2 gpurun_out/eval_bench.err
