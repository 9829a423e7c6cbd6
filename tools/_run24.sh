cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/t24.log 2>&1
tail -4 gpurun_out/t24.log
timeout 300 python tools/eval_bench.py > gpurun_out/eval_bench.json 2> gpurun_out/eval_bench.err
cat gpurun_out/eval_bench.json; tail -2 gpurun_out/eval_bench.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench24.json 2> gpurun_out/bench24.err
cut -c1-330 gpurun_out/bench24.json
