cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or head or model or backbone" > gpurun_out/t18.log 2>&1
tail -3 gpurun_out/t18.log
timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench18.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench18.json 2> gpurun_out/bench18.err
cut -c1-330 gpurun_out/bench18.json
