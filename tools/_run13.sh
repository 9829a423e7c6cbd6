cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" > gpurun_out/t13.log 2>&1
tail -5 gpurun_out/t13.log
timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench13.log 2>&1
BPB_MULTI_TILE=0 timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench13_single.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 --graph 0 --no-cpu-baseline > gpurun_out/bench13.json 2> gpurun_out/bench13.err
BPB_MULTI_TILE=0 timeout 300 python bench.py --steps 8 --warmup 3 --graph 0 --no-cpu-baseline > gpurun_out/bench13_single.json 2> gpurun_out/bench13_single.err
cut -c1-330 gpurun_out/bench13.json; cut -c1-330 gpurun_out/bench13_single.json
