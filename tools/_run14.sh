cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv" > gpurun_out/t14.log 2>&1
tail -5 gpurun_out/t14.log
timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench14.log 2>&1
BPB_MULTI_TILE=0 timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench14_single.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 --graph 0 --no-cpu-baseline > gpurun_out/bench14.json 2> gpurun_out/bench14.err
BPB_MULTI_TILE=0 timeout 300 python bench.py --steps 8 --warmup 3 --graph 0 --no-cpu-baseline > gpurun_out/bench14_single.json 2> gpurun_out/bench14_single.err
cut -c1-330 gpurun_out/bench14.json; cut -c1-330 gpurun_out/bench14_single.json
