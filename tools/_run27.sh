cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python tools/conv_sweep.py 2>&1 | cut -c1-120 > gpurun_out/conv_sweep2.log
cat gpurun_out/conv_sweep2.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or backbone" 2>&1 | tail -2
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c88-200
