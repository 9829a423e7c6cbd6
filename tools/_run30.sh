cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x 2>&1 | tail -6
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c88-200
BPB_FUSE_FINALIZE=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c88-200
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --graph 1 2>/dev/null | cut -c88-200
