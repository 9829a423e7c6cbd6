cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -k "model or backbone or module or full_size" 2>&1 | tail -3
timeout 200 python tools/fwd_bench.py 2>/dev/null | tee gpurun_out/fwd_bench.json
