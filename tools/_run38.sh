cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "two_rank" > gpurun_out/t38.log 2>&1
tail -30 gpurun_out/t38.log
