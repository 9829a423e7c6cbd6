cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python tools/conv_sweep.py > gpurun_out/conv_sweep.log 2>&1
cat gpurun_out/conv_sweep.log
