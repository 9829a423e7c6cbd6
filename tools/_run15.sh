cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/conv_pmc.py 64 32 32 32 3 > gpurun_out/pmc15_b1.log 2>&1
timeout 200 python tools/conv_pmc.py 32 16 64 64 3 > gpurun_out/pmc15_b2.log 2>&1
timeout 200 python tools/conv_pmc.py 64 32 64 256 1 > gpurun_out/pmc15_1x1.log 2>&1
cat gpurun_out/pmc15_b1.log gpurun_out/pmc15_b2.log gpurun_out/pmc15_1x1.log
