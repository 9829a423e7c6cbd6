cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "2 512" "4 512" "4 256" "2 768" "2 1024" "8 256" "3 512"; do set -- $cfg; echo "TPB=$1 BLOCKS=$2"; BPB_WGRAD_TPB=$1 BPB_WGRAD_BLOCKS=$2 timeout 100 python tools/conv_bench.py 2>/dev/null | cut -c1-62; done > gpurun_out/wgrad_sweep.log 2>&1
