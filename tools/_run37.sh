cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or backbone" 2>&1 | tail -2
for cfg in "2 1" "2 0" "4 1"; do set -- $cfg; echo "NTW_MAX=$1 DMA1=$2"
BPB_WGRAD_NTW_MAX=$1 BPB_WGRAD_DMA1=$2 timeout 200 python bench.py --backbone resnet50 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c88-200
BPB_WGRAD_NTW_MAX=$1 BPB_WGRAD_DMA1=$2 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c88-200
done
timeout 200 python tools/conv_bench.py 2>/dev/null | cut -c1-62 | grep "k1"
