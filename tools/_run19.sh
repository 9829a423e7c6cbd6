cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or backbone" > gpurun_out/t19.log 2>&1
tail -3 gpurun_out/t19.log
timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench19.log 2>&1
BPB_WGRAD_DMA3=1 timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench19_dma3.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench19.json 2> gpurun_out/bench19.err
cut -c1-330 gpurun_out/bench19.json
