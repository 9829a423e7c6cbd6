cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x > gpurun_out/t22.log 2>&1
tail -3 gpurun_out/t22.log
for cfg in "1 1 8" "0 1 8" "1 0 8" "0 0 8" "1 1 4"; do set -- $cfg
  echo "WGRAD_STREAMS=$1 INTERLEAVE=$2 HWQ=$3"
  BPB_WGRAD_STREAMS=$1 BPB_INTERLEAVE=$2 GPU_MAX_HW_QUEUES=$3 timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline | cut -c88-200
  BPB_WGRAD_STREAMS=$1 BPB_INTERLEAVE=$2 GPU_MAX_HW_QUEUES=$3 timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --graph 0 | cut -c88-200
done > gpurun_out/streams_sweep.log 2>&1
cat gpurun_out/streams_sweep.log
