cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for q in 1 2 3 4 6; do
  echo "HWQ=$q"
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c88-200
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --graph 0 2>/dev/null| cut -c88-200
done > gpurun_out/hwq_sweep.log 2>&1
echo "default"; timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --graph 0 2>/dev/null | cut -c88-200 >> gpurun_out/hwq_sweep.log
grep -v amdgpu gpurun_out/hwq_sweep.log | sed 's/"unit".*"ms_per_step"/ms/' | cut -c1-80
