cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof1; cd /tmp
BPB_SINGLE_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r01s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof1.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof1.err
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py gpurun_out/prof1/r01s_results.db gpurun_out/r01s_kernel_stats.csv
cut -c1-200 gpurun_out/bench_prof1.json; head -5 gpurun_out/r01s_kernel_stats.csv | cut -c1-120
find gpurun_out/prof1 -size +8M -delete
