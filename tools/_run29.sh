cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01g -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --graph 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/bench_profg.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_profg.err
cd $GRAFT_REPO_ROOT
python tools/timeline.py gpurun_out/prof/r01g_results.db 0.3 > gpurun_out/timeline_graph.log 2>&1
cut -c88-200 gpurun_out/bench_profg.json; cat gpurun_out/timeline_graph.log
find gpurun_out/prof -size +40M -delete
