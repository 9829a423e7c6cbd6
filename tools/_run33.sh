cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/t33.log 2>&1
tail -3 gpurun_out/t33.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke33.log 2>&1; tail -2 gpurun_out/smoke33.log
timeout 600 python bench.py > gpurun_out/bench33.json 2> gpurun_out/bench33.err
cut -c1-330 gpurun_out/bench33.json
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_hbm.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r01_pmc_hbm.json > gpurun_out/pmc_hbm.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof/r01_results.db gpurun_out/r01_kernel_stats.csv > gpurun_out/prof_summary.log 2>&1
python tools/timeline.py gpurun_out/prof/r01_results.db 0.4 > gpurun_out/timeline.log 2>&1
timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench33.log 2>&1
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof -size +8M -delete
head -6 gpurun_out/pmc_hbm.log; tail -2 gpurun_out/prof_summary.log
