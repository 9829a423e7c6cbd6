cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/diag_noise.py hr32_k5_full train > gpurun_out/diag_noise.log 2>&1
timeout 200 python -m pytest tests/test_gpu_kernels.py -q -x -k "full_backbones" > gpurun_out/t11.log 2>&1
timeout 200 python tools/conv_bench.py > gpurun_out/conv_bench11.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 --graph 0 --no-cpu-baseline > gpurun_out/bench11.json 2> gpurun_out/bench11.err
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --graph 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --graph 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/pmc_fetch FETCH_SIZE gpurun_out/pmc_fetch_summary.csv > gpurun_out/pmc_fetch_summary.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_write WRITE_SIZE gpurun_out/pmc_write_summary.csv > gpurun_out/pmc_write_summary.log 2>&1
find gpurun_out/pmc_fetch gpurun_out/pmc_write -size +4M -delete
tail -3 gpurun_out/diag_noise.log; tail -3 gpurun_out/t11.log; cat gpurun_out/bench11.json | cut -c1-400
