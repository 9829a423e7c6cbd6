cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/fwd_bench.py > gpurun_out/fwd_bench.json 2>/dev/null; cat gpurun_out/fwd_bench.json
timeout 200 python tools/fwd_bench.py resnet50 5 256 128 64 2>/dev/null | tee gpurun_out/fwd_bench_r50.json
