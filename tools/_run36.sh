cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python bench.py --backbone resnet50 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r50.json 2> gpurun_out/bench_r50.err
cut -c88-220 gpurun_out/bench_r50.json; tail -2 gpurun_out/bench_r50.err
timeout 400 python bench.py --backbone hrnet48 --parts 8 --height 384 --width 128 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_w48.json 2> gpurun_out/bench_w48.err
cut -c88-220 gpurun_out/bench_w48.json; tail -2 gpurun_out/bench_w48.err
