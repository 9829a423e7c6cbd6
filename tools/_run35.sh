cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "mask_preprocessing or full_size or interchanges or stitch" > gpurun_out/t35.log 2>&1
tail -15 gpurun_out/t35.log
