cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc32a -- python $GRAFT_REPO_ROOT/tools/wgrad_pmc.py 64 32 32 32 3 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc32a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc32b -- python $GRAFT_REPO_ROOT/tools/wgrad_pmc.py 64 32 32 32 3 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc32b.log 2>&1
cd $GRAFT_REPO_ROOT
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE; do echo -n "$c "; python tools/pmc_summary.py gpurun_out/pmc32a $c | grep conv_wgrad | awk '{print $NF}'; done > gpurun_out/pmc32_summary.log 2>&1
for c in SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES; do echo -n "$c "; python tools/pmc_summary.py gpurun_out/pmc32b $c | grep conv_wgrad | awk '{print $NF}'; done >> gpurun_out/pmc32_summary.log 2>&1
cat gpurun_out/pmc32_summary.log; tail -1 gpurun_out/pmc32a.log
find gpurun_out/pmc32a gpurun_out/pmc32b -size +2M -delete
