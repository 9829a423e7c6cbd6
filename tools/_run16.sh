cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
export CONV_PMC_ONLY=0
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc16a -- python $GRAFT_REPO_ROOT/tools/conv_pmc.py 64 32 32 32 3 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc16a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc16b -- python $GRAFT_REPO_ROOT/tools/conv_pmc.py 64 32 32 32 3 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc16b.log 2>&1
cd $GRAFT_REPO_ROOT
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE; do echo $c; python tools/pmc_summary.py gpurun_out/pmc16a $c | grep conv_igemm; done > gpurun_out/pmc16a_summary.log 2>&1
for c in SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT; do echo $c; python tools/pmc_summary.py gpurun_out/pmc16b $c | grep conv_igemm; done > gpurun_out/pmc16b_summary.log 2>&1
cat gpurun_out/pmc16a_summary.log gpurun_out/pmc16b_summary.log; tail -3 gpurun_out/pmc16a.log gpurun_out/pmc16b.log
find gpurun_out/pmc16a gpurun_out/pmc16b -size +2M -delete
