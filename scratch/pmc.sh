cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/quick_bench.py 65536,1,-1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/quick_bench.py 65536,1,-1 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc1 $GRAFT_REPO_ROOT/gpurun_out/pmc2
