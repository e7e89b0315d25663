#!/bin/bash
# PMC passes over tools/prof_flash.py (fused attention kernels at the headline shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_flash
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python $R/tools/prof_flash.py > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- python $R/tools/prof_flash.py > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT --kernel-trace -d $OUT/p3 -o p3 --output-format csv -- python $R/tools/prof_flash.py > $OUT/p3.log 2>&1
tail -2 $OUT/p3.log
