#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/pmc_w4
cd /tmp && export TMPDIR=/tmp; cd $R
O=gpurun_out/pmc_w4
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d $O -o a -- python scripts/w4_check.py pmc > $O/a.log 2>&1
python profiles/rocpd_stats.py --pmc $O/a_results.db "gemm_pp_kernel,gemm_w4_kernel,Cijk" > gpurun_out/r04_pmc_w4_a.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD --kernel-trace -d $O -o b -- python scripts/w4_check.py pmc > $O/b.log 2>&1
python profiles/rocpd_stats.py --pmc $O/b_results.db "gemm_pp_kernel,gemm_w4_kernel,Cijk" > gpurun_out/r04_pmc_w4_b.txt 2>&1
rm -f $O/*.db
cut -c1-40,90-160 gpurun_out/r04_pmc_w4_a.txt | head -30; cut -c1-40,90-160 gpurun_out/r04_pmc_w4_b.txt | head -30
