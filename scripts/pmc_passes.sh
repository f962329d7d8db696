#!/bin/bash
# the separate rocprofv3 --pmc passes behind profiles/rNN_pmc.json (run on the GPU box from the repo root):
#   bash scripts/pmc_passes.sh && python scripts/pmc_json.py gpurun_out/pmc_rNN > gpurun_out/rNN_pmc.json
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-recipe --no-dp-emulation"   # (counter passes serialise the kernels: with the sharded-step emulation legs in, three passes did not fit a 30-minute GPU call)
R_=${PMC_ROUND:-r04}; O=gpurun_out/pmc_$R_; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O -o sq -- $B > $O/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write -- $B > $O/write.log 2>&1
for t in sq fetch write; do python profiles/rocpd_stats.py --pmc $O/${t}_results.db "gemm_pp_kernel,adamw_k,sumsq,swiglu" > $O/$t.txt 2>&1; done
rm -f $O/*.db
python scripts/pmc_json.py $O > gpurun_out/${R_}_pmc.json; for t in sq fetch write; do cp $O/$t.txt gpurun_out/${R_}_pmc_$t.txt; done
head -8 $O/sq.txt; head -6 $O/fetch.txt; head -6 $O/write.txt; head -30 gpurun_out/${R_}_pmc.json
