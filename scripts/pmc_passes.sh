#!/bin/bash
# the separate rocprofv3 --pmc passes behind profiles/r02_pmc.json (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-secondary"
O=gpurun_out/pmc2; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O -o sq -- $B > $O/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write -- $B > $O/write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o calfetch -- python scripts/pmc_calibrate.py > $O/calfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o calwrite -- python scripts/pmc_calibrate.py > $O/calwrite.log 2>&1
for t in sq fetch write calfetch calwrite; do python profiles/rocpd_stats.py --pmc $O/${t}_results.db "gemm_pp_kernel,adamw_k,sumsq,swiglu" > $O/$t.txt 2>&1; done
rm -f $O/*.db
head -12 $O/sq.txt; head -6 $O/fetch.txt; head -6 $O/write.txt; cat $O/calfetch.txt | head -8; cat $O/calwrite.txt | head -8; cat $O/calfetch.log | tail -3
