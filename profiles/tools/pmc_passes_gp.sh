#!/bin/bash
# PMC passes over the GP objective at BASELINE configs[4] (1000 x 2000), program directly after `--`
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${PMC_OUT:-r4_pmc_gp}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/exits.txt
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/bench_gp.py --tracks 1000 --nobs 2000 --evals 2 --cpu-evals 0 > $O/$name.log 2>&1; rc=$?; echo "$name exit $rc" >> $O/exits.txt; return $rc; }
pass stats --stats && pass fetch --pmc FETCH_SIZE && pass write --pmc WRITE_SIZE && pass mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY && pass tcc --pmc TCC_HIT_sum TCC_MISS_sum
cat $O/exits.txt
if [ -n "$GP_FIT" ]; then cd $R && python bench_gp.py --nobs 2000 --fit-tracks 64 --fit-restarts 15 > $O/fit_restarts.json 2> $O/fit_restarts.err; fi
[ -n "$GP_FIT" ] && tail -c 1500 $O/fit_restarts.json || true
