#!/bin/bash
# Round-5 profile of the bench command with per-step launches (--sequence 0: the steady-state configuration; counter passes serialise kernels): rocprofv3 kernel stats + separate --pmc passes (never combined with the
# sys / hip / hsa trace domains).  Run on the GPU box from the repo root:  bash profiles/tools/pmc_passes_r05.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${PMC_OUT:-r5_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --cpu-tracks 0 --no-gp --sequence 0"
: > $O/exits.txt
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/bench.py $ARGS > $O/$name.log 2>&1; rc=$?; echo "$name exit $rc" >> $O/exits.txt; return $rc; }
pass stats --stats && pass fetch --pmc FETCH_SIZE && pass write --pmc WRITE_SIZE \
 && pass mix --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM \
 && pass busy --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
 && pass f64 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 \
 && pass tcc --pmc TCC_HIT_sum TCC_MISS_sum
cat $O/exits.txt
# the same command without a profiler, and at the driver's flags
cd $R
python3 bench.py --steps 100 --warmup 10 --no-gp > $O/bench_k100.json 2> $O/bench_k100.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
# the driver's form runs its forward passes as scheduled launches (7 + 13 steps) (bench.py --sequence auto): its own kernel stats
cd /tmp
# (--sequence-form single: the form the untimed comparison of --sequence auto usually ends on, pinned so that the stats show one form)
rm -rf $O/stats_driver_form
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_driver_form -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-tracks 0 --no-gp --no-fleet --sequence-form single > $O/stats_driver_form.log 2>&1; echo "stats_driver_form exit $?" >> $O/exits.txt
cd $R
