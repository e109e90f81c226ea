#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${PMC_OUT:-r2_pmc3}
mkdir -p $O
cd $R
python -m pytest tests/test_pipeline.py -m gpu -x -q > $O/pytest_pipeline.log 2>&1 || exit 1
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
python bench.py --tracks 12500 --cpu-tracks 0 > $O/bench_12500.json 2> $O/bench_12500.err || exit 1
python bench.py --force-dist --cpu-tracks 0 --tracks 12500 > $O/bench_12500_rccl1.json 2> $O/bench_12500_rccl1.err || exit 1
cd /tmp && export TMPDIR=/tmp
: > $O/exits.txt
pass() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-tracks 0 > $O/$name.log 2>&1; rc=$?; echo "$name exit $rc" >> $O/exits.txt; return $rc; }
pass stats --stats && pass fetch --pmc FETCH_SIZE && pass write --pmc WRITE_SIZE && pass mix --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SMEM && pass busy --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY && pass f64 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 && pass tcc --pmc TCC_HIT_sum TCC_MISS_sum
cat $O/exits.txt
