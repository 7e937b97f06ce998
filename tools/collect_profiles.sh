#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + the HBM-traffic PMC passes for bench.py's workload.
# Counters are collected in their OWN passes (no trace domains mixed with --pmc), FETCH_SIZE and WRITE_SIZE separately
# (TCC slot budget), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Outputs land in gpurun_out/ and are then
# summarised into profiles/ by tools/summarise_profiles.py (run in the authoring container).
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_traced.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $GRAFT_REPO_ROOT/bench.py --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_l2.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq1.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq2.log
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.log
echo done
