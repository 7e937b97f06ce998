#!/bin/bash
# GPU box: rocprofv3 kernel trace + the PMC passes for ONE bench.py command (the arguments after the tag), every pass in its own run (no trace domains mixed
# with --pmc; FETCH_SIZE and WRITE_SIZE separately), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Outputs under gpurun_out/TAG; tools/summarise_job.py
# condenses them into profiles/ (tracked).      usage: tools/profile_job.sh TAG [bench.py arguments]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B "$@" > $OUT/bench_traced.json 2> $OUT/trace.log || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B "$@" --no-single-render --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.log || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B "$@" --no-single-render --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.log || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $B "$@" --no-single-render --no-cpu-baseline > /dev/null 2> $OUT/pmc_l2.log || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 $B "$@" --no-single-render --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq1.log || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- python3 $B "$@" --no-single-render --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq2.log || exit 1
python3 $B "$@" > $OUT/bench.json 2> $OUT/bench.log || exit 1
echo "$@" > $OUT/args.txt
echo done $TAG
