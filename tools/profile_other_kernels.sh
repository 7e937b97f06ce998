#!/bin/bash
# GPU box: rocprofv3 kernel stats + SQ counters for the TLAS render job and for the query / Whitted kernels.  usage: tools/profile_other_kernels.sh TAG
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_other
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_q -- python3 $GRAFT_REPO_ROOT/tools/other_kernels.py > $OUT/other_kernels.txt 2> $OUT/trace_q.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_q1 -- python3 $GRAFT_REPO_ROOT/tools/other_kernels.py > /dev/null 2> $OUT/pmc_q1.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_q2 -- python3 $GRAFT_REPO_ROOT/tools/other_kernels.py > /dev/null 2> $OUT/pmc_q2.log
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-single-render --scene tlas_scene.xml --kind 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_t -- $B > $OUT/tlas_bench.json 2> $OUT/trace_t.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_t1 -- $B > /dev/null 2> $OUT/pmc_t1.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_t2 -- $B > /dev/null 2> $OUT/pmc_t2.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_t3 -- $B > /dev/null 2> $OUT/pmc_t3.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_t4 -- $B > /dev/null 2> $OUT/pmc_t4.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_t5 -- $B > /dev/null 2> $OUT/pmc_t5.log
echo done
