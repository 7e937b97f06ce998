#!/bin/bash
# GPU box: rocprofv3 kernel stats + SQ counters of the query / Whitted kernels (tools/other_kernels.py: 2^20 rays through find_nearest_kernel on the bunny and the two-level
# scene, through the KD-tree and the grid; whitted_kernel on config 1), condensed by tools/summarise_query_kernels.py TAG -> profiles/TAG_query_kernels.json.
#   tools/profile_query_kernels.sh TAG
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_query
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_q -- python3 $GRAFT_REPO_ROOT/tools/other_kernels.py > $OUT/other_kernels.txt 2> $OUT/trace_q.log || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_q1 -- python3 $GRAFT_REPO_ROOT/tools/other_kernels.py > /dev/null 2> $OUT/pmc_q1.log || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_q2 -- python3 $GRAFT_REPO_ROOT/tools/other_kernels.py > /dev/null 2> $OUT/pmc_q2.log || exit 1
cd $GRAFT_REPO_ROOT && python3 tools/summarise_query_kernels.py $TAG
