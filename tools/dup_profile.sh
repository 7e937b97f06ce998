#!/bin/bash
# Diagnostic: dynamic instruction cost of the SHADE phase's regions.  For every build/variants/libcrt_d*.so (built with
# -DCRT_DUP=n, see kernels.hip) collect SQ_INSTS_VALU / SQ_INSTS_SALU of the render kernel; the difference against d0 is region n.
OUT=$GRAFT_REPO_ROOT/gpurun_out/dup
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in $GRAFT_REPO_ROOT/build/variants/libcrt_d*.so; do
  n=$(basename $f .so)
  CRT_LIB_PATH=$f rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $OUT/$n -- python3 $GRAFT_REPO_ROOT/tools/dup_profile.py "$@" > /dev/null 2> $OUT/$n.log
  echo "$n $(python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$n | grep render_tiles)"
done
