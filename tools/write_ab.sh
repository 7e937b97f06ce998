#!/bin/bash
# GPU box: WRITE_SIZE (own PMC pass) of the render + accumulate kernels for each library build under build/variants/ (CRT_LIB_PATH selects the build)
OUT=$GRAFT_REPO_ROOT/gpurun_out/write_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in $GRAFT_REPO_ROOT/build/variants/*.so; do
  n=$(basename $f .so)
  export CRT_LIB_PATH=$f
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-single-render > $OUT/$n.json 2> $OUT/$n.log
  echo "== $n"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$n
done
