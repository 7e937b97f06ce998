#!/bin/bash
# GPU box: SQ counter passes for one bench.py configuration (args after the tag go to bench.py).  usage: tools/pmc_quick.sh TAG [bench args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/b1.json 2> $OUT/pmc_sq1.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/b2.json 2> $OUT/pmc_sq2.log
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LEVEL_WAVES --output-format csv -d $OUT/pmc_sq3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/b3.json 2> $OUT/pmc_sq3.log
python3 - <<PY
import csv, glob, collections
for d in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "render" not in k or "true" in k: continue
            print(k, {c: "%.4g" % x for c, x in v.items()})
PY
