#!/bin/bash
# GPU box: SQ counters of one pool-only job with render_pool_kernel and with render_duo_kernel (CRT_POOL_DUO=1).   tools/duo_pmc.sh TAG [K]
TAG=$1; K=${2:-32}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in pool duo; do
  if [ $v = duo ]; then export CRT_POOL_DUO=1; else unset CRT_POOL_DUO; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/$v.1 -- python3 $GRAFT_REPO_ROOT/tools/pool_job.py $K > $OUT/$v.json 2> $OUT/$v.1.log || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/$v.2 -- python3 $GRAFT_REPO_ROOT/tools/pool_job.py $K > /dev/null 2> $OUT/$v.2.log || exit 1
done
python3 - <<PY
import csv, glob, collections, json
for v in ("pool", "duo"):
    m = collections.defaultdict(float)
    for d in ("1", "2"):
        for f in glob.glob("$OUT/%s.%s/**/*counter_collection.csv" % (v, d), recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(f)):
                if "render_pool_kernel" in r["Kernel_Name"] or "render_duo_kernel" in r["Kernel_Name"]: per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            last = per[sorted(per, key=int)[-1]] if per else {}
            for c, x in last.items(): m[c] = x
    t = json.load(open("$OUT/%s.json" % v)); K = t["windows"]
    print(v, "job %.1f ms" % t["job_ms"], {c: round(x / K / 1e6, 1) for c, x in sorted(m.items())}, "lanes %.3f" % (m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]) if m.get("SQ_ACTIVE_INST_VALU") else 0))
PY
