#!/bin/bash
# GPU box: per library variant of build/variants (or the names given), one untraced pool-only job (time) + two SQ counter passes of the same job
# (instruction counts and lane utilisation of render_pool_kernel), summarised per 64-frame window into gpurun_out/$TAG/variants.json.
#   tools/variant_pmc.sh TAG K scene kind W H [variant names...]
TAG=$1; K=${2:-32}; SCENE=${3:-bunny_scene.xml}; KIND=${4:-0}; W=${5:-1280}; H=${6:-720}; shift 6
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
NAMES="$@"; if [ -z "$NAMES" ]; then NAMES=$(ls $ROOT/build/variants | sed -n 's/^libcrt_\(.*\)\.so$/\1/p'); fi
cd /tmp && export TMPDIR=/tmp
export CRT_RENDER_KERNEL=pool_always CRT_SPLIT_OFF=1
for n in $NAMES; do
  export CRT_LIB_PATH=$ROOT/build/variants/libcrt_$n.so
  python3 $ROOT/tools/pool_job.py $K $SCENE $KIND $W $H > $OUT/$n.time.json 2> $OUT/$n.time.err || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/$n.pmc1 -- python3 $ROOT/tools/pool_job.py $K $SCENE $KIND $W $H > /dev/null 2> $OUT/$n.pmc1.log || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/$n.pmc2 -- python3 $ROOT/tools/pool_job.py $K $SCENE $KIND $W $H > /dev/null 2> $OUT/$n.pmc2.log || exit 1
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for n in "$NAMES".split():
    t = json.load(open("$OUT/%s.time.json" % n)); K = t["windows"]
    acc = collections.defaultdict(list)
    for d in ("pmc1", "pmc2"):
        for f in glob.glob("$OUT/%s.%s/**/*counter_collection.csv" % (n, d), recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(f)):
                if "render_pool_kernel" in r["Kernel_Name"]: per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            for disp in per.values():
                for c, v in disp.items(): acc[c].append(v)
    m = {c: sum(v) / len(v) for c, v in acc.items()}
    rays = t["rays_per_window"]
    o = dict(t)
    if m:
        o.update({"valu_per_window_M": round(m["SQ_INSTS_VALU"] / K / 1e6, 1), "salu_per_window_M": round(m["SQ_INSTS_SALU"] / K / 1e6, 1), "lds_per_window_M": round(m["SQ_INSTS_LDS"] / K / 1e6, 1),
                  "vmem_per_window_M": round((m["SQ_INSTS_VMEM_RD"] + m["SQ_INSTS_VMEM_WR"]) / K / 1e6, 1), "smem_per_window_M": round(m["SQ_INSTS_SMEM"] / K / 1e6, 1), "branch_per_window_M": round(m.get("SQ_INSTS_BRANCH", 0) / K / 1e6, 1),
                  "valu_per_ray": round(m["SQ_INSTS_VALU"] / K / rays, 3), "all_instr_per_ray": round((m["SQ_INSTS_VALU"] + m["SQ_INSTS_SALU"] + m["SQ_INSTS_LDS"] + m["SQ_INSTS_VMEM_RD"] + m["SQ_INSTS_VMEM_WR"] + m["SQ_INSTS_SMEM"] + m.get("SQ_INSTS_BRANCH", 0)) / K / rays, 3),
                  "lane_utilisation": round(m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]), 4), "waves": round(m["SQ_WAVES"])})
    out[n] = o
    print(n, json.dumps(o))
json.dump(out, open("$OUT/variants.json", "w"), indent=1)
PY
