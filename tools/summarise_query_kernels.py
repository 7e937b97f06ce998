#!/usr/bin/env python3
"""gpurun_out/TAG_query (tools/profile_query_kernels.sh) -> profiles/TAG_query_kernels.json: per kernel the rocprofv3 --stats figures (calls, average / longest ms) and, from
the largest dispatch of the two SQ counter passes, lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU), the share of wave time spent waiting
(SQ_WAIT_ANY / SQ_WAVE_CYCLES), wavefronts and VALU instructions."""
import collections, csv, glob, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(REPO, "gpurun_out", tag + "_query")
out = collections.defaultdict(dict)
def short(k): return k.split("(")[0].replace("void ", "").strip()
for f in glob.glob(src + "/trace_q/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"]); out[k].update(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) * 1e-6, max_ms=float(r["MaxNs"]) * 1e-6)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_q1", "pmc_q2"):
    for f in glob.glob(src + "/" + d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)): per[(short(r["Kernel_Name"]), r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), v in per.items():
            for c, x in v.items(): pmc[k][c].append(x)
for k, v in pmc.items():
    m = {c: max(x) for c, x in v.items()}
    if "SQ_ACTIVE_INST_VALU" in m and m["SQ_ACTIVE_INST_VALU"] > 0: out[k]["lane_utilisation"] = m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0: out[k]["wait_fraction"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"):
        if c in m: out[k][c + "_largest_dispatch"] = m[c]
keep = {k: v for k, v in out.items() if any(s in k for s in ("find_nearest", "whitted"))}
txt = os.path.join(src, "other_kernels.txt")
res = {"source": "tools/profile_query_kernels.sh %s: rocprofv3 --kernel-trace --stats and two --pmc passes of tools/other_kernels.py (2^20 rays per query call)" % tag, "kernels": keep,
       "host_timings": open(txt).read().strip() if os.path.exists(txt) else None}
json.dump(res, open(os.path.join(REPO, "profiles", tag + "_query_kernels.json"), "w"), indent=1, sort_keys=True)
for k, v in sorted(keep.items()): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("calls", "avg_ms", "lane_utilisation", "wait_fraction")})
