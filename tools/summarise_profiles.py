#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
   profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of `bench.py`
   profiles/<tag>_pmc_summary.json     per-kernel mean PMC values per dispatch
   profiles/<tag>_bench.json           bench.py's JSON line of the same run
   profiles/hbm_traffic.json           HBM bytes / VALU instructions of the job's render kernel launch, per 64-frame window (bench.py's roofline.traffic / valu_issue)
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B... the guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, and on gfx950
FETCH_SIZE under-reports wide coalesced streaming reads by 2x; this kernel's reads are 16-byte-per-lane record fetches with
divergent addresses (not a calibrated pattern), so both the raw and the x2-corrected figure are stored."""
import collections, csv, glob, json, os, shutil, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(REPO, "gpurun_out", tag)
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
pmc = collections.defaultdict(dict)
for d in sorted(glob.glob(src + "/pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, vals in v.items():
                pmc[k][c] = dict(mean_per_dispatch=sum(vals) / len(vals), max_dispatch=max(vals), dispatches=len(vals))
json.dump(pmc, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
for name in ("bench.json", "bench_traced.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, tag + "_" + name), "w").write(lines[-1] + "\n")
# the job's render kernel: the non-counting variant (template argument COUNT = false) with the largest dispatch
cands = [k for k in pmc if ("render_pool_kernel" in k or "render_tiles_kernel" in k) and "false" in k and "SQ_INSTS_VALU" in pmc[k]]
rk = sorted(cands, key=lambda k: -pmc[k]["SQ_INSTS_VALU"]["max_dispatch"])[:1]
WINDOWS = int(sys.argv[2]) if len(sys.argv) > 2 else 64      # 64-frame windows of bench.py's job launch (its --steps)
if rk and "FETCH_SIZE" in pmc[rk[0]] and "WRITE_SIZE" in pmc[rk[0]]:
    # bench.py's job is ONE launch over all its windows; its warm-up and the single-render probes are smaller launches,
    # so the job's launch is the largest dispatch.  Everything is stored PER WINDOW so that bench.py can scale it to any --steps.
    k = rk[0]
    f = pmc[k]["FETCH_SIZE"]["max_dispatch"] * 1024
    w = pmc[k]["WRITE_SIZE"]["max_dispatch"] * 1024
    per = dict(fetch_bytes_raw=f / WINDOWS, write_bytes=w / WINDOWS,
               valu_wave_instructions=pmc[k]["SQ_INSTS_VALU"]["max_dispatch"] / WINDOWS, salu_wave_instructions=pmc[k]["SQ_INSTS_SALU"]["max_dispatch"] / WINDOWS)
    out = dict(kernel=k, windows_in_measured_launch=WINDOWS, per_window=per,
               note="FETCH_SIZE*1024 and WRITE_SIZE*1024 of the job's launch (largest dispatch) / its windows; FETCH_SIZE not x2-corrected because the record fetches are divergent 16-B loads, not the calibrated wide streaming pattern",
               source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_*, separate passes, tools/collect_profiles.sh " + tag)
    if "TCC_HIT_sum" in pmc[k]:
        h, m = pmc[k]["TCC_HIT_sum"]["max_dispatch"], pmc[k]["TCC_MISS_sum"]["max_dispatch"]
        out["l2_hit_rate"] = h / (h + m)
    if "SQ_THREAD_CYCLES_VALU" in pmc[k]:
        out["lane_utilisation"] = pmc[k]["SQ_THREAD_CYCLES_VALU"]["max_dispatch"] / (64 * pmc[k]["SQ_ACTIVE_INST_VALU"]["max_dispatch"])
    ak = [x for x in pmc if "accumulate_kernel" in x]
    if ak and "FETCH_SIZE" in pmc[ak[0]]:
        out["accumulate_kernel_fetch_bytes_x2_per_window"] = 2 * pmc[ak[0]]["FETCH_SIZE"]["max_dispatch"] * 1024 / WINDOWS   # wide coalesced streaming reads: the guide's x2 correction applies
        out["accumulate_kernel_write_bytes_per_window"] = pmc[ak[0]]["WRITE_SIZE"]["max_dispatch"] * 1024 / WINDOWS
    out["workload"] = ["bunny_scene.xml", 0, 1280, 720, 64]      # bench.py defaults: scene, kind, W, H, spp per step
    json.dump(out, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    print(out)
print("profiles written for", tag)
