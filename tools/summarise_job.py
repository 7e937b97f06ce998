#!/usr/bin/env python3
"""gpurun_out/TAG (tools/profile_job.sh) -> tracked summaries:
   profiles/TAG_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the bench.py command
   profiles/TAG_bench.json / TAG_bench_traced.json   the command's JSON lines (untraced / traced run)
   profiles/TAG_job_counters.json     PMC counters of the TIMED JOB's render kernels (every kernel of its launches: render_pool_kernel and, in a split job, the
                                      block-table render_tiles_kernel beside it), summed and per kernel
   profiles/job_counters.json         the same, keyed by job shape: what bench.py's roofline.traffic / valu_issue / useful_lane_issue_frac quote (counters_from)
The timed job's dispatches = the non-counting (COUNT = false) render kernels after the warm-up job's (one kernel: its launch measures the tile costs, nothing is split yet).
FETCH_SIZE / WRITE_SIZE: x 1024 bytes (the guide's unit); FETCH_SIZE is NOT x2-corrected here: the render kernels' reads are divergent 16-byte record fetches, not the
wide coalesced streaming pattern the guide's gfx950 correction was calibrated on (the accumulate kernel's is: stored x2)."""
import collections, csv, glob, json, os, shutil, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, "gpurun_out", tag); dst = os.path.join(REPO, "profiles")
base = tag.replace("/", "_")
for f in glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, os.path.join(dst, base + "_kernel_stats.csv"))
lines = {}
for name in ("bench.json", "bench_traced.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        ls = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if ls:
            open(os.path.join(dst, base + "_" + name), "w").write(ls[-1] + "\n"); lines[name] = json.loads(ls[-1])
b = lines.get("bench.json") or lines.get("bench_traced.json")
warm_kernels = 1
job = collections.defaultdict(lambda: collections.defaultdict(float)); acc = collections.defaultdict(float)
for d in sorted(glob.glob(src + "/pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].strip()
            per.setdefault((int(r["Dispatch_Id"]), k), collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
        render = [(did, k) for (did, k) in sorted(per) if ("render_pool_kernel" in k or "render_tiles_kernel" in k or "render_narrow_kernel" in k) and "true" not in k.split("<")[-1].split(",")[1:2][0]]
        for (did, k) in render[warm_kernels:]:
            for c, v in per[(did, k)].items(): job[k][c] += v
        for (did, k) in per:
            if "accumulate_kernel" in k:
                for c, v in per[(did, k)].items(): acc[c] = max(acc[c], v)
tot = collections.defaultdict(float)
for k in job:
    for c, v in job[k].items(): tot[c] += v
out = {"source": "rocprofv3 --pmc passes of `bench.py %s` (tools/profile_job.sh %s), counters of the timed job's render kernels" % (open(os.path.join(src, "args.txt")).read().strip() if os.path.exists(os.path.join(src, "args.txt")) else "", tag),
       "kernels": {k: dict(v) for k, v in job.items()}, "total": dict(tot)}
if "FETCH_SIZE" in tot: out["fetch_bytes_raw"] = tot["FETCH_SIZE"] * 1024
if "WRITE_SIZE" in tot: out["write_bytes"] = tot["WRITE_SIZE"] * 1024
if "SQ_INSTS_VALU" in tot: out["valu_wave_instructions"] = tot["SQ_INSTS_VALU"]; out["salu_wave_instructions"] = tot.get("SQ_INSTS_SALU", 0)
if tot.get("SQ_ACTIVE_INST_VALU"): out["lane_utilisation"] = tot["SQ_THREAD_CYCLES_VALU"] / (64 * tot["SQ_ACTIVE_INST_VALU"])
if tot.get("TCC_HIT_sum"): out["l2_hit_rate"] = tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"])
if acc.get("FETCH_SIZE"): out["accumulate_kernel_fetch_bytes_x2"] = 2 * acc["FETCH_SIZE"] * 1024
if b:
    cfg = b.get("config", {}); out["bench_value"] = b.get("value"); out["rays_per_step"] = cfg.get("rays_per_step_rank0")
    key = b.get("job_shape")
    out["job_shape"] = key
    json.dump(out, open(os.path.join(dst, base + "_job_counters.json"), "w"), indent=1)
    if key:
        p = os.path.join(dst, "job_counters.json")
        allc = json.load(open(p)) if os.path.exists(p) else {}
        allc[key] = dict(out, file="profiles/%s_job_counters.json" % base)
        json.dump(allc, open(p, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if k not in ("kernels", "total")}, indent=1))
