#!/usr/bin/env python3
"""Prints the few numbers of a bench.py JSON line one usually wants (stdin or file argument)."""
import json, sys
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
for l in src:
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("steps %d: %.0f Mrays/s, %.3f ms/step | %s %.1f ms job launch, accumulate %.2f ms/step, frac %.3f, valu %.3f | single %s | cpu %s" % (
            d["steps"], d["value"], d["ms_per_step"], r["kernel"], r["job_launch_ms"], r["accumulate_kernel_ms_per_step"], r["frac"], r["valu_issue"]["frac"],
            d["single_render"] and d["single_render"]["ms"], d.get("cpu_baseline") and d["cpu_baseline"]["value"]))
    elif l.startswith("[crt]"): print(l.rstrip())
