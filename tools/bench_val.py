#!/usr/bin/env python3
"""stdin: bench.py's JSON line -> value, ms per step, single render (short)"""
import json, sys
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print(d["config"].get("workload", "")[:60], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms/step |", (d.get("single_render") or {}).get("ms"), (d.get("single_render") or {}).get("first_ms"))
