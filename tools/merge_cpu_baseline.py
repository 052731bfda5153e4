#!/usr/bin/env python
"""Carry the `cpu_baseline` block of an earlier bench line of the same workload over into a newer one that was run with
--no-cpu-baseline (the CPU oracle's speed does not depend on the GPU kernels; re-running it costs ~200 s of GPU-box time per
workload).  usage: merge_cpu_baseline.py <older.json> <newer.json> <out.json>"""
import json
import sys

old = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
new = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
assert old["config"]["workload"] == new["config"]["workload"], "different workloads"
if old.get("cpu_baseline") and not new.get("cpu_baseline"):
    new["cpu_baseline"] = old["cpu_baseline"]
    new["cpu_baseline_source"] = "carried over from an earlier round-3 run of this workload (host-only measurement, 64 threads of the GPU box's CPU)"
open(sys.argv[3], "w").write(json.dumps(new) + "\n")
