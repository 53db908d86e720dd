#!/usr/bin/env python3
"""Print selected fields of a bench.py JSON line read from stdin: tools/jline.py [label]"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "%.0f sims/s  %.3f ms/step  trunk %.1f us  %.1f TF/s  errors %s" % (
    d["value"], d["ms_per_step"], r.get("us_per_launch", float("nan")), r["achieved"], d["config"].get("trees_with_error_status")))
