#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats CSV: name (short), calls, average us, percentage."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    name = r["Name"].split("(")[0][-48:]
    print("%-48s %6s %10.1f us %6s %%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
