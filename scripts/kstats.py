#!/usr/bin/env python3
"""Average kernel times of a rocprofv3 --kernel-trace --stats output directory: python3 scripts/kstats.py DIR [pattern]"""
import csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Name"]:
            print(f'{row["Name"].split("(")[0][:48]:48s} calls={row["Calls"]:>4s} avg_ms={float(row["AverageNs"]) / 1e6:9.3f} total_ms={float(row["TotalDurationNs"]) / 1e6:10.2f}')
