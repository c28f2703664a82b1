#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(7), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'].rjust(7))
