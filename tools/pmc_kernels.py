#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 PMC counters collected in separate passes (one directory per pass under <root>):
pmc_kernels.py <root> [kernel substring ...].  FETCH_SIZE is doubled per MI355X_MICROARCH.md's gfx950 correction
(128-byte requests tallied at 64 bytes); FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them."""
import csv
import glob
import sys
from collections import defaultdict

root, subs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").replace("(anonymous namespace)::", "").replace("void ", "")
        if subs and not any(s in k for s in subs):
            continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    parts = []
    for c, v in sorted(cs.items()):
        v = v[len(v) // 5:]
        m = sum(v) / len(v)
        if c == "FETCH_SIZE":
            parts.append(f"fetch {2 * m / 1024:.2f} MiB (2 x FETCH_SIZE)")
        elif c == "WRITE_SIZE":
            parts.append(f"write {m / 1024:.2f} MiB")
        else:
            parts.append(f"{c} {m:.0f}")
    h, ms = cs.get("TCC_HIT_sum"), cs.get("TCC_MISS_sum")
    if h and ms:
        hh, mm = sum(h) / len(h), sum(ms) / len(ms)
        parts.append(f"L2 hit rate {hh / (hh + mm):.3f}")
    print(f"{k:60s} n={len(next(iter(cs.values())))}: " + ", ".join(parts))
