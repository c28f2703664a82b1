"""Per-kernel duration AND gap to the previous kernel's end from a rocprofv3 kernel_trace.csv (one stream, graph replays):
separates launch-boundary time from kernel bodies.  usage: trace_gaps.py <dir> [skip_first_n]"""
import csv, glob, sys, collections
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[skip:]
st = collections.defaultdict(lambda: [0, 0, 0, 0])
prev_end = None
for s, e, n in rows:
    k = n.split("(")[0][-60:]
    q = st[k]
    q[0] += 1
    q[1] += e - s
    if prev_end is not None and 0 <= s - prev_end < 200000:
        q[2] += s - prev_end
        q[3] += 1
    prev_end = e
print(f"{'kernel':62s} {'calls':>6s} {'avg_dur_us':>10s} {'avg_gap_us':>10s}")
tot_d = tot_g = 0
for k, (c, dur, gap, gc) in sorted(st.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:62s} {c:6d} {dur / c / 1e3:10.2f} {gap / max(gc, 1) / 1e3:10.2f}")
    tot_d += dur; tot_g += gap
print(f"total kernel time {tot_d / 1e6:.3f} ms, total gaps {tot_g / 1e6:.3f} ms, span {(rows[-1][1] - rows[0][0]) / 1e6:.3f} ms")
