#!/usr/bin/env python3
"""Concurrency of the kernels of a multi-stream bench run, from a rocprofv3 --kernel-trace database (rocpd sqlite):
how much of the busiest window had 0 / 1 / 2 / 3+ kernels resident, per-kernel mean durations in that window against their
single-stream durations, and the gaps between consecutive kernels of one stream (dependent-launch boundaries).
usage: timeline.py <results.db> [window_ms]      (window: the LAST window_ms of the trace = the timed steps; default 40)"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(con.execute(f"select name, start, end, {qcol or '0'}, grid_x / workgroup_x from kernels order by start"))
t_end = max(r[2] for r in rows)
t0 = t_end - int(win_ms * 1e6)
rows = [r for r in rows if r[1] >= t0]
print(f"{len(rows)} kernels in the last {win_ms} ms; queues: {sorted(set(r[3] for r in rows))}")
ev = []
for n, s, e, q, g in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
dur = defaultdict(float)
lvl, last = 0, ev[0][0]
for t, d in ev:
    dur[min(lvl, 4)] += t - last
    last = t; lvl += d
tot = sum(dur.values())
print("kernels resident: " + ", ".join(f"{k}{'+' if k == 4 else ''}: {100 * v / tot:.1f} %" for k, v in sorted(dur.items())))
by = defaultdict(list)
for n, s, e, q, g in rows:
    by[(n.split("(")[0].replace("void ", "")[:44], int(g))].append((e - s) / 1e3)
print("kernel (workgroups): calls, mean us, total ms")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"   {k[0]:44s} {k[1]:6d}: {len(v):5d} {sum(v) / len(v):8.2f} {sum(v) / 1e3:8.2f}")
print(f"sum of kernel durations {sum(sum(v) for v in by.values()) / 1e3:.2f} ms over a window of {tot / 1e6:.2f} ms")
# gaps inside one queue: end of a kernel -> start of the next kernel of the same queue
gaps = defaultdict(list)
prev = {}
for n, s, e, q, g in rows:
    if q in prev:
        gaps[q].append((s - prev[q]) / 1e3)
    prev[q] = max(e, prev.get(q, 0))
for q, v in gaps.items():
    v = sorted(v)
    pos = [x for x in v if x > 0]
    print(f"queue {q}: {len(v)} boundaries, gap p10/p50/p90 {v[len(v) // 10]:.2f}/{v[len(v) // 2]:.2f}/{v[9 * len(v) // 10]:.2f} us, sum of positive gaps {sum(pos) / 1e3:.2f} ms")
