#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results.db: per-kernel calls / total / average over the
last `window` seconds of the trace (the timed bench steps), plus GPU busy fraction."""
import collections, sqlite3, sys
db, win_s, step_ms = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
tend = rows[-1][2]; win = win_s * 1e9; steps = win / (step_ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0]); busy = 0
for n, s, e in rows:
    if s > tend - win:
        k = n.split('(')[0]
        k = k.replace('_ZN2dh', '').split('EEv')[0][:44]
        agg[k][0] += 1; agg[k][1] += e - s; busy += e - s
print(f"window {win_s}s = {steps:.2f} steps of {step_ms} ms; kernels/step {sum(v[0] for v in agg.values())/steps:.0f}; busy {busy/win:.3f}; busy ms/step {busy/1e6/steps:.2f}")
print(f"{'kernel':46s} {'calls/step':>10s} {'ms/step':>8s} {'avg us':>8s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{k:46s} {v[0]/steps:10.1f} {v[1]/1e6/steps:8.2f} {v[1]/v[0]/1e3:8.1f} {100*v[1]/busy:6.1f}")
