#!/usr/bin/env python3
"""Timeline of the last run in a rocprofv3 kernel-trace database of tools/trace_cfg1.py: time per kernel,
idle time of the main stream (everything but the sequential-sum chain), the chain's launches."""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select d.start,d.end,s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
last = max(i for i, r in enumerate(rows) if 'k_pack' in r[2] or 'k_diag' in r[2])
# the last run: from the last big gap before the end
gaps = [(rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)]
cut = max(i for g, i in gaps if g > 300000) + 1
run = rows[cut:]
t0 = run[0][0]; t1 = max(r[1] for r in run)
print("span %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(run)))
agg = collections.defaultdict(lambda: [0, 0])
for a, b, nm in run:
    agg[nm.split('(')[0][:48]][0] += b - a; agg[nm.split('(')[0][:48]][1] += 1
for nm, (t, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:12]:
    print("%8.1f us %4d  %s" % (t / 1e3, n, nm))
main = sorted((r[0], r[1], r[2]) for r in run if 'k_seq_chain' not in r[2])
ce = main[0][1]; prev = main[0][2]; tot = 0
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 8000
for a, b, nm in main[1:]:
    if a > ce:
        tot += a - ce
        if a - ce > thr:
            print("  gap %.1f us after %s before %s at %.2f ms" % ((a - ce) / 1e3, prev[:30], nm[:30], (a - t0) / 1e6))
    if b > ce: ce = b; prev = nm
print("main-stream idle total %.1f us" % (tot / 1e3))
