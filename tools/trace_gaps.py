"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd sqlite): how much of a
multigrid cycle is spent BETWEEN its ~4000 dispatches rather than in them.
    python tools/trace_gaps.py run_results.db        (profiles/r03_trace_gaps.txt)
Gaps above 200 us (host work between solves, synchronisations) are listed apart; the histogram is of the rest."""
import sqlite3
import sys
import numpy as np

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select start, end, name from kernels order by start").fetchall()
st = np.array([r[0] for r in rows], dtype=np.int64); en = np.array([r[1] for r in rows], dtype=np.int64)
gap = (st[1:] - np.maximum.accumulate(en)[:-1]) / 1e3        # us; negative = overlap
dur = (en - st) / 1e3
small = gap[(gap <= 200)]
print(f"# {sys.argv[1]}: {len(rows)} dispatches, kernel time {dur.sum() / 1e3:.1f} ms, span {(en.max() - st.min()) / 1e6:.1f} ms")
print(f"gaps <= 200 us: {len(small)}, sum {small.clip(min=0).sum() / 1e3:.1f} ms, mean {small.mean():.2f} us, median {np.median(small):.2f} us; "
      f"overlapping (negative) {int((small < 0).sum())}")
print(f"gaps  > 200 us: {int((gap > 200).sum())}, sum {gap[gap > 200].sum() / 1e3:.1f} ms")
edges = [-1e9, 0, 0.5, 1, 1.5, 2, 3, 5, 10, 50, 200]
h, _ = np.histogram(small, bins=edges)
for a, b, n in zip(edges[:-1], edges[1:], h):
    print(f"  {a if a > -1e8 else '-inf':>6} .. {b:<5} us: {n}")
# by duration class of the FOLLOWING kernel
for lo, hi in ((0, 10), (10, 30), (30, 100), (100, 1e9)):
    m = (dur[1:] >= lo) & (dur[1:] < hi) & (gap <= 200)
    if m.any():
        print(f"before kernels of {lo}-{hi if hi < 1e8 else 'inf'} us: {int(m.sum())} gaps, mean {gap[m].mean():.2f} us, kernel time {dur[1:][m].sum() / 1e3:.1f} ms, gap time {gap[m].clip(min=0).sum() / 1e3:.1f} ms")
# which kernels follow the gaps of 3-200 us, and which precede them
import collections
names = [r[2].split('(anonymous namespace)::')[-1][:48] for r in rows]
after, before = collections.Counter(), collections.Counter()
tafter = collections.Counter()
for i in np.nonzero((gap > 3) & (gap <= 200))[0]:
    after[names[i + 1]] += 1; tafter[names[i + 1]] += gap[i]; before[names[i]] += 1
print("gaps of 3-200 us, by the kernel that FOLLOWS (count, gap ms):")
for k, n in after.most_common(12):
    print(f"  {k:50s} {n:6d} {tafter[k] / 1e3:8.2f}")
print("... by the kernel that PRECEDES:")
for k, n in before.most_common(8):
    print(f"  {k:50s} {n:6d}")
