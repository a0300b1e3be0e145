"""Concurrency of one graph-replayed step from a rocprofv3 kernel trace: busy time (union of kernel intervals), idle
gaps, average number of kernels in flight.   python tools/debug/timeline.py <results.db> [n_last_steps]"""
import sqlite3, sys

db, nlast = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# steps end with k_merge_border*; take the last complete ones
ends = [i for i, r in enumerate(rows) if r[0].startswith("void k_merge_border") or "k_merge_border" in r[0]]
for s in range(len(ends) - nlast, len(ends)):
    lo = ends[s - 1] + 1
    hi = ends[s]
    ks = [r for r in rows[lo:hi + 1] if "k_" in r[0]]
    t0, t1 = min(r[1] for r in ks), max(r[2] for r in ks)
    ev = sorted([(r[1], 1) for r in ks] + [(r[2], -1) for r in ks])
    busy = 0; depth = 0; last = t0; area = 0
    for t, d in ev:
        if depth > 0:
            busy += t - last
        area += depth * (t - last)
        depth += d; last = t
    print(f"step: span {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, "
          f"sum of durations {sum(r[2] - r[1] for r in ks) / 1e6:.3f} ms, mean kernels in flight {area / max(busy, 1):.2f}, kernels {len(ks)}")

# kernel census of the last step
import collections
lo, hi = ends[-2] + 1, ends[-1]
cnt = collections.Counter(r[0].split("(")[0][:70] for r in rows[lo:hi + 1])
dur = collections.Counter()
for r in rows[lo:hi + 1]:
    dur[r[0].split("(")[0][:70]] += r[2] - r[1]
for k, v in cnt.most_common():
    print(f"{v:4d}  {dur[k] / 1e3:8.1f} us  {k}")
