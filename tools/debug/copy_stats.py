"""Durations of the large H2D copies of a rocprofv3 --memory-copy-trace database, in time order (groups of 20).
   python tools/debug/copy_stats.py results.db"""
import sqlite3, sys

con = sqlite3.connect(sys.argv[1])
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
mc = [n for n in names if "memory_cop" in n and "rocpd" not in n][0]
cs = con.execute(f"select start, end, size from {mc} where size > 1000000 order by start").fetchall()
print(len(cs), "large copies")
for g in range(0, len(cs), 20):
    grp = cs[g:g + 20]
    d = [(e - s) / 1e6 for s, e, _ in grp]
    span = (grp[-1][1] - grp[0][0]) / 1e6
    print(f"copies {g:4d}..: {grp[0][2] / 1e6:5.1f} MB each, mean {sum(d) / len(d):.3f} ms, max {max(d):.3f}, span of the group {span:.2f} ms")
