"""Summarise a rocprofv3 rocpd database (``--kernel-trace --stats``) into a per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db [steps] > profiles/r01_x.md
"""
import sqlite3
import sys


def main(db, steps=1):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3, max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name "
                       "order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"Total kernel time {tot:.2f} ms over {steps} step(s) = {tot / steps:.2f} ms/step")
    own = sum(r[2] for r in rows if r[0].startswith("k_") or r[0].startswith("void k_"))
    print(f"hhsr kernels only (`k_*`; the rest of the total is the synthetic burst's generation — GEMMs, sin / cos, elementwise — "
          f"and torch copies): {own:.2f} ms = {own / steps:.2f} ms/step\n")
    print("| kernel | calls | total ms | % | avg us | min us | max us | vgpr | sgpr | lds B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for n, c, t, a, mn, mx, v, s, l in rows:
        if t / tot < 0.001:
            continue
        print(f"| `{n[:70]}` | {c} | {t:.2f} | {100 * t / tot:.1f} | {a:.1f} | {mn:.1f} | {mx:.1f} | {v} | {s} | {l} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
