"""Per-kernel HBM traffic and bandwidth of one bench step: PMC passes (FETCH_SIZE / WRITE_SIZE, counters only)
joined with the un-instrumented kernel durations of a --kernel-trace run.

    tools/hbm_table.sh            # on the GPU box: writes gpurun_out/hbm/{fetch,write}/..., gpurun_out/hbm/kt.db
    python tools/hbm_table.py gpurun_out/hbm > profiles/r01_hbm_per_kernel.md

HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB): gfx950 reports half of the bytes of wide coalesced reads in
FETCH_SIZE (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are included, so this is fabric-side traffic.
"""
import collections
import csv
import glob
import sqlite3
import sys

PEAK = 8000.0  # GB/s


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:44]


def main(root, steps):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[k][r["Counter_Name"]] += 1
    con = sqlite3.connect(root + "/kt.db")
    dur = {}
    for name, c, tot in con.execute("select name, count(*), sum(end-start)/1e3 from kernels group by name"):
        k = short(name)
        a = dur.setdefault(k, [0, 0.0])
        a[0] += c
        a[1] += tot
    rows = []
    for k, d in cnt.items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d or k not in dur or k.startswith("Cijk"):
            continue
        fetch = d["FETCH_SIZE"] / nd[k]["FETCH_SIZE"]
        write = d["WRITE_SIZE"] / nd[k]["WRITE_SIZE"]
        byts = (2 * fetch + write) * 1024.0
        calls, tot_us = dur[k]
        us = tot_us / calls
        rows.append((tot_us / steps, k, calls / steps, us, byts, byts / (us * 1e-6) / 1e9))
    rows.sort(reverse=True)
    print("# HBM-side traffic and bandwidth per kernel — 12 MP x 20 frames x2, one MI355X\n")
    print("`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (counters only) joined with the kernel durations of a "
          "separate `--kernel-trace` run of the same command (`bench.py --no-cpu-baseline --streams 1`).  "
          "Bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction), per launch; peak 8 TB/s.\n")
    print("| kernel | launches / step | us / launch | ms / step | HBM MB / launch | GB/s | % of peak |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    tb = tt = 0.0
    for ms, k, c, us, b, gbs in rows:
        if ms / 1e3 < 0.02:
            continue
        print(f"| `{k}` | {c:.0f} | {us:.1f} | {ms / 1e3:.2f} | {b / 1e6:.1f} | {gbs:.0f} | {100 * gbs / PEAK:.0f} |")
        tb += b * c
        tt += ms
    print(f"\nAll listed kernels: {tb / 1e9:.1f} GB per step in {tt / 1e3:.1f} ms of kernel time = {tb / (tt * 1e-6) / 1e9:.0f} GB/s "
          f"average ({100 * tb / (tt * 1e-6) / 1e9 / PEAK:.0f} % of the 8 TB/s peak).")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
