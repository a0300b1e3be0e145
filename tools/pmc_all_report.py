"""Per-kernel bottleneck table from tools/pmc_all.sh: launch duration (kernel trace), VALU / LDS busy fractions, wave
stalls, HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md).

    python tools/pmc_all_report.py gpurun_out/<name> <steps incl. warm-up> > profiles/r02_kernel_bottlenecks.md
"""
import collections
import csv
import glob
import sqlite3
import sys

SIMDS, XCDS = 1024, 8


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:44]


def main(root, steps):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    con = sqlite3.connect(root + "/kt.db")
    dur = {}
    for name, c, tot in con.execute("select name, count(*), sum(end-start)/1e3 from kernels group by name"):
        dur[short(name)] = (c, tot)
    print("| kernel | launches / step | us / launch | ms / step | VALU busy | LDS busy | waves parked | issue stall | "
          "VALU inst / wave | HBM MB / launch | GB/s | LDS conflict share |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    rows = []
    for k, (c, tot) in dur.items():
        if k not in cnt or not k.startswith("k_"):  # the include regex also matches library GEMMs ("Cijk_...": the
            continue                                 # synthetic burst generator of bench.py, outside the step)
        per = {n: v / max(1, len(disp[k][n])) for n, v in cnt[k].items()}
        us = tot / c
        cyc = per.get("GRBM_GUI_ACTIVE", 0) / XCDS  # cycles of the launch
        valu = 4 * per.get("SQ_ACTIVE_INST_VALU", 0) / (cyc * SIMDS) if cyc else 0
        lds = per.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256) if cyc else 0
        wc = per.get("SQ_WAVE_CYCLES", 0)
        parked = per.get("SQ_WAIT_ANY", 0) / wc if wc else 0
        stall = per.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0
        ipw = per.get("SQ_INSTS_VALU", 0) / per["SQ_WAVES"] if per.get("SQ_WAVES") else 0
        mb = (2 * per.get("FETCH_SIZE", 0) + per.get("WRITE_SIZE", 0)) * 1024 / 1e6
        conf = per.get("SQ_LDS_BANK_CONFLICT", 0) / per["SQ_LDS_IDX_ACTIVE"] if per.get("SQ_LDS_IDX_ACTIVE") else 0
        rows.append((tot / steps, k, c / steps, us, valu, lds, parked, stall, ipw, mb, mb / us * 1e3 if us else 0, conf))
    for r in sorted(rows, reverse=True):
        print(f"| `{r[1]}` | {r[2]:.0f} | {r[3]:.1f} | {r[0] / 1e3:.2f} | {100 * r[4]:.0f} % | {100 * r[5]:.0f} % | "
              f"{100 * r[6]:.0f} % | {100 * r[7]:.0f} % | {r[8]:.0f} | {r[9]:.1f} | {r[10]:.0f} | {100 * r[11]:.0f} % |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
