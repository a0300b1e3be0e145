"""Summarise the rocprofv3 --pmc passes written by tools/pmc_merge.sh for one kernel.

    python tools/pmc_report.py gpurun_out/<name> <kernel-substring> > profiles/r01_x_pmc.md

Counters are summed over the dispatches of the matching kernel and divided by the number of dispatches seen
in each pass (one launch per step).  HBM-side traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB): on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import glob
import json
import sys


def main(root, needle):
    tot = collections.defaultdict(float)
    ndisp = collections.defaultdict(set)
    name = grid = None
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if needle not in r["Kernel_Name"]:
                continue
            name, grid = r["Kernel_Name"], r["Grid_Size"]
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
    if not tot:
        sys.exit("no dispatch of a kernel matching %r under %s" % (needle, root))
    per = {k: v / len(ndisp[k]) for k, v in tot.items()}
    print(f"# PMC counters per launch: `{name[:100]}`\n")
    print(f"grid {grid} threads; {max(len(v) for v in ndisp.values())} dispatch(es) per pass averaged; separate "
          f"`rocprofv3 --pmc` passes, counters only (tools/pmc_merge.sh).\n")
    print("| counter | per launch |\n|---|---:|")
    for k in sorted(per):
        print(f"| {k} | {per[k]:.4g} |")
    out = {"kernel": needle}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        traffic = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
        out.update(traffic_bytes_per_launch=traffic, fetch_kb=per["FETCH_SIZE"], write_kb=per["WRITE_SIZE"])
        print(f"\nHBM-side traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB = {traffic / 1e9:.3f} GB")
    if "SQ_INSTS_VALU" in per:
        out["valu_wave_insts_per_launch"] = per["SQ_INSTS_VALU"]
        print(f"VALU wave-instructions per launch: {per['SQ_INSTS_VALU']:.4g}")
    if "TCC_HIT_sum" in per and "TCC_REQ_sum" in per:
        print(f"L2 hit rate: {per['TCC_HIT_sum'] / per['TCC_REQ_sum']:.3f}")
    print("\n```json\n" + json.dumps(out) + "\n```")
    return out


def write_record(out, path, source, workload, note):
    """The record bench.py's roofline.traffic is read from: keyed to the hash of the kernel's source file, so a later
    edit of the kernel invalidates it instead of leaving a stale number."""
    import hashlib

    h = hashlib.sha256()
    for f in source.split(","):  # (the kernel's translation unit and the internal header it is written against)
        h.update(open(f, "rb").read())
    out = dict(out, workload=workload, source=source, source_sha16=h.hexdigest()[:16], collected_by=note)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    rec = main(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 3:  # ... <record.json> <source file> <workload> <note>
        write_record(rec, *sys.argv[3:7])
