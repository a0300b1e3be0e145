"""FETCH_SIZE / WRITE_SIZE per kernel of tools/ubench/traffic_calib against the true byte counts.
   python tools/ubench/traffic_calib_report.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>"""
import collections, csv, glob, sys

TRUE = 24576 * 24576 * 4.0
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:3]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | true bytes | FETCH_SIZE x 1024 | 2 x FETCH / true | WRITE_SIZE x 1024 | WRITE / true |")
print("|---|---:|---:|---:|---:|---:|")
for k in ("read_b128", "read_b32", "read_window", "write_b128", "write_b32", "write_tile384", "write_tile_dwords"):
    c = rows.get(k, {})
    f = sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [0]))) * 1024
    w = sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [0]))) * 1024
    print(f"| `{k}` | {TRUE:.4g} | {f:.4g} | {2 * f / TRUE:.3f} | {w:.4g} | {w / TRUE:.3f} |")
