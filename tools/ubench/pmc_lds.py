"""Summarise SQ_LDS_* counters per kernel dispatch of a rocprofv3 --pmc run (csv)."""
import collections, csv, glob, sys
rows = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60])
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, name), c in sorted(rows.items()):
    print(d, name, {k: f"{v:.3g}" for k, v in c.items()})
