"""Summary of a tests/test_fuzz_parity.py report run for PARITY.md.
    HHSR_FUZZ_REPORT=/tmp/fuzz.txt python -m pytest tests/test_fuzz_parity.py -m gpu -q      # on an MI355X
    python tools/fuzz_report.py /tmp/fuzz.txt >> PARITY.md"""
import re
import sys

rows = []
for line in open(sys.argv[1]):
    m = re.match(r"case (\S+) \((.*?)\): flipped (\d+)(?: \(NOT one cluster\))?,(?: ica (\d+),)? nan (\d+), flow (\S+), r (\S+) / injected (\S+); injected image max (\S+) "
                 r"\((\d+) > 1e-4, (\d+) outside rejecting regions\); own flows: flow-sensitive (\d+) \(max (\S+)\), other (\d+) "
                 r"\(max (\S+), (\d+) outside", line)
    if m:
        g = m.groups()
        rows.append(dict(id=g[0], desc=g[1], flipped=int(g[2]), ica=int(g[3] or 0), nan=int(g[4]), flow=float(g[5].rstrip(",")),
                         r=float(g[6]), r_inj=float(g[7]), inj=float(g[8]), n_inj=int(g[9]), sens=int(g[11]),
                         sens_max=float(g[12]), other=int(g[13]), other_max=float(g[14])))
n = len(rows)
print(f"\n## Randomised end-to-end sweep (tests/test_fuzz_parity.py, {n} cases, HIP main() vs oracle.main())\n")
print(f"* NaN pattern mismatches: {sum(r['nan'] for r in rows)}; tiles that follow another block-matching decision (flow differs by > 1e-3 px): "
      f"{sum(r['flipped'] for r in rows)} (in case{'s' if sum(1 for r in rows if r['flipped']) != 1 else ''} "
      f"{', '.join(r['id'] for r in rows if r['flipped']) or '-'})")
print(f"* tiles whose flow differs by 1e-4 ... 1e-3 px (ill-conditioned ICA): {sum(r['ica'] for r in rows)} "
      f"(in case{'s' if sum(1 for r in rows if r['ica']) != 1 else ''} {', '.join(r['id'] for r in rows if r['ica']) or '-'})")
print(f"* flow, all other tiles: max {max(r['flow'] for r in rows):.1e} px (asserted 1e-4); robustness r: max "
      f"{max(r['r'] for r in rows):.1e} own flows / {max(r['r_inj'] for r in rows):.1e} oracle flows injected (asserted 1e-4)")
clean = [r for r in rows if r["n_inj"] == 0]
print(f"* image, oracle flows injected: {len(clean)} cases <= {max(r['inj'] for r in clean):.2e}; "
      + "; ".join(f"case {r['id']}: {r['n_inj']} values > 1e-4, max {r['inj']:.2e}" for r in rows if r["n_inj"]))
print("* image, own flows, outside the footprint of the flipped tile: "
      f"{sum(1 for r in rows if r['sens'] == 0 and r['other'] == 0)} cases <= 1e-4; flow-sensitive values (agree once the "
      "flows are the oracle's): "
      + "; ".join(f"case {r['id']} ({r['desc']}): {r['sens']} values, max {r['sens_max']:.1e}" for r in rows if r["sens"])
      + "; other values > 1e-4: " + ("; ".join(f"case {r['id']}: {r['other']}, max {r['other_max']:.1e}" for r in rows if r["other"]) or "none"))
nfail = sum(1 for line in open(sys.argv[1]) if "ASSERTIONS FAILED" in line)
print(f"* cases violating an assertion of the test: {nfail}")
