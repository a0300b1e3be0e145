"""Instruction budget of a kernel by SOURCE REGION, through the line table of a -gline-tables-only build.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -DHHSR_X2_BUDGET -S --cuda-device-only \\
          csrc/hhsr_merge_x2.hip -o /tmp/x2.s
    python tools/isa_budget.py /tmp/x2.s 'k_merge_x2ILb0ELb1' csrc/hhsr_merge_x2.hip

The source carries `//@ name` tags: a tag opens a region that lasts to the next tag.  Every instruction of the kernel is
attributed to the region of the source line its `.loc` names (instructions of helpers inlined from other files — LDS
load wrappers, the geometry helpers — go to the region of the last line of the tagged file seen before them), and counted by
rate class (tools/ubench/valu_rate.hip, cycles per wave64 instruction per SIMD): full 2.2-2.8 (fma, mul, add, mov, integer
add / shift / logic), half 4-5 (min / max / cmp / cndmask / cvt / floor / mul_lo / v_min3_u32), quarter 8.2 (exp, rcp,
sqrt); LDS reads / writes, global loads / stores, scalar instructions and waits are listed next to them.
Static counts: a region inside the frame loop is executed once per frame and thread unless it is one arm of a branch.
"""
import collections
import re
import sys

QUARTER = ("v_exp_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32")
HALF = ("v_min", "v_max", "v_cmp", "v_cndmask", "v_cvt", "v_floor", "v_mul_lo", "v_mul_hi", "v_med3", "v_fract", "v_rndne",
        "v_trunc", "v_ceil", "v_readfirstlane", "v_readlane")
COLS = ["full", "half", "quarter", "lds_rd", "lds_wr", "vmem_rd", "vmem_wr", "scalar", "wait"]


def rate(op):
    if op.startswith(QUARTER):
        return "quarter"
    if op.startswith(HALF):
        return "half"
    if op.startswith("v_"):
        return "full"
    if op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle")):
        return "lds_rd"
    if op.startswith("ds_"):
        return "lds_wr"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_rd"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem_wr"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop")):
        return "wait"
    return "scalar"


def regions(src):
    tags, cur = {}, "outside"
    for i, line in enumerate(open(src), 1):
        m = re.match(r"\s*//@ (\w+)", line)
        if m:
            cur = m.group(1)
        tags[i] = cur
    return tags


def by_line(path, rx, src, top):
    """--lines N: the N source lines of `src` with the most VALU issue cycles attributed (static), with their text."""
    base = src.split("/")[-1]
    text = open(src).read().split("\n")
    files, cur, lineno = {}, None, 0
    counts = collections.defaultdict(collections.Counter)
    for line in open(path):
        t = line.strip()
        m = re.match(r"\.file\s+(\d+)\s+(?:\"[^\"]*\"\s+)?\"([^\"]+)\"", t)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if rx.search(m.group(1)) else None
            lineno = 0
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        if not cur:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            if files.get(int(m.group(1)), "").endswith(base):
                lineno = int(m.group(2))
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        counts[lineno][rate(t.split()[0])] += 1
    cyc = lambda c: 2.5 * c["full"] + 4.5 * c["half"] + 8.2 * c["quarter"]  # noqa: E731
    total = sum(cyc(c) for c in counts.values())
    print(f"total VALU issue cycles (static) {total:.0f}")
    for ln, c in sorted(counts.items(), key=lambda kv: -cyc(kv[1]))[:top]:
        print(f"{ln:5d} {cyc(c):7.0f} {100 * cyc(c) / total:5.1f}%  full {c['full']:4d} half {c['half']:4d} quarter {c['quarter']:3d} "
              f"lds {c['lds_rd'] + c['lds_wr']:3d} vmem {c['vmem_rd'] + c['vmem_wr']:3d} | {text[ln - 1].strip()[:110] if 0 < ln <= len(text) else ''}")


def main():
    path, rx, src = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3]
    if "--lines" in sys.argv:
        return by_line(path, rx, src, int(sys.argv[sys.argv.index("--lines") + 1]))
    tags = regions(src)
    base = src.split("/")[-1]
    files, cur, fileno, region = {}, None, None, "outside"
    counts, order = {}, []
    for line in open(path):
        t = line.strip()
        m = re.match(r"\.file\s+(\d+)\s+(?:\"[^\"]*\"\s+)?\"([^\"]+)\"", t)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if rx.search(m.group(1)) else None
            region = "outside"
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        if not cur:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            if files.get(int(m.group(1)), "").endswith(base):
                region = tags.get(int(m.group(2)), "outside")
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if region not in counts:
            counts[region] = collections.Counter()
            order.append(region)
        counts[region][rate(op)] += 1
    print("| region | " + " | ".join(COLS) + " | VALU | VALU issue cycles (2.5 / 4.5 / 8.2) |")
    print("|---|" + "---:|" * (len(COLS) + 2))
    for region in order:
        c = counts[region]
        valu = c["full"] + c["half"] + c["quarter"]
        cyc = 2.5 * c["full"] + 4.5 * c["half"] + 8.2 * c["quarter"]
        print(f"| {region} | " + " | ".join(str(c[k]) for k in COLS) + f" | {valu} | {cyc:.0f} |")


if __name__ == "__main__":
    main()
