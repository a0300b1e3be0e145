"""Static instruction mix of the kernels in a hipcc -S listing whose mangled name matches a regex.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only file.hip -o /tmp/x.s && python tools/isa_count.py /tmp/x.s k_align
"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
rx = re.compile(sys.argv[2])
cur, out = None, {}
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1) if rx.search(m.group(1)) else None
        if cur:
            out[cur] = []
        continue
    if l.startswith(".Lfunc_end"):
        cur = None
    if cur and l.startswith("\t") and not l.strip().startswith((".", ";")):
        out[cur].append(l.split()[0])
for k, ins in out.items():
    c = collections.Counter()
    for i in ins:
        p = i.split("_")[0]
        c["f64" if "f64" in i else p] += 1
    print(k[:60], len(ins), dict(c))
