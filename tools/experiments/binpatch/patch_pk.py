#!/usr/bin/env python3
"""Replace every v_pk_fma_f32 with a broadcast first operand (op_sel_hi:[0,1,1] or op_sel:[1,0,0]) by two v_fma_f32, in place.
   python patch_pk.py in.s out.s [kernel-substring] [first:last]   (only the sites first..last of each kernel, 0-based)"""
import re, sys
src, dst = sys.argv[1], sys.argv[2]
only = sys.argv[3] if len(sys.argv) > 3 else ""
lo, hi = (int(x) for x in sys.argv[4].split(":")) if len(sys.argv) > 4 else (0, 10**9)
def pair(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok.strip())
    assert m and int(m.group(2)) == int(m.group(1)) + 1, tok
    return int(m.group(1)), int(m.group(2))
out, cur, count = [], None, {}
for line in open(src).read().split("\n"):
    m = re.match(r"^(_Z\w+):", line)
    if m: cur = m.group(1)
    mm = re.match(r"^\tv_pk_fma_f32 (v\[\d+:\d+\]), (v\[\d+:\d+\]), (v\[\d+:\d+\]), (v\[\d+:\d+\])\s*(op_sel_hi:\[0,1,1\]|op_sel:\[1,0,0\])\s*$", line)
    if cur and only in cur and mm:
        k = count.get(cur, 0)
        count[cur] = k + 1
        if lo <= k <= hi:
            d, a, b, c = (pair(mm.group(i)) for i in (1, 2, 3, 4))
            w = a[0] if mm.group(5).startswith("op_sel_hi") else a[1]
            first = (d[0], w, b[0], c[0]); second = (d[1], w, b[1], c[1])
            # the packed instruction reads everything before it writes: keep that when vDst overlaps a source of the other half
            if first[0] in (second[1], second[2], second[3]):
                assert second[0] not in (first[1], first[2], first[3]), line
                first, second = second, first
            out.append("\tv_fma_f32 v%d, v%d, v%d, v%d" % first)
            out.append("\tv_fma_f32 v%d, v%d, v%d, v%d" % second)
            continue
    out.append(line)
open(dst, "w").write("\n".join(out))
for k, v in count.items(): print(k[:80], v)
