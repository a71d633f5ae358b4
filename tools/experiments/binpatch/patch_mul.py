#!/usr/bin/env python3
"""v_pk_mul_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]  ->  v_mul_f32 Dlo, Alo, Bhi ; v_mul_f32 Dhi, Ahi, Blo   (in place)
   python patch_mul.py in.s out.s [kernel-substring]"""
import re, sys
src, dst = sys.argv[1], sys.argv[2]
only = sys.argv[3] if len(sys.argv) > 3 else ""
def pair(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok.strip()); assert m, tok
    return int(m.group(1)), int(m.group(2))
out, cur, n = [], None, {}
for line in open(src).read().split("\n"):
    m = re.match(r"^(_Z\w+):", line)
    if m: cur = m.group(1)
    mm = re.match(r"^\tv_pk_mul_f32 (v\[\d+:\d+\]), (v\[\d+:\d+\]), (v\[\d+:\d+\]) op_sel:\[0,1\] op_sel_hi:\[1,0\]\s*$", line)
    if cur and only in cur and mm:
        d, a, b = (pair(mm.group(i)) for i in (1, 2, 3))
        first, second = (d[0], a[0], b[1]), (d[1], a[1], b[0])
        if first[0] in second[1:]:
            assert second[0] not in first[1:], line
            first, second = second, first
        out += ["\tv_mul_f32_e32 v%d, v%d, v%d" % first, "\tv_mul_f32_e32 v%d, v%d, v%d" % second]
        n[cur] = n.get(cur, 0) + 1
        continue
    out.append(line)
open(dst, "w").write("\n".join(out))
for k, v in n.items(): print(k[40:80], v)
