#!/usr/bin/env python3
"""python patch_ins.py in.s out.s kernel-substring rule [rule ...]
   rule = before:<regex>:<text>  |  after:<regex>:<text>  |  afterlast:<regex>:<text> (behind a match whose NEXT instruction does not match)
          | beforefirst:<regex>:<text> (in front of a match whose PREVIOUS instruction does not match)
   optional last field a-b = only matches number a..b (0-based, per kernel and rule); \\n in <text> = a new line"""
import re, sys
src, dst, only = sys.argv[1], sys.argv[2], sys.argv[3]
rules = []
for r in sys.argv[4:]:
    f = r.split(":")
    rng = (0, 10**9)
    if len(f) > 3 and re.fullmatch(r"\d+-\d+", f[-1]):
        a, b = f[-1].split("-"); rng = (int(a), int(b)); f = f[:-1]
    rules.append((f[0], re.compile(f[1]), ":".join(f[2:]).replace("\\n", "\n\t"), rng))
lines = open(src).read().split("\n")
def is_ins(l): return l.startswith("\t") and not l.startswith(("\t.", "\t;")) and l.strip()
idx = [i for i, l in enumerate(lines) if is_ins(l)]
nxt = {a: b for a, b in zip(idx, idx[1:])}; prv = {b: a for a, b in zip(idx, idx[1:])}
cur, out, cnt = None, [], {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m: cur = m.group(1)
    pre, post = [], []
    if cur and only in cur and is_ins(l):
        op = l.strip()
        for ri, (kind, rx, text, rng) in enumerate(rules):
            if not rx.search(op): continue
            if kind == "afterlast" and i in nxt and rx.search(lines[nxt[i]].strip()): continue
            if kind == "beforefirst" and i in prv and rx.search(lines[prv[i]].strip()): continue
            k = cnt.get((cur, ri), 0); cnt[(cur, ri)] = k + 1
            if not (rng[0] <= k <= rng[1]): continue
            (pre if kind.startswith("before") else post).append("\t" + text)
    out += pre + [l] + post
open(dst, "w").write("\n".join(out))
for (k, ri), v in sorted(cnt.items()): print(k[40:75], "rule", ri, "matches", v)
