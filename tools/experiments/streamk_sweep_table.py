import re, sys
rows, cur = {}, None
for l in open(sys.argv[1]):
    m = re.match(r"== cfg (\d) sk (\d)", l)
    if m:
        cur = (int(m[1]), int(m[2])); continue
    m = re.search(r"M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+([\d.]+) ms", l)
    if m:
        rows.setdefault((int(m[1]), int(m[2]), int(m[3])), {})[cur] = float(m[4]) * 1e3
print("shape (M,N,K)         | 128x128 dp/sk | 128x64 dp/sk  | 64x64 dp/sk   (us)")
for k, v in rows.items():
    print(f"{str(k):22s}| {v.get((0,0),0):6.1f} {v.get((0,1),0):6.1f} | {v.get((1,0),0):6.1f} {v.get((1,1),0):6.1f} | {v.get((2,0),0):6.1f} {v.get((2,1),0):6.1f}")
