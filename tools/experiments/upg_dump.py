#!/usr/bin/env python3
"""Round 6 (profiles/r6_two_models.txt section 9): WHAT is wrong when the fp32 fused Up block goes wrong beside a bf16 GEMM?
Needs a DEBUG build of the failing ir_fused.hip (CASYNC_LIB; `git show 15b33ac^:calipsync_amd/csrc/ir_fused.hip` + ir_common.h / common.h of
that commit, with tools/experiments/binpatch/upg_dump.diff applied, linked with the tree's other objects) whose UPG kernel, given an ODD
pointer through casync_debug_ir_stamps, writes per lane, tile and chunk the expand accumulator, the E value it stores, the four G taps
it read, its four bilinear weights and its tap offset (32 floats).  One clean launch,
then launches beside the looping GEMM until the output differs; the two dumps are compared."""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import _lib, pack, recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
sd_np = recipe.make_state_dict()
folded = pack.fold(sd_np)
net = Model(6, "hubert").to(dev)
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
x, a = recipe.make_inputs_range(0, 96)
net(torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev))
torch.cuda.synchronize()
F32 = lambda k: torch.from_numpy(folded[k].astype(np.float32)).contiguous().to(dev)
frames, h, cin, cexp, c_lo = 32, 160, 64, 128, 32
prefix = "up4.conv.double_conv.0"
G_ = torch.randn(frames * 80 * 80, cexp, device=dev)
skip = torch.randn(frames, h, h, cin - c_lo, device=dev)
w1b, b1, wd, bd, w2, b2 = (F32(prefix + k) for k in (".pw1b.w", ".pw1.b", ".dw.w", ".dw.b", ".pw2.w", ".pw2.b"))
out = torch.empty(frames, h, h, 32, device=dev)
NWG, NCH, NT = frames * 200, 4, 6            # 8 x 16 output tiles; 4 chunks; MT1 * NT1 tiles per wave and chunk
dump = torch.zeros(NWG * NCH * NT * 256 * 32, device=dev)   # per lane, tile slot and chunk: acc, E, the four G taps, the four weights, the tap offset


def upg(s, dump_ptr):
    lib.casync_op_set_dtype(0)
    lib.casync_debug_ir_stamps(dump_ptr | 1 if dump_ptr else 0)
    st = lib.casync_op_ir_fused_upg(G_.data_ptr(), cexp, skip.data_ptr(), cin - c_lo, w1b.data_ptr(), b1.data_ptr(), wd.data_ptr(), bd.data_ptr(),
                                    w2.data_ptr(), b2.data_ptr(), out.data_ptr(), 32, frames, h, h, cin, 32, s)
    lib.casync_debug_ir_stamps(0)
    assert st == 0, lib.casync_last_error()


upg(torch.cuda.current_stream().cuda_stream, dump.data_ptr())
torch.cuda.synchronize()
ref_out, ref_dump = out.clone(), dump.clone()
A = torch.randn(25600, 512, device=dev).to(torch.bfloat16)
W = (torch.randn(1024, 512, device=dev) / 512 ** 0.5).to(torch.bfloat16)
bias = torch.randn(1024, device=dev)
Cs = [torch.empty(25600, 1024, device=dev, dtype=torch.bfloat16) for _ in range(4)]
stop, found = [False], []


def load():
    torch.cuda.set_device(0)
    ss = [torch.cuda.Stream() for _ in range(4)]
    lib.casync_op_set_dtype(1)
    while not stop[0]:
        for i, s in enumerate(ss):
            for _ in range(2):
                lib.casync_op_pw_gemm(A.data_ptr(), 512, W.data_ptr(), bias.data_ptr(), Cs[i].data_ptr(), 1024, 25600, 1024, 512, 1, 0, 0, 0, 0, 0, 0, 0, s.cuda_stream)
        for s in ss:
            s.synchronize()


def work():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(int(os.environ.get("LAUNCHES", "300"))):
            dump.zero_()
            upg(s.cuda_stream, dump.data_ptr())
            s.synchronize()
            if not torch.equal(out, ref_out):
                found.append((i, out.clone(), dump.clone()))
                if len(found) >= 4:
                    return


tl, tw = threading.Thread(target=load), threading.Thread(target=work)
tl.start(); tw.start(); tw.join(); stop[0] = True; tl.join()
print(f"launches with a wrong output: {len(found)}", flush=True)
lrelu_inv = lambda e: torch.where(e < 0, e / 0.01, e)
for i, o, d in found:
    npx = int(((o - ref_out).abs().amax(-1) > 0).sum())
    dd, rr = d.view(NWG, NCH, NT, 256, 32), ref_dump.view(NWG, NCH, NT, 256, 32)
    lane = torch.arange(256, device=dev)
    # live lanes: halo pixel 16 * (wave * 3 + i) + l15 < 180
    live = torch.stack([(16 * ((lane // 64) * 3 + t // 2) + (lane % 16)) < 180 for t in range(NT)])      # [NT, 256]
    bad_e = (dd[..., 4:8] != rr[..., 4:8]).any(-1) & live
    bad_acc = (dd[..., 0:4] != rr[..., 0:4]).any(-1) & live
    bad_g = (dd[..., 8:24] != rr[..., 8:24]).any(-1) & live
    bad_w = (dd[..., 24:29] != rr[..., 24:29]).any(-1) & live
    print(f"launch {i}: {npx} output pixels wrong; LIVE dump entries with a wrong accumulator {int(bad_acc.sum())}, wrong G taps {int(bad_g.sum())}, wrong weights / offset {int(bad_w.sum())}, wrong E {int(bad_e.sum())}")
    idx = bad_e.nonzero()
    groups = {}
    for wg, ch, t, ln in idx.tolist():
        groups.setdefault((wg, ch, t, ln // 64), []).append(ln % 64)
    shown = 0
    for (wg, ch, t, wave), lanes in groups.items():
        if shown >= 8:
            break
        shown += 1
        ln = wave * 64 + lanes[0]
        e_b, e_g, acc = dd[wg, ch, t, ln, 4:8], rr[wg, ch, t, ln, 4:8], rr[wg, ch, t, ln, 0:4]
        taps_b, taps_g = dd[wg, ch, t, ln, 8:24].view(4, 4), rr[wg, ch, t, ln, 8:24].view(4, 4)
        w_b, w_g = dd[wg, ch, t, ln, 24:28], rr[wg, ch, t, ln, 24:28]
        print(f"   workgroup {wg} chunk {ch} wave {wave} slot {t} (i={t // 2}, n={t % 2}): lanes {sorted(set(lanes))[:4]}..{max(lanes)} ({len(lanes)}); "
              f"taps wrong {bool((taps_b != taps_g).any())}, weights wrong {bool((w_b != w_g).any())}")
        v = acc.double()
        steps = [v.clone()]
        for k in range(4):
            v = v + taps_b[k].double() * w_b[k].double()
            steps.append(v.clone())
        e_re = torch.nn.functional.leaky_relu(v.float(), 0.01)
        print(f"      lane {lanes[0]}: E good {e_g.tolist()}\n               E got  {e_b.tolist()}\n               E recomputed from the dumped taps and weights {e_re.tolist()}")
        pre_b = lrelu_inv(e_b).double()
        for k in range(5):
            if torch.allclose(pre_b, steps[k], rtol=1e-5, atol=1e-6):
                print(f"      E got = lrelu(accumulator + the first {k} of the four taps)")
        for drop in range(4):
            alt = acc.double() + sum(taps_b[k].double() * w_b[k].double() for k in range(4) if k != drop)
            if torch.allclose(pre_b, alt, rtol=1e-5, atol=1e-6):
                print(f"      E got = the sum WITHOUT tap {drop}")
        # ONE component is wrong: which single input, replaced by which other value of this lane's dump, explains it?
        pre_g = lrelu_inv(e_g).double()
        for e in range(4):
            if e_b[e] == e_g[e]:
                continue
            for k in range(4):
                wk = w_b[k].double()
                if abs(float(wk)) < 1e-12:
                    continue
                x = (pre_b[e] - (pre_g[e] - wk * taps_b[k, e].double())) / wk            # the tap value that would give the wrong sum
                row = dd[wg, :, :, ln, :].reshape(-1, 32).double()                        # everything this lane dumped (all chunks, all slots)
                hit = ((row - x).abs() <= 1e-5 * max(1.0, abs(float(x)))).nonzero()
                for r, c in hit.tolist()[:4]:
                    what = ["acc", "E", "tap0", "tap1", "tap2", "tap3", "weights"][min(c // 4, 6)] if c < 28 else "offset"
                    print(f"      element {e}: right if tap {k} had been {float(x):.7g} = this lane's {what}[{c % 4}] of chunk {r // NT} slot {r % NT}")
            for k in range(4):      # or a stale WEIGHT
                tk = taps_b[k, e].double()
                if abs(float(tk)) < 1e-12:
                    continue
                x = (pre_b[e] - (pre_g[e] - w_b[k].double() * tk)) / tk
                row = dd[wg, :, :, ln, 24:28].reshape(-1, 4).double()
                hit = ((row - x).abs() <= 1e-5 * max(1.0, abs(float(x)))).nonzero()
                for r, c in hit.tolist()[:4]:
                    print(f"      element {e}: right if weight {k} had been {float(x):.7g} = this lane's weight {c} of chunk {r // NT} slot {r % NT}")
        if (taps_b != taps_g).any():
            print(f"      taps good {taps_g.tolist()}\n      taps got  {taps_b.tolist()}")
