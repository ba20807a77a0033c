"""Per-layer rates of the finetune step's convolutions at batch B on the hand-written kernels: forward, data gradient
(the same kernel on the rotated weights) and weight gradient (kernel + slab reduction), each timed alone with HIP events.
The layer list is DtoidNet's (models/dtoid/network.py: DenseNet-121 blocks at 120x160 / 60x80 / 30x40 / 29x39, the
correlation / fusion / decoder / cls / reg convolutions).  python tools/train_layers_bench.py [--batch 8] [--what wgrad,fwd,dgrad]"""
import re
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd.dtoid import train_ops as T  # noqa: E402


def layers():
    """(name, count, Cin, Cout, H, W, taps, src_hw or None, in_cs extra) -- `count` identical launches per step."""
    out = []
    c, hw = 64, (120, 160)
    for bi, n in enumerate((6, 12, 24, 16)):
        for li in range(n):
            cl = c + 32 * li
            out.append(("b%d.l%02d.1x1" % (bi + 1, li + 1), 1, cl, 128, hw[0], hw[1], 1, None, c + 32 * n - cl))
        out.append(("b%d.3x3" % (bi + 1), n, 128, 32, hw[0], hw[1], 9, None, 0))
        c += 32 * n
        if bi < 3:
            out.append(("t%d.1x1" % (bi + 1), 1, c, c // 2, hw[0], hw[1], 1, None, 0))
            c //= 2
            hw = (hw[0] // 2, hw[1] // 2) if bi < 2 else (29, 39)
    out.append(("c1.1x1", 1, 1024, 640, 29, 39, 1, None, 0))
    out += [("corr.640-256", 3, 640, 256, 29, 39, 9, None, 0), ("cf.768-512", 1, 768, 512, 29, 39, 9, None, 0),
            ("s1/cls1/reg1.512-256", 3, 512, 256, 29, 39, 9, None, 0), ("trunk.256-256", 6, 256, 256, 29, 39, 9, None, 0),
            ("cls.out.256-48", 1, 256, 48, 29, 39, 9, None, 0), ("reg.out.256-96", 1, 256, 96, 29, 39, 9, None, 0),
            ("s2.256-128", 1, 256, 128, 58, 78, 9, (29, 39), 0), ("s3.128-64", 1, 128, 64, 116, 156, 9, (58, 78), 0),
            ("s4.64-32", 1, 64, 32, 232, 312, 9, (116, 156), 0), ("s5.32-16", 1, 32, 16, 480, 640, 9, (232, 312), 0)]
    return out


def timed(fn, reps):
    """ms per call: the better of two back-to-back groups of `reps` calls (one group now and then catches a stall that is
    not the kernel's -- 5 ms on a 0.03 ms layer, always at the same place of a long run and gone when the layer runs alone)."""
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        best = t if best is None else min(best, t)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--fwd-kind", default="fwd", choices=["fwd", "fwd_x6", "fwd_exact"],
                    help="arithmetic of the direct kernel's forward launches: split-bf16, three-way split, exact f32")
    ap.add_argument("--wino", action="store_true", help="3x3 layers (no fused up-sampling): forward and data gradient on csrc/wino.hip")
    a = ap.parse_args()
    B = a.batch
    what = a.what.split(",")
    tot = {w: [0.0, 0.0] for w in what}
    rows = []
    print("%-22s %3s %5s %5s %9s | " % ("layer", "n", "Cin", "Cout", "HxW") + " | ".join("%-7s ms   TF/s" % w for w in what))
    for name, count, cin, cout, H, W, taps, src, extra in layers():
        if a.only and not re.search(a.only, name):
            continue
        Hs, Ws = (H, W) if src is None else src
        k = 3 if taps == 9 else 1
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, cin + extra, Hs, Ws, generator=g).cuda().contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, cout, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
        ps, pt = torch.rand(cin).cuda() + 0.5, torch.randn(cin).cuda() * 0.1
        flops = 2.0 * B * H * W * cin * cout * taps
        res = {}
        wino = a.wino and taps == 9 and src is None
        if "fwd" in what:
            wpk = T._pack(w, "wino_fwd" if wino else a.fwd_kind)
            out = T.empty_nhwc(B, cout, H, W, "cuda")
            res["fwd"] = timed(lambda: T.conv_raw(x, wpk, B, H, W, cin, cout, taps, out, pre=(ps, pt), pre_relu=True,
                                                  in_cs=cin + extra, src_hw=(Hs, Ws) if src else (0, 0), wino=wino), a.reps)
        if "dgrad" in what and src is None and cout % 16 == 0:
            wpk = T._pack(w, "wino_dgrad" if wino else "dgrad")
            dx = T.empty_nhwc(B, cin, H, W, "cuda")
            res["dgrad"] = timed(lambda: T.conv_raw(dy, wpk, B, H, W, cout, cin, taps, dx, wino=wino), a.reps)
        elif "dgrad" in what and cout % 16 == 0:
            wpk = T._pack(w, "dgrad")
            dx = T.empty_nhwc(B, cin, H, W, "cuda")
            res["dgrad"] = timed(lambda: T.conv_raw(dy, wpk, B, H, W, cout, cin, taps, dx), a.reps)
        if "wgrad" in what:
            dw = torch.empty_like(w)
            res["wgrad"] = timed(lambda: T.wgrad_raw(x, dy, B, H, W, cin, cout, taps, dw, pre=(ps, pt), pre_relu=True,
                                                     in_cs=cin + extra, src_hw=(Hs, Ws) if src else (0, 0)), a.reps)
        cells = []
        for wname in what:
            if wname in res:
                tot[wname][0] += res[wname] * count
                tot[wname][1] += flops * count
                cells.append("%10.3f %6.1f" % (res[wname], flops / res[wname] / 1e9))
            else:
                cells.append("%10s %6s" % ("-", "-"))
        rows.append({"layer": name, "count": count, "cin": cin, "cout": cout, "H": H, "W": W, "taps": taps,
                     "gflop": flops / 1e9, **{k2: v for k2, v in res.items()}})
        print("%-22s %3d %5d %5d %4dx%-4d | " % (name, count, cin, cout, H, W) + " | ".join(cells), flush=True)
    print("TOTAL per step: " + "; ".join("%s %.2f ms (%.1f TFLOP/s over %.0f GFLOP)" % (w, t, f / t / 1e9 if t else 0, f / 1e9)
                                         for w, (t, f) in tot.items()))
    if a.json:
        json.dump({"batch": B, "rows": rows, "total": {w: {"ms": t, "gflop": f / 1e9} for w, (t, f) in tot.items()}},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
