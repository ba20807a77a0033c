"""Times every kernel of the training stem alone on the chip (HIP events around `reps` back-to-back launches) at the finetune
batch, with the bytes / flops each moves:   python tools/stem_bench.py [--batch 8] [--reps 20]
Output: one line per kernel: ms, GB/s of algorithmic bytes (or TFLOP/s), launches."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import _lib  # noqa: E402
from ossid_code_amd.dtoid import ops, train_ops as T  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    B, H, W, C = a.batch, 480, 640, 64
    Ho, Wo = H // 2, W // 2
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(dev)
    kern = (torch.randn(B, C, 3, 3, generator=g) * 0.2).to(dev)
    bn = torch.nn.BatchNorm2d(C).to(dev).train()
    s = _lib.stream
    F = _lib.fn
    x0 = ops.stem_conv(img, conv)
    gx0 = torch.randn_like(x0)
    m = torch.empty_like(x0)
    n_px = B * Ho * Wo
    t_bytes = n_px * C * 4                                      # one pass over the [B,240,320,64] tensor: 157 MB at batch 8
    res = {}

    def rec(name, ms, bytes_=None, flops=None):
        res[name] = {"ms": round(ms, 4)}
        if bytes_:
            res[name]["GB/s"] = round(bytes_ / ms / 1e6, 1)
        if flops:
            res[name]["TFLOP/s"] = round(flops / ms / 1e9, 1)
        print("%-28s %8.4f ms  %s" % (name, ms, {k: v for k, v in res[name].items() if k != "ms"}), flush=True)

    flops = 2.0 * n_px * C * 147
    rec("stem_conv_fwd", timed(lambda: ops.stem_conv(img, conv), a.reps), img.numel() * 4 + t_bytes, flops)
    wsb = F("ossid_stem_conv_wgrad_workspace_bytes")(B, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dw = torch.empty_like(conv.weight)
    rec("stem_conv_wgrad(+reduce)", timed(lambda: _lib.check(F("ossid_stem_conv_wgrad")(
        img.data_ptr(), gx0.data_ptr(), B, 3, H, W, 64, 7, 2, 3, None, None, ws.data_ptr(), wsb, dw.data_ptr(), 0, s()), "wgrad"), a.reps),
        img.numel() * 4 + t_bytes, flops)
    P = F("ossid_dw_add_stats_partials")(B, Ho, Wo, C)
    part = torch.empty(P * 2 * C, dtype=torch.float32, device=dev)
    pivot = torch.empty(C, dtype=torch.float32, device=dev)
    rec("dw_add+stats", timed(lambda: _lib.check(F("ossid_dw_add_stats_nhwc")(
        x0.data_ptr(), kern.data_ptr(), C * 9, B, Ho, Wo, C, 0, m.data_ptr(), part.data_ptr(), pivot.data_ptr(), s()), "dw"), a.reps),
        2 * t_bytes)
    rec("dw_add (data gradient)", timed(lambda: _lib.check(F("ossid_dw_add_nhwc")(
        gx0.data_ptr(), kern.data_ptr(), C * 9, B, Ho, Wo, C, 1, m.data_ptr(), s()), "dw"), a.reps), 2 * t_bytes)
    _lib.check(F("ossid_dw_add_stats_nhwc")(x0.data_ptr(), kern.data_ptr(), C * 9, B, Ho, Wo, C, 0, m.data_ptr(), part.data_ptr(),
                                            pivot.data_ptr(), s()), "dw")
    f = T.bn_fold_fwd((part, P, pivot), C, n_px, bn.weight, bn.bias, bn.eps, 0.1, bn.running_mean, bn.running_var)
    rec("bn_fold_fwd(P=%d)" % P, timed(lambda: T.bn_fold_fwd((part, P, pivot), C, n_px, bn.weight, bn.bias, bn.eps, 0.1, None, None), a.reps))
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    pooled = T.empty_nhwc(B, C, Hp, Wp, dev)
    idx = torch.empty(B * Hp * Wp * C, dtype=torch.uint8, device=dev)
    rec("stem_pool_fwd", timed(lambda: _lib.check(F("ossid_stem_pool_fwd")(
        m.data_ptr(), f[0].data_ptr(), f[1].data_ptr(), B, Ho, Wo, C, pooled.data_ptr(), idx.data_ptr(), s()), "pool"), a.reps),
        t_bytes + pooled.numel() * 5)
    dp = torch.randn_like(pooled)
    P2 = F("ossid_stem_pool_bwd_partials")(B, Ho, Wo, C)
    part2 = torch.empty(P2 * 2 * C, dtype=torch.float32, device=dev)
    rec("stem_pool_bwd sums", timed(lambda: _lib.check(F("ossid_stem_pool_bwd")(
        m.data_ptr(), idx.data_ptr(), dp.data_ptr(), f[0].data_ptr(), f[1].data_ptr(), None, None, B, Ho, Wo, C, part2.data_ptr(), None,
        s()), "pb"), a.reps), t_bytes + pooled.numel() * 5)
    r = torch.zeros(4, C, device=dev)
    rec("bn_fold_bwd(P=%d)" % P2, timed(lambda: T.bn_fold_bwd(None, None, bn.weight, f[2], f[3], C, n_px, r[0], r[1], r[2], r[3],
                                                            partials=(part2, P2)), a.reps))
    dm = torch.empty_like(m)
    rec("stem_pool_bwd apply", timed(lambda: _lib.check(F("ossid_stem_pool_bwd")(
        m.data_ptr(), idx.data_ptr(), dp.data_ptr(), f[0].data_ptr(), f[1].data_ptr(), r[2].data_ptr(), r[3].data_ptr(), B, Ho, Wo, C,
        None, dm.data_ptr(), s()), "pb"), a.reps), 2 * t_bytes + pooled.numel() * 5)
    wk = torch.empty(F("ossid_dw_bwd_k_workspace_floats")(B, Ho, Wo, C), dtype=torch.float32, device=dev)
    dk = torch.empty(B, C, 3, 3, device=dev)
    rec("dw_bwd_k(+finalize)", timed(lambda: _lib.check(F("ossid_dw_bwd_k_nhwc")(
        x0.data_ptr(), dm.data_ptr(), B, Ho, Wo, C, wk.data_ptr(), dk.data_ptr(), s()), "bk"), a.reps), 2 * t_bytes)
    # a plain 157 MB copy for scale
    rec("copy of one tensor (torch)", timed(lambda: m.copy_(x0), a.reps), 2 * t_bytes)
    res["total_ms"] = round(sum(v["ms"] for k, v in res.items() if not k.startswith("copy")), 4)
    print("total", res["total_ms"])
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
