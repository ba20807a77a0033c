"""Where a conv_nhwc launch spends its time, from per-wave s_memrealtime stamps (100 MHz) (library built with -DOSSID_TIMING):
entry / main loop start / main loop end / exit + HW_ID of every wave, grouped per SIMD.
  OSSID_HIPCC_EXTRA=-DOSSID_TIMING python -c "from ossid_code_amd import _build; _build.build_lib(force=True)"
  python tools/conv_timeline.py --cin 768 --cout 512 --hw 29 39 --batch 8 --taps 9"""
import argparse
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd.dtoid import train_ops as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=768)
    ap.add_argument("--cout", type=int, default=512)
    ap.add_argument("--hw", type=int, nargs=2, default=[29, 39])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--taps", type=int, default=9)
    ap.add_argument("--pre", action="store_true")
    ap.add_argument("--wino", action="store_true", help="the Winograd kernel (csrc/wino.hip) instead of the direct one")
    a = ap.parse_args()
    B, cin, cout, (H, W), taps = a.batch, a.cin, a.cout, a.hw, a.taps
    k = 3 if taps == 9 else 1
    x = torch.randn(B, cin, H, W).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k) * 0.05).cuda()
    wpk = T._pack(w, "wino_fwd" if a.wino else "fwd")
    out = T.empty_nhwc(B, cout, H, W, "cuda")
    pre = (torch.rand(cin).cuda() + 0.5, torch.randn(cin).cuda() * 0.1) if a.pre else None
    buf = torch.zeros(6 << 20, dtype=torch.int64, device="cuda")
    tb = buf.view(torch.float32)
    run = lambda: T.conv_raw(x, wpk, B, H, W, cin, cout, taps, out, pre=pre, pre_relu=a.pre, epi={"timing_buf": tb}, wino=a.wino)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    r = buf.cpu().numpy().reshape(-1, 6)
    r = r[r[:, 0] != 0]
    if os.environ.get("RAW"):
        print(r[:6]); print(r[-3:])
    t0, t1, t2, t3, hw, xcc = (r[:, i].astype(np.int64) for i in range(6))
    if os.environ.get("RAW"):
        for xc in np.unique(xcc):
            m = xcc == xc
            print("xcc", xc, "n", m.sum(), "t0 pct", np.percentile(t0[m], [0, 1, 50, 99, 100]).astype(np.int64), "t3 pct", np.percentile(t3[m], [0, 1, 50, 99, 100]).astype(np.int64))
    cyc = xcc >> 8
    xcc = xcc & 0xFF
    ghz = cyc / np.maximum(t2 - t1, 1) / 10.0
    print("shader clock over the main loops (s_memtime / s_memrealtime): mean %.3f GHz  p10 %.3f  p90 %.3f" % (ghz.mean(), np.percentile(ghz, 10), np.percentile(ghz, 90)))
    base = t0.min()
    span = t3.max() - base
    flops = 2.0 * B * H * W * cin * cout * taps
    print("waves %d (workgroups %d); span %d ticks; event time %.3f ms -> %.1f ticks/us; %.1f TFLOP/s over the span" %
          (len(r), len(r) // 4, span, ms, span / (ms * 1e3), flops / (ms * 1e-3) / 1e12))
    tick_us = 100.0                       # s_memrealtime: 100 MHz
    for name, d in (("prologue", t1 - t0), ("main loop", t2 - t1), ("epilogue", t3 - t2), ("whole wave", t3 - t0)):
        print("  %-10s mean %8.1f us  p10 %8.1f  p90 %8.1f" % (name, d.mean() / tick_us, np.percentile(d, 10) / tick_us, np.percentile(d, 90) / tick_us))
    simd = (xcc & 0xF) * (1 << 16) + ((hw >> 4) & 0xFFF)          # HW_ID without the wave slot: simd, pipe, cu, sh, se
    per = collections.defaultdict(list)
    for i in range(len(r)):
        per[int(simd[i])].append(i)
    print("SIMDs seen %d, waves per SIMD: mean %.2f min %d max %d" % (len(per), len(r) / len(per), min(map(len, per.values())), max(map(len, per.values()))))

    def union(iv):
        iv = sorted(iv)
        tot, cs, ce = 0, None, None
        for s, e in iv:
            if cs is None:
                cs, ce = s, e
            elif s <= ce:
                ce = max(ce, e)
            else:
                tot += ce - cs
                cs, ce = s, e
        return tot + (ce - cs if cs is not None else 0)

    res, loop, loopsum, first, last = [], [], [], [], []
    for k2, idx in per.items():
        res.append(union([(t0[i], t3[i]) for i in idx]))
        loop.append(union([(t1[i], t2[i]) for i in idx]))
        loopsum.append(sum(t2[i] - t1[i] for i in idx))
        first.append(min(t0[i] for i in idx) - base)
        last.append(span - (max(t3[i] for i in idx) - base))
    res, loop, loopsum, first, last = map(np.array, (res, loop, loopsum, first, last))
    print("per SIMD, as a fraction of the span: some wave resident %.3f | some wave in its main loop %.3f | idle before the first wave %.3f | idle after the last %.3f" %
          (res.mean() / span, loop.mean() / span, first.mean() / span, last.mean() / span))
    print("mean waves concurrently in the main loop while any is: %.2f" % (loopsum.sum() / loop.sum()))
    # MFMA time owed per SIMD: every wave issues the same count
    n_mfma_wave = flops / 4096.0 / len(r) * (16.0 / 36.0 if a.wino else 1.0)     # Winograd: 16 of the 36 multiplies
    # s_memtime runs at a fixed 100 MHz-class clock on some parts: derive shader cycles from the event time at 2.4 GHz
    cyc_per_tick = ghz.mean() * 1000.0 / tick_us
    owed = np.array([len(idx) for idx in per.values()]) * n_mfma_wave * 64.0 / cyc_per_tick
    print("MFMA pipe time owed / time some wave is in its main loop (per SIMD mean): %.3f   (ticks -> cycles x %.2f at the measured clock)" %
          ((owed / loop).mean(), cyc_per_tick))


if __name__ == "__main__":
    main()
