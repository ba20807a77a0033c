"""A/B timing of the hand-written conv kernel on the DTOID head/decoder shapes (one library per process; choose it with
OSSID_HIP_LIB=path/to/libossid_hip.so).  python tools/conv_ab.py [--nt 21]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def timeit(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=21)
    a = ap.parse_args()
    from ossid_code_amd.dtoid import ops
    shapes = [("corr 640->256", 640, 256, 29, 39), ("cf 768->512", 768, 512, 29, 39), ("dec 512->256", 512, 256, 29, 39),
              ("trunk 256->256", 256, 256, 29, 39), ("cls 256->48", 256, 48, 29, 39), ("reg 256->96", 256, 96, 29, 39),
              ("dec 256->128 @58x78", 256, 128, 58, 78), ("dec 128->64 @116x156", 128, 64, 116, 156),
              ("dec 64->32 @232x312", 64, 32, 232, 312), ("dec 32->16 @480x640", 32, 16, 480, 640)]
    out = {}
    tot = 0.0
    for name, ci, co, h, w in shapes:
        B = a.nt
        x = torch.randn(B, ci, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
        cv = torch.nn.Conv2d(ci, co, 3, padding=1).cuda()
        if os.environ.get("OSSID_AB_ZEROS"):          # all-zero operands: how much of the rate is a power / clock matter?
            x.zero_()
            with torch.no_grad():
                cv.weight.zero_()
        pk = ops.PackedConv3x3(cv)
        t = timeit(lambda: pk(x))
        fl = 2.0 * B * h * w * co * ci * 9
        out[name] = {"ms": round(t * 1e3, 4), "TF": round(fl / t / 1e12, 1)}
        tot += t
    from ossid_code_amd import _lib
    if hasattr(ops, "SegTail") and hasattr(_lib.lib(), "ossid_seg_tail_fwd"):
        c1, bn, c2 = torch.nn.Conv2d(32, 16, 3, padding=1).cuda(), torch.nn.BatchNorm2d(16).cuda().eval(), \
            torch.nn.Conv2d(16, 1, 3, padding=1).cuda()
        tail = ops.SegTail(c1, bn, c2)
        x = torch.randn(a.nt, 32, 232, 312, device="cuda").contiguous(memory_format=torch.channels_last)
        t = timeit(lambda: tail(x, size=(480, 640)))
        fl = 2.0 * a.nt * 480 * 640 * 9 * (16 * 32 + 16)
        out["tail 32->16->1 fused @480x640"] = {"ms": round(t * 1e3, 4), "TF": round(fl / t / 1e12, 1)}
    out["sum_ms"] = round(tot * 1e3, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
