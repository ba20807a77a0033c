"""A DenseNet-121 block at test time, alone on the chip, replayed from a hipGraph: the one-launch-per-layer form
(csrc/dense.hip) against the two-launch form (csrc/conv.hip), us per block.
  python tools/dense_bench.py [--blocks b2,b3,b4] [--reps 200]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {"b1": (64, 6, 120, 160), "b2": (128, 12, 60, 80), "b3": (256, 24, 30, 40), "b4": (512, 16, 29, 39)}


def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", default="b2,b3,b4")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--eager", action="store_true", help="no graph: a few eager passes of the fused form only (for rocprofv3 --kernel-trace)")
    a = ap.parse_args()
    from ossid_code_amd.dtoid import ops
    from ossid_code_amd.dtoid.backbones import DenseBlock
    out = {}
    for name in a.blocks.split(","):
        C0, L, H, W = SHAPES[name]
        B = a.batch
        torch.manual_seed(0)
        blk = DenseBlock(L, C0).cuda().eval()
        P = ops.PackedConv
        layers = [(P(l.conv1, pre_bn=l.norm1, pre_relu=True), P(l.conv2, pre_bn=l.norm2, pre_relu=True)) for l in blk.values()]
        table = ops.dense_block_table(layers, 32)
        ctot = C0 + 32 * L
        buf = torch.randn(B, ctot, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        tmp = torch.empty((B, 128, H, W), device="cuda").contiguous(memory_format=torch.channels_last)

        def two():
            c = C0
            for c1, c2 in layers:
                c1.run(buf, B, H, W, tmp, in_cs=ctot)
                c2.run(tmp, B, H, W, buf, out_cs=ctot, out_coff=c)
                c += 32

        def fused():
            ops.dense_block_fused(buf, B, H, W, C0, layers, table)
        if a.eager:
            with torch.no_grad():
                for _ in range(5):
                    fused()
            torch.cuda.synchronize()
            continue
        with torch.no_grad():
            g2, gf = graphed(two), graphed(fused)
            out[name] = {"two_launch_us": wall(g2.replay, a.reps), "fused_us": wall(gf.replay, a.reps), "layers": L}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
