"""Times ossid_conv_wgrad_group on the 3x3 problems of each dense block at the finetune batch (A/B: OSSID_WGRAD_FEWCH=1 keeps
them on the general grouped kernel, 3 = default sends them to csrc/wgrad_fc.hip).  python tools/wgrad_group_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd.dtoid import train_ops as T  # noqa: E402


def main():
    B = 8
    for name, H, W, L, C0 in (("b1", 120, 160, 6, 64), ("b2", 60, 80, 12, 128), ("b3", 30, 40, 24, 256), ("b4", 29, 39, 16, 512)):
        Ct = C0 + 32 * L
        G = torch.randn(B, Ct, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        items = []
        for li in range(L):
            c = C0 + 32 * li
            y1 = torch.randn(B, 128, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
            items.append(dict(x=y1, dy=T.flat(G, c), B=B, H=H, W=W, cin=128, cout=32, taps=9, dw=torch.empty(32, 128, 3, 3, device="cuda"),
                              pre=(torch.rand(128, device="cuda") + 0.5, torch.randn(128, device="cuda")), pre_relu=True, dy_cs=Ct))
        for _ in range(3):
            T.wgrad_group(items)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            T.wgrad_group(items)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = L * 2.0 * B * H * W * 32 * 128 * 9
        print("%s  %2d layers  %7.3f ms  %6.1f TFLOP/s" % (name, L, ms, fl / ms / 1e9), flush=True)
        # the same block's 1x1 layers (c -> 128 on the channel prefix of the resident buffer)
        buf = torch.randn(B, Ct, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        items, fl = [], 0.0
        for li in range(L):
            c = C0 + 32 * li
            dz = torch.randn(B, 128, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
            f = torch.rand(2, c, device="cuda") + 0.5
            items.append(dict(x=buf, dy=dz, B=B, H=H, W=W, cin=c, cout=128, taps=1, dw=torch.empty(128, c, 1, 1, device="cuda"),
                              pre=(f[0], f[1] - 1.0), pre_relu=True, in_cs=Ct))
            fl += 2.0 * B * H * W * 128 * c
        for _ in range(3):
            T.wgrad_group(items)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            T.wgrad_group(items)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%s  %2d layers 1x1  %7.3f ms  %6.1f TFLOP/s" % (name, L, ms, fl / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
