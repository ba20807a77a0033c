"""The fused decoder tail (ossid_seg_tail_fwd) at the test-time size -- 21 templates, 232 x 312 -> 480 x 640 -- against float64:
error and time per launch. Run on the default build and on -DOSSID_SEGTAIL_F32 (exact-f32 first convolution)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from ossid_code_amd.dtoid import ops

torch.manual_seed(0)
c1 = torch.nn.Conv2d(32, 16, 3, padding=1).cuda()
bn = torch.nn.BatchNorm2d(16).cuda().eval()
c2 = torch.nn.Conv2d(16, 1, 3, padding=1).cuda()
x = torch.randn(21, 32, 232, 312, device="cuda")
tail = ops.SegTail(c1, bn, c2)
with torch.no_grad():
    got = tail(x, size=(480, 640))
    up = F.interpolate(x.double(), size=(480, 640))
    mid = F.elu(F.conv2d(up, c1.weight.double(), c1.bias.double(), padding=1))
    mid = (mid - bn.running_mean.double()[None, :, None, None]) / torch.sqrt(bn.running_var.double()[None, :, None, None] + bn.eps)
    mid = mid * bn.weight.double()[None, :, None, None] + bn.bias.double()[None, :, None, None]
    w64 = F.conv2d(mid, c2.weight.double(), c2.bias.double(), padding=1)
    print("max err vs f64: %.3e (scale %.2f)" % (float((got.double() - w64).abs().max()), float(w64.abs().max())))
    for _ in range(3):
        tail(x, size=(480, 640))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        tail(x, size=(480, 640))
    torch.cuda.synchronize()
    print("seg_tail 21 x 480x640: %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
