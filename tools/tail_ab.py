"""Timing of the fused decoder tail alone (library chosen with OSSID_HIP_LIB)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ossid_code_amd.dtoid import ops
c1, bn, c2 = torch.nn.Conv2d(32, 16, 3, padding=1).cuda(), torch.nn.BatchNorm2d(16).cuda().eval(), torch.nn.Conv2d(16, 1, 3, padding=1).cuda()
tail = ops.SegTail(c1, bn, c2)
x = torch.randn(21, 32, 232, 312, device="cuda").contiguous(memory_format=torch.channels_last)
for _ in range(5):
    tail(x, size=(480, 640))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tail(x, size=(480, 640))
torch.cuda.synchronize()
print("%s tail ms %.4f" % (os.environ.get("OSSID_HIP_LIB", "default"), (time.perf_counter() - t0) / 20 * 1e3))
