"""A/B of the Winograd launch's tail split (csrc/wino.hip wino_plan_tail): the same layer with and without the scratch
buffer that enables it.  python tools/wino_tail_ab.py [B Cin Cout H W]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import _lib  # noqa: E402
from ossid_code_amd.dtoid import ops, train_ops as T  # noqa: E402

shapes = [(21, 768, 512, 29, 39), (21, 640, 256, 29, 39), (21, 512, 512, 29, 39), (21, 256, 256, 29, 39), (8, 768, 512, 29, 39),
          (8, 256, 256, 29, 39), (8, 512, 256, 29, 39)]
if len(sys.argv) == 6:
    shapes = [tuple(int(v) for v in sys.argv[1:])]
for B, Cin, Cout, H, W in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wpk = T._pack(w, "wino_fwd")
    out = [T.empty_nhwc(B, Cout, H, W, x.device) for _ in range(2)]

    def run(split, o):
        d = _lib.ConvDesc()
        d.x, d.wpk, d.out = x.data_ptr(), wpk.data_ptr(), o.data_ptr()
        d.in_batch_stride = -1
        d.batch, d.height, d.width, d.cin, d.cout, d.taps = B, H, W, Cin, Cout, 9
        if split:
            ops.wino_workspace((d,), x.device)
        _lib.check(_lib.fn("ossid_conv3x3_wino_fwd")(T._byref(d), _lib.stream()), "wino")
        return d
    res = {}
    for split in (False, True):
        d = run(split, out[split])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(split, out[split])
        e1.record()
        torch.cuda.synchronize()
        res[split] = e0.elapsed_time(e1) / 20
    need = _lib.fn("ossid_conv3x3_wino_workspace_bytes")(T._byref(d))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    err = [float((o.double() - ref).abs().max() / ref.abs().max()) for o in out]
    print("B %2d %4d->%4d %dx%d  whole %.3f ms  split %.3f ms  (scratch %.1f MB)  err %.1e %.1e" %
          (B, Cin, Cout, H, W, res[False], res[True], need / 1e6, err[0], err[1]))
