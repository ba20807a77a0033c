"""Where a dense_layer launch (csrc/dense.hip) spends its time: per-wave s_memrealtime stamps (100 MHz) from a library built
with -DOSSID_DENSE_TIMING. Stamps: 0 entry, 1 all loads issued, 2 patch staged (barrier passed), 3 3x3 MFMAs done, 4 slab
published, 5 shares' stores issued, 6 stores acknowledged.
  OSSID_HIPCC_EXTRA=-DOSSID_DENSE_TIMING python -c "from ossid_code_amd import _build; _build.build_lib(force=True)"
  python tools/dense_timeline.py --block b3 --layer 0"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import _lib  # noqa: E402
from ossid_code_amd.dtoid import ops  # noqa: E402
from ossid_code_amd.dtoid.backbones import DenseBlock  # noqa: E402

SHAPES = {"b1": (64, 6, 120, 160), "b2": (128, 12, 60, 80), "b3": (256, 24, 30, 40), "b4": (512, 16, 29, 39)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block", default="b3")
    ap.add_argument("--layer", type=int, default=0)
    a = ap.parse_args()
    C0, L, H, W = SHAPES[a.block]
    blk = DenseBlock(L, C0).cuda().eval()
    P = ops.PackedConv
    layers = [(P(l.conv1, pre_bn=l.norm1, pre_relu=True), P(l.conv2, pre_bn=l.norm2, pre_relu=True)) for l in blk.values()]
    table = ops.dense_block_table(layers, 32)
    ctot = C0 + 32 * L
    buf = torch.randn(1, ctot, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.randn(L, H * W, 128, device="cuda")
    c2 = layers[a.layer][1]
    lib = _lib.lib()
    fn = lib.ossid_dense_debug_stamps
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_size_t], ctypes.c_int

    def run():
        _lib.check(_lib.fn("ossid_dense_layer")(y.data_ptr(), buf.data_ptr(), 1, H, W, ctot, C0, a.layer, L, c2.wpk.data_ptr(),
                                                c2.pre_scale.data_ptr(), c2.pre_shift.data_ptr(), table.data_ptr(), _lib.stream()),
                   "ossid_dense_layer")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    host = np.zeros(8192 * 8, dtype=np.uint64)
    # a few filler launches in front so that the timed one starts on a busy queue, as inside the block
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert fn(host.ctypes.data, host.nbytes) == 0
    r = host.reshape(-1, 8).astype(np.int64)
    r = r[r[:, 0] != 0]
    base = r[:, 0].min()
    names = ["entry", "loads issued", "patch staged", "3x3 done", "slab published", "shares stored", "stores acked"]
    print("%d waves; times in us from the first wave's entry (100 MHz clock): min / median / max" % len(r))
    for i, nm in enumerate(names):
        col = r[:, i]
        col = col[col != 0]
        if len(col) == 0:
            continue
        d = (col - base) / 100.0
        print("  %-16s %6.2f %6.2f %6.2f   (%d waves)" % (nm, d.min(), np.median(d), d.max(), len(col)))
    print("per-wave phase lengths (median us): " + ", ".join(
        "%s %.2f" % (names[i + 1], np.median((r[:, i + 1] - r[:, i])[(r[:, i + 1] != 0) & (r[:, i] != 0)]) / 100.0)
        for i in range(6) if ((r[:, i + 1] != 0) & (r[:, i] != 0)).any()))


if __name__ == "__main__":
    main()
