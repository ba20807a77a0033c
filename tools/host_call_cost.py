"""Host-side cost of one call across the C ABI through ctypes (the binding of ossid_code_amd/_lib.py), measured without
waiting for the GPU: the time to ENQUEUE n launches of a tiny kernel, per launch -- next to the same for a torch op and
for the full Python wrapper (train_ops.chan_op: descriptor fill + call). INTEGRATION.md quotes these numbers."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import _lib  # noqa: E402
from ossid_code_amd.dtoid import train_ops as T  # noqa: E402


def per_call(fn, n=2000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return t * 1e6


def main():
    x = torch.randn(64, 32, device="cuda")
    out = torch.empty_like(x)
    d = _lib.ChanOpDesc()
    d.g, d.out, d.n_rows, d.channels = x.data_ptr(), out.data_ptr(), 64, 32
    f = _lib.fn("ossid_chan_op")
    stream = _lib.stream()
    res = {
        "ctypes_raw_call_us": per_call(lambda: f(ctypes.byref(d), stream)),
        "ctypes_call_plus_stream_lookup_us": per_call(lambda: f(ctypes.byref(d), _lib.stream())),
        "python_wrapper_chan_op_us": per_call(lambda: T.chan_op(x, 64, 32, out=out)),
        "torch_tiny_op_us": per_call(lambda: torch.add(x, 1.0, out=out)),
    }
    print(json.dumps(res))


if __name__ == "__main__":
    main()
