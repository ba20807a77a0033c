"""Diagnosis of round 1's host SIGSEGV (gpurun_out/prof_dtoid_fwd3.log: launch_conv -> hipLaunchKernel -> crash, during a
torch.cuda.graph capture under `rocprofv3 --kernel-trace`). Run it once plainly and once under rocprofv3:

    python3 tools/diag_capture_under_profiler.py
    rocprofv3 --kernel-trace -d gpurun_out/diag_prof -- python3 tools/diag_capture_under_profiler.py

It prints (1) which copies of the HIP / HSA runtimes and of the rocprofiler libraries the process has mapped, (2) whether
a hand-written convolution launched into a CAPTURING stream works (capture, replay, compare with the eager result).
Each step is printed before it runs, so the log of a crash names the step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def say(*a):
    print("[diag]", *a, flush=True)


def mapped():
    libs = set()
    for line in open("/proc/self/maps"):
        path = line.strip().split()[-1]
        if any(k in path for k in ("amdhip64", "hsa-runtime", "rocprofiler", "roctracer", "libossid")):
            libs.add(path)
    return sorted(libs)


def main():
    say("LD_PRELOAD =", os.environ.get("LD_PRELOAD"), "| ROCP_TOOL_LIBRARIES =", os.environ.get("ROCP_TOOL_LIBRARIES"))
    from ossid_code_amd.dtoid import network, ops
    say("Network.under_profiler =", network.Network.under_profiler, "| use_graph =", network.Network.use_graph)
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(256, 256, 3, padding=1).cuda()
    pk = ops.PackedConv3x3(conv, act=True)
    x = torch.randn(4, 256, 29, 39, device="cuda")
    say("eager launch")
    ref = pk(x)
    torch.cuda.synchronize()
    for p in mapped():
        say("mapped:", p)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        say("warm-up on a side stream")
        pk(x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    say("capture begins")
    with torch.cuda.graph(graph):
        say("launch into the capturing stream")
        out = pk(x)
        say("launched")
    say("capture ended; replay")
    graph.replay()
    torch.cuda.synchronize()
    say("replayed; max |graph - eager| =", float((out - ref).abs().max()))
    say("OK")


if __name__ == "__main__":
    main()
