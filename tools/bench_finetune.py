"""Times the DTOID finetune step (DtoidNet.forward + loss + backward + fused AMSGrad) at a given batch, eager and
hipGraph-replayed, on the hand-written training kernels or the nn.Module path.  python tools/bench_finetune.py --impl hip"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import dtoid  # noqa: E402
from ossid_code_amd.dtoid import finetune  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="hip")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--phases", action="store_true", help="also time forward and backward separately (eager)")
    ap.add_argument("--with-inference", action="store_true",
                    help="run a test-time frame first (captures the dense-forward hipGraph in this process), as the online stream does")
    ap.add_argument("--sync-each", action="store_true",
                    help="synchronise after every step (a loop that reads the loss every step: the host never runs ahead)")
    ap.add_argument("--main-priority", type=int, default=None,
                    help="run the step on a stream of this priority instead of the default stream (-1 = high): do the side "
                         "streams' kernels then stay out of the critical path's way?")
    a = ap.parse_args()
    cfg = dtoid.DtoidConfig()
    torch.manual_seed(0)
    m = dtoid.DtoidNet(cfg).cuda().train()
    m.model.use_hip_training = a.impl == "hip"
    flat = finetune.FlatParams(m)
    opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
    g = torch.Generator().manual_seed(1)
    B = a.batch
    mask = torch.zeros(B, 1, 480, 640)
    mask[:, :, 120:240, 160:320] = 1
    batch = {"img": torch.rand(B, 3, 480, 640, generator=g), "limg": torch.rand(B, 3, 124, 124, generator=g),
             "lmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
             "gimg": torch.rand(B, 3, 124, 124, generator=g),
             "gmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
             "bbox_gt": torch.tensor([[[160.0, 120.0, 320.0, 240.0, 1.0]]]).repeat(B, 1, 1),
             "heatmap": torch.rand(B, 1, 29, 39, generator=g).double(), "mask": mask}
    batch = {k: v.cuda() for k, v in batch.items()}

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    res = {"impl": a.impl, "batch": B}
    if a.with_inference:
        m.eval()
        test = {"img": torch.rand(1, 3, 480, 640, generator=g).cuda(), "obj_id": torch.tensor([1]),
                "limg": torch.rand(1, 21, 3, 124, 124, generator=g).cuda(),
                "lmask": (torch.rand(1, 21, 1, 124, 124, generator=g) > 0.5).float().cuda()}
        with torch.no_grad():
            for _ in range(3):
                m.forwardTestTime(test)
        torch.cuda.synchronize()
        m.train()
    if a.main_priority is not None:
        hp = torch.cuda.Stream(priority=a.main_priority)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
        res["main_priority"] = a.main_priority
    if a.sync_each:
        def one():
            float(finetune.finetune_step(m, batch, opt))          # .item(): the host waits for the step
        res["eager_sync_each_ms"] = timed(one, 2, a.reps)
    res["eager_ms"] = timed(lambda: finetune.finetune_step(m, batch, opt), 2, a.reps)
    # host time to ENQUEUE one eager step (no synchronisation inside): the eager step is launch-bound once this
    # approaches the device time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    finetune.finetune_step(m, batch, opt)
    res["eager_host_enqueue_ms"] = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    if a.phases:
        def fwd():
            return m(batch)["loss"]
        res["forward_ms"] = timed(fwd, 1, a.reps)

        def fb():
            flat.detach_grads()
            fwd().backward()
        res["forward_backward_ms"] = timed(fb, 1, a.reps)
    if not a.no_graph:
        graphed = finetune.GraphedForwardBackward(m, flat, batch)
        res["graph_ms"] = timed(lambda: finetune.finetune_step(m, batch, opt, graphed=graphed), 2, a.reps)
        res["tflops_nominal"] = B * 258e9 / res["graph_ms"] / 1e9
    res["loss"] = float(finetune.finetune_step(m, batch, opt))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
