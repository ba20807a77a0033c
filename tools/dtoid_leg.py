"""Runs exactly `--calls` calls of ONE DTOID bench leg (the shapes of bench.py's dtoid object), eagerly, for profiling:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- python3 tools/dtoid_leg.py --leg forward --calls 3
legs: forward (1 image x 21 templates), forward_batch (32 x 21), forward_pairs (32 pairs), finetune (batch 8 step).
The first call fills the template cache, packs the weights and records every launch plan; a marker launch (torch.cuda._sleep's
spin kernel) in front of every call lets tools/pmc_dtoid_traffic.py take the LAST call alone."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ossid_code_amd import dtoid  # noqa: E402
from ossid_code_amd.dtoid import finetune  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", required=True, choices=("forward", "forward_batch", "forward_pairs", "finetune"))
    ap.add_argument("--calls", type=int, default=3)
    ap.add_argument("--nt", type=int, default=21)
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda")
    cfg = dtoid.DtoidConfig()
    torch.manual_seed(0)
    m = dtoid.DtoidNet(cfg).to(dev).eval()
    m.model.use_graph = False                       # eager: a graph replay makes no per-kernel records
    g = torch.Generator().manual_seed(1)
    nt, B32, B = a.nt, a.images, a.batch
    test = {"img": torch.rand(1, 3, 480, 640, generator=g).to(dev), "obj_id": torch.tensor([1]),
            "limg": torch.rand(1, nt, 3, 124, 124, generator=g).to(dev),
            "lmask": (torch.rand(1, nt, 1, 124, 124, generator=g) > 0.5).float().to(dev)}
    if a.leg == "forward":
        fn = lambda: m.forwardTestTime(test)                                          # noqa: E731
    elif a.leg == "forward_batch":
        test32 = dict(test, img=torch.rand(B32, 3, 480, 640, generator=g).to(dev))
        fn = lambda: m.forwardTestTimeBatch(test32)                                   # noqa: E731
    elif a.leg == "forward_pairs":
        pairs = [torch.rand(B32, 3, 480, 640, generator=g), torch.rand(B32, 3, 124, 124, generator=g),
                 (torch.rand(B32, 1, 124, 124, generator=g) > 0.5).float(), torch.rand(B32, 3, 124, 124, generator=g),
                 (torch.rand(B32, 1, 124, 124, generator=g) > 0.5).float()]
        pairs = [dtoid.normalizeImageRange(p.to(dev)) if p.shape[1] == 3 else p.to(dev) for p in pairs]

        def fn():
            with torch.no_grad():
                m.model(*pairs)
    else:
        flat = finetune.FlatParams(m)
        opt = finetune.FusedAMSGrad(flat, lr=1e-4, weight_decay=1e-6)
        mask = torch.zeros(B, 1, 480, 640)
        mask[:, :, 120:240, 160:320] = 1
        batch = {"img": torch.rand(B, 3, 480, 640, generator=g), "limg": torch.rand(B, 3, 124, 124, generator=g),
                 "lmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
                 "gimg": torch.rand(B, 3, 124, 124, generator=g),
                 "gmask": (torch.rand(B, 1, 124, 124, generator=g) > 0.5).float(),
                 "bbox_gt": torch.tensor([[[160.0, 120.0, 320.0, 240.0, 1.0]]]).repeat(B, 1, 1),
                 "heatmap": torch.rand(B, 1, 29, 39, generator=g).double(), "mask": mask}
        batch = {k: v.to(dev) for k, v in batch.items()}
        m.train()
        fn = lambda: finetune.finetune_step(m, batch, opt)                            # noqa: E731
    for _ in range(a.calls + 1):
        torch.cuda._sleep(1000)                      # marker launch (spin kernel): tools/pmc_dtoid_traffic.py cuts the run at these
        fn()
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    print("done", a.leg, a.calls + 1)


if __name__ == "__main__":
    main()
