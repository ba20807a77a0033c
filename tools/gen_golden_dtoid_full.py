"""Writes tests/golden/dtoid_head_full.npz: an EVAL-mode capture of the reference's test-time head at the REAL sizes
(480x640 image, 29x39 feature grid; SURVEY.md 8c "at full 29x39 for shape truth"), produced by running the reference's
own classes and its own `Network.forward_all_templates` (models/dtoid/network.py:473-581) in the build container.

What runs from /root/reference: CorrelationModel, ClassificationModel, RegressionModel, BBoxTransform, ClipBoxes,
generate_anchors/shift and -- unbound, on a holder object -- the body of Network.forward_all_templates (template loop,
torch.cat bookkeeping, obj_indices, decode+clip, view reshapes, topk 1000, NMS call, [:topk], seg/heat gather).
What does NOT (absent offline): the image backbone (torchvision DenseNet) -- the holder's
`image_feature_extractor` returns the seeded feature map -- and `torchvision.ops.boxes.nms`, for which the greedy NMS of
oracle/dtoid_oracle.py (torchvision's published algorithm) is installed on the placeholder module; `Anchors.forward`
(unconditional .cuda(), anchors.py:42) is replaced by the same two reference functions it calls, on the CPU.

Only data is stored (inputs re-derived from seeds at test time; dense outputs stored whole where small, strided where
large). Run from the repo root:   python tools/gen_golden_dtoid_full.py

`--nt21` writes tests/golden/dtoid_head_full_nt21.npz instead: BASELINE configs[2]'s 21 templates in ONE chunk (the chunk
size of DtoidNet.forwardTestTime is 120, models/dtoid/__init__.py:92), the template count at which the product's own
dispatch picks the 128-channel Winograd workgroups, the tail split and the direct `dot` convolution -- with dense scores /
deltas stored row-strided (the whole tensors would be 14 MB) and the reference's detection list beside them.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402
from gen_golden_dtoid import seeded_state  # noqa: E402

IMG = (480, 640)
GRID = (29, 39)
CHUNKS = (2, 1)          # template chunks, as DtoidNet.forwardTestTime hands them over (3 templates in all)
SEED = 4321
TOPK = 20
X2_CH_STRIDE, SEG_STRIDE, SEGPRED_STRIDE = 8, 4, 8


# the 21-template variant: (chunks, x2 channel stride, seg pixel stride, post_seg pixel stride, cls / reg row stride)
NT21 = dict(chunks=(21,), x2s=64, segs=8, sps=8, rows=13, seed=SEED + 100, name="dtoid_head_full_nt21.npz")


def seeded_inputs(seed=SEED + 10, chunks=CHUNKS):
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(1, 640, *GRID, generator=g)
    tmpl = [torch.randn(n, 640, 7, 7, generator=g) for n in chunks]
    return feat, tmpl


def main(variant=None):
    network, loss_mod, anchors_mod, utils = ref_import.load()
    from oracle import dtoid_oracle
    sys.modules["torchvision.ops.boxes"].nms = lambda b, s, t: dtoid_oracle.nms(b, s, t)
    network.torchvision = sys.modules["torchvision"]
    torch.manual_seed(0)
    corr = network.CorrelationModel(IMG, 640)
    cls = network.ClassificationModel(512, num_anchors=24)
    reg = network.RegressionModel(512, num_anchors=24)
    for i, m in enumerate((corr, cls, reg)):
        m.load_state_dict(seeded_state(m, SEED + i))
        m.eval()
    feat, tmpl = seeded_inputs() if variant is None else seeded_inputs(variant["seed"], variant["chunks"])
    base = anchors_mod.generate_anchors(base_size=30, ratios=np.array([0.5, 1, 2]), scales=np.array([1, 2, 3, 4, 5, 6, 7, 8]))

    class Holder:       # the attributes Network.forward_all_templates reads from `self`
        pass
    h = Holder()
    h.image_feature_extractor = lambda image, g: feat
    h.correlation_model, h.classification, h.regression = corr, cls, reg
    h.anchors = lambda shapes: torch.from_numpy(anchors_mod.shift(tuple(shapes[0]), 16, base).astype(np.float32))[None]
    h.regressBoxes = network.BBoxTransform(mean=torch.zeros(4), std=torch.tensor([0.1, 0.1, 0.2, 0.2]))
    h.clipBoxes = network.ClipBoxes()
    image = torch.zeros(1, 3, *IMG)
    score, boxes, obj, seg_pred, heat_pred = network.Network.forward_all_templates(
        h, image, tmpl, [torch.zeros(1, 64, 3, 3)], topk=TOPK)
    # the dense tensors behind it, template by template (same modules, same inputs)
    with torch.no_grad():
        x2, heat, seg, c, r = [], [], [], [], []
        for t in tmpl:
            a, b, s = corr(feat.expand(t.shape[0], -1, -1, -1), t, True)
            x2.append(a), heat.append(b), seg.append(s)
            c.append(cls(a)[0]), r.append(reg(a))
        x2, heat, seg, c, r = (torch.cat(v, 0) for v in (x2, heat, seg, c, r))
    x2s, segs, sps, rows, chunks, name = (X2_CH_STRIDE, SEG_STRIDE, SEGPRED_STRIDE, 1, CHUNKS, "dtoid_head_full.npz") \
        if variant is None else tuple(variant[k] for k in ("x2s", "segs", "sps", "rows", "chunks", "name"))
    out = dict(
        x2=x2[:, ::x2s].numpy(), heat=heat.numpy(), seg=seg[:, :, ::segs, ::segs].numpy(),
        cls=c[:, ::rows].numpy(), reg=r[:, ::rows].numpy(),
        post_score=score.numpy(), post_boxes=boxes.numpy(), post_obj=obj.numpy(),
        post_seg=seg_pred[:, ::sps, ::sps].numpy(), post_heat=heat_pred.numpy(),
        seed=SEED, topk=TOPK, chunks=np.asarray(chunks))
    if variant is not None:
        out.update(input_seed=variant["seed"], strides=np.asarray([x2s, segs, sps, rows]))
    path = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print(path, os.path.getsize(path), "bytes; kept", len(score), "boxes; top score", float(score[0]),
          "templates fired", sorted(set(obj.reshape(-1).tolist())))


if __name__ == "__main__":
    main(NT21 if "--nt21" in sys.argv else None)
