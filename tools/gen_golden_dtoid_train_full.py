"""Writes tests/golden/dtoid_head_train_full.npz: a TRAIN-mode forward + backward of the reference's DTOID head at the REAL
sizes of the finetune step (480x640 image, 29x39 feature grid, batch 8 = `--finetune_batch_size` default,
scripts/online_learning.py:708), produced by running the reference's own CorrelationModel / ClassificationModel /
RegressionModel (models/dtoid/network.py:96-157, 282-371), DetectionLoss (models/dtoid/loss.py:46-175) and the loss
weighting of DtoidNet.forward (models/dtoid/__init__.py:213-221) in the build container (SURVEY.md 8c recipe).

Why a second training fixture: tests/golden/dtoid_head.npz is a 4x5 grid at batch 2, where none of the product's tilings
(Winograd form, split-K plans, grouped weight gradients, few-channel decoder tilings) is the dispatcher's own choice.
At 29x39 / batch 8 they are, with no threshold overridden.

Only data is stored: inputs and weights are re-derived from seeds at test time; outputs are stored strided where large,
every convolution weight gradient as a flat strided sample, bias / BatchNorm-parameter gradients and every BatchNorm
running statistic whole. Run from the repo root (~1 min of CPU):   python tools/gen_golden_dtoid_train_full.py
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
B = 8
SEED = 8642
# how the large tensors are thinned (the test applies the same slices)
X2_CH, SEG_PX, CLS_ROW, GF_CH, GT_CH = 16, 8, 7, 16, 4
W_STRIDE = 251            # flat stride over a convolution weight gradient (prime: walks every (co, ci, tap) residue)


def seeded_inputs(seed=SEED + 10):
    """feat [B,640,29,39], tmpl [B,640,7,7], one box per image (x1,y1,x2,y2,class 1), heat-map target (float64, as the
    reference's dataset hands it over), binary mask target [B,1,480,640]."""
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(B, 640, *GRID, generator=g)
    tmpl = torch.randn(B, 640, 7, 7, generator=g)
    cx = 80 + 480 * torch.rand(B, generator=g)
    cy = 60 + 360 * torch.rand(B, generator=g)
    w = 40 + 160 * torch.rand(B, generator=g)
    h = 40 + 160 * torch.rand(B, generator=g)
    ann = torch.stack([(cx - w / 2).clamp(min=0), (cy - h / 2).clamp(min=0), (cx + w / 2).clamp(max=IMG[1] - 1),
                       (cy + h / 2).clamp(max=IMG[0] - 1), torch.ones(B)], 1)[:, None]
    heat_t = torch.rand(B, 1, *GRID, generator=g).double()
    mask_t = torch.zeros(B, 1, *IMG)
    for b in range(B):
        x1, y1, x2, y2 = (int(v) for v in ann[b, 0, :4])
        mask_t[b, 0, y1:y2, x1:x2] = 1.0
    return feat, tmpl, ann, heat_t, mask_t


def head_state(module, seed, is_cls=False):
    """seeded_state, with the classification output layer near its reference initialisation (network.py:408-419: zero weights,
    bias -log(99)): object probabilities around 0.01 as in a detector being finetuned, so that the focal term does not drown
    the other three losses in the gradient that reaches the shared trunk."""
    sd = seeded_state(module, seed)
    if is_cls:
        sd["output.weight"] = 0.2 * sd["output.weight"]
        sd["output.bias"] = sd["output.bias"] - float(np.log(99.0))
    return sd


def weight_sample(t):
    return t.reshape(-1)[::W_STRIDE]


def main():
    network, loss_mod, anchors_mod, utils = ref_import.load()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    corr = network.CorrelationModel(IMG, 640)
    cls = network.ClassificationModel(512, num_anchors=24)
    reg = network.RegressionModel(512, num_anchors=24)
    mods = (("corr", corr), ("cls", cls), ("reg", reg))
    for i, (_, m) in enumerate(mods):
        m.load_state_dict(head_state(m, SEED + i, is_cls=m is cls))
        m.train()                     # BatchNorm in training mode, as in finetuneDtoid (online_learning.py:656)
    feat, tmpl, ann, heat_t, mask_t = seeded_inputs()
    feat.requires_grad_(True)
    tmpl.requires_grad_(True)
    x2, heat, seg = corr(feat, tmpl)
    c, _ = cls(x2)
    r = reg(x2)
    base = anchors_mod.generate_anchors(base_size=30, ratios=np.array([0.5, 1, 2]), scales=np.array([1, 2, 3, 4, 5, 6, 7, 8]))
    anc = torch.from_numpy(anchors_mod.shift(GRID, 16, base).astype(np.float32))[None]
    lc, lr = loss_mod.DetectionLoss()(c, r, anc, ann)
    l_center = torch.nn.L1Loss()(heat_t, heat)
    l_seg = torch.nn.BCELoss()(torch.sigmoid(seg), mask_t)
    total = 20 * l_seg + 20 * l_center + lc + lr
    total.sum().backward()
    out = dict(
        x2=x2.detach()[:, ::X2_CH].numpy(), heat=heat.detach().numpy(), seg=seg.detach()[:, :, ::SEG_PX, ::SEG_PX].numpy(),
        cls=c.detach()[:, ::CLS_ROW].numpy(), reg=r.detach()[:, ::CLS_ROW].numpy(),
        loss_cls=lc.detach().numpy(), loss_reg=lr.detach().numpy(), loss_center=l_center.detach().numpy(),
        loss_seg=l_seg.detach().numpy(),
        grad_feat=feat.grad[:, ::GF_CH].numpy(), grad_tmpl=tmpl.grad[:, ::GT_CH].numpy(),
        seed=SEED, batch=B)
    n_w = n_s = n_b = 0
    for prefix, m in mods:
        for name, p in m.named_parameters():
            key = "g.%s.%s" % (prefix, name)
            if p.grad is None:
                continue
            if p.dim() == 4:
                out[key] = weight_sample(p.grad).numpy()
                n_w += 1
            else:
                out[key] = p.grad.numpy()
                n_s += 1
        for name, b in m.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                out["b.%s.%s" % (prefix, name)] = b.numpy()
                n_b += 1
    path = os.path.join(ROOT, "tests", "golden", "dtoid_head_train_full.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print(path, os.path.getsize(path), "bytes;", n_w, "conv weight gradients (sampled),", n_s, "vector gradients,", n_b,
          "running statistics; losses", float(lc), float(lr), float(l_center), float(l_seg))


if __name__ == "__main__":
    main()
