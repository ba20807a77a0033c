"""The TRAINING head against the full-size reference fixture tests/golden/dtoid_head_train_full.npz (29x39 grid, 480x640
masks, batch 8; produced by tools/gen_golden_dtoid_train_full.py from the reference's own CorrelationModel /
ClassificationModel / RegressionModel / DetectionLoss in train mode, forward AND backward).
CPU: this repo's nn.Module restatement. GPU (-m gpu): the PRODUCT training path (Network._head_train_hip: FusedConv /
BNFold / Winograd / split-bf16 data + weight gradients, grouped launches, side streams) with the dispatch exactly as the
finetune step chooses it -- no threshold is overridden. Bounds are the reduced fixture's (rtol 2e-3 / atol 2e-4
elementwise) plus a scale-relative bound per tensor, stated at each check."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden_dtoid_train_full as gen  # noqa: E402

from oracle import dtoid_oracle  # noqa: E402
from ossid_code_amd import dtoid  # noqa: E402

F = np.load(os.path.join(ROOT, "tests", "golden", "dtoid_head_train_full.npz"))
IMG, GRID, B, SEED = gen.IMG, gen.GRID, int(F["batch"]), int(F["seed"])


def build_net(device):
    torch.manual_seed(0)
    net = dtoid.Network(img_size=IMG, heatmap_size=GRID)
    for i, m in enumerate((net.correlation_model, net.classification, net.regression)):
        m.load_state_dict(gen.head_state(m, SEED + i, is_cls=m is net.classification))
    return net.to(device).train()


def run(net, device, product):
    corr, cls, reg = net.correlation_model, net.classification, net.regression
    feat, tmpl, ann, heat_t, mask_t = (t.to(device) for t in gen.seeded_inputs(SEED + 10))
    feat.requires_grad_(True)
    tmpl.requires_grad_(True)
    if product:
        # (views, not the leaves: in the product these are the backbone's / encoder's outputs -- a LEAF read on several streams
        # gets its AccumulateGrad node on one of them and gradients from the others, which torch warns about)
        c, r, anc, heat, seg = net._head_train_hip(feat.view_as(feat), tmpl.view_as(tmpl))
        x2 = None
    else:
        x2, heat, seg = corr(feat, tmpl)
        c, r = cls(x2)[0], reg(x2)
        anc = net.anchors([list(GRID)], device=device)
    lc, lr = dtoid.DetectionLoss()(c, r, anc, ann)
    l_center = torch.nn.L1Loss()(heat_t, heat)
    l_seg = torch.nn.BCELoss()(torch.sigmoid(seg), mask_t)
    (20 * l_seg + 20 * l_center + lc + lr).sum().backward()
    out = dict(heat=heat, seg=seg[:, :, ::gen.SEG_PX, ::gen.SEG_PX], cls=c[:, ::gen.CLS_ROW], reg=r[:, ::gen.CLS_ROW],
               loss_cls=lc, loss_reg=lr, loss_center=l_center, loss_seg=l_seg,
               grad_feat=feat.grad[:, ::gen.GF_CH], grad_tmpl=tmpl.grad[:, ::gen.GT_CH])
    if x2 is not None:
        out["x2"] = x2[:, ::gen.X2_CH]
    for prefix, m in (("corr", corr), ("cls", cls), ("reg", reg)):
        for name, p in m.named_parameters():
            key = "g.%s.%s" % (prefix, name)
            if key in F.files:
                assert p.grad is not None, key
                out[key] = gen.weight_sample(p.grad) if p.dim() == 4 else p.grad
            else:
                assert p.grad is None or float(p.grad.abs().max()) == 0, key
        for name, b in m.named_buffers():
            if "b.%s.%s" % (prefix, name) in F.files:
                out["b.%s.%s" % (prefix, name)] = b
    return out


def compare(out, rtol, atol, scale_tol):
    """Elementwise allclose(rtol, atol) AND max |got - want| <= scale_tol * max |want| per tensor (the elementwise form is
    vacuous for gradients whose entries are smaller than atol; the scale-relative one is not)."""
    bad = []
    for k, v in out.items():
        got = v.detach().cpu().numpy()
        want = F[k]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        err = float(np.abs(got.astype(np.float64) - want).max())
        scale = max(float(np.abs(want).max()), 1e-12)
        if not np.allclose(got, want, rtol=rtol, atol=atol) or err > scale_tol * scale:
            bad.append((k, "%.2e of scale %.2e" % (err / scale, scale)))
    assert not bad, bad
    stored = [k for k in F.files if k not in ("seed", "batch", "x2")]
    assert sorted(k for k in out if k != "x2") == sorted(stored)           # every stored tensor was compared


def test_module_path_matches_full_size_training_fixture():
    """This repo's nn.Modules (same state_dict keys as the reference's) on the CPU: same arithmetic in the same order, so the
    bound is torch-CPU's own thread-partition noise."""
    net = build_net("cpu")
    with dtoid_oracle.cpu_ops():
        out = run(net, "cpu", product=False)
    compare(out, rtol=2e-4, atol=2e-5, scale_tol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [True, False])
def test_hip_training_head_matches_full_size_reference_fixture(hiplib, streams):
    """Network._head_train_hip at the finetune step's own sizes with the product dispatch untouched: at batch 8 on 29x39 the
    plain 3x3 layers clear the Winograd threshold (train_ops.wino_fits), the weight gradients run their split-K plans and
    few-channel decoder tilings, and (streams) the branches and weight gradients run on their side streams."""
    from ossid_code_amd.dtoid import train_ops
    # the dispatch this test is about: what the finetune step itself would choose
    assert train_ops.wino_fits(B, GRID[0], GRID[1], 768, 512, 9) and train_ops.wino_fits(B, GRID[0], GRID[1], 256, 256, 9)
    net = build_net("cuda")
    net.use_train_streams = streams
    out = run(net, "cuda", product=True)
    torch.cuda.synchronize()
    # rtol / atol: the reduced fixture's bounds (test_hip_training_head_matches_reference_golden); scale_tol: split-bf16
    # products (5-7e-6 per layer) through ~12 layers forward and back, and BatchNorm statistics over 9 048 values
    compare(out, rtol=2e-3, atol=2e-4, scale_tol=2e-3)
    assert int(net.correlation_model.ns3.num_batches_tracked) == 1 and int(net.correlation_model.nf.num_batches_tracked) == 1
