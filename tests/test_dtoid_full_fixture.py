"""The test-time head against the FULL-SIZE reference fixture tests/golden/dtoid_head_full.npz (480x640 image, 29x39
grid, 3 templates in chunks of 2 + 1; produced by tools/gen_golden_dtoid_full.py from the reference's own classes and its
own Network.forward_all_templates body). CPU: the nn.Module path and the post-processing. GPU (-m gpu): the PRODUCT path
-- FusedHead on csrc/conv.hip + csrc/segtail.hip with the three reassociations, both branches of the `dot` form --
one hop from the reference. Tolerances: fp32, sums reordered (MFMA tiles vs torch-CPU), stated at each check."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden_dtoid import seeded_state  # noqa: E402

from oracle import dtoid_oracle  # noqa: E402
from ossid_code_amd import dtoid  # noqa: E402

F = np.load(os.path.join(ROOT, "tests", "golden", "dtoid_head_full.npz"))
IMG, GRID = (480, 640), (29, 39)
SEED, TOPK, CHUNKS = int(F["seed"]), int(F["topk"]), tuple(int(c) for c in F["chunks"])
X2S, SEGS, SPS = 8, 4, 8          # strides the generator stored x2 channels / seg pixels / post_seg pixels with


def inputs(device="cpu"):
    g = torch.Generator().manual_seed(SEED + 10)
    feat = torch.randn(1, 640, *GRID, generator=g)
    tmpl = [torch.randn(n, 640, 7, 7, generator=g) for n in CHUNKS]
    return feat.to(device), [t.to(device) for t in tmpl]


def build_net(device="cpu"):
    """A dtoid.Network whose head carries the fixture's seeded weights (the backbone is not part of this fixture)."""
    torch.manual_seed(0)
    net = dtoid.Network(img_size=IMG, heatmap_size=GRID)
    for i, m in enumerate((net.correlation_model, net.classification, net.regression)):
        m.load_state_dict(seeded_state(m, SEED + i))
    return net.to(device).eval()


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-6))


def check_dense(x2, heat, seg, cls, reg, tol):
    assert tuple(x2.shape) == (3, 512, 29, 39) and tuple(seg.shape) == (3, 1, 480, 640)
    assert tuple(cls.shape) == (3, 27144, 2) and tuple(reg.shape) == (3, 27144, 4) and tuple(heat.shape) == (3, 1, 29, 39)
    for name, got, want in (("x2", x2[:, ::X2S], F["x2"]), ("heat", heat, F["heat"]), ("seg", seg[:, :, ::SEGS, ::SEGS], F["seg"]),
                            ("cls", cls, F["cls"]), ("reg", reg, F["reg"])):
        assert rel(got, want) < tol, (name, rel(got, want))


def check_post(out, seg_tol=1e-4):
    score, boxes, obj, seg, heat = out
    assert score.shape[0] == F["post_score"].shape[0] == TOPK
    assert np.array_equal(obj.cpu().numpy(), F["post_obj"])                      # which template fired, in score order
    assert np.allclose(score.cpu().numpy(), F["post_score"], rtol=1e-5, atol=1e-6)
    assert np.allclose(boxes.cpu().numpy(), F["post_boxes"], rtol=1e-5, atol=1e-3)      # pixels
    assert rel(seg[:, ::SPS, ::SPS], F["post_seg"]) < seg_tol and rel(heat, F["post_heat"]) < seg_tol


def test_module_path_and_postprocessing_match_full_size_reference_fixture():
    net = build_net()
    feat, tmpl = inputs()
    with dtoid_oracle.cpu_ops(), torch.no_grad():
        parts = [net.correlation_model(feat.expand(t.shape[0], -1, -1, -1), t, True) for t in tmpl]
        x2, heat, seg = (torch.cat([p[i] for p in parts]) for i in range(3))
        cls, reg = net.classification(x2)[0], net.regression(x2)
        check_dense(x2, heat, seg, cls, reg, 2e-5)
        # D12's post-processing on the reference's own dense scores / deltas: decode, clip, top-1000, NMS, [:topk], gather
        out = net.postprocess(torch.from_numpy(F["cls"]), torch.from_numpy(F["reg"]), seg, heat, GRID, IMG, topk=TOPK)
    check_post(out)


@pytest.mark.gpu
@pytest.mark.parametrize("dot_gemm,wino", [(False, False), (True, False), (False, True), (True, True)])
def test_fused_head_product_path_matches_full_size_reference_fixture(hiplib, dot_gemm, wino, monkeypatch):
    """FusedHead (hand-written MFMA convolutions, fused decoder tail, sub / dot / phase reassociations), chunk by chunk as
    forward_all_templates drives it, directly against the reference's outputs. wino: every plain 3x3 layer with >= 64 output
    channels on the Winograd kernel (at the fixture's two templates per chunk the product's dispatch would keep the direct
    kernel: the workgroup threshold is lowered to force it), trunks through detection() (merged first layer, paired launches)."""
    from ossid_code_amd.dtoid import ops
    monkeypatch.setattr(ops, "WINO_MIN_WGS", 1 if wino else 10 ** 9)
    net = build_net("cuda")
    feat, tmpl = inputs("cuda")
    fused = net._fused_head()
    fused.DOT_GEMM_MIN_TEMPLATES = 1 if dot_gemm else 10 ** 6
    frame = {}
    with torch.no_grad():
        parts = []
        for t in tmpl:
            x2, heat, seg = fused.correlation(feat, t, None, frame)
            parts.append((x2, heat, seg) + (fused.detection(x2) if wino else (fused.classification(x2), fused.regression(x2))))
        x2, heat, seg, cls, reg = (torch.cat([p[i] for p in parts]) for i in range(5))
    check_dense(x2, heat, seg, cls, reg, 1e-4)
    with torch.no_grad():
        # post-processing on the device (HIP decode / top-k / NMS / gather) from the reference's dense scores and deltas
        out = net.postprocess(torch.from_numpy(F["cls"]).cuda(), torch.from_numpy(F["reg"]).cuda(), seg, heat, GRID, IMG,
                              topk=TOPK)
    check_post(out, seg_tol=2e-4)


@pytest.mark.gpu
def test_whole_test_time_call_on_fixture_weights_finds_the_reference_detections(hiplib):
    """End to end through forward_all_templates (graph replay included) with the image backbone swapped for the fixture's
    feature map: the detection list equals the reference's wherever the scores are separated by more than the fp32
    reordering noise."""
    net = build_net("cuda")
    feat, tmpl = inputs("cuda")
    net.use_fused_backbone = False
    net.image_feature_extractor.forward = lambda image, g: feat
    with torch.no_grad():
        out = net.forward_all_templates(torch.zeros(1, 3, *IMG, device="cuda"), tmpl,
                                        [torch.zeros(1, 64, 3, 3, device="cuda")], topk=TOPK)
    score, boxes, obj = out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy()
    assert np.allclose(score, F["post_score"], rtol=2e-4, atol=1e-6)
    gaps = np.abs(np.diff(F["post_score"])) > 1e-4                        # rows whose rank cannot flip under 1e-4 noise
    stable = np.concatenate([[True], gaps]) & np.concatenate([gaps, [True]])
    assert stable.sum() >= TOPK // 2
    assert np.array_equal(obj[stable], F["post_obj"][stable])
    assert np.allclose(boxes[stable], F["post_boxes"][stable], rtol=1e-4, atol=0.05)


# ---- BASELINE configs[2]'s template count: 21 templates in ONE chunk (tests/golden/dtoid_head_full_nt21.npz) ---------------
F21 = np.load(os.path.join(ROOT, "tests", "golden", "dtoid_head_full_nt21.npz"))
X2S21, SEGS21, SPS21, ROWS21 = (int(v) for v in F21["strides"])
CHUNKS21 = tuple(int(c) for c in F21["chunks"])


def inputs21(device="cpu"):
    g = torch.Generator().manual_seed(int(F21["input_seed"]))
    feat = torch.randn(1, 640, *GRID, generator=g)
    tmpl = [torch.randn(n, 640, 7, 7, generator=g) for n in CHUNKS21]
    return feat.to(device), [t.to(device) for t in tmpl]


def check_dense21(x2, heat, seg, cls, reg, tol):
    n = sum(CHUNKS21)
    assert tuple(x2.shape) == (n, 512, 29, 39) and tuple(seg.shape) == (n, 1, 480, 640)
    assert tuple(cls.shape) == (n, 27144, 2) and tuple(reg.shape) == (n, 27144, 4)
    for name, got, want in (("x2", x2[:, ::X2S21], F21["x2"]), ("heat", heat, F21["heat"]),
                            ("seg", seg[:, :, ::SEGS21, ::SEGS21], F21["seg"]), ("cls", cls[:, ::ROWS21], F21["cls"]),
                            ("reg", reg[:, ::ROWS21], F21["reg"])):
        assert rel(got, want) < tol, (name, rel(got, want))


def check_detections21(out, score_rtol):
    """The reference's detection list, wherever its scores are separated by more than the f32 reordering noise."""
    score, boxes, obj = out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy()
    assert score.shape[0] == F21["post_score"].shape[0] == TOPK
    assert np.allclose(score, F21["post_score"], rtol=score_rtol, atol=1e-6)
    gaps = np.abs(np.diff(F21["post_score"])) > 1e-4
    stable = np.concatenate([[True], gaps]) & np.concatenate([gaps, [True]])
    assert stable.sum() >= TOPK // 2
    assert np.array_equal(obj[stable], F21["post_obj"][stable])
    assert np.allclose(boxes[stable], F21["post_boxes"][stable], rtol=1e-4, atol=0.05)
    # the segmentation / heat map rows gathered for those detections
    seg, heat = out[3][:, ::SPS21, ::SPS21].cpu().numpy(), out[4].cpu().numpy()
    for i in np.nonzero(stable)[0]:
        assert rel(seg[i], F21["post_seg"][i]) < 2e-4 and rel(heat[i], F21["post_heat"][i]) < 2e-4, i


def test_module_path_matches_nt21_reference_fixture():
    net = build_net()
    feat, tmpl = inputs21()
    with dtoid_oracle.cpu_ops(), torch.no_grad():
        parts = [net.correlation_model(feat.expand(t.shape[0], -1, -1, -1), t, True) for t in tmpl]
        x2, heat, seg = (torch.cat([p[i] for p in parts]) for i in range(3))
        cls, reg = net.classification(x2)[0], net.regression(x2)
        check_dense21(x2, heat, seg, cls, reg, 2e-5)
        check_detections21(net.postprocess(cls, reg, seg, heat, GRID, IMG, topk=TOPK), 1e-5)


@pytest.mark.gpu
def test_fused_head_product_dispatch_matches_nt21_reference_fixture(hiplib):
    """FusedHead exactly as forward_all_templates drives it at 21 templates, with NO threshold overridden: the product's own
    choices -- Winograd with 128 output channels per workgroup (>= 128 tile groups), its tail split, the direct `dot`
    convolution (21 < DOT_GEMM_MIN_TEMPLATES), merged first trunk layer and paired trunk launches -- against the reference."""
    from ossid_code_amd.dtoid import ops
    net = build_net("cuda")
    feat, tmpl = inputs21("cuda")
    fused = net._fused_head()
    n_t = sum(CHUNKS21)
    # what this test is about: the dispatcher's own decisions at this size (they are not forced here)
    assert fused.cf.use_wino(n_t, GRID[0], GRID[1]) and n_t < fused.DOT_GEMM_MIN_TEMPLATES
    frame = {}
    with torch.no_grad():
        parts = []
        for t in tmpl:
            x2, heat, seg = fused.correlation(feat, t, None, frame)
            parts.append((x2, heat, seg) + fused.detection(x2))
        x2, heat, seg, cls, reg = (torch.cat([p[i] for p in parts]) for i in range(5))
        check_dense21(x2, heat, seg, cls, reg, 1e-4)
        check_detections21(net.postprocess(cls, reg, seg, heat, GRID, IMG, topk=TOPK), 2e-4)


@pytest.mark.gpu
def test_whole_test_time_call_at_21_templates_finds_the_reference_detections(hiplib):
    """forward_all_templates (graph capture + replay) on the fixture's weights and 21 templates, backbone swapped for the
    fixture's feature map."""
    net = build_net("cuda")
    feat, tmpl = inputs21("cuda")
    net.use_fused_backbone = False
    net.image_feature_extractor.forward = lambda image, g: feat
    with torch.no_grad():
        out = net.forward_all_templates(torch.zeros(1, 3, *IMG, device="cuda"), tmpl,
                                        [torch.zeros(1, 64, 3, 3, device="cuda")], topk=TOPK)
    check_detections21(out, 2e-4)
