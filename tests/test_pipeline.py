"""batch producer, bbox/heat map, visibility mask + IoU, splat renderer, BOP csv.
CPU tests pin the numpy oracle (against the reference's own heatmapGaussain via the golden file) and the host-side csv
writer; GPU tests compare the kernels with the oracle."""
import csv
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as po
from ossid_code_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "dtoid_head.npz"))


def test_oracle_heatmap_matches_reference_golden():
    got = po.heatmap_gaussian(29, 39, 12.3, 7.9, np.sqrt(1.5))
    assert got.shape == (29, 39) and np.array_equal(got, G["gauss"])


def test_bop_csv_format(tmp_path):
    from ossid_code_amd.pipeline import save_results_bop
    pose = np.eye(4)
    pose[:3, 3] = [0.1, -0.2, 0.8]
    path = save_results_bop([{"scene_id": 2, "im_id": 5, "obj_id": 9, "pose": pose, "score": 21.5, "time": 0.3}],
                            str(tmp_path), "my_exp", "lmo")
    assert os.path.basename(path) == "my-exp_lmo-test.csv"
    rows = list(csv.DictReader(open(path)))
    assert list(rows[0].keys()) == ["scene_id", "im_id", "obj_id", "score", "R", "t", "time"]
    assert rows[0]["t"] == "100.0 -200.0 800.0" and rows[0]["R"].split(" ")[0] == "1.0" and rows[0]["score"] == "21.5"
    assert pose[2, 3] == 0.8                                  # caller's pose untouched


def _frame(seed=0):
    d = synth.make_scoring_inputs(N=2, M=900, seed=seed)
    rng = np.random.default_rng(seed)
    mask = np.zeros((480, 640), np.uint8)
    mask[150:331, 200:401] = 255
    mask[rng.random(mask.shape) < 0.3] = 0
    return d, mask


@pytest.mark.gpu
@pytest.mark.parametrize("out_hw", [None, (240, 320), (224, 224), (496, 656)])
def test_batch_producer_matches_oracle(hiplib, out_hw):
    from ossid_code_amd.pipeline import make_dtoid_sample
    d, mask = _frame()
    H, W = (480, 640) if out_hw is None else out_hw
    s = make_dtoid_sample(d["img"], d["depth"], mask, d["cam_K"], out_hw=out_hw, heatmap_hw=(29, 39))
    im, m, xyz = po.process_data(d["img"], mask / 255.0, d["depth"], d["cam_K"], H, W)
    assert s["img"].shape == (3, H, W) and s["xyz"].shape == (3, H, W) and s["mask"].shape == (1, H, W)
    if out_hw is None:                                        # same size: exact
        assert np.array_equal(s["img"].cpu().numpy(), im) and np.array_equal(s["mask"].cpu().numpy(), m)
        assert np.array_equal(s["xyz"].cpu().numpy(), xyz)
    else:                                                     # float bilinear: same formula, fused multiply-adds aside
        assert np.abs(s["img"].cpu().numpy() - im).max() <= 1.0 / 255 + 1e-6      # rounding to uint8 may flip one step
        assert np.allclose(s["mask"].cpu().numpy(), m, atol=1e-6) and np.allclose(s["xyz"].cpu().numpy(), xyz, atol=1e-5)
    box = po.mask_bbox(m[0])
    assert s["bbox_gt"].cpu().numpy().astype(int).tolist() == [box.tolist()]
    scale = 29.0 / H
    want = po.heatmap_gaussian(29, 39, (box[0] + box[2]) / 2.0 * scale, (box[1] + box[3]) / 2.0 * scale, np.sqrt(1.5))
    assert s["heatmap"].dtype == torch.float64 and np.allclose(s["heatmap"].cpu().numpy()[0], want, rtol=1e-12, atol=1e-15)


@pytest.mark.gpu
def test_empty_mask_gives_padding_label(hiplib):
    from ossid_code_amd.pipeline import make_dtoid_sample
    d, mask = _frame()
    s = make_dtoid_sample(d["img"], d["depth"], np.zeros_like(mask), d["cam_K"])
    assert float(s["bbox_gt"][0, 4]) == -1 and float(s["heatmap"].abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [0, 1, 2])
def test_splat_renderer_and_visibility(hiplib, radius):
    from ossid_code_amd.pipeline import render_depth_points, visibility_and_iou
    d, _ = _frame(3)
    pose = d["pose_hypos"][0]
    H, W = 120, 160
    K = d["cam_K"].copy()
    K[:2] *= 0.25
    got = render_depth_points(pose, d["model_points"], K, (H, W), radius=radius).cpu().numpy()
    want = po.render_depth_points(pose, d["model_points"], K, H, W, radius)
    assert np.array_equal(got, want) and (got > 0).sum() > 20
    rng = np.random.default_rng(0)
    d_obs = np.where(rng.random((H, W)) < 0.1, 0, want + rng.normal(0, 0.01, (H, W))).astype(np.float32)
    gt = want > 0
    gtv = gt & (rng.random((H, W)) < 0.8)
    pm, vm, iou, iou_v = visibility_and_iou(d_obs, got, gt, gtv)
    wpm, wvm, wiou, wiou_v = po.visib_and_iou(d_obs, want, gt, gtv, 15 / 1000.0)
    assert np.array_equal(pm.cpu().numpy(), wpm) and np.array_equal(vm.cpu().numpy(), wvm)
    assert iou == wiou == 1.0 and abs(iou_v - wiou_v) < 1e-12


def test_det_results_and_expand_box(tmp_path):
    from ossid_code_amd import pipeline
    pipeline.save_det_results({(2, 7): [(5, 10, 20, 30, 40, 0.5), (6, 1, 2, 3, 4, 1.0)], (2, 8): [(5, 1, 2, 3, 4)]},
                              str(tmp_path))
    assert (tmp_path / "s000002_i000007.txt").read_text() == "obj_000005 0.500000 10 20 30 40\nobj_000006 1.000000 1 2 3 4\n"
    assert (tmp_path / "s000002_i000008.txt").read_text() == "obj_000005 1 2 3 4\n"
    # utils/__init__.py:11-16 restated independently
    x1, y1, x2, y2 = pipeline.expand_box(600, 10, 640, 50, 480, 640, 1.2)
    assert (x1, y1, x2, y2) == (620 - 24, 30 - 24, 639, 30 + 24)
    assert pipeline.expand_box(0, 0, 10, 10, 480, 640, 2.0)[:2] == (0, 0)


def test_template_view_selection_rules():
    """Rounded-linspace thinning at test time and nearest-rotation candidates at train time
    (datasets/dtoid_bop_dataset.py:294-318), checked against scipy's Rotation for the quaternion part."""
    from scipy.spatial.transform import Rotation
    from ossid_code_amd import pipeline
    rng = np.random.default_rng(0)
    rots = Rotation.random(40, random_state=1)
    bank = pipeline.TemplateBank.__new__(pipeline.TemplateBank)
    bank.n_local_test, bank.sample_from = 10, 5
    bank.img, bank.mask, bank.quats = {3: np.zeros((40, 1))}, {}, {3: rots.as_quat()}
    assert list(bank.test_views(3)) == list(np.linspace(0, 39, 10).round().astype(int))
    for _ in range(10):
        gt = Rotation.random(random_state=int(rng.integers(1 << 30)))
        q = gt.as_quat()
        want = np.argsort(2 * np.arccos(np.minimum(np.abs(rots.as_quat() @ q), 1 - 1e-7)), kind="stable")
        got = bank.nearest_views(3, gt.as_matrix())
        assert list(got[:5]) == list(want[:5])
        assert bank.train_view(3, gt.as_matrix(), rng) in want[:5]
        qq = pipeline._rotmat_to_quat(gt.as_matrix())
        assert min(np.abs(qq - q).max(), np.abs(qq + q).max()) < 1e-12


@pytest.mark.gpu
def test_pseudo_label_set_rows_are_d14_batches():
    import torch
    from ossid_code_amd import pipeline, synth
    d = synth.make_scoring_inputs(N=4, M=256, seed=3)
    g = torch.Generator().manual_seed(0)
    bank = pipeline.TemplateBank(n_local_test=4, sample_from=3)
    bank.add(1, (torch.rand(9, 124, 124, 3, generator=g) * 255).to(torch.uint8), torch.rand(9, 124, 124, generator=g) > 0.5,
             grid_quats=np.random.default_rng(0).normal(size=(9, 4)))
    mask = np.zeros((480, 640), np.float32)
    mask[100:200, 300:420] = 1
    for mode in ("train", "test"):
        ps = pipeline.PseudoLabelSet(bank, mode=mode)
        ps.add(1, 2, 3, d["img"], d["depth"], d["cam_K"], mask, 25.0, rot=np.eye(3))
        ps.add(1, 2, 4, d["img"], d["depth"], d["cam_K"], mask, 21.0)
        assert len(ps) == 2
        row = ps[0]
        ref = pipeline.make_dtoid_sample(d["img"], d["depth"], mask, d["cam_K"])
        for k in ("img", "xyz", "mask", "bbox_gt", "heatmap"):
            assert torch.equal(row[k], ref[k]), k
        assert row["bbox_gt"].tolist() == [[300.0, 100.0, 419.0, 199.0, 1.0]]
        assert row["gimg"].shape == (3, 124, 124) and row["gmask"].shape == (1, 124, 124)
        assert float(row["gimg"].max()) <= 1.0
        if mode == "train":
            assert row["limg"].shape == (3, 124, 124)
            batch = pipeline.collate([ps[0], ps[1]])
            assert batch["img"].shape == (2, 3, 480, 640) and batch["limg"].shape == (2, 3, 124, 124)
        else:
            assert row["limg"].shape == (4, 3, 124, 124) and row["lmask"].shape == (4, 1, 124, 124)
        m2 = np.zeros((480, 640), np.float32)
        m2[10:20, 30:50] = 1
        ps.updateZephyrMask(1, 2, 3, m2, 30.0)
        assert ps[0]["bbox_gt"].tolist() == [[30.0, 10.0, 49.0, 19.0, 1.0]] and ps[0]["zephyr_score"] == 30.0
