"""cfg-5: the per-frame online loop chained on the device -- detect, ADD/ADI, score, pseudo-label, finetune
trigger -- on a small synthetic stream. The finetune here is a stand-in (a fused-optimizer step on a synthetic gradient): the
real forward/backward step has its own tests; what is checked is the plumbing between the stages and that the detector's
packed plans follow the weight update."""
import numpy as np
import pytest
import torch

from ossid_code_amd import dtoid, pipeline, synth, zephyr
from ossid_code_amd.dtoid import finetune
from ossid_code_amd.stream import OnlineStream

pytestmark = pytest.mark.gpu


class _Args:
    dataset, no_valid_proj, no_valid_depth, inconst_ratio_th, interp = "HSVD_diff_uv_norm", True, True, 100, 0


def test_online_stream_chains_all_stages(hiplib):
    torch.manual_seed(0)
    det = dtoid.DtoidNet(dtoid.DtoidConfig()).cuda().eval()
    flat = finetune.FlatParams(det)
    opt = finetune.FusedAMSGrad(flat, lr=1e-3)
    ds = zephyr.ScoreDataset([], "", "lmo", _Args(), mode="test")
    scorer = synth.random_pn2_state(zephyr.PointNet2SSG(ds.dim_point, _Args(), num_class=1), 0).to(0).eval()
    g = torch.Generator().manual_seed(1)
    limg = torch.rand(3, 3, 124, 124, generator=g)
    lmask = (torch.rand(3, 1, 124, 124, generator=g) > 0.5).float()
    frames = []
    for f in range(5):
        d = synth.make_scoring_inputs(64, 512, seed=200 + f)
        d.update(limg=limg, lmask=lmask, obj_id=1, pose_gt=d["pose_hypos"][0].copy())
        frames.append(d)
    calls = []

    def finetune_fn(samples):
        calls.append(len(samples))
        for _, smp in samples:                          # every pseudo-labelled sample is a D14 batch row on the device
            assert smp["img"].shape == (3, 480, 640) and smp["img"].is_cuda and smp["bbox_gt"].shape == (1, 5)
            assert smp["heatmap"].dtype == torch.float64 and float(smp["mask"].sum()) > 0
        flat.grad.normal_(0, 1e-3, generator=torch.Generator(device="cuda").manual_seed(len(calls)))
        opt.step()
        det.clearCache()

    stream = OnlineStream(det, scorer, ds, confident_threshold=-1e30, finetune_fn=finetune_fn)
    results, win = stream.run(frames, finetune_interval=2)
    assert len(results) == 5 and calls == [2, 4]        # cumulative training set, trigger at every 2nd confident frame
    assert [f for f, _ in win.committed] == [0, 1, 2, 3, 4] and win.discarded == 0
    for r, fr in zip(results, frames):
        assert r["pred_pose"].shape == (4, 4) and np.isfinite(r["pred_score"]) and r["confident"]
        assert r["pred_mask_visib"].shape == (480, 640) and r["pred_mask_visib"].dtype == torch.bool
        assert r["dtoid_bbox"].shape[1] == 4
        # the best hypothesis by score is one of the given hypotheses, and its ADD is what pose_errors says
        k = int(np.argmin([np.abs(np.asarray(h) - r["pred_pose"]).max() for h in fr["pose_hypos"]]))
        assert np.abs(np.asarray(fr["pose_hypos"][k]) - r["pred_pose"]).max() < 1e-6
    # the detector really changed between frame 1 and frame 2 (weights updated, plans refreshed in place)
    net = det.model
    assert net.__dict__.get("_plan_epoch", 0) >= 2
    # a render of the ground-truth pose covers the object: visibility mask of frame 0 is non-empty
    assert int(results[0]["pred_mask_visib"].sum()) > 0
    assert set(stream.times) == {"detect", "pose_err", "score", "pseudo_label"} and stream.n_processed == 5
