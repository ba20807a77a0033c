"""The per-frame online loop of OSSID, hot-path steps only, kept on the device end to end -- the synthetic counterpart of
scripts/online_learning.py:314-591 (SURVEY.md 8d cfg-5, 8e "full online stream"):

    detect (DtoidNet.forwardTestTime)                       :346
    -> pose hypotheses (GIVEN: Halcon PPF / SIFT are out of scope, :416-446)
    -> per-hypothesis ADD/ADI (:452) -> Zephyr score (networkInference, :464) -> argmax (:466-469)
    -> predicted depth (point-splat renderer for pyrender, :485) -> visibility mask (:500)
    -> if score > threshold: pseudo-label joins the finetune set (:506-516)
    -> when the set reaches the next multiple of finetune_interval: finetune DTOID (:517-533)

Multi-GPU (one process per GPU): the loop is sequential by construction -- the finetune trigger depends on the running count
of confident frames and later frames must see the updated detector -- so frames are dispatched in SPECULATIVE WINDOWS of one
frame per rank with frozen weights, results are committed in frame order, and when the trigger falls inside a window the
frames behind it are discarded, all ranks run the data-parallel finetune (gradient mean over RCCL), and the next window
starts right after the trigger frame. With deterministic per-frame work this reproduces the single-process result.
"""
import time

import numpy as np
import torch

from . import pipeline
from .hostutil import K2meta
from .scoring import networkInference, pose_errors


class SpeculativeWindow:
    """In-order commit logic of the speculative multi-GPU stream, free of any device work (unit-testable):

        w = SpeculativeWindow(n_frames, world, finetune_interval)
        while not w.done:
            frames = w.window()                  # <= world frame ids, one per rank, all scored with the current weights
            trigger = w.commit(confident_flags)  # flags of those frames, in order; returns the training-set size if a
                                                 # finetune fires now (the frames behind the trigger frame are re-issued)
    """

    def __init__(self, n_frames, world, finetune_interval, cumulative=True):
        self.n_frames, self.world, self.interval, self.cumulative = n_frames, world, finetune_interval, cumulative
        self.next_frame = 0
        self.train_set = []
        self.next_finetune = finetune_interval
        self.committed = []           # (frame, confident) in commit order
        self.discarded = 0

    @property
    def done(self):
        return self.next_frame >= self.n_frames

    def window(self):
        return list(range(self.next_frame, min(self.next_frame + self.world, self.n_frames)))

    def commit(self, flags):
        frames = self.window()
        assert len(flags) == len(frames)
        for i, (f, c) in enumerate(zip(frames, flags)):
            self.committed.append((f, bool(c)))
            self.next_frame = f + 1
            if c:
                self.train_set.append(f)
                if len(self.train_set) == self.next_finetune:
                    self.discarded += len(frames) - i - 1      # scored with weights that are about to change
                    size = len(self.train_set)
                    if self.cumulative:
                        self.next_finetune += self.interval
                    else:
                        self.train_set = []
                    return size
        return None


class OnlineStream:
    """One GPU's worth of the loop. `detector` is a dtoid.DtoidNet (eval), `scorer` a zephyr.PointNet2SSG (eval),
    `score_dataset` a zephyr.ScoreDataset; `finetune_fn(samples)` is called with the accumulated pseudo-labelled samples."""

    def __init__(self, detector, scorer, score_dataset, confident_threshold=20.0, symmetric=False, finetune_fn=None):
        self.detector, self.scorer, self.dataset = detector, scorer, score_dataset
        self.threshold, self.symmetric, self.finetune_fn = confident_threshold, symmetric, finetune_fn
        self.times = {k: 0.0 for k in ("detect", "pose_err", "score", "pseudo_label")}

    def _timed(self, key, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        self.times[key] += time.perf_counter() - t0
        return out

    def process(self, frame):
        """frame: dict with img uint8 [H,W,3], depth [H,W], cam_K, limg [n_t,3,124,124], lmask [n_t,1,124,124], obj_id,
        pose_hypos [N,4,4], pose_gt [4,4], model_points/normals/colors [M,3]. Returns the per-frame result dict."""
        dev = next(self.detector.parameters()).device
        img_t = torch.from_numpy(np.ascontiguousarray(frame["img"])).to(dev).permute(2, 0, 1).float().div_(255.0)[None]
        batch = {"img": img_t, "obj_id": torch.tensor([int(frame["obj_id"])]), "limg": frame["limg"][None].to(dev),
                 "lmask": frame["lmask"][None].to(dev)}
        det = self._timed("detect", lambda: self.detector.forwardTestTime(batch))
        pp_err = self._timed("pose_err", lambda: pose_errors(frame["pose_hypos"], frame["pose_gt"],
                                                             frame["model_points"], self.symmetric))
        data = {k: frame[k] for k in ("img", "depth", "cam_K", "model_points", "model_normals", "model_colors",
                                      "pose_hypos")}
        data["pp_err"] = pp_err
        poses, scores, errs, uv = self._timed("score", lambda: networkInference(self.scorer, self.dataset, data))
        best = int(scores.argmax())
        pred_pose, pred_score = poses[best], float(scores.max())
        H, W = frame["depth"].shape

        def pseudo():
            pred_depth = pipeline.render_depth_points(pred_pose, frame["model_points"], frame["cam_K"], (H, W), radius=1)
            return pipeline.visibility_and_iou(frame["depth"], pred_depth)[:2]
        pred_mask, pred_mask_visib = self._timed("pseudo_label", pseudo)
        confident = pred_score > self.threshold
        sample = None
        if confident:
            sample = pipeline.make_dtoid_sample(frame["img"], frame["depth"], pred_mask_visib.float(), frame["cam_K"])
        return {"pred_pose": pred_pose, "pred_score": pred_score, "pred_err": float(np.asarray(errs)[best]),
                "confident": confident, "dtoid_score": det["pred_scores"][:1], "dtoid_bbox": det["pred_bbox"][:1],
                "pred_mask_visib": pred_mask_visib, "sample": sample}

    def run(self, frames, finetune_interval=8):
        """Sequential loop on this GPU (world 1); returns (results, window bookkeeping)."""
        win = SpeculativeWindow(len(frames), 1, finetune_interval)
        results, samples = [], []
        while not win.done:
            f = win.window()[0]
            r = self.process(frames[f])
            results.append(r)
            if r["sample"] is not None:
                samples.append((frames[f], r["sample"]))
            fired = win.commit([r["confident"]])
            if fired is not None and self.finetune_fn is not None:
                self.finetune_fn(samples)
        return results, win
