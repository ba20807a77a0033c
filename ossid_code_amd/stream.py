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
        self.n_processed = 0

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
        self.n_processed += 1
        img_t = torch.from_numpy(np.ascontiguousarray(frame["img"])).to(dev).permute(2, 0, 1).float().div_(255.0)[None]
        batch = {"img": img_t, "obj_id": torch.tensor([int(frame["obj_id"])]), "limg": frame["limg"][None].to(dev),
                 "lmask": frame["lmask"][None].to(dev)}
        det = self._timed("detect", lambda: self.detector.forwardTestTime(batch))
        pp_err = self._timed("pose_err", lambda: pose_errors(frame["pose_hypos"], frame["pose_gt"],
                                                             frame["model_points"], self.symmetric))
        data = {k: frame[k] for k in ("img", "depth", "cam_K", "model_points", "model_normals", "model_colors",
                                      "pose_hypos")}
        data["pp_err"] = pp_err
        poses, scores, errs, _uv = self._timed("score", lambda: networkInference(self.scorer, self.dataset, data))
        best = int(scores.argmax())
        pred_pose, pred_score = poses[best], float(scores.max())
        H, W = frame["depth"].shape

        def pseudo():
            pred_depth = pipeline.render_depth_points(pred_pose, frame["model_points"], frame["cam_K"], (H, W), radius=1)
            return pipeline.visibility_and_iou(frame["depth"], pred_depth)[:2]
        _pred_mask, pred_mask_visib = self._timed("pseudo_label", pseudo)
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


def run_speculative(frames, process_fn, finetune_fn, finetune_interval, dist=None, group=None):
    """The multi-GPU stream, SPMD: every rank calls this with the same `frames`. process_fn(frame) -> (confident, sample)
    scores one frame with this rank's (replicated) weights; finetune_fn(train_set) is entered by ALL ranks together with the
    identical, frame-ordered list of (frame_id, sample) and is expected to run the data-parallel finetune (GradSync).
    Per window: one frame per rank, an all_gather of the confident flags, in-order commit, an all_gather_object-free
    sample exchange (each confident sample is broadcast from the rank that produced it: a 480x640 sample is ~5 MB, one
    xGMI hop). Returns the committed [(frame_id, confident)] and the SpeculativeWindow (for .discarded)."""
    world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    win = SpeculativeWindow(len(frames), world, finetune_interval)
    train = []
    while not win.done:
        ids = win.window()
        mine = ids[rank] if rank < len(ids) else None
        confident, sample = process_fn(frames[mine]) if mine is not None else (False, None)
        if world > 1:
            flag = torch.tensor([1 if confident else 0], dtype=torch.int32)
            if dist.get_backend(group) == "nccl":
                flag = flag.cuda()
            flags = [torch.zeros_like(flag) for _ in range(world)]
            dist.all_gather(flags, flag, group=group)
            flags = [bool(int(f)) for f in flags][: len(ids)]
        else:
            flags = [bool(confident)]
        n_before = len(win.committed)
        fired = win.commit(flags)
        for f, c in win.committed[n_before:]:
            if not c:
                continue
            src = f - ids[0]
            if world > 1:
                sample_f = _broadcast_sample(sample if src == rank else None, src, dist, group)
            else:
                sample_f = sample
            train.append((f, sample_f))
        if fired is not None:
            finetune_fn(train)
    return win.committed, win


def _broadcast_sample(sample, src, dist, group):
    """Sends a dict of tensors from rank `src` to everyone: the (small) key/shape/dtype header as an object, the payload as
    tensor broadcasts on the group's device."""
    header = [None]
    if sample is not None:
        header = [[(k, tuple(v.shape), v.dtype, v.is_cuda) for k, v in sample.items() if torch.is_tensor(v)]]
    dist.broadcast_object_list(header, src=src, group=group)
    out = {}
    for k, shape, dtype, on_gpu in header[0]:
        if sample is not None:
            t = sample[k].contiguous()
        else:           # same device kind as the sender's tensor (device tensors stay on the device: RCCL, or gloo's CUDA path)
            t = torch.empty(shape, dtype=dtype, device="cuda" if on_gpu else "cpu")
        dist.broadcast(t, src=src, group=group)
        out[k] = t
    return out
