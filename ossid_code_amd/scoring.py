"""Zephyr scoring entry points with the call signatures of ossid.utils.zephyr_utils
(/root/reference/python/ossid/utils/zephyr_utils.py:10-47 networkInference, :49-71 filterHypoByMask), so
scripts/online_learning.py:464 can call them unchanged.

What differs from the reference is where the work happens: the 5x5 blur + /255 that the reference does with
OpenCV on the host (:13-14) is part of the frame-staging kernel, the featurizer runs on the GPU instead of the
CPU, point_x goes from featurizer to scorer without leaving HBM, and the mask test of filterHypoByMask is a
device-side gather over the projected pixels.
"""
import time

import numpy as np
import torch

from .hostutil import K2meta, to_np
from .zephyr import score_dataset as _sd

_MODEL_KEYS = ("model_points", "model_colors", "model_normals")


def _as_tensor(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))


def networkInference(model, dataset, data, return_time=False):
    """data: img uint8 [H,W,3], depth [H,W] (m), cam_K [3,3], pose_hypos [N,4,4], model_points/colors/normals [M,3],
    optional pp_err [N].  Returns (poses [N',4,4], scores [N',1], pp_err [N'], uv_original [N',M,2][, seconds])."""
    frame = _as_tensor(data["img"])
    pack = {"img": frame, "_blur_on_device": frame.dtype == torch.uint8,
            "depth": _as_tensor(data["depth"]), "transforms": _as_tensor(data["pose_hypos"]),
            "meta_data": K2meta(data["cam_K"])}
    for key in _MODEL_KEYS:
        pack[key] = _as_tensor(data[key])
    pack["pp_err"] = data["pp_err"] if "pp_err" in data else torch.zeros(len(data["pose_hypos"]))

    with torch.no_grad():
        t0 = time.time()
        point_x, uv_original = dataset.getPointNetData(pack, return_uv_original=True)
        scores = to_np(model({"point_x": point_x.to(model.device)}))  # the D2H copy synchronises the stream
        elapsed = time.time() - t0

    # getPointNetData drops hypotheses with too many free-space violations and leaves the survivors in the dict
    out = (to_np(pack["transforms"]), scores, pack["pp_err"], uv_original)
    return out + (elapsed,) if return_time else out


def networkInferenceMany(model, dataset, frames, streams=2, return_time=False):
    """ADDITIVE API: networkInference for a list of frames (dicts as networkInference takes), kept `streams` frames in
    flight on alternating HIP streams: the VALU/LDS-bound sampling kernels of one frame (FPS, ball query) overlap the
    MFMA-bound MLP kernels of another (+3.6 % hypotheses/s measured at 1000 x 2048, `bench.py --streams 2`). Results
    are exactly those of per-frame networkInference calls (each frame's kernels run in order on its own stream, the
    scorer's scratch is per stream); device->host copies happen once, after the last launch. Returns a list of the
    per-frame tuples."""
    dev = model.device
    pool = [torch.cuda.Stream(device=dev) for _ in range(max(1, int(streams)))]
    cur = torch.cuda.current_stream(dev)
    pending = []
    t0 = time.time()
    with torch.no_grad():
        for i, data in enumerate(frames):
            frame = _as_tensor(data["img"])
            pack = {"img": frame, "_blur_on_device": frame.dtype == torch.uint8,
                    "depth": _as_tensor(data["depth"]), "transforms": _as_tensor(data["pose_hypos"]),
                    "meta_data": K2meta(data["cam_K"])}
            for key in _MODEL_KEYS:
                pack[key] = _as_tensor(data[key])
            pack["pp_err"] = data["pp_err"] if "pp_err" in data else torch.zeros(len(data["pose_hypos"]))
            st = pool[i % len(pool)]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                point_x, uv_original = dataset.getPointNetData(pack, return_uv_original=True)
                scores = model({"point_x": point_x.to(dev)})
            pending.append((pack, scores, uv_original))
        for st in pool:
            cur.wait_stream(st)
        out = []
        for pack, scores, uv in pending:
            res = (to_np(pack["transforms"]), to_np(scores), pack["pp_err"], uv)
            out.append(res)
    elapsed = time.time() - t0
    return [r + (elapsed / max(1, len(out)),) for r in out] if return_time else out


def filterHypoByMask(model_points, meta_data, pose_hypos, mask, th=0.5):
    """Boolean [N]: hypotheses whose model points land inside `mask` (h x w, {0,1}) for more than a fraction th."""
    dev = _sd._dev()
    T = _sd._f32(pose_hypos, dev).reshape(-1, 4, 4)
    P = _sd._f32(model_points, dev)
    N, M = int(T.shape[0]), int(P.shape[0])
    if N == 0 or M == 0:
        return np.zeros(N, dtype=bool)
    uv = torch.empty(N, M, 2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _sd._lib.fn("ossid_zephyr_project_uv")(T.data_ptr(), P.data_ptr(), N, M, *_sd._cam(meta_data),
                                                    uv.data_ptr(), _sd._lib.stream())
    _sd._lib.check(rc, "ossid_zephyr_project_uv")
    m = _as_tensor(np.asarray(mask)).to(dev)
    h, w = m.shape
    x, y = uv[..., 0].long(), uv[..., 1].long()
    inside = (x >= 0) & (x < w) & (y >= 0) & (y < h)
    hit = m[y.clamp(0, h - 1), x.clamp(0, w - 1)].to(torch.float64) * inside
    return (hit.sum(-1) / float(M) > th).cpu().numpy()


def pose_errors(pose_hypos, pose_gt, model_points, symmetric=False):
    """ADD (symmetric=False) or ADI (True) of every hypothesis against the ground-truth pose, float64 numpy [N]:
    the device-side form of online_learning.py:452's per-hypothesis Python loop over zephyr.utils.metrics.add / adi."""
    dev = _sd._dev()
    T = _as_tensor(np.asarray(pose_hypos, dtype=np.float64)).to(dev).reshape(-1, 4, 4).contiguous()
    G = _as_tensor(np.asarray(pose_gt, dtype=np.float64)).to(dev).reshape(4, 4).contiguous()
    P = _as_tensor(np.asarray(model_points, dtype=np.float64)).to(dev).contiguous()
    N, M = int(T.shape[0]), int(P.shape[0])
    err = torch.empty(N, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = _sd._lib.fn("ossid_pose_errors")(T.data_ptr(), G.data_ptr(), P.data_ptr(), N, M, int(bool(symmetric)),
                                              err.data_ptr(), _sd._lib.stream())
    _sd._lib.check(rc, "ossid_pose_errors")
    return err.cpu().numpy()
