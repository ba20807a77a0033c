"""ScoreDataset.getPointNetData and projectPointsUv -- host-side mirrors of the Zephyr featurizer.

Reference interface (the implementation lives in the un-vendored `zephyr` package; SPEC.md states what
this build computes):
  ScoreDataset([], "", name, zephyr_args, mode='test'), .dim_point
                                   /root/reference/python/ossid/scripts/online_learning.py:206-207
  dataset.getPointNetData(scoring_data, return_uv_original=True) -> (point_x, uv_original), filtering
  scoring_data['transforms'] / ['pp_err'] in place           utils/zephyr_utils.py:31,39-43
  zephyr.utils.projectPointsUv(pose_hypos, model_points, meta_data) -> int [N, M, 2]
                                                              utils/zephyr_utils.py:58
All device work goes through libossid_hip.so (include/ossid_hip.h); torch only owns the buffers.
"""
import numpy as np
import torch

from .. import _lib

DIM_POINT = 8  # (x, y, 0, dH, dS, dV, dD, cosN)
INCONST_MARGIN = 0.02  # metres, SPEC.md 3.4


def _dev(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError("the OSSID hot path needs a GPU: torch.cuda.is_available() is False and there is no "
                           "CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _f32(x, dev):
    """numpy / torch (any float dtype, any device) -> contiguous float32 tensor on dev."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()


def _cam(meta):
    return tuple(float(np.float32(meta[k])) for k in ("camera_fx", "camera_fy", "camera_cx", "camera_cy"))


class FrameCache:
    """Device-resident staged frame + model table, so a caller scoring several hypothesis sets against the same
    frame/object uploads and converts them once (bigger batches, fewer host round trips)."""

    def __init__(self, rgbd, tab, H, W, M):
        self.rgbd, self.tab, self.H, self.W, self.M = rgbd, tab, H, W, M


def stage_frame(img, depth, dev=None, blur=False):
    """img: uint8 [H,W,3] (blurred on the GPU when blur) or float [H,W,3] in [0,1]; depth [H,W] metres."""
    dev = _dev(dev)
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
    H, W = int(img.shape[0]), int(img.shape[1])
    depth = _f32(depth, dev)
    rgbd = torch.empty(H, W, 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if img.dtype == torch.uint8:
            img = img.to(dev).contiguous()
            rc = _lib.fn("ossid_zephyr_prep_frame_u8")(img.data_ptr(), depth.data_ptr(), H, W, int(bool(blur)),
                                                       rgbd.data_ptr(), _lib.stream())
        else:
            if blur:
                raise ValueError("blur is defined on the uint8 image (cv2.GaussianBlur, zephyr_utils.py:13)")
            img = _f32(img, dev)
            rc = _lib.fn("ossid_zephyr_prep_frame_f32")(img.data_ptr(), depth.data_ptr(), H, W, rgbd.data_ptr(),
                                                        _lib.stream())
    _lib.check(rc, "ossid_zephyr_prep_frame")
    return rgbd


def stage_model(points, normals, colors, dev=None):
    dev = _dev(dev)
    p, n, c = _f32(points, dev), _f32(normals, dev), _f32(colors, dev)
    M = int(p.shape[0])
    tab = torch.empty(M, 12, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.fn("ossid_zephyr_prep_model")(p.data_ptr(), n.data_ptr(), c.data_ptr(), M, tab.data_ptr(),
                                                _lib.stream())
    _lib.check(rc, "ossid_zephyr_prep_model")
    return tab


def inconst_count(rgbd, transforms, tab, cam, margin=INCONST_MARGIN):
    H, W = rgbd.shape[:2]
    N, M = transforms.shape[0], tab.shape[0]
    cnt = torch.empty(N, dtype=torch.int32, device=rgbd.device)
    with torch.cuda.device(rgbd.device):
        rc = _lib.fn("ossid_zephyr_inconst_count")(rgbd.data_ptr(), H, W, transforms.data_ptr(), N, tab.data_ptr(), M,
                                                   *cam, float(margin), cnt.data_ptr(), _lib.stream())
    _lib.check(rc, "ossid_zephyr_inconst_count")
    return cnt


def featurize(rgbd, transforms, tab, cam, sel=None, interp=0, want_uv=True):
    """-> point_x [N', M, 8] float32, uv_original [N', M, 2] int32 (or None)."""
    H, W = rgbd.shape[:2]
    M = tab.shape[0]
    n = int(sel.shape[0]) if sel is not None else int(transforms.shape[0])
    dev = rgbd.device
    px = torch.empty(n, M, DIM_POINT, dtype=torch.float32, device=dev)
    uv = torch.empty(n, M, 2, dtype=torch.int32, device=dev) if want_uv else None
    with torch.cuda.device(dev):
        rc = _lib.fn("ossid_zephyr_featurize")(rgbd.data_ptr(), H, W, transforms.data_ptr(),
                                               None if sel is None else sel.data_ptr(), n, tab.data_ptr(), M, *cam,
                                               int(interp), px.data_ptr(), None if uv is None else uv.data_ptr(),
                                               _lib.stream())
    _lib.check(rc, "ossid_zephyr_featurize")
    return px, uv


def projectPointsUv(pose_hypos, model_points, meta_data):
    """zephyr.utils.projectPointsUv: (N,4,4), (M,3), camera dict -> integer pixel coordinates [N, M, 2]
    (numpy int64, [..., 0] = x / column, [..., 1] = y / row), as utils/zephyr_utils.py:58-65 consumes them."""
    dev = _dev()
    T = _f32(pose_hypos, dev).reshape(-1, 4, 4)
    P = _f32(model_points, dev)
    N, M = int(T.shape[0]), int(P.shape[0])
    uv = torch.empty(N, M, 2, dtype=torch.int32, device=dev)
    if N and M:
        with torch.cuda.device(dev):
            rc = _lib.fn("ossid_zephyr_project_uv")(T.data_ptr(), P.data_ptr(), N, M, *_cam(meta_data), uv.data_ptr(),
                                                    _lib.stream())
        _lib.check(rc, "ossid_zephyr_project_uv")
    return uv.cpu().numpy().astype(np.int64)


class ScoreDataset:
    """Only the surface the OSSID loop touches: the constructor, .dim_point and getPointNetData."""

    def __init__(self, datapoints, dataset_root, dataset_name, args, mode="train"):
        self.datapoints, self.dataset_root, self.dataset_name = datapoints, dataset_root, dataset_name
        self.args, self.mode = args, mode
        name = getattr(args, "dataset", "HSVD_diff_uv_norm")
        if name != "HSVD_diff_uv_norm" or not getattr(args, "no_valid_proj", True) or \
                not getattr(args, "no_valid_depth", True):
            raise NotImplementedError(
                "only dataset='HSVD_diff_uv_norm' with no_valid_proj and no_valid_depth is built "
                "(the configuration of scripts/online_learning.py:191-196)")
        self.inconst_ratio_th = float(getattr(args, "inconst_ratio_th", 100))
        self.interp = int(getattr(args, "interp", 0))  # build option: 0 nearest pixel, 1 bilinear (SPEC.md 3.3)
        self.dim_point = DIM_POINT

    def __len__(self):
        return len(self.datapoints)

    def getPointNetData(self, data, return_uv_original=False):
        with torch.no_grad():
            dev = _dev()
            cam = _cam(data["meta_data"])
            cache = data.get("_frame_cache")
            if cache is None:
                blur = bool(data.get("_blur_on_device", False))
                rgbd = stage_frame(data["img"], data["depth"], dev, blur=blur)
                tab = stage_model(data["model_points"], data["model_normals"], data["model_colors"], dev)
            else:
                rgbd, tab = cache.rgbd, cache.tab
            T = _f32(data["transforms"], dev).reshape(-1, 4, 4)
            N = int(T.shape[0])
            sel = None
            if self.mode == "test" and self.inconst_ratio_th < 100 and N > 0:
                # drop hypotheses with too many free-space violations; the caller reads the filtered
                # transforms / pp_err back from the dict (utils/zephyr_utils.py:39-43)
                cnt = inconst_count(rgbd, T, tab, cam)
                keep = cnt.double() * 100.0 <= self.inconst_ratio_th * float(tab.shape[0])
                sel = torch.nonzero(keep).flatten().to(torch.int32)
                keep_cpu = keep.cpu()
                tr = data["transforms"]
                data["transforms"] = tr[keep_cpu.to(tr.device)] if torch.is_tensor(tr) else tr[keep_cpu.numpy()]
                pe = data.get("pp_err")
                if pe is not None:
                    data["pp_err"] = pe[keep_cpu.to(pe.device)] if torch.is_tensor(pe) else \
                        np.asarray(pe)[keep_cpu.numpy()]
            px, uv = featurize(rgbd, T, tab, cam, sel=sel, interp=self.interp, want_uv=return_uv_original)
            if return_uv_original:
                data["uv_original"] = uv
                return px, uv
            return px
