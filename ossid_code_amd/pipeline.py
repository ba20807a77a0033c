"""The steps either side of the hot path that SURVEY.md 8(f) ranks next, kept on the device so that a frame never bounces
through host memory between detector, scorer and pseudo-label:

  make_dtoid_sample      datasets/dtoid_bop_dataset.py:256-338 (__getitem__) + utils/data.py:7-83 (processData)
  visibility_and_iou     scripts/online_learning.py:485-500, :557-558 (render depth, estimate_visib_mask_gt, IoUs)
  render_depth_points    depth-only point-splat renderer in place of pyrender (online_learning.py:485)
  save_results_bop       utils/bop_utils.py:9-52 (BOP csv)

All paths cite /root/reference/python/ossid. Compute goes through libossid_hip.so (csrc/pipeline.hip).
"""
import csv
import os

import numpy as np
import torch

from . import _lib
from .zephyr.score_dataset import _dev, _f32

HEATMAP_SIGMA = float(np.sqrt(1.5))   # dtoid_bop_dataset.py:286


def make_dtoid_sample(img, depth, mask, cam_K, out_hw=None, heatmap_hw=(29, 39)):
    """img uint8 [Ho,Wo,3], depth [Ho,Wo] (m), mask [Ho,Wo] (non-zero = object; uint8 0/255 or float 0..1), cam_K [3,3]
    -> dict of device tensors with the reference's keys and layouts: img [3,H,W] in [0,1], xyz [3,H,W], mask [1,H,W],
    bbox_gt [1,5] (x1,y1,x2,y2,label), heatmap [1,hh,hw] float64. out_hw None keeps the input size (BOP frames)."""
    dev = _dev()
    img = torch.as_tensor(np.ascontiguousarray(img)) if not torch.is_tensor(img) else img
    if img.dtype != torch.uint8:
        raise ValueError("img must be uint8 (utils/data.py:22)")
    img = img.to(dev).contiguous()
    depth = _f32(depth, dev)
    mask = _f32(mask, dev)
    if float(mask.max()) > 1.0:
        mask = mask / 255.0                                   # dtoid_bop_dataset.py:242
    Ho, Wo = int(depth.shape[0]), int(depth.shape[1])
    H, W = (Ho, Wo) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
    K = np.asarray(cam_K, dtype=np.float64)
    out_img = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    out_xyz = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    out_mask = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    bbox = torch.empty(5, dtype=torch.int32, device=dev)
    hh, hw = int(heatmap_hw[0]), int(heatmap_hw[1])
    heat = torch.empty(1, hh, hw, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        s = _lib.stream()
        _lib.check(_lib.fn("ossid_dtoid_prep_sample")(img.data_ptr(), depth.data_ptr(), mask.data_ptr(), Ho, Wo,
                                                      float(np.float32(K[0, 0])), float(np.float32(K[1, 1])),
                                                      float(np.float32(K[0, 2])), float(np.float32(K[1, 2])), H, W,
                                                      out_img.data_ptr(), out_xyz.data_ptr(), out_mask.data_ptr(), s),
                   "ossid_dtoid_prep_sample")
        _lib.check(_lib.fn("ossid_mask_bbox_heatmap")(out_mask.data_ptr(), H, W, hh, hw, float(hh) / float(H),
                                                      HEATMAP_SIGMA, bbox.data_ptr(), heat.data_ptr(), s),
                   "ossid_mask_bbox_heatmap")
    return {"img": out_img, "xyz": out_xyz, "mask": out_mask, "bbox_gt": bbox.to(torch.float32)[None],
            "heatmap": heat}


def collate(samples):
    """datasets/utils.py:35-46 for device samples: stack every tensor key."""
    return {k: torch.stack([s[k] for s in samples], 0) for k in samples[0] if torch.is_tensor(samples[0][k])}


def render_depth_points(pose, model_points, cam_K, hw, radius=1):
    """Depth image [H,W] (m, 0 = background) of the model cloud at `pose`: every point splats a (2r+1)^2 square into a
    z-buffer. Stands in for the mesh renderer of online_learning.py:485 when only the silhouette/depth is needed."""
    dev = _dev()
    T = _f32(np.asarray(pose, dtype=np.float64).reshape(4, 4), dev)
    P = _f32(model_points, dev)
    H, W = int(hw[0]), int(hw[1])
    K = np.asarray(cam_K, dtype=np.float64)
    zbuf = torch.empty(H * W, dtype=torch.int32, device=dev)
    depth = torch.empty(H, W, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.fn("ossid_render_depth_points")(T.data_ptr(), P.data_ptr(), int(P.shape[0]),
                                                  float(np.float32(K[0, 0])), float(np.float32(K[1, 1])),
                                                  float(np.float32(K[0, 2])), float(np.float32(K[1, 2])), H, W, int(radius),
                                                  zbuf.data_ptr(), depth.data_ptr(), _lib.stream())
    _lib.check(rc, "ossid_render_depth_points")
    return depth


def visibility_and_iou(depth_obs, depth_pred, gt_mask=None, gt_mask_visib=None, delta=15 / 1000.0):
    """-> pred_mask, pred_mask_visib (bool [H,W] on the device), iou, iou_visib (python floats; nan without a gt mask)."""
    dev = _dev()
    dob, dpr = _f32(depth_obs, dev), _f32(depth_pred, dev)
    H, W = int(dob.shape[0]), int(dob.shape[1])
    u8 = lambda m: None if m is None else torch.as_tensor(np.ascontiguousarray(np.asarray(m) > 0).astype(np.uint8)).to(dev)  # noqa: E731
    g, gv = u8(gt_mask), u8(gt_mask_visib)
    pm = torch.empty(H, W, dtype=torch.uint8, device=dev)
    vm = torch.empty(H, W, dtype=torch.uint8, device=dev)
    cnt = torch.empty(4, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.fn("ossid_visib_mask_iou")(dob.data_ptr(), dpr.data_ptr(), None if g is None else g.data_ptr(),
                                             None if gv is None else gv.data_ptr(), H, W, float(delta), pm.data_ptr(),
                                             vm.data_ptr(), cnt.data_ptr(), _lib.stream())
    _lib.check(rc, "ossid_visib_mask_iou")
    c = cnt.cpu().numpy().astype(float)
    iou = c[0] / c[1] if g is not None and c[1] > 0 else float("nan")
    iou_v = c[2] / c[3] if gv is not None and c[3] > 0 else float("nan")
    return pm.bool(), vm.bool(), iou, iou_v


def save_results_bop(results, output_folder, result_name, dataset_name, split_name="test", pose_key="pose",
                     score_key="score", time_key="time"):
    """BOP-challenge csv (scene_id,im_id,obj_id,score,R,t,time; translation metres -> millimetres), the wire format
    utils/bop_utils.py:9-52 hands to the unchanged evaluator. Returns the path."""
    name = "%s_%s-%s.csv" % (result_name.replace("_", "-"), dataset_name, split_name)
    path = os.path.join(output_folder, name)
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["scene_id", "im_id", "obj_id", "score", "R", "t", "time"])
        w.writeheader()
        for r in results:
            mat = np.array(r[pose_key], dtype=np.float64, copy=True)
            mat[:3, 3] *= 1000.0
            w.writerow({"scene_id": r["scene_id"], "im_id": r["im_id"], "obj_id": r["obj_id"],
                        "score": r.get(score_key, 1), "R": " ".join(str(v) for v in mat[:3, :3].flatten()),
                        "t": " ".join(str(v) for v in mat[:3, 3].flatten()), "time": r.get(time_key, -1)})
    return path
