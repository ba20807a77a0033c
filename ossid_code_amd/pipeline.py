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


def save_det_results(results, folder):
    """Per-image detection text files for the mAP evaluator (utils/detection.py:14-33): `results` maps (scene_id, im_id)
    to rows (obj_id, x1, y1, x2, y2) for ground truth or (obj_id, x1, y1, x2, y2, score) for detections."""
    os.makedirs(folder, exist_ok=True)
    for (scene_id, im_id), rows in results.items():
        with open(os.path.join(folder, "s%06d_i%06d.txt" % (scene_id, im_id)), "w") as f:
            for row in rows:
                if len(row) == 5:
                    f.write("obj_%06d %d %d %d %d\n" % tuple(row))
                elif len(row) == 6:
                    obj_id, x1, y1, x2, y2, score = row
                    f.write("obj_%06d %04f %d %d %d %d\n" % (obj_id, score, x1, y1, x2, y2))


def expand_box(x1, y1, x2, y2, img_h, img_w, expand_ratio):
    """Grow a box about its centre and clip to the image (utils/__init__.py:11-16; online_learning.py:365 uses 1.2)."""
    cx, cy, hw, hh = (x1 + x2) / 2, (y1 + y2) / 2, (x2 - x1) / 2 * expand_ratio, (y2 - y1) / 2 * expand_ratio
    return max(0, cx - hw), max(0, cy - hh), min(img_w - 1, cx + hw), min(img_h - 1, cy + hh)


def _rotmat_to_quat(Rm):
    """xyzw quaternion of a rotation matrix (scipy's convention, sign-free use only)."""
    Rm = np.asarray(Rm, dtype=np.float64)
    t = np.trace(Rm)
    cand = np.array([Rm[0, 0], Rm[1, 1], Rm[2, 2], t])
    k = int(cand.argmax())
    if k == 3:
        q = np.array([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1], 1.0 + t])
    else:
        i, j, l = k, (k + 1) % 3, (k + 2) % 3
        q = np.empty(4)
        q[i] = 1.0 - t + 2.0 * Rm[i, i]
        q[j] = Rm[j, i] + Rm[i, j]
        q[l] = Rm[l, i] + Rm[i, l]
        q[3] = Rm[l, j] - Rm[j, l]
    return q / np.linalg.norm(q)


class TemplateBank:
    """The object templates, resident on the device in the layout DtoidNet takes (datasets/template_dataset.py:60-117:
    RGB /255 -> float32 [V,3,124,124], mask [V,1,124,124]) plus the two view-selection rules of
    datasets/dtoid_bop_dataset.py:294-318: test = all views, thinned to n_local_test by rounded linspace; train = one of the
    `sample_from` views whose grid rotation is closest (quaternion angle) to the pose."""

    def __init__(self, n_local_test=160, sample_from=10):
        self.n_local_test, self.sample_from = int(n_local_test), int(sample_from)
        self.img, self.mask, self.quats = {}, {}, {}

    def add(self, obj_id, imgs, masks, grid_quats=None):
        dev = _dev()
        imgs = torch.as_tensor(np.ascontiguousarray(imgs)) if not torch.is_tensor(imgs) else imgs
        if imgs.dtype == torch.uint8:                                    # [V,h,w,3] uint8 as read from disk
            imgs = imgs.to(dev).permute(0, 3, 1, 2).to(torch.float32) / 255.0
        self.img[obj_id] = imgs.to(dev, torch.float32).contiguous()
        m = torch.as_tensor(np.ascontiguousarray(masks)) if not torch.is_tensor(masks) else masks
        m = m.to(dev, torch.float32)
        self.mask[obj_id] = (m if m.dim() == 4 else m[:, None]).contiguous()
        if grid_quats is not None:
            self.quats[obj_id] = np.asarray(grid_quats, dtype=np.float64)

    def view(self, obj_id, view_id):
        return self.img[obj_id][int(view_id)], self.mask[obj_id][int(view_id)]

    def test_views(self, obj_id):
        n = int(self.img[obj_id].shape[0])
        if n > self.n_local_test:
            return np.linspace(0, n - 1, self.n_local_test).round().astype(int)
        return np.arange(n)

    def all_local(self, obj_id):
        ids = torch.as_tensor(self.test_views(obj_id), device=self.img[obj_id].device)
        return self.img[obj_id][ids], self.mask[obj_id][ids]

    def nearest_views(self, obj_id, rot_gt):
        """View ids sorted by 2*acos(min(|q_view . q_gt|, 1-1e-7)) (utils/__init__.py:18-32)."""
        q = _rotmat_to_quat(rot_gt)
        dot = np.minimum(np.abs(self.quats[obj_id] @ q), 1.0 - 1e-7)
        return np.argsort(2.0 * np.arccos(dot), kind="stable")

    def train_view(self, obj_id, rot_gt, rng):
        return int(rng.choice(self.nearest_views(obj_id, rot_gt)[: self.sample_from]))


class PseudoLabelSet:
    """The finetune set of the online loop, on the device: frames whose Zephyr score passed the threshold, each with the
    pseudo ground-truth mask the scorer produced (DtoidBopDataset with zephyr_results, dtoid_bop_dataset.py:230-235,
    :256-338; filled at online_learning.py:506-516). Items are the D14 batch dict rows, built by make_dtoid_sample."""

    def __init__(self, bank, mode="train", seed=0):
        self.bank, self.mode, self.rng = bank, mode, np.random.default_rng(seed)
        self.keys, self.frames = [], {}

    def __len__(self):
        return len(self.keys)

    def add(self, obj_id, scene_id, im_id, img, depth, cam_K, mask, score, rot=None):
        key = (obj_id, scene_id, im_id)
        if key not in self.frames:
            self.keys.append(key)
        self.frames[key] = {"img": img, "depth": depth, "cam_K": cam_K, "pred_mask_visib": mask, "score": score,
                            "rot": rot, "sample": None}

    def updateZephyrMask(self, obj_id, scene_id, im_id, mask, score):
        fr = self.frames[(obj_id, scene_id, im_id)]
        fr["pred_mask_visib"], fr["score"], fr["sample"] = mask, score, None

    def __getitem__(self, idx):
        obj_id, scene_id, im_id = key = self.keys[idx]
        fr = self.frames[key]
        if fr["sample"] is None:                                          # geometry of a frame is computed once
            m = fr["pred_mask_visib"]
            m = m.float() if torch.is_tensor(m) else np.asarray(m, dtype=np.float32)
            fr["sample"] = make_dtoid_sample(fr["img"], fr["depth"], m, fr["cam_K"])
        out = dict(fr["sample"])
        n_views = int(self.bank.img[obj_id].shape[0])
        out["gimg"], out["gmask"] = self.bank.view(obj_id, self.rng.integers(n_views))
        if self.mode == "train":
            if fr["rot"] is not None and obj_id in self.bank.quats:
                lv = self.bank.train_view(obj_id, fr["rot"], self.rng)
            else:
                lv = int(self.rng.integers(n_views))
            out["limg"], out["lmask"] = self.bank.view(obj_id, lv)
        else:
            out["limg"], out["lmask"] = self.bank.all_local(obj_id)
        out.update(obj_id=int(obj_id), scene_id=scene_id, im_id=im_id, zephyr_score=fr["score"])
        return out
