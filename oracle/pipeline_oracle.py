"""numpy restatements of the SURVEY.md 8(f) steps -- TEST INFRASTRUCTURE ONLY (tests/ may import this; the product never
does). Each function cites the reference lines it follows (/root/reference/python/ossid/...); the Gaussian heat map is
pinned against the reference's own heatmapGaussain through tests/golden/dtoid_head.npz."""
import numpy as np


def depth2xyz(depth, K):
    """utils/__init__.py:241-255"""
    h, w = depth.shape
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    z = depth.astype(np.float32)
    x = (xs.astype(np.float32) - np.float32(K[0, 2])) * z / np.float32(K[0, 0])
    y = (ys.astype(np.float32) - np.float32(K[1, 2])) * z / np.float32(K[1, 1])
    return np.stack([x, y, z], 2)


def _axis(n_dst, n_src):
    if n_dst == n_src:
        i = np.arange(n_dst)
        return i, i, np.zeros(n_dst, np.float32)
    s = (np.arange(n_dst, dtype=np.float32) + np.float32(0.5)) * (np.float32(n_src) / np.float32(n_dst)) - np.float32(0.5)
    f = np.floor(s)
    i0 = f.astype(int)
    w = (s - f).astype(np.float32)
    i1 = i0 + 1
    low = i0 < 0
    i0, i1, w = np.where(low, 0, i0), np.where(low, 0, i1), np.where(low, np.float32(0), w)
    i1 = np.minimum(i1, n_src - 1)
    i0 = np.minimum(i0, n_src - 1)
    return i0, i1, w


def resize_bilinear(a, H, W):
    """cv2.resize(..., INTER_LINEAR) on float data: pixel centres aligned, borders clamped (utils/data.py:45-47)."""
    a = a.astype(np.float32)
    y0, y1, wy = _axis(H, a.shape[0])
    x0, x1, wx = _axis(W, a.shape[1])
    if a.ndim == 3:
        wy_, wx_ = wy[:, None, None], wx[None, :, None]
    else:
        wy_, wx_ = wy[:, None], wx[None, :]
    one = np.float32(1)
    p00, p10, p01, p11 = a[y0][:, x0], a[y0][:, x1], a[y1][:, x0], a[y1][:, x1]
    return (p00 * ((one - wx_) * (one - wy_)) + p10 * (wx_ * (one - wy_))) + (p01 * ((one - wx_) * wy_) + p11 * (wx_ * wy_))


def process_data(img, mask, depth, K, H, W):
    """utils/data.py:7-83 (no crop, no warp): img [3,H,W] float32 in [0,1], mask [1,H,W], xyz [3,H,W]"""
    xyz = depth2xyz(depth, K)
    if (H, W) == img.shape[:2]:
        im = img.astype(np.float32)
        m, x = mask.astype(np.float32), xyz
    else:
        im = np.floor(resize_bilinear(img, H, W) + np.float32(0.5))
        m, x = resize_bilinear(mask, H, W), resize_bilinear(xyz, H, W)
    return (im.transpose(2, 0, 1) / np.float32(255)).astype(np.float32), m[None].astype(np.float32), \
        x.transpose(2, 0, 1).astype(np.float32)


def mask_bbox(mask):
    """dtoid_bop_dataset.py:274-279"""
    nz = np.stack(mask.nonzero(), 1)
    if len(nz) == 0:
        return np.array([1 << 30, 1 << 30, -1, -1, -1])
    (y1, x1), (y2, x2) = nz.min(0), nz.max(0)
    return np.array([x1, y1, x2, y2, 1])


def heatmap_gaussian(h, w, cx, cy, sigma):
    """utils/__init__.py:354-367"""
    x, y = np.meshgrid(np.arange(int(round(w))), np.arange(int(round(h))))
    dst = np.sqrt((x - cx) ** 2 + (y - cy) ** 2)
    return np.exp(-(dst ** 2 / (2.0 * sigma ** 2)))


def render_depth_points(T, pts, K, H, W, radius):
    T, p = T.astype(np.float32), pts.astype(np.float32)
    fx, fy, cx, cy = (np.float32(K[0, 0]), np.float32(K[1, 1]), np.float32(K[0, 2]), np.float32(K[1, 2]))
    cam = np.stack([((T[i, 0] * p[:, 0] + T[i, 1] * p[:, 1]) + T[i, 2] * p[:, 2]) + T[i, 3] for i in range(3)], 1)
    depth = np.full((H, W), np.inf, np.float32)
    for X, Y, Z in cam:
        if not Z > np.float32(1e-6):
            continue
        u, v = int(np.floor((X / Z) * fx + cx)), int(np.floor((Y / Z) * fy + cy))
        for vv in range(max(v - radius, 0), min(v + radius, H - 1) + 1):
            for uu in range(max(u - radius, 0), min(u + radius, W - 1) + 1):
                depth[vv, uu] = min(depth[vv, uu], Z)
    depth[np.isinf(depth)] = 0
    return depth


def visib_and_iou(d_obs, d_pred, gt, gt_visib, delta):
    """bop_toolkit_lib.visibility (mode 'bop19') + online_learning.py:557-558"""
    d_obs, d_pred = d_obs.astype(np.float32), d_pred.astype(np.float32)
    pm = d_pred > 0
    vm = np.logical_and(np.logical_or((d_pred - d_obs) <= np.float32(delta), d_obs == 0), pm)
    iou = np.logical_and(pm, gt).sum() / float(np.logical_or(pm, gt).sum())
    iou_v = np.logical_and(vm, gt_visib).sum() / float(np.logical_or(vm, gt_visib).sum())
    return pm, vm, iou, iou_v
