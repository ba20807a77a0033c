"""Synthetic Zephyr inputs of SURVEY.md 8(d) cfg-2 (the bench workload and the parity-test inputs).

640x480 RGB-D frame, LM camera, a colour-textured sphere of radius 5 cm at (0, 0, 0.8) m rendered into the
depth map, M model points on that sphere (Fibonacci lattice, radial normals, colours read off the frame at the
ground-truth projection), and N pose hypotheses = perturbations of the ground truth in the style of
/root/reference/python/ossid/utils/__init__.py:82-98 (axis ~ N(0,1) normalised, angle ~ N(0, 0.2 rad),
translation noise ~ N(0, 1 cm)); hypothesis 0 is the ground truth itself.
"""
import numpy as np

CAM_K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]])
RADIUS = 0.05
T_GT = np.array([0.0, 0.0, 0.8])


def _smooth_field(rng, H, W, C, sigma):
    """Low-pass of U{0..255}: separable box filters applied three times (~ Gaussian), numpy only."""
    f = rng.integers(0, 256, size=(H, W, C)).astype(np.float64)
    k = max(1, int(sigma))
    for _ in range(3):
        for axis, n in ((0, H), (1, W)):
            c = np.cumsum(np.concatenate([np.zeros_like(np.take(f, [0], axis)), f], axis), axis)
            idx_hi = np.clip(np.arange(n) + k + 1, 0, n)
            idx_lo = np.clip(np.arange(n) - k, 0, n)
            f = (np.take(c, idx_hi, axis) - np.take(c, idx_lo, axis)) / \
                (idx_hi - idx_lo).reshape([-1 if a == axis else 1 for a in range(3)])
    f -= f.min()
    f /= max(f.max(), 1e-9)
    return f


def _rodrigues(axis, angle):
    x, y, z = axis
    Kx = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * (Kx @ Kx)


def make_frame(seed=42, H=480, W=640):
    """-> img uint8 [H,W,3], depth float32 [H,W] (metres, 0 = invalid)."""
    rng = np.random.default_rng(seed)
    img = (_smooth_field(rng, H, W, 3, 6) * 255.0).round().astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    depth = 0.6 + 0.6 * (0.5 * xx / W + 0.5 * yy / H)            # tilted plane, z in [0.6, 1.2]
    for _ in range(6):                                             # six Gaussian bumps
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        s, a = rng.uniform(20, 60), rng.uniform(-0.08, 0.08)
        depth += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    # the object: a sphere in front of the background
    fx, fy, cx0, cy0 = CAM_K[0, 0], CAM_K[1, 1], CAM_K[0, 2], CAM_K[1, 2]
    dx, dy = (xx + 0.5 - cx0) / fx, (yy + 0.5 - cy0) / fy
    a = dx * dx + dy * dy + 1.0
    b = -2.0 * T_GT[2]
    c = T_GT[2] ** 2 - RADIUS ** 2
    disc = b * b - 4 * a * c
    hit = disc > 0
    z = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0.0))) / (2 * a), 0.0)
    depth = np.where(hit & (z < depth), z, depth)
    depth += rng.uniform(-0.002, 0.002, size=depth.shape)
    depth[rng.random(depth.shape) < 0.05] = 0.0                    # 5 % invalid pixels
    return img, depth.astype(np.float32)


def make_model(img, M=2048):
    """-> points, normals, colors [M,3] float64 (colours in [0,1], read at the ground-truth projection)."""
    i = np.arange(M) + 0.5
    phi = np.arccos(1 - 2 * i / M)
    theta = np.pi * (1 + 5 ** 0.5) * i
    n = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], 1)
    pts = RADIUS * n
    cam = pts + T_GT
    u = np.clip((cam[:, 0] / cam[:, 2] * CAM_K[0, 0] + CAM_K[0, 2]).astype(int), 0, img.shape[1] - 1)
    v = np.clip((cam[:, 1] / cam[:, 2] * CAM_K[1, 1] + CAM_K[1, 2]).astype(int), 0, img.shape[0] - 1)
    colors = img[v, u].astype(np.float64) / 255.0
    return pts, n, colors


def make_hypotheses(N=1000, seed=42):
    """-> pose_hypos [N,4,4] float64; row 0 is the ground truth."""
    rng = np.random.default_rng(seed + 1)
    T = np.tile(np.eye(4), (N, 1, 1))
    T[:, :3, 3] = T_GT
    for k in range(1, N):
        axis = rng.normal(0, 1.0, 3)
        axis /= np.linalg.norm(axis)
        T[k, :3, :3] = _rodrigues(axis, rng.normal(0, 0.2))
        T[k, :3, 3] += rng.normal(0, 0.01, 3)
    return T


def make_scoring_inputs(N=1000, M=2048, seed=42, H=480, W=640):
    """The dict networkInference takes (utils/zephyr_utils.py:10; packed at online_learning.py:455-459)."""
    img, depth = make_frame(seed, H, W)
    pts, nrm, col = make_model(img, M)
    return {"img": img, "depth": depth, "cam_K": CAM_K.copy(), "model_points": pts, "model_normals": nrm,
            "model_colors": col, "pose_hypos": make_hypotheses(N, seed)}


def random_pn2_state(model, seed=0):
    """Random-init weights (there is no checkpoint offline) with non-trivial BatchNorm statistics, in place."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    return model
