"""Independent numpy (float32, explicit operation order) restatement of SPEC.md sections 2-3 -- test infrastructure
that pins oracle/zephyr_oracle.c from a second implementation (vectorised arrays instead of scalar loops)."""
import numpy as np

f32 = np.float32


def blur5_u8(img):
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    H, W = img.shape[:2]
    p = np.pad(img.astype(np.int64), ((2, 2), (2, 2), (0, 0)), mode="reflect")  # reflect == BORDER_REFLECT_101
    rows = sum(k[i] * p[:, i:i + W] for i in range(5))
    S = sum(k[i] * rows[i:i + H] for i in range(5))
    return ((S + 128) >> 8).astype(np.uint8)


def rgb_to_hsv(rgb):
    rgb = rgb.astype(f32)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    mx, mn = rgb.max(-1), rgb.min(-1)
    delta = mx - mn
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(mx > 0, delta / mx, f32(0))
        hr, hg, hb = (g - b) / delta, f32(2) + (b - r) / delta, f32(4) + (r - g) / delta
    h = np.where(r == mx, hr, np.where(g == mx, hg, hb))
    h = np.where(delta > 0, h, f32(0)) / f32(6)
    h = np.where(h < 0, h + f32(1), h)
    return np.stack([h, s.astype(f32), mx], -1).astype(f32)


def project(T, pts, K):
    """-> cam [N,M,3] f32, uf, vf [N,M] f32, uv [N,M,2] int32"""
    T, p = T.astype(f32), pts.astype(f32)
    fx, fy, cx, cy = (f32(K[0, 0]), f32(K[1, 1]), f32(K[0, 2]), f32(K[1, 2]))
    R, t = T[:, None, :3, :3], T[:, None, :3, 3]
    x, y, z = p[None, :, 0], p[None, :, 1], p[None, :, 2]
    cam = np.stack([((R[..., i, 0] * x + R[..., i, 1] * y) + R[..., i, 2] * z) + t[..., i] for i in range(3)], -1)
    ok = cam[..., 2] > f32(1e-6)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        zs = np.where(ok, cam[..., 2], f32(1))
        uf = (cam[..., 0] / zs) * fx + cx
        vf = (cam[..., 1] / zs) * fy + cy
    ok &= np.isfinite(uf) & np.isfinite(vf) & (np.abs(uf) < f32(1e9)) & (np.abs(vf) < f32(1e9))
    u = np.where(ok, np.trunc(np.where(ok, uf, 0)), -1).astype(np.int32)
    v = np.where(ok, np.trunc(np.where(ok, vf, 0)), -1).astype(np.int32)
    return cam.astype(f32), uf, vf, np.stack([u, v], -1)


def featurize(rgbd, T, pts, nrm, col_rgb, K):
    """nearest-pixel mode; -> point_x [N,M,8] f32, uv_original [N,M,2] i32, inconst counts [N] (margin 0.02)"""
    H, W = rgbd.shape[:2]
    cam, uf, vf, uv = project(T, pts, K)
    u, v = uv[..., 0], uv[..., 1]
    inb = (u >= 0) & (u < W) & (v >= 0) & (v < H)
    uc, vc = np.where(inb, u, 0), np.where(inb, v, 0)
    obs = rgbd[vc, uc]
    ohsv = rgb_to_hsv(obs[..., :3])
    mhsv = rgb_to_hsv(col_rgb.astype(f32))[None]
    dh = np.abs(ohsv[..., 0] - mhsv[..., 0])
    dh = np.minimum(dh, f32(1) - dh)
    od = obs[..., 3]
    dd = np.where(od > 0, od - cam[..., 2], f32(0))
    Tn = T.astype(f32)
    R = Tn[:, None, :3, :3]
    n = nrm.astype(f32)[None]
    nr = np.stack([(R[..., i, 0] * n[..., 0] + R[..., i, 1] * n[..., 1]) + R[..., i, 2] * n[..., 2] for i in range(3)], -1)
    dot = (nr[..., 0] * cam[..., 0] + nr[..., 1] * cam[..., 1]) + nr[..., 2] * cam[..., 2]
    ln = np.sqrt((cam[..., 0] * cam[..., 0] + cam[..., 1] * cam[..., 1]) + cam[..., 2] * cam[..., 2])
    with np.errstate(divide="ignore", invalid="ignore"):
        cosn = np.where(ln > 0, dot / ln, f32(0))
    M = pts.shape[0]
    mu = (uc.sum(1).astype(np.int64).astype(f32) / f32(M))[:, None]
    mv = (vc.sum(1).astype(np.int64).astype(f32) / f32(M))[:, None]
    du, dv = uc.astype(f32) - mu, vc.astype(f32) - mv
    ext = np.maximum(np.abs(du).max(1), np.abs(dv).max(1))[:, None]
    ext = np.where(ext > 0, ext, f32(1))
    px = np.stack([du / ext, dv / ext, np.zeros_like(du), dh, ohsv[..., 1] - mhsv[..., 1], ohsv[..., 2] - mhsv[..., 2],
                   dd, cosn], -1).astype(f32)
    inconst = (inb & (od > 0) & ((od - cam[..., 2]) > f32(0.02))).sum(1).astype(np.int32)
    return px, uv, inconst
