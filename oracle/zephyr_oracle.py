"""ctypes front end of oracle/zephyr_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED for the Zephyr half (see the C file's header and DESIGN.md): the oracle is the
executable form of SPEC.md, anchored on the reference's call sites
(/root/reference/python/ossid/utils/zephyr_utils.py:10-71, scripts/online_learning.py:187-227).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libzephyr_oracle.so")
_lib = None

# (kpad, cout) of the 12 folded dense layers, SPEC.md 4.3
LAYER_K = (8, 64, 64, 136, 128, 128, 264, 256, 512, 1024, 512, 256)
LAYER_C = (64, 64, 128, 128, 128, 256, 256, 512, 1024, 512, 256, 1)


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "zephyr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "CC=gcc"], stdout=subprocess.DEVNULL)
    return _SO


class _PN2(C.Structure):
    _fields_ = [("W", C.c_void_p * 12), ("b", C.c_void_p * 12),
                ("npoint1", C.c_int), ("nsample1", C.c_int), ("npoint2", C.c_int), ("nsample2", C.c_int),
                ("radius1", C.c_float), ("radius2", C.c_float)]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def set_product_mode(mode):
    """0 = SPEC.md's fmaf chains (the oracle); 2 = EXPERIMENT: emulate six-product split-bf16 arithmetic in SA1 / SA2
    (tools/six_product_emulation.py only; no parity test and no product code path uses it)."""
    return lib().ozr_set_product_mode(C.c_int(int(mode)))


def set_threads(n):
    return lib().ozr_set_threads(C.c_int(int(n)))


def blur5_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    rc = lib().ozr_blur5_u8(_p(img), H, W, ch, _p(out))
    assert rc == 0
    return out


def u8_to_unit(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = np.empty(a.shape, np.float32)
    lib().ozr_u8_to_unit(_p(a), C.c_size_t(a.size), _p(out))
    return out


def pack_rgbd(rgb, depth):
    rgb, depth = _f32(rgb), _f32(depth)
    H, W = depth.shape
    out = np.empty((H, W, 4), np.float32)
    lib().ozr_pack_rgbd(_p(rgb), _p(depth), H, W, _p(out))
    return out


def rgb_to_hsv(rgb):
    rgb = _f32(rgb)
    out = np.empty_like(rgb)
    lib().ozr_rgb_to_hsv(_p(rgb), C.c_size_t(rgb.size // 3), _p(out))
    return out


def prep_model(pts, nrm, rgb):
    pts, nrm, rgb = _f32(pts), _f32(nrm), _f32(rgb)
    M = pts.shape[0]
    tab = np.empty((M, 12), np.float32)
    lib().ozr_prep_model(_p(pts), _p(nrm), _p(rgb), M, _p(tab))
    return tab


def _cam(K):
    K = np.asarray(K, dtype=np.float64)
    return [C.c_float(np.float32(v)) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2])]


def project_uv(T, pts, K):
    T, pts = _f32(T).reshape(-1, 16), _f32(pts)
    N, M = T.shape[0], pts.shape[0]
    uv = np.empty((N, M, 2), np.int32)
    rc = lib().ozr_project_uv(_p(T), _p(pts), N, M, *_cam(K), _p(uv))
    assert rc == 0
    return uv


def inconst_count(rgbd, T, tab, K, margin=0.02):
    rgbd, T, tab = _f32(rgbd), _f32(T).reshape(-1, 16), _f32(tab)
    H, W = rgbd.shape[:2]
    N, M = T.shape[0], tab.shape[0]
    cnt = np.empty((N,), np.int32)
    rc = lib().ozr_inconst_count(_p(rgbd), H, W, _p(T), N, _p(tab), M, *_cam(K), C.c_float(margin), _p(cnt))
    assert rc == 0
    return cnt


def featurize(rgbd, T, tab, K, sel=None, interp=0, want_uv=True):
    rgbd, T, tab = _f32(rgbd), _f32(T).reshape(-1, 16), _f32(tab)
    H, W = rgbd.shape[:2]
    M = tab.shape[0]
    if sel is not None:
        sel = np.ascontiguousarray(sel, dtype=np.int32)
        n = sel.shape[0]
    else:
        n = T.shape[0]
    px = np.empty((n, M, 8), np.float32)
    uv = np.empty((n, M, 2), np.int32) if want_uv else None
    rc = lib().ozr_featurize(_p(rgbd), H, W, _p(T), _p(sel), n, _p(tab), M, *_cam(K), int(interp), _p(px), _p(uv))
    assert rc == 0, rc
    return px, uv


def _pn2_struct(weights, cfg):
    """weights: list of 12 (W[cout,kpad], b[cout]) float32 pairs (folded, canonical channel order)."""
    st = _PN2()
    keep = []
    for i, (W, b) in enumerate(weights):
        W, b = _f32(W), _f32(b)
        assert W.shape == (LAYER_C[i], LAYER_K[i]), (i, W.shape)
        assert b.shape == (LAYER_C[i],)
        keep += [W, b]
        st.W[i] = W.ctypes.data
        st.b[i] = b.ctypes.data
    st.npoint1, st.nsample1 = cfg.get("npoint1", 512), cfg.get("nsample1", 64)
    st.npoint2, st.nsample2 = cfg.get("npoint2", 128), cfg.get("nsample2", 64)
    st.radius1, st.radius2 = cfg.get("radius1", 0.2), cfg.get("radius2", 0.4)
    return st, keep


def pn2_score(point_x, weights, cfg=None, debug=False):
    cfg = cfg or {}
    point_x = _f32(point_x)
    B, M, D = point_x.shape
    assert D == 8
    st, keep = _pn2_struct(weights, cfg)
    scores = np.empty((B,), np.float32)
    dbg = {}
    args = [None] * 7
    if debug:
        np1, ns1, np2, ns2 = st.npoint1, st.nsample1, st.npoint2, st.nsample2
        dbg = dict(fps1=np.empty((B, np1), np.int32), ball1=np.empty((B, np1, ns1), np.int32),
                   feat1=np.empty((B, np1, 128), np.float32), fps2=np.empty((B, np2), np.int32),
                   ball2=np.empty((B, np2, ns2), np.int32), feat2=np.empty((B, np2, 256), np.float32),
                   feat3=np.empty((B, 1024), np.float32))
        args = [_p(dbg[k]) for k in ("fps1", "ball1", "feat1", "fps2", "ball2", "feat2", "feat3")]
    rc = lib().ozr_pn2_score(_p(point_x), B, M, C.byref(st), _p(scores), *args)
    assert rc == 0, rc
    del keep
    return (scores, dbg) if debug else scores


def fps(xyz, npoint):
    xyz = _f32(xyz)
    B, n, stride = xyz.shape
    idx = np.empty((B, npoint), np.int32)
    rc = lib().ozr_fps(_p(xyz), stride, B, n, npoint, _p(idx))
    assert rc == 0
    return idx


def ball_query(xyz, centres, radius, nsample):
    xyz, centres = _f32(xyz), _f32(centres)
    B, n, stride = xyz.shape
    npoint = centres.shape[1]
    idx = np.empty((B, npoint, nsample), np.int32)
    rc = lib().ozr_ball_query(_p(xyz), stride, B, n, _p(centres), npoint, C.c_float(radius), nsample, _p(idx))
    assert rc == 0
    return idx
