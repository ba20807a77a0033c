"""Host-side odds and ends shared by the mirrors (camera dict <-> matrix, tensor -> numpy)."""
import numpy as np
import torch

_CAM_KEYS = (("camera_fx", (0, 0)), ("camera_fy", (1, 1)), ("camera_cx", (0, 2)), ("camera_cy", (1, 2)))


def K2meta(cam_K):
    """3x3 intrinsics -> the camera dict Zephyr passes around (reference: ossid/utils/__init__.py:148-156)."""
    meta = {key: cam_K[r, c] for key, (r, c) in _CAM_KEYS}
    meta["camera_scale"] = 1.0
    return meta


def meta2K(meta):
    K = np.eye(3)
    for key, (r, c) in _CAM_KEYS:
        K[r, c] = meta[key]
    return K


def to_np(x):
    """Tensor (any device) -> numpy; numpy and python scalars pass through (ossid/utils/__init__.py:166-173)."""
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return x
