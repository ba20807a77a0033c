"""zephyr.options.getOptions / zephyr.utils.K2meta, meta2K -- the bits of Zephyr's option plumbing the OSSID loop
touches (/root/reference/python/ossid/scripts/online_learning.py:187-209; K2meta also utils/__init__.py:148-156)."""
import argparse

import numpy as np


def getOptions():
    """An argparse parser holding the Zephyr options online_learning.py reads or overwrites (:187-209)."""
    p = argparse.ArgumentParser()
    p.add_argument("--model_name", type=str, default="pn2")
    p.add_argument("--dataset", type=str, default="HSVD_diff_uv_norm")
    p.add_argument("--no_valid_proj", action="store_true")
    p.add_argument("--no_valid_depth", action="store_true")
    p.add_argument("--inconst_ratio_th", type=float, default=100)
    p.add_argument("--dataset_root", type=str, nargs="*", default=[""])
    p.add_argument("--dataset_name", type=str, nargs="*", default=["lmo"])
    p.add_argument("--resume_path", type=str, default=None)
    p.add_argument("--test_dataset", action="store_true")
    p.add_argument("--dim_point", type=int, default=8)
    p.add_argument("--extra_bottleneck_dim", type=int, default=0)
    p.add_argument("--interp", type=int, default=0, help="build option: 0 nearest pixel, 1 bilinear gather")
    return p


from ..hostutil import K2meta, meta2K  # noqa: E402,F401
