"""Mirrors of the un-vendored `zephyr` interfaces that OSSID's online loop calls
(/root/reference/python/ossid/scripts/online_learning.py:28-41, utils/zephyr_utils.py:8)."""
from .pointnet2 import PointNet2SSG, fold_pn2, pack_pn2  # noqa: F401
from .score_dataset import ScoreDataset, projectPointsUv, stage_frame, stage_model, featurize, inconst_count  # noqa: F401
from .options import getOptions, K2meta, meta2K  # noqa: F401
