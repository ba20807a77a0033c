"""Drop-in wiring for the unchanged reference caller, scripts/online_learning.py.

    import ossid_code_amd.compat as compat
    compat.install()              # before `import ossid.scripts.online_learning`

install() registers this package's mirrors under the module paths the script imports from
(/root/reference/python/ossid/scripts/online_learning.py:18-41), so that

    from ossid.models.dtoid import DtoidNet
    from ossid.utils.zephyr_utils import networkInference
    from zephyr.datasets.score_dataset import ScoreDataset
    from zephyr.models.pointnet2 import PointNet2SSG
    from zephyr.options import getOptions
    from zephyr.utils import K2meta, meta2K, projectPointsUv

resolve to the MI355X path. Only the hot-path names are provided; everything else the script imports (Halcon PPF,
ICP, the renderer, BOP tooling, datasets) stays with the reference / zephyr installation -- when a real `zephyr` or
`ossid` package is importable, just these attributes are overridden on it, nothing else is shadowed.
"""
import importlib
import sys
import types


def _module(name):
    try:
        return importlib.import_module(name)
    except Exception:
        mod = types.ModuleType(name)
        mod.__path__ = []
        sys.modules[name] = mod
        parent, _, child = name.rpartition(".")
        if parent:
            setattr(_module(parent), child, mod)
        return mod


def install():
    from . import dtoid, hostutil, scoring, zephyr
    table = {
        "zephyr.datasets.score_dataset": {"ScoreDataset": zephyr.ScoreDataset},
        "zephyr.models.pointnet2": {"PointNet2SSG": zephyr.PointNet2SSG},
        "zephyr.options": {"getOptions": zephyr.getOptions},
        "zephyr.utils": {"projectPointsUv": zephyr.projectPointsUv, "K2meta": hostutil.K2meta,
                         "meta2K": hostutil.meta2K},
        "ossid.utils.zephyr_utils": {"networkInference": scoring.networkInference,
                                     "filterHypoByMask": scoring.filterHypoByMask},
        "ossid.models.dtoid": {"DtoidNet": dtoid.DtoidNet},
    }
    for modname, attrs in table.items():
        mod = _module(modname)
        for k, v in attrs.items():
            setattr(mod, k, v)
    return sorted(table)
