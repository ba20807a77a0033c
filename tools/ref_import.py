"""Imports the runnable part of the reference DTOID head from /root/reference (build container only) with inert
placeholder modules for packages that are absent offline (cv2, torchvision, pytorch_lightning). Recipe from
SURVEY.md 8c. Used ONLY by tools/gen_golden_dtoid.py to produce fixtures; nothing of the reference is copied."""
import sys
import types

REF = "/root/reference/python"


def load():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ("cv2", "torchvision", "torchvision.models", "torchvision.transforms",
                 "torchvision.transforms.transforms", "torchvision.ops", "torchvision.ops.boxes"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    tv.models, tv.transforms, tv.ops = (sys.modules["torchvision.models"], sys.modules["torchvision.transforms"],
                                        sys.modules["torchvision.ops"])
    sys.modules["torchvision.transforms"].transforms = sys.modules["torchvision.transforms.transforms"]
    sys.modules["torchvision.ops"].boxes = sys.modules["torchvision.ops.boxes"]
    sys.modules["torchvision.ops"].nms = None
    import numpy.lib
    if "numpy.lib.type_check" not in sys.modules or not hasattr(sys.modules["numpy.lib.type_check"], "imag"):
        m = types.ModuleType("numpy.lib.type_check")
        import numpy as np
        m.imag = np.imag
        sys.modules["numpy.lib.type_check"] = m
    for pkg, path in (("ossid.models", REF + "/ossid/models"), ("ossid.models.dtoid", REF + "/ossid/models/dtoid")):
        if pkg not in sys.modules:
            mod = types.ModuleType(pkg)
            mod.__path__ = [path]
            sys.modules[pkg] = mod
    import ossid.models.dtoid.network as network
    import ossid.models.dtoid.loss as loss
    import ossid.models.dtoid.anchors as anchors
    import ossid.utils as utils
    return network, loss, anchors, utils
