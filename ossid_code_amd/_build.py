"""Builds libossid_hip.so (the C-ABI boundary, include/ossid_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the GPU box.
-ffp-contract=off: SPEC.md fixes the operation order of every float op; fused multiply-adds appear
only where the source writes fmaf / an MFMA.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("OSSID_HIP_LIB") or os.path.join(_HERE, "libossid_hip.so")
SOURCES = ["zephyr.hip", "pn2.hip", "dtoid.hip", "conv.hip", "segtail.hip", "pipeline.hip", "train.hip", "wino.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libossid_hip.so cannot be built")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(CSRC, "common.h"), os.path.join(_HERE, "..", "include", "ossid_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False):
    """Compile every HIP source into ossid_code_amd/libossid_hip.so; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    extra = os.environ.get("OSSID_HIPCC_EXTRA", "").split()       # A/B builds of ablation switches (-DOSSID_...)
    cmd = [_hipcc()] + FLAGS + extra + ["-o", LIB_PATH + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
