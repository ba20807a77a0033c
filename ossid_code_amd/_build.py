"""Builds libossid_hip.so (the C-ABI boundary, include/ossid_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container as well as on the GPU box.
-ffp-contract=off: SPEC.md fixes the operation order of every float op; fused multiply-adds appear
only where the source writes fmaf / an MFMA.

Each csrc/*.hip is compiled to its own object (in parallel, only when it or a header is newer than the object),
then the objects are linked into the shared library: a one-file edit rebuilds in the time of that file.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("OSSID_HIP_LIB") or os.path.join(_HERE, "libossid_hip.so")
OBJ_DIR = os.path.join(_HERE, "build")
SOURCES = ["zephyr.hip", "pn2.hip", "dtoid.hip", "conv.hip", "segtail.hip", "pipeline.hip", "train.hip", "wino.hip", "stem.hip", "wgrad_fc.hip", "dense.hip", "wgrad_t9.hip", "dense_bwd.hip", "seq.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libossid_hip.so cannot be built")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return hs + [os.path.join(_HERE, "..", "include", "ossid_hip.h")]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + _headers()
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


LAST_BUILD = {"compiled": 0, "linked": False}     # what the last build_lib() call did (objects compiled; library linked)


def build_lib(force=False, verbose=False):
    """Compile every HIP source into ossid_code_amd/libossid_hip.so; returns its path. LAST_BUILD says what was done."""
    LAST_BUILD.update(compiled=0, linked=False)
    if not force and not is_stale():
        return LIB_PATH
    extra = os.environ.get("OSSID_HIPCC_EXTRA", "").split()       # A/B builds of ablation switches (-DOSSID_...)
    tag = ("_" + "".join(c if c.isalnum() else "_" for c in " ".join(extra))) if extra else ""
    obj_dir = OBJ_DIR + tag
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    hdr_t = max(os.path.getmtime(h) for h in _headers() if os.path.exists(h))
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc] + FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    workers = max(1, min(len(jobs), int(os.environ.get("OSSID_BUILD_JOBS", "0")) or (os.cpu_count() or 2)))
    if jobs:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(run, jobs))
    run([hipcc] + FLAGS + ["-shared", "-o", LIB_PATH + ".tmp"] + objs)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    LAST_BUILD.update(compiled=len(jobs), linked=True)
    return LIB_PATH


if __name__ == "__main__":
    import sys
    print(build_lib(force="--force" in sys.argv, verbose=True))
