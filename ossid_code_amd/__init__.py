"""ossid_code_amd -- MI355X (gfx950) native hot path of r-pad/OSSID_code.

Scope: Zephyr per-hypothesis pose scoring (project -> gather -> featurize -> PointNet++
score) and the DTOID detector forward/backward used for online finetuning, behind the reference's own call
signatures. Compute runs in hand-written HIP kernels in libossid_hip.so (C ABI: include/ossid_hip.h); this
package is the thin Python host layer that mirrors the reference interfaces. There is no CPU fallback.
"""
from ._build import build_lib  # noqa: F401

__version__ = "0.1.0"
