"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol
include/ossid_hip.h declares (no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ossid_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ossid_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ("ossid_zephyr_featurize", "ossid_zephyr_project_uv", "ossid_pn2_score", "ossid_zephyr_prep_frame_u8"):
        assert must in names


def test_library_exports_every_declared_symbol(hiplib):
    handle = ctypes.CDLL(os.path.join(ROOT, "ossid_code_amd", "libossid_hip.so"))
    missing = [n for n in _declared() if not hasattr(handle, n)]
    assert not missing, missing


def test_python_prototypes_cover_the_header(hiplib):
    assert set(_declared()) == set(hiplib.exported_symbols())


def test_abi_version_and_struct_layouts_agree_between_header_and_binding(hiplib, tmp_path):
    """OSSID_ABI_VERSION of the header = the binding's = what the built library reports, and every descriptor struct has
    the same size and the same offset of its LAST field when the header is compiled by gcc as in the ctypes mirror."""
    import subprocess
    text = open(os.path.join(ROOT, "include", "ossid_hip.h")).read()
    ver = int(re.search(r"#define\s+OSSID_ABI_VERSION\s+(\d+)", text).group(1))
    assert ver == hiplib.ABI_VERSION == hiplib.lib().ossid_abi_version(None, 0)
    pairs = [("ossid_pn2_weights", hiplib.PN2Weights), ("ossid_conv_desc", hiplib.ConvDesc), ("ossid_wgrad_desc", hiplib.WgradDesc),
             ("ossid_chan_op_desc", hiplib.ChanOpDesc), ("ossid_pack_row", hiplib.PackRow)]
    src = tmp_path / "layout.c"
    body = "".join('printf("%%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));\n' % (c, c, st._fields_[-1][0]) for c, st in pairs)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ossid_hip.h"\nint main(void) {\n%sreturn 0;\n}\n' % body)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)]).decode().split("\n")
    for (c, st), line in zip(pairs, lines):
        size, off = (int(v) for v in line.split())
        assert size == ctypes.sizeof(st), (c, size, ctypes.sizeof(st))
        assert off == getattr(st, st._fields_[-1][0]).offset, (c, off)


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    import torch
    from ossid_code_amd.zephyr import PointNet2SSG
    m = PointNet2SSG(8).eval()
    with pytest.raises(RuntimeError):
        m({"point_x": torch.zeros(1, 600, 8)})


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "ossid_code_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "zephyr_oracle" not in \
                    src.replace("oracle/zephyr_oracle.c", ""), f


def test_compat_install_resolves_reference_import_paths():
    """The names scripts/online_learning.py imports for the hot path resolve to this package after install()."""
    import subprocess
    import sys
    code = ("import ossid_code_amd.compat as c; c.install();"
            "from zephyr.datasets.score_dataset import ScoreDataset;"
            "from zephyr.models.pointnet2 import PointNet2SSG;"
            "from zephyr.options import getOptions;"
            "from zephyr.utils import K2meta, projectPointsUv;"
            "from ossid.utils.zephyr_utils import networkInference;"
            "from ossid.models.dtoid import DtoidNet;"
            "a = getOptions().parse_args([]); a.dataset='HSVD_diff_uv_norm'; a.no_valid_proj=True; a.no_valid_depth=True;"
            "d = ScoreDataset([], '', 'lmo', a, mode='test'); assert d.dim_point == 8;"
            "m = PointNet2SSG(d.dim_point, a, num_class=1); print('ok', DtoidNet.__module__, networkInference.__module__)")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "ok ossid_code_amd.dtoid.model ossid_code_amd.scoring" in out.stdout


def test_seq_replay_passes_every_argument_class_through_the_c_loop(hiplib):
    """ossid_seq_replay (csrc/seq.hip) re-issues recorded calls through ONE call shape: integer-class arguments beyond the
    sixth and the stream on the stack, floats and doubles in xmm registers, negative 32-bit values, NULL. The probe entry
    point writes back what it received; the same ops through the Python loop must agree."""
    import ctypes as C
    lib = hiplib.lib()

    class FakeStream:
        def __init__(self, h):
            self.cuda_stream = h

        def wait_stream(self, other):
            pass
    out = [(C.c_double * 15)(), (C.c_double * 15)()]
    anchor = C.create_string_buffer(64)
    args = lambda o: (-7, 0.1, anchor, 1e300, -(1 << 40), 2 ** 31 - 1, -2.5, (1 << 63) + 5, -1, -123456, 77, -3.25e-7, 0, 1 << 50, o)
    for use_c, o in ((True, out[0]), (False, out[1])):
        seq = hiplib.Seq()
        seq.ops.append((lib.ossid_seq_probe, args(o), 1, "ossid_seq_probe"))
        seq.ops.append(("wait", 0, 0))                     # same stream on both sides: nothing to do, no HIP call
        seq.ops.append((lib.ossid_fill_zero, (None, 0), 0, "ossid_fill_zero"))     # zero bytes: returns before any HIP call
        old = hiplib.SEQ_C
        hiplib.SEQ_C = use_c
        try:
            seq.run((FakeStream(0x1000), FakeStream(0xABCDEF0123)))
            assert (seq._compiled is not None) == use_c
        finally:
            hiplib.SEQ_C = old
    want = [-7, float(C.c_float(0.1).value), C.addressof(anchor), 1e300, -(1 << 40), 2 ** 31 - 1, -2.5, float((1 << 63) + 5), -1,
            -123456, 77, -3.25e-7, 0, float(1 << 50), float(0xABCDEF0123)]
    assert list(out[0]) == want
    assert list(out[1]) == want
    # a failing op reports its status and index
    seq = hiplib.Seq()
    seq.ops.append((lib.ossid_fill_zero, (None, 0), 0, "ossid_fill_zero"))
    seq.ops.append((lib.ossid_fill_zero, (None, 8), 0, "ossid_fill_zero"))          # NULL with bytes: OSSID_EINVAL
    with pytest.raises(RuntimeError, match="ossid_fill_zero failed with status -22 .*op 1"):
        seq.run((FakeStream(0),))
