"""Resource usage of the shipped gfx950 kernels, read from the code objects inside libmmtpsm.so (no GPU): the
plane-fed implicit GEMM ships exactly its three product forms -- no main-loop ablation (DBG) arm, no copy-wave-split (AF) arm --
and none of them spills vector registers or uses scratch memory."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd", "tools"))


@pytest.fixture(scope="module")
def table(tmp_path_factory):
    import codeobj
    if not os.path.exists(codeobj.LIB) or not os.path.exists(os.path.join(codeobj.LLVM, "clang-offload-bundler")):
        pytest.skip("library or LLVM tools not present")
    t = codeobj.kernel_table(workdir=str(tmp_path_factory.mktemp("co")))
    return t, codeobj.demangle(sorted(t))


def test_every_translation_unit_is_seen(table):
    t, d = table
    names = " ".join(d.values())
    for k in ("conv_pg_kernel", "conv3x3_strip_kernel", "conv1x1_rows_kernel", "wgrad_pl_kernel", "stem_fused_kernel", "roi_align_kernel",
              "nms_sweep_kernel", "ema_kernel"):
        assert k in names, k


def test_plane_fed_kernel_ships_three_forms_without_spills(table):
    t, d = table
    pg = {d[n]: t[n] for n in t if "conv_pg_kernel<" in d[n]}
    forms = sorted(re.search(r"conv_pg_kernel<([^>]*)>", n).group(1).replace(" ", "") for n in pg)
    assert forms == ["1,4,3,0,false", "2,2,4,0,false", "4,1,4,0,false"], forms
    for n, r in pg.items():
        # no vector register lives in scratch memory.  (The K-group forms reserve a 20-byte private segment no instruction touches --
        # scalar spills go to lanes of a vector register --; the 256-row form, which carries the time, reserves none.)
        assert r["vgpr_spill"] == 0 and r["scratch"] <= 32, (n, r)
        if "<4, 1, 4" in n:
            assert r["scratch"] == 0, (n, r)
        assert r["vgpr"] <= 168, (n, r)   # 768 threads = three waves per SIMD


def test_hot_kernels_use_no_scratch(table):
    """the kernels that carry the step: nothing of theirs lives in scratch memory (the slow exact path of the fp16 split is a real
    call with a 288-byte frame in the kernels that have one: bounded here)"""
    t, d = table
    for n, r in t.items():
        name = d[n]
        if any(k in name for k in ("conv3x3_strip_kernel", "wgrad_pl_kernel", "stem_fused_kernel", "conv3x3_c64_kernel")):
            assert r["vgpr_spill"] == 0, (name, r)
