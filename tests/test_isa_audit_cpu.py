"""Static guard on the weight-gradient kernel's code generation (CPU: hipcc cross-compiles gfx950, ~30 s).  The first version of
conv_wgrad_kernel chose its operand per LANE, which made the buffer descriptor a vector value and wrapped every global load in a
v_readfirstlane "waterfall" loop; nothing functional catches that -- the results are right, the kernel is 40 % slower."""
import os
import shutil
import tempfile

import pytest

from tools import isa_audit


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_conv_wgrad_kernel_has_no_waterfall_loops_and_no_scratch():
    with tempfile.TemporaryDirectory() as tmp:
        src, rows, err = isa_audit.audit_source("conv_bwd.hip", False, tmp)
    assert rows is not None, err
    by_name = {k: (vg, ag, sc, water, mfma, pk) for k, vg, ag, sc, water, mfma, pk in rows}
    wg = [k for k in by_name if "conv_wgrad_kernelILi0" in k]
    assert len(wg) == 1
    vg, ag, sc, water, mfma, pk = by_name[wg[0]]
    assert water == 0 and sc == 0
    assert vg + ag <= 256            # two waves per SIMD (512-thread workgroup)
    assert mfma == 156               # prologue 12 + six intervals of 24: the six-interval body was not re-rolled or duplicated
    assert pk == 0                   # built without the SLP vectorizer (build.SOURCE_FLAGS)
    for k, (vg, ag, sc, water, mfma, pk) in by_name.items():
        assert water == 0 and sc == 0, k
