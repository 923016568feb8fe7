"""CPU: the C-ABI library loads and exports every symbol include/opp_hip.h declares (no
compute calls without a GPU); handle/queries that do not touch the device work."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "opp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(opp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from onepose_plus_plus_amd.build import build
    from onepose_plus_plus_amd import _lib
    build(verbose=False)
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from onepose_plus_plus_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), "libopp_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"


def test_no_torch_types_in_abi():
    src = open(os.path.join(ROOT, "include", "opp_hip.h")).read()
    assert "torch" not in src.lower().replace("pytorch", "") and "at::" not in src and "#include <hip" not in src


def test_handle_and_weight_table(lib):
    from onepose_plus_plus_amd import OnePosePlus_model, default_config
    from onepose_plus_plus_amd.params import param_spec
    cfg = default_config()
    m = OnePosePlus_model(cfg)
    ctx = ctypes.c_void_p()
    ccfg = m._c_config()
    assert lib.opp_create(ctypes.byref(ccfg), ctypes.byref(ctx)) == 0, lib.opp_last_error()
    names = [lib.opp_weight_name(ctx, i).decode() for i in range(lib.opp_num_weights(ctx))]
    spec = [(k, s) for k, s, kind in param_spec(cfg) if kind != "bn_count"]
    assert names == [k for k, _ in spec]                      # reference state-dict order
    for i, (k, s) in enumerate(spec):
        n = 1
        for d in s:
            n *= d
        assert lib.opp_weight_numel(ctx, i) == n, k
    assert lib.opp_packed_weights_bytes(ctx) > 40e6          # 10.2 M params + channel padding
    ws = lib.opp_forward_coarse_workspace_bytes(ctx, 512, 512, 5000)
    assert 100e6 < ws < 2e9
    assert lib.opp_fine_workspace_bytes(ctx, 0) > 0
    lib.opp_destroy(ctx)


def test_create_rejects_unsupported_config(lib):
    from onepose_plus_plus_amd import OnePosePlus_model, default_config
    m = OnePosePlus_model(default_config())
    ccfg = m._c_config()
    ccfg.coarse_d_model = 192
    ctx = ctypes.c_void_p()
    assert lib.opp_create(ctypes.byref(ccfg), ctypes.byref(ctx)) != 0
    assert b"d_model" in lib.opp_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from onepose_plus_plus_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.OppError):
        _lib.load()


def test_conv_packed_k(lib):
    """Host-only size query: ks*ks*pad32(cin), or the K-tail packing (n full groups x 9 taps + 2 tail chunks)."""
    assert lib.opp_conv_packed_k(128, 3) == 9 * 128
    assert lib.opp_conv_packed_k(196, 3) == (6 * 9 + 2) * 32
    assert lib.opp_conv_packed_k(196, 1) == 224
    assert lib.opp_conv_packed_k(97, 3) == (3 * 9 + 2) * 32
    assert lib.opp_conv_packed_k(101, 3) == 9 * 128        # five tail channels do not fit 4 per tap
    assert lib.opp_conv_packed_k(4, 3) == 9 * 32           # no full group to attach the tail to


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """`include/opp_hip.h` compiles as C99 and as C++ (no torch / HIP types in the signatures), and a C program links
    against libopp_hip.so and calls host-only entry points (what a cgo / JNI / N-API binding would do)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    gxx = shutil.which("g++")
    if not gcc or not gxx:
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    libdir = os.path.join(root, "onepose_plus_plus_amd")
    src = tmp_path / "use.c"
    src.write_text(
        '#include <stdio.h>\n#include "opp_hip.h"\n'
        "int main(void) {\n"
        "  opp_config cfg; (void)cfg;\n"
        '  printf("%d %d %zu\\n", opp_conv_packed_k(196, 3), opp_conv_packed_k(128, 3), opp_focal_loss_workspace_bytes((size_t)1 << 20));\n'
        "  /* a NULL handle is an argument error, reported through the C error channel */\n"
        "  int rc = opp_num_weights(NULL);\n"
        '  printf("%d\\n", rc);\n'
        "  return 0;\n}\n")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(src)], check=True)
    cpp = tmp_path / "use.cpp"
    cpp.write_text('#include "opp_hip.h"\nint main() { return opp_conv_packed_k(4, 3) == 288 ? 0 : 1; }\n')
    subprocess.run([gxx, "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(cpp)], check=True)
    exe = tmp_path / "use"
    subprocess.run([gcc, "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lopp_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "1792" and out[1] == "1152" and int(out[2]) > 0


def test_stale_library_is_refused(monkeypatch):
    """The library carries the sha256 of the sources it was built from (csrc/version.hip); `_lib.load()` recomputes it from the
    sources next to the .so and refuses a binary built from anything else -- the GPU box runs the prebuilt .so of the snapshot."""
    from onepose_plus_plus_amd import _lib, build
    lib = _lib.load()
    assert lib.opp_source_hash().decode() == build.source_hash() and len(build.source_hash()) == 64
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 64)          # "the sources changed after the build"
    with pytest.raises(_lib.OppError, match="built from other sources"):
        _lib.load()
    monkeypatch.setenv("OPP_ALLOW_STALE_LIB", "1")                         # explicit override for tooling
    assert _lib.load() is not None
