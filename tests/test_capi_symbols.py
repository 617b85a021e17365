"""CPU: the C-ABI library loads and exports every symbol include/dc_density.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dc_density.h")).read()
    return sorted(set(re.findall(r"DC_API\s+[\w\s\*]+?\b(dc_hip_\w+)\s*\(", text)))


def test_header_declares_expected_set():
    from clustering_amd import capi
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    import ctypes
    from clustering_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert capi.lib.dc_hip_abi_version() == capi.ABI_VERSION == 5
    assert capi.lib.dc_hip_last_error() is not None


@pytest.mark.parametrize("order", ["avx", "fma"])
def test_the_builds_say_which_summation_order_they_reproduce(order):
    """libdcdensity.so = the reference's default build ("sse2"), lib_avx/libdcdensity.so = a reference built with
    -DCPU_ACCELERATION=AVX ("avx"), lib_fma/ = with -DNATIVE_COMPILATION on an AVX2 + FMA host ("fma"); same ABI, same
    symbols, same source digest"""
    import ctypes
    from clustering_amd import capi
    if "DC_LIB_PATH" not in os.environ:
        assert capi.lib.dc_hip_canon_order().decode() == capi.CANON_ORDER
    other = os.path.join(ROOT, "clustering_amd", "lib_" + order, "libdcdensity.so")
    assert os.path.exists(other), "build() makes all three libraries"
    lib = ctypes.CDLL(other)
    lib.dc_hip_canon_order.restype = ctypes.c_char_p
    lib.dc_hip_build_digest.restype = ctypes.c_char_p
    assert lib.dc_hip_canon_order().decode() == order
    assert lib.dc_hip_abi_version() == capi.ABI_VERSION
    assert lib.dc_hip_build_digest().decode() == _digest_module().source_digest()
    for name in declared_symbols():
        assert hasattr(lib, name), name


def _digest_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("dc_digest", os.path.join(ROOT, "clustering_amd", "csrc", "digest.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_library_carries_the_digest_of_the_sources_it_was_built_from():
    """dc_hip_build_digest(): what bench.py ties its line and the counter profiles to.  A library older than the sources
    in the tree fails here (rebuild: __graft_entry__.build())."""
    from clustering_amd import capi
    assert capi.lib.dc_hip_build_digest().decode() == _digest_module().source_digest()


def test_digest_ignores_comments_and_sees_code(tmp_path):
    """a comment edit of dc_prep.hpp leaves the digest alone (no profile regeneration), a code edit changes it"""
    import shutil
    dg = _digest_module()
    src = os.path.join(ROOT, "clustering_amd", "csrc")
    a, inc = tmp_path / "csrc", tmp_path / "include"
    a.mkdir()
    shutil.copytree(os.path.join(ROOT, "include"), inc)
    for f in os.listdir(src):
        if f.endswith(dg.EXTS) or f == "Makefile":
            shutil.copy(os.path.join(src, f), a / f)
    base = dg.source_digest((str(a), str(inc)))
    assert base == dg.source_digest()
    prep = a / "dc_prep.hpp"
    text = prep.read_text()
    prep.write_text("// a remark\n" + text.replace("\n", "   // trailing remark\n", 1) + "/* block\n comment */\n")
    assert dg.source_digest((str(a), str(inc))) == base
    prep.write_text(text + "\nstatic const int dc_digest_probe = 1;\n")
    assert dg.source_digest((str(a), str(inc))) != base
    prep.write_text(text.replace('"', "'", 0) + 'static const char* dc_probe = "// not a comment";\n')
    assert dg.source_digest((str(a), str(inc))) != base


def test_make_rebuilds_the_sweeps_when_the_preparation_header_changes():
    """header dependencies come from the compiler (-MMD): a touched dc_prep.hpp must rebuild dc_mfma.o"""
    import subprocess
    csrc = os.path.join(ROOT, "clustering_amd", "csrc")
    if not os.path.exists(os.path.join(ROOT, "clustering_amd", "lib", "obj", "dc_mfma.d")):
        import pytest
        pytest.skip("no dependency files: the library was not built by this Makefile in this tree")
    out = subprocess.run(["make", "-n", "-W", "dc_prep.hpp", "../lib/libdcdensity.so"], cwd=csrc, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "-o ../lib/obj/dc_mfma.o" in out.stdout
    assert "-o ../lib/obj/dc_sort.o" not in out.stdout   # (dc_sort.hip does not include it)


def test_no_device_is_a_status_not_a_crash():
    from clustering_amd import capi
    import torch
    if torch.cuda.is_available():
        return
    assert capi.device_count() == 0
    import numpy as np, ctypes as C
    c = np.zeros((4, 3), np.float32)
    pops = np.zeros(4, np.uint32)
    r = np.array([1.0], np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = capi.lib.dc_hip_populations(vp(c), 4, 3, vp(r), 1, 0, 4, 0, vp(pops))
    assert rc in (-2, -3)          # DC_ERR_NO_DEVICE / DC_ERR_HIP: the product has no CPU path


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "clustering_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "capi.py" and False, (dirpath, f)


def test_sessions_without_a_device_fail_with_a_status():
    """dc_hip_session_open on a box without a GPU: a status and a message, never a crash or a CPU path."""
    import ctypes as C
    import numpy as np
    from clustering_amd import capi
    if capi.device_count() > 0:
        import pytest
        pytest.skip("needs a box without a GPU")
    c = np.zeros((8, 3), dtype=np.float32)
    h = C.c_void_p(0)
    rc = capi.lib.dc_hip_session_open(c.ctypes.data_as(C.c_void_p), 8, 3, None, 0, C.byref(h))
    assert rc == -2 and not h.value                       # DC_ERR_NO_DEVICE
    assert b"no HIP device" in capi.lib.dc_hip_last_error()
    assert capi.lib.dc_hip_session_open(c.ctypes.data_as(C.c_void_p), 8, 0, None, 0, C.byref(h)) == -1   # n_cols = 0
    assert capi.lib.dc_hip_session_devices(None) == 0 and capi.lib.dc_hip_session_uses_rccl(None) == 0
    capi.lib.dc_hip_session_close(None)                   # a null session is a no-op
    pops = np.zeros(8, dtype=np.uint32)
    r = np.array([0.5], dtype=np.float32)
    assert capi.lib.dc_hip_session_populations(None, r.ctypes.data_as(C.c_void_p), 1, pops.ctypes.data_as(C.c_void_p)) == -1
    assert capi.lib.dc_hip_density_all(c.ctypes.data_as(C.c_void_p), 8, 3, r.ctypes.data_as(C.c_void_p), 1, 0, 0,
                                       pops.ctypes.data_as(C.c_void_p), None, None, None, None, None) == -2


def test_population_scale_rule_on_the_host():
    """tests/cpp/test_scale_host.hip (built with the library, no device code runs): the population sweeps' scale is
    the largest power of two with a guard band <= 1 for extents and radii over 60 orders of magnitude, everything
    fits fp16 at that scale, and the two fp16 pieces reproduce a value to 2^-22 (2^-20 absolute)."""
    import subprocess
    exe = os.path.join(ROOT, "clustering_amd", "bin", "test_scale_host")
    if not os.path.exists(exe):
        pytest.skip("clustering_amd/bin/test_scale_host not built")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
