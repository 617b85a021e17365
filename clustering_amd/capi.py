"""ctypes binding of the C ABI in include/dc_density.h (clustering_amd/lib/libdcdensity.so).

There is deliberately no fallback: if the HIP library is missing or cannot be loaded,
importing this module raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C clustering_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DC_LIB_PATH: another build of the SAME library (measurement variants under clustering_amd/lib/variants/) instead of
# copying it over the product.  DC_CANON_ORDER=avx: the build with the summation order of a reference compiled
# with -DCPU_ACCELERATION=AVX (`make -C clustering_amd/csrc CANON=avx` -> clustering_amd/lib_avx/; dc_hip_canon_order());
# DC_CANON_ORDER=fma: that of a -DNATIVE_COMPILATION build on an AVX2 + FMA host (CANON=fma -> clustering_amd/lib_fma/)
CANON_ORDER = os.environ.get("DC_CANON_ORDER", "sse2")
_CANON_DIRS = {"sse2": "lib", "avx": "lib_avx", "fma": "lib_fma"}
if CANON_ORDER not in _CANON_DIRS:
    raise ImportError(f"DC_CANON_ORDER={CANON_ORDER!r}: expected 'sse2' (the reference's default build), 'avx' or 'fma'")
LIB_PATH = os.environ.get("DC_LIB_PATH") or os.path.join(_HERE, _CANON_DIRS[CANON_ORDER], "libdcdensity.so")

DC_OK = 0
ABI_VERSION = 5               # include/dc_density.h: DC_HIP_ABI_VERSION this binding was written against
FLAG_STATS_VALID = 0x100      # DC_FLAG_STATS_VALID
VARIANT_AUTO, VARIANT_DIRECT, VARIANT_MFMA, VARIANT_MFMA_PRUNED = 0, 1, 2, 3
VARIANTS = {"auto": VARIANT_AUTO, "direct": VARIANT_DIRECT, "mfma": VARIANT_MFMA,
            "pruned": VARIANT_MFMA_PRUNED, "mfma32": 4}

# every symbol include/dc_density.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = (
    "dc_hip_last_error", "dc_hip_abi_version", "dc_hip_build_digest", "dc_hip_canon_order", "dc_hip_device_count", "dc_hip_workspace_bytes",
    "dc_hip_populations_dev", "dc_hip_free_energies_dev", "dc_hip_nearest_neighbors_dev",
    "dc_hip_sigma2_dev", "dc_hip_workspace_counters_dev", "dc_hip_workspace_mfma_counters_dev", "dc_hip_workspace_layout_status_dev", "dc_hip_sweep_timing", "dc_hip_last_sweep_ms", "dc_hip_workspace_components_dev", "dc_hip_populations", "dc_hip_nearest_neighbors", "dc_hip_density_all",
    "dc_hip_radius_pairs_dev", "dc_hip_radius_pairs", "dc_hip_radius_min_edge_dev", "dc_hip_radius_forest",
    "dc_hip_populations_segment_dev", "dc_hip_nearest_neighbors_segment_dev",
    "dc_hip_neighbors_pack_dev", "dc_hip_neighbors_unpack_dev", "dc_hip_neighbors_block_rows",
    "dc_hip_neighbors_block_pack_dev", "dc_hip_neighbors_block_unpack_dev", "dc_hip_radius_min_edge_segment_dev",
    "dc_hip_session_open", "dc_hip_session_close", "dc_hip_session_devices", "dc_hip_session_uses_rccl",
    "dc_hip_session_merge_mode", "dc_hip_session_merge_note",
    "dc_hip_session_counters", "dc_hip_session_populations", "dc_hip_session_free_energies",
    "dc_hip_session_set_free_energies", "dc_hip_session_nearest_neighbors", "dc_hip_session_radius_pairs",
    "dc_hip_session_radius_forest",
)


class DensityLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise DensityLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()). "
            "clustering_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    lib.dc_hip_last_error.restype = C.c_char_p
    lib.dc_hip_last_error.argtypes = []
    lib.dc_hip_abi_version.restype = i32
    lib.dc_hip_abi_version.argtypes = []
    if lib.dc_hip_abi_version() != ABI_VERSION:
        raise DensityLibraryError(
            f"{LIB_PATH} has ABI version {lib.dc_hip_abi_version()}, this binding needs {ABI_VERSION}: rebuild "
            "the library (python -c 'import __graft_entry__ as g; g.build()')")
    lib.dc_hip_build_digest.restype = C.c_char_p
    lib.dc_hip_build_digest.argtypes = []
    lib.dc_hip_canon_order.restype = C.c_char_p
    lib.dc_hip_canon_order.argtypes = []
    if "DC_LIB_PATH" not in os.environ and lib.dc_hip_canon_order().decode() != CANON_ORDER:
        raise DensityLibraryError(f"{LIB_PATH} reproduces the {lib.dc_hip_canon_order().decode()!r} summation order, "
                                  f"DC_CANON_ORDER asks for {CANON_ORDER!r}: rebuild it")
    lib.dc_hip_device_count.restype = i32
    lib.dc_hip_workspace_bytes.restype = sz
    lib.dc_hip_workspace_bytes.argtypes = [sz, sz, sz]
    lib.dc_hip_populations_dev.restype = i32
    lib.dc_hip_populations_dev.argtypes = [vp, sz, sz, C.POINTER(C.c_float), sz, sz, sz, vp, vp, sz,
                                           i32, vp]
    lib.dc_hip_free_energies_dev.restype = i32
    lib.dc_hip_free_energies_dev.argtypes = [vp, sz, vp, C.POINTER(C.c_uint32), vp]
    lib.dc_hip_nearest_neighbors_dev.restype = i32
    lib.dc_hip_nearest_neighbors_dev.argtypes = [vp, sz, sz, vp, sz, sz, vp, vp, vp, vp, vp, sz,
                                                 i32, vp]
    lib.dc_hip_sigma2_dev.restype = i32
    lib.dc_hip_sigma2_dev.argtypes = [vp, sz, C.POINTER(C.c_double), vp]
    lib.dc_hip_workspace_counters_dev.restype = i32
    lib.dc_hip_workspace_counters_dev.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp]
    lib.dc_hip_workspace_components_dev.restype = i32
    lib.dc_hip_workspace_components_dev.argtypes = [vp, sz, sz, C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                                    C.POINTER(C.c_float), C.POINTER(C.c_float), vp]
    lib.dc_hip_sweep_timing.restype = i32
    lib.dc_hip_sweep_timing.argtypes = [i32]
    lib.dc_hip_last_sweep_ms.restype = i32
    lib.dc_hip_last_sweep_ms.argtypes = [i32, C.POINTER(C.c_float)]
    lib.dc_hip_populations.restype = i32
    lib.dc_hip_populations.argtypes = [vp, sz, sz, vp, sz, sz, sz, i32, vp]
    lib.dc_hip_nearest_neighbors.restype = i32
    lib.dc_hip_nearest_neighbors.argtypes = [vp, sz, sz, vp, sz, sz, i32, vp, vp, vp, vp]
    lib.dc_hip_populations_segment_dev.restype = i32
    lib.dc_hip_populations_segment_dev.argtypes = [vp, sz, sz, C.POINTER(C.c_float), sz, sz, sz, vp, vp, sz,
                                                   i32, vp]
    lib.dc_hip_nearest_neighbors_segment_dev.restype = i32
    lib.dc_hip_nearest_neighbors_segment_dev.argtypes = [vp, sz, sz, vp, sz, sz, vp, vp, vp, vp, vp, sz,
                                                         i32, vp]
    lib.dc_hip_radius_pairs_dev.restype = i32
    lib.dc_hip_radius_pairs_dev.argtypes = [vp, sz, sz, C.c_float, vp, vp, sz, vp, vp, sz, vp]
    lib.dc_hip_radius_pairs.restype = i32
    lib.dc_hip_radius_pairs.argtypes = [vp, sz, sz, C.c_float, i32, vp, sz, C.POINTER(C.c_uint64)]
    lib.dc_hip_neighbors_pack_dev.restype = i32
    lib.dc_hip_neighbors_pack_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    lib.dc_hip_neighbors_unpack_dev.restype = i32
    lib.dc_hip_neighbors_unpack_dev.argtypes = [vp, sz, vp, vp, vp, vp, vp]
    lib.dc_hip_neighbors_block_rows.restype = sz
    lib.dc_hip_neighbors_block_rows.argtypes = [sz, sz, sz]
    lib.dc_hip_neighbors_block_pack_dev.restype = i32
    lib.dc_hip_neighbors_block_pack_dev.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz, vp, sz, i32, vp, vp]
    lib.dc_hip_neighbors_block_unpack_dev.restype = i32
    lib.dc_hip_neighbors_block_unpack_dev.argtypes = [vp, sz, sz, sz, vp, sz, i32, vp, vp, vp, vp, vp]
    lib.dc_hip_radius_min_edge_dev.restype = i32
    lib.dc_hip_radius_min_edge_dev.argtypes = [vp, sz, sz, C.c_float, vp, vp, vp, vp, vp, sz, vp]
    lib.dc_hip_radius_min_edge_segment_dev.restype = i32
    lib.dc_hip_radius_min_edge_segment_dev.argtypes = [vp, sz, sz, C.c_float, vp, vp, sz, sz, vp, vp, vp, sz, vp]
    lib.dc_hip_radius_forest.restype = i32
    lib.dc_hip_radius_forest.argtypes = [vp, sz, sz, C.c_float, vp, i32, vp, C.POINTER(sz),
                                         C.POINTER(C.c_uint32)]
    lib.dc_hip_density_all.restype = i32
    lib.dc_hip_density_all.argtypes = [vp, sz, sz, vp, sz, sz, i32, vp, vp, vp, vp, vp, vp]
    lib.dc_hip_session_open.restype = i32
    lib.dc_hip_session_open.argtypes = [vp, sz, sz, C.POINTER(C.c_int), i32, C.POINTER(vp)]
    lib.dc_hip_session_close.restype = None
    lib.dc_hip_session_close.argtypes = [vp]
    lib.dc_hip_session_devices.restype = i32
    lib.dc_hip_session_devices.argtypes = [vp]
    lib.dc_hip_session_uses_rccl.restype = i32
    lib.dc_hip_session_uses_rccl.argtypes = [vp]
    lib.dc_hip_session_merge_mode.restype = i32
    lib.dc_hip_session_merge_mode.argtypes = [vp]
    lib.dc_hip_workspace_mfma_counters_dev.restype = i32
    lib.dc_hip_workspace_mfma_counters_dev.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp]
    lib.dc_hip_workspace_layout_status_dev.restype = i32
    lib.dc_hip_workspace_layout_status_dev.argtypes = [vp, C.POINTER(C.c_int), vp]
    lib.dc_hip_session_merge_note.restype = C.c_char_p
    lib.dc_hip_session_merge_note.argtypes = [vp]
    lib.dc_hip_session_counters.restype = i32
    lib.dc_hip_session_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.dc_hip_session_populations.restype = i32
    lib.dc_hip_session_populations.argtypes = [vp, vp, sz, vp]
    lib.dc_hip_session_free_energies.restype = i32
    lib.dc_hip_session_free_energies.argtypes = [vp, sz, vp, C.POINTER(C.c_uint32)]
    lib.dc_hip_session_set_free_energies.restype = i32
    lib.dc_hip_session_set_free_energies.argtypes = [vp, vp]
    lib.dc_hip_session_nearest_neighbors.restype = i32
    lib.dc_hip_session_nearest_neighbors.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_double)]
    lib.dc_hip_session_radius_pairs.restype = i32
    lib.dc_hip_session_radius_pairs.argtypes = [vp, C.c_float, vp, sz, C.POINTER(C.c_uint64)]
    lib.dc_hip_session_radius_forest.restype = i32
    lib.dc_hip_session_radius_forest.argtypes = [vp, C.c_float, vp, vp, C.POINTER(sz), C.POINTER(C.c_uint32)]
    return lib


lib = _load()


def check(rc, what=""):
    if rc != DC_OK:
        msg = lib.dc_hip_last_error().decode("utf-8", "replace")
        raise DensityLibraryError(f"{what or 'dc_hip call'} failed (status {rc}): {msg}")


def device_count():
    n = lib.dc_hip_device_count()
    if n < 0:
        check(n, "dc_hip_device_count")
    return n
