"""ctypes front-end of the CPU oracle (oracle/dc_oracle.c).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this module; nothing under clustering_amd/ does.
See the header of dc_oracle.c for what the oracle restates and how it is pinned
("parity unpinned" bitwise: the reference ships no vectors and is unbuildable
here without stand-ins).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile the oracle, its timing build and the fast-math probe (gcc only)."""
    want = [os.path.join(_BUILD, n) for n in
            ("libdc_oracle.so", "libdc_oracle_fast.so", "libfastmath_probe.so", "libscreening_oracle.so",
             "libdc_oracle_avx.so", "libfastmath_probe_avx.so", "libdc_oracle_fma.so", "libfastmath_probe_fma.so")]
    srcs = [os.path.join(_HERE, n) for n in ("dc_oracle.c", "fastmath_probe.cpp", "screening_oracle.cpp",
                                             "Makefile")]
    stale = force or any(not os.path.exists(w) for w in want)
    if not stale:
        newest = max(os.path.getmtime(s) for s in srcs)
        stale = any(os.path.getmtime(w) < newest for w in want)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return want


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    """One loaded build of dc_oracle.c (canonical by default, ``fast=True`` = timing build, ``order="avx"`` / ``"fma"`` = the
    summation order of a reference built with -DCPU_ACCELERATION=AVX / with -DNATIVE_COMPILATION on an AVX2 + FMA host, for
    libraries built with `make CANON=avx` / `make CANON=fma`)."""

    def __init__(self, fast=False, order="sse2"):
        assert order in ("sse2", "avx", "fma") and not (fast and order != "sse2")
        name = "libdc_oracle_fast.so" if fast else {"sse2": "libdc_oracle.so", "avx": "libdc_oracle_avx.so", "fma": "libdc_oracle_fma.so"}[order]
        path = os.path.join(_BUILD, name)
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.dco_dist2.restype = C.c_float
        L.dco_dist2.argtypes = [_f32p, _f32p, C.c_size_t]
        L.dco_is_fast_build.restype = C.c_int
        L.dco_num_threads.restype = C.c_int
        L.dco_populations_brute.restype = None
        L.dco_populations_brute.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p, C.c_size_t,
                                            C.c_size_t, C.c_size_t, _u64p]
        L.dco_populations_boxgrid.restype = None
        L.dco_populations_boxgrid.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p, C.c_size_t,
                                              _u64p]
        L.dco_free_energies.restype = None
        L.dco_free_energies.argtypes = [_u64p, C.c_size_t, _f32p]
        L.dco_nearest_neighbors.restype = None
        L.dco_nearest_neighbors.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p, C.c_size_t,
                                            C.c_size_t, _u64p, _f32p, _u64p, _f32p]
        L.dco_sigma2.restype = C.c_double
        L.dco_sigma2.argtypes = [_f32p, C.c_size_t]
        L.dco_lumping_radius.restype = C.c_float
        L.dco_lumping_radius.argtypes = [C.c_double]
        assert bool(L.dco_is_fast_build()) == bool(fast)

    @property
    def threads(self):
        return int(self.lib.dco_num_threads())

    @staticmethod
    def _coords(coords):
        c = np.ascontiguousarray(coords, dtype=np.float32)
        assert c.ndim == 2
        return c

    def dist2(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        return np.float32(self.lib.dco_dist2(_ptr(x, _f32p), _ptr(y, _f32p), x.size))

    def populations(self, coords, radii, i_from=0, i_to=None, boxgrid=False):
        """-> uint64 [n_radii, n_rows] in the order of ``radii``."""
        c = self._coords(coords)
        n, d = c.shape
        r = np.ascontiguousarray(radii, dtype=np.float32).reshape(-1)
        out = np.zeros((r.size, n), dtype=np.uint64)
        if n == 0 or r.size == 0:
            return out
        if boxgrid:
            assert i_from == 0 and i_to in (None, n)
            self.lib.dco_populations_boxgrid(_ptr(c, _f32p), n, d, _ptr(r, _f32p), r.size,
                                             _ptr(out, _u64p))
        else:
            self.lib.dco_populations_brute(_ptr(c, _f32p), n, d, _ptr(r, _f32p), r.size, i_from,
                                           n if i_to is None else i_to, _ptr(out, _u64p))
        return out

    def free_energies(self, pops):
        p = np.ascontiguousarray(pops, dtype=np.uint64).reshape(-1)
        fe = np.empty(p.size, dtype=np.float32)
        if p.size:
            self.lib.dco_free_energies(_ptr(p, _u64p), p.size, _ptr(fe, _f32p))
        return fe

    def nearest_neighbors(self, coords, fe, i_from=0, i_to=None):
        """-> (nn_idx u64, nn_d2 f32, hd_idx u64, hd_d2 f32); rows outside the range = sentinel."""
        c = self._coords(coords)
        n, d = c.shape
        f = np.ascontiguousarray(fe, dtype=np.float32).reshape(-1)
        assert f.size == n
        nn_idx = np.full(n, n + 1, dtype=np.uint64)
        hd_idx = np.full(n, n + 1, dtype=np.uint64)
        nn_d2 = np.full(n, np.finfo(np.float32).max, dtype=np.float32)
        hd_d2 = np.full(n, np.finfo(np.float32).max, dtype=np.float32)
        if n:
            self.lib.dco_nearest_neighbors(_ptr(c, _f32p), n, d, _ptr(f, _f32p), i_from,
                                           n if i_to is None else i_to, _ptr(nn_idx, _u64p),
                                           _ptr(nn_d2, _f32p), _ptr(hd_idx, _u64p),
                                           _ptr(hd_d2, _f32p))
        return nn_idx, nn_d2, hd_idx, hd_d2

    def sigma2(self, nn_d2):
        a = np.ascontiguousarray(nn_d2, dtype=np.float32)
        return float(self.lib.dco_sigma2(_ptr(a, _f32p), a.size))

    def lumping_radius(self, sigma2):
        return np.float32(self.lib.dco_lumping_radius(float(sigma2)))


class ScreeningOracle:
    """oracle/screening_oracle.cpp: the reference's screening / microstate assignment, quadratic scans."""

    def __init__(self):
        path = os.path.join(_BUILD, "libscreening_oracle.so")
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.dso_screening.restype = None
        L.dso_screening.argtypes = [_f32p, _f32p, C.c_float, _f32p, C.c_size_t, C.c_size_t, _u64p, _u64p]
        L.dso_assign_low_density.restype = None
        L.dso_assign_low_density.argtypes = [_u64p, _u64p, _f32p, C.c_size_t, _u64p]
        L.dso_sorted_names.restype = None
        L.dso_sorted_names.argtypes = [_u64p, C.c_size_t, _u64p]

    def screening(self, fe, nn_d2, threshold, coords, initial=None):
        c = np.ascontiguousarray(coords, dtype=np.float32)
        n, d = c.shape
        fe = np.ascontiguousarray(fe, dtype=np.float32)
        nn_d2 = np.ascontiguousarray(nn_d2, dtype=np.float32)
        out = np.zeros(n, dtype=np.uint64)
        ini = None if initial is None else np.ascontiguousarray(initial, dtype=np.uint64)
        self.lib.dso_screening(_ptr(fe, _f32p), _ptr(nn_d2, _f32p), float(np.float32(threshold)), _ptr(c, _f32p),
                               n, d, None if ini is None else _ptr(ini, _u64p), _ptr(out, _u64p))
        return out

    def assign_low_density(self, initial, hd_idx, fe):
        ini = np.ascontiguousarray(initial, dtype=np.uint64)
        hd = np.ascontiguousarray(hd_idx, dtype=np.uint64)
        fe = np.ascontiguousarray(fe, dtype=np.float32)
        out = np.zeros(ini.size, dtype=np.uint64)
        self.lib.dso_assign_low_density(_ptr(ini, _u64p), _ptr(hd, _u64p), _ptr(fe, _f32p), ini.size,
                                        _ptr(out, _u64p))
        return out

    def sorted_names(self, clustering):
        cl = np.ascontiguousarray(clustering, dtype=np.uint64)
        out = np.zeros(cl.size, dtype=np.uint64)
        self.lib.dso_sorted_names(_ptr(cl, _u64p), cl.size, _ptr(out, _u64p))
        return out


class Probe:
    """oracle/fastmath_probe.cpp: the reference's loop shape under the reference's flags (``order="avx"``: with the
    -mavx that -DCPU_ACCELERATION=AVX adds; ``"fma"``: with -mavx2 -mfma, a -march=native build on such a host)."""

    def __init__(self, order="sse2"):
        assert order in ("sse2", "avx", "fma")
        path = os.path.join(_BUILD, {"sse2": "libfastmath_probe.so", "avx": "libfastmath_probe_avx.so", "fma": "libfastmath_probe_fma.so"}[order])
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.probe_pairwise_d2.restype = None
        L.probe_pairwise_d2.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p]
        L.probe_free_energies.restype = None
        L.probe_free_energies.argtypes = [_u64p, C.c_size_t, _f32p]
        L.probe_box_index.restype = None
        L.probe_box_index.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float,
                                      C.POINTER(C.c_int)]

    @staticmethod
    def _aligned(a, align=32):
        """copy into a 32-byte aligned buffer (the reference's _mm_malloc, tools.hxx:96)."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        raw = np.empty(a.nbytes + align, dtype=np.uint8)
        off = (-raw.ctypes.data) % align
        out = raw[off:off + a.nbytes].view(np.float32).reshape(a.shape)
        out[...] = a
        return out

    def pairwise_d2(self, coords):
        c = self._aligned(coords)
        n, d = c.shape
        out = np.empty((n, n), dtype=np.float32)
        self.lib.probe_pairwise_d2(_ptr(c, _f32p), n, d, _ptr(out, _f32p))
        return out

    def free_energies(self, pops):
        p = np.ascontiguousarray(pops, dtype=np.uint64).reshape(-1)
        fe = np.empty(p.size, dtype=np.float32)
        self.lib.probe_free_energies(_ptr(p, _u64p), p.size, _ptr(fe, _f32p))
        return fe
