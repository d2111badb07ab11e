"""oracle/ref_shim.py -- TEST INFRASTRUCTURE ONLY.

Imports the *reference* package (wolfidan/cosmo_pol, mounted read-only at
/root/reference) by path, in this container only, so that
  (i)  oracle/cosmo_pol_oracle (our CPU restatement) can be checked against it,
  (ii) tests/golden/*.npz fixtures can be generated (oracle/gen_golden.py).

Nothing of the reference is copied: the shim only supplies
  * NumPy-2 / SciPy compatibility aliases the 2017-era code expects,
  * empty stand-ins for third-party modules that are not installed here
    (pycosmo, pyproj, pyart, h5py, netCDF4),
  * ctypes-backed stand-ins for the two SWIG extension modules, backed by
    oracle/_ref/lib*_ref.so (built by oracle/Makefile from the reference's own
    .c files where they lie),
  * geodesy injection: pyproj.Geod / pycosmo.WGS_to_COSMO are bound to the
    oracle's own implementations (the real packages' source is not under
    /root/reference -> "parity unpinned" at that boundary).

The GPU box has no /root/reference: nothing under tests -m gpu, smoke() or
bench.py imports this file.
"""
import ctypes
import os
import sys
import types
import warnings

import numpy as np

REF_ROOT = os.environ.get("COSMO_POL_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_loaded = {}


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "cosmo_pol"))


def _build_ref_libs():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _c_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _make_interp_module():
    lib = ctypes.CDLL(os.path.join(_HERE, "_ref", "libinterp_ref.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    lib.get_all_radar_pts.restype = fp
    lib.get_all_radar_pts.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                      fp, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, fp, ctypes.c_int, fp, ctypes.c_int]
    lib.binary_search.restype = ctypes.c_int
    lib.binary_search.argtypes = [fp, ctypes.c_int, ctypes.c_float]

    def P(a):
        return a.ctypes.data_as(fp)

    def get_all_radar_pts(n, coords, heights, data, zlevels, llc, res):
        # mirrors the SWIG typemaps of interpolation_c.i:15-21 (ARGOUT first,
        # IN_ARRAYs converted to contiguous float32)
        coords = _c_f32(coords)
        heights = _c_f32(heights)
        data = _c_f32(data)
        zlevels = _c_f32(zlevels)
        llc = _c_f32(llc)
        res = _c_f32(res)
        out = np.empty(int(n), dtype=np.float32)
        lib.get_all_radar_pts(P(out), int(n), P(coords), coords.shape[0], coords.shape[1],
                              P(heights), heights.shape[0],
                              P(data), data.shape[0], data.shape[1], data.shape[2],
                              P(zlevels), zlevels.shape[0], zlevels.shape[1], zlevels.shape[2],
                              P(llc), llc.shape[0], P(res), res.shape[0])
        return [None, out]

    def binary_search(arr, key):
        arr = _c_f32(arr)
        return lib.binary_search(P(arr), arr.shape[0], float(key))

    m = types.ModuleType("_interpolation_c")
    m.get_all_radar_pts = get_all_radar_pts
    m.binary_search = binary_search
    m.trilinear_interp = None
    m._lib = lib
    return m


def _make_doppler_module():
    lib = ctypes.CDLL(os.path.join(_HERE, "_ref", "libdoppler_ref.so"))
    fp = ctypes.POINTER(ctypes.c_float)
    lib.get_refl.restype = fp
    lib.get_refl.argtypes = ([fp, ctypes.c_int] + [fp, ctypes.c_int, ctypes.c_int] * 5
                             + [fp, ctypes.c_int] * 2)
    m = types.ModuleType("_doppler_c")

    def get_refl(n, Da, Db, rcs, D, N, step_D, Dmin):
        # mirrors the SWIG typemaps of doppler_c.i:16-23 (ARGOUT first, IN_ARRAYs converted
        # to contiguous float32); returns [return value, refl] as the SWIG wrapper does
        Da, Db, rcs, D, N, step_D, Dmin = [_c_f32(a) for a in (Da, Db, rcs, D, N, step_D, Dmin)]
        out = np.empty(int(n), dtype=np.float32)

        def P(a):
            return a.ctypes.data_as(fp)
        lib.get_refl(P(out), int(n), P(Da), Da.shape[0], Da.shape[1], P(Db), Db.shape[0],
                     Db.shape[1], P(rcs), rcs.shape[0], rcs.shape[1], P(D), D.shape[0], D.shape[1],
                     P(N), N.shape[0], N.shape[1], P(step_D), step_D.shape[0], P(Dmin), Dmin.shape[0])
        return [None, out]

    m.get_refl = get_refl
    m._lib = lib
    return m


class _Geod(object):
    """Stand-in for pyproj.Geod(ellps='WGS84'): fwd(lon, lat, az, dist)."""

    def __init__(self, ellps="WGS84"):
        assert ellps == "WGS84"

    def fwd(self, lon, lat, az, dist):
        from cosmo_pol_oracle import geodesy
        lat2, lon2 = geodesy.wgs84_direct(float(lat), float(lon), float(az),
                                          np.asarray([dist], dtype=np.float64))
        return float(lon2[0]), float(lat2[0]), 0.0


def _wgs_to_cosmo(coords, sp):
    from cosmo_pol_oracle import geodesy
    if len(coords) == 3 and np.isscalar(coords[0]):
        # single point [lat, lon, alt] (atm_refraction.py:101): 1-D (rlat, rlon)
        return geodesy.wgs_to_rotated(np.array([float(coords[0])]), np.array([float(coords[1])]),
                                      float(sp[0]), float(sp[1]))[0]
    lats, lons = coords
    return geodesy.wgs_to_rotated(np.asarray(lats, dtype=np.float64),
                                  np.asarray(lons, dtype=np.float64),
                                  float(sp[0]), float(sp[1]))


def load_reference():
    """Returns the imported reference package `cosmo_pol` (cached)."""
    if "pkg" in _loaded:
        return _loaded["pkg"]
    if not reference_available():
        raise RuntimeError("reference not available at %s" % REF_ROOT)
    _build_ref_libs()
    sys.dont_write_bytecode = True
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    # --- NumPy 2 / SciPy compat aliases (removed upstream names) ---
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    if not hasattr(np, "warnings"):
        np.warnings = warnings
    if not hasattr(np, "float128"):
        np.float128 = np.longdouble
    import scipy.integrate
    if not hasattr(scipy.integrate, "trapz"):
        scipy.integrate.trapz = scipy.integrate.trapezoid
    if not hasattr(np, "trapz"):
        np.trapz = np.trapezoid
    import scipy.ndimage
    if "scipy.ndimage.filters" not in sys.modules:
        filt = types.ModuleType("scipy.ndimage.filters")
        filt.gaussian_filter = scipy.ndimage.gaussian_filter
        sys.modules["scipy.ndimage.filters"] = filt
        scipy.ndimage.filters = filt

    # --- empty stand-ins for missing third-party packages ---
    for name in ["pycosmo", "h5py", "pyproj", "netCDF4", "pyart", "pyart.graph",
                 "pyart.graph.radardisplay", "pyart.filters", "pyart.config",
                 "pyart.core", "pyart.correct", "pyart.io", "pyart.util",
                 "pyart.core.transforms", "pyart.graph.common"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["pyart.graph.radardisplay"].RadarDisplay = object
    sys.modules["pyart.core"].Radar = object
    sys.modules["pyart"].core = sys.modules["pyart.core"]
    sys.modules["pyart"].graph = sys.modules["pyart.graph"]
    sys.modules["pyart"].config = sys.modules["pyart.config"]
    sys.modules["pyart"].filters = sys.modules["pyart.filters"]
    sys.modules["pyart.graph"].radardisplay = sys.modules["pyart.graph.radardisplay"]
    sys.modules["pyart.graph"].common = sys.modules["pyart.graph.common"]
    sys.modules["pyart.core"].transforms = sys.modules["pyart.core.transforms"]
    sys.modules["pyart.config"].get_metadata = lambda *a, **k: {}
    sys.modules["pyart.config"].get_field_name = lambda *a, **k: a[0] if a else None

    # --- geodesy injection (the oracle's own implementations) ---
    sys.modules["pyproj"].Geod = _Geod
    sys.modules["pycosmo"].WGS_to_COSMO = _wgs_to_cosmo

    # --- SWIG extension stand-ins ---
    interp = _make_interp_module()
    dopp = _make_doppler_module()
    sys.modules["_interpolation_c"] = interp
    sys.modules["cosmo_pol.interpolation._interpolation_c"] = interp
    sys.modules["_doppler_c"] = dopp
    sys.modules["cosmo_pol.scatter._doppler_c"] = dopp

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import cosmo_pol  # noqa: F401  (the reference, by path)
        import cosmo_pol.config.cfg
        import cosmo_pol.constants
        import cosmo_pol.interpolation
        import cosmo_pol.scatter
        import cosmo_pol.hydrometeors
        import cosmo_pol.lookup
    _loaded["pkg"] = cosmo_pol
    _loaded["interp_module"] = interp
    return cosmo_pol


import contextlib


@contextlib.contextmanager
def numpy1_linspace():
    """NumPy < 1.16 semantics of `np.linspace` for ONE-ELEMENT array endpoints, for the duration of a reference call.

    The 2017 code writes `D[:, j] = np.linspace(h.d_min, h.d_max, h.nbins_D)` with `d_min`, `d_max` of shape (1,) for the
    melting species (doppler_scatter.py:687-689; the `f_wet` setter, hydrometeors.py:335-342).  The NumPy it was written
    for computed `arange(num) * step + start` and returned shape (num,); NumPy >= 1.16 returns (num, 1) and the assignment
    raises ValueError -- Doppler scheme 3 with the melting scheme is dead under this container's NumPy 2.2.  Same
    arithmetic (arange * step + start, last element = stop), only the shape of the result differs.  Used by
    oracle/gen_golden.py for the `d3_melt*` cases ONLY, and said so in their fixtures (`numpy1_linspace` = 1)."""
    real = np.linspace

    def linspace(start, stop, num=50, *args, **kw):
        s, e = np.asarray(start), np.asarray(stop)
        if s.shape == (1,) and e.shape == (1,):
            return real(s[0], e[0], num, *args, **kw)
        return real(start, stop, num, *args, **kw)
    np.linspace = linspace
    try:
        yield
    finally:
        np.linspace = real


class KeyListDict(dict):
    """dict whose keys()/values() return lists (py2 idiom the reference uses,
    interpolation.py:118-119,411)."""

    def keys(self):
        return list(dict.keys(self))

    def values(self):
        return list(dict.values(self))


class ModelVar(object):
    """Duck type of a pycosmo variable (interpolation.py:547-561)."""

    def __init__(self, name, data, zlevels, proj_info, resolution, time=None):
        self.name = name
        self.data = data
        self.attributes = {"z-levels": zlevels, "proj_info": proj_info,
                           "resolution": resolution, "time": time}


def configure_reference(conf_overrides):
    """Sets cfg.CONFIG of the reference from DEFAULTS + overrides (no YAML),
    re-derives the constants; returns the checked config."""
    import copy
    load_reference()
    from cosmo_pol.config import cfg
    from cosmo_pol.constants import global_constants
    conf = copy.deepcopy(cfg.DEFAULTS)
    for sec, d in conf_overrides.items():
        conf.setdefault(sec, {})
        conf[sec].update(d)
    # the reference's sanity_check fits Gaussians whenever the key is present and that fit
    # raises under Python 3 (antenna_fit.py:47): keep both keys out of it, set them after
    late = {k: conf["integration"].pop(k) for k in ("antenna_diagram", "antenna_params")
            if k in conf["integration"]}
    conf["radar"].setdefault("type", "ground")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        cfg.CONFIG = cfg.sanity_check(conf)
    for k, v in late.items():
        if v is not None:
            cfg.CONFIG["integration"][k] = np.array(v) if k == "antenna_params" else v
    global_constants.update()
    return cfg.CONFIG
