"""Model files -> the arrays RadarOperator.load_model_arrays stages.

The reference reads a COSMO GRIB file (and its c-file of constant fields) through
pycosmo (`pc.open_file(...).get_variable(vars, get_proj_info=True, assign_heights=True,
cfile_name=...)`, cosmo_pol/radar_operator.py:217-309): it asks the file for
P, T, QV, QR, QC, QI, QS, QG, U, V, W (+ QH, QNH, QNR, QNS, QNG for the 2-moment scheme)
and gets back the DERIVED variables of BASE_VARIABLES (:35-36) -- mass and number
DENSITIES `Q*_v`, `QN*_v`, the air density RHO, the refractivity N -- each carrying
`attributes['z-levels']` [nz, ny, nx], `['proj_info']` (Lo1, La1, Lo2, La2,
Latitude_of_southern_pole, Longitude_of_southern_pole), `['resolution']` and `['time']`
(interpolation.py:547-561, radar_operator.py:191).  pycosmo and every GRIB decoder are
absent from this environment and the reference ships no sample file, so GRIB stays
refused; what IS read here, with the same variable names and the same attributes:

  * NetCDF classic (`.nc`, through scipy.io.netcdf_file), COSMO's own netCDF conventions:
    variables [time,] level, rlat, rlon; coordinate variables `rlon`, `rlat` (degrees, rotated);
    a `rotated_pole` variable with `grid_north_pole_latitude / _longitude`; the half-level
    heights `HHL` [level1 = nz + 1, rlat, rlon] in the file or in the c-file;
  * `.npz` archives with the same variable names, `rlon`, `rlat`, `HHL` (or full-level
    `z-levels`), and `proj_info` either as the six scalars named above or as
    `grid_north_pole_latitude / _longitude`.

Either kind may hold the derived variables directly (U, V, W, QR_v, ..., RHO, T [, N]) -- they
are taken as they are -- or the raw model output, from which they are derived here:

  RHO   = P / (R_d T (1 + (R_v / R_d - 1) QV - QC - QR - QS - QG - QI))   moist air with its condensate
  Q*_v  = Q* RHO,  QN*_v = QN* RHO                                        kg m-3, m-3
  N     = 77.6 / T (P / 100 + 4810 e / T),  e = QV P / 100 / (0.622 + 0.378 QV)    refractivity (N-units)

(COSMO's thermodynamic constants; *parity unpinned*: pycosmo's own expressions are not in
/root/reference.)  W on half levels (nz + 1) is averaged onto full levels; `level 0 = model
top` as in COSMO output.  Full-level heights are the means of the adjacent half levels.
"""
import os

import numpy as np

R_D, R_V = 287.05, 461.51
BASE_VARIABLES = ['U', 'V', 'W', 'QR_v', 'QS_v', 'QG_v', 'QI_v', 'RHO', 'T']
BASE_VARIABLES_2MOM = ['QH_v', 'QNH_v', 'QNR_v', 'QNS_v', 'QNG_v', 'QNI_v']
RAW_BASE = ['P', 'T', 'QV', 'QR', 'QC', 'QI', 'QS', 'QG', 'U', 'V', 'W']       # radar_operator.py:234
RAW_2MOM = ['QH', 'QNH', 'QNR', 'QNS', 'QNG']                                   # :235
PROJ_KEYS = ['Lo1', 'La1', 'Lo2', 'La2', 'Latitude_of_southern_pole', 'Longitude_of_southern_pole']


class _Npz(object):
    def __init__(self, path):
        self.z = np.load(path, allow_pickle=False)

    def names(self):
        return set(self.z.files)

    def get(self, k):
        return np.asarray(self.z[k])

    def attr(self, var, k):
        # (attributes are scalar entries of the archive, whatever variable they would hang on in a NetCDF file)
        if k in self.z.files and self.z[k].ndim == 0 and var in (None, 'rotated_pole'):
            return self.z[k].item()
        return None

    def close(self):
        self.z.close()


class _NetCDF(object):
    def __init__(self, path):
        from scipy.io import netcdf_file
        self.f = netcdf_file(path, 'r', mmap=False)

    def names(self):
        return set(self.f.variables)

    def get(self, k):
        v = self.f.variables[k]
        a = np.array(v[:], dtype=np.float64 if v.data.dtype.kind == 'f' and v.data.dtype.itemsize == 8 else None)
        scale, off = getattr(v, 'scale_factor', None), getattr(v, 'add_offset', None)
        if scale is not None or off is not None:
            a = a * (1.0 if scale is None else scale) + (0.0 if off is None else off)
        return a

    def attr(self, var, k):
        obj = self.f.variables.get(var) if var else self.f
        val = getattr(obj, k, None) if obj is not None else None
        if isinstance(val, bytes):
            val = val.decode()
        if isinstance(val, np.ndarray) and val.size == 1:
            val = val.reshape(-1)[0]
        return val

    def close(self):
        self.f.close()


def _open(path):
    if not os.path.exists(path):
        raise IOError('model file %r does not exist' % (path,))
    with open(path, 'rb') as f:
        magic = f.read(8)
    if magic[:4] == b'GRIB':
        raise NotImplementedError(
            'GRIB input needs pycosmo (cosmo_pol/radar_operator.py:229), which is not installable here and of '
            'which /root/reference holds no source: convert the file to NetCDF classic (e.g. `fxconvert nc` / '
            '`cdo -f nc copy`) or pass arrays to RadarOperator.load_model_arrays')
    if magic[:3] == b'CDF':
        return _NetCDF(path)
    if magic[:4] == b'\x89HDF':
        raise NotImplementedError('NetCDF-4 / HDF5 files need netCDF4 or h5py (absent here): write NetCDF classic '
                                  '(`nccopy -k classic`) or an .npz archive (cosmo_pol_amd/model_io.py)')
    if magic[:2] == b'PK':
        return _Npz(path)
    raise ValueError('model file %r is neither NetCDF classic, an .npz archive nor GRIB' % (path,))


def _levels_first(a, nz_hint=None):
    """[time = 1,] level, y, x -> [level, y, x] float32."""
    a = np.asarray(a)
    while a.ndim > 3 and a.shape[0] == 1:
        a = a[0]
    if a.ndim != 3:
        raise ValueError('model variables must be [level, rlat, rlon] (with an optional time axis of length 1): '
                         'got shape %s' % (a.shape,))
    return np.ascontiguousarray(a, dtype=np.float32)


def _proj_info(src, rlon, rlat):
    """The six keys of the reference's proj_info from what the file offers."""
    got = {k: src.attr(None, k) for k in PROJ_KEYS}
    if all(v is not None for v in got.values()):
        return {k: float(v) for k, v in got.items()}
    np_lat = np_lon = None
    for var in ('rotated_pole', None):
        if np_lat is None:
            np_lat = src.attr(var, 'grid_north_pole_latitude')
            np_lon = src.attr(var, 'grid_north_pole_longitude')
    if np_lat is None or rlon is None or rlat is None:
        raise ValueError('the model file gives no projection: need Lo1, La1, Lo2, La2, Latitude_of_southern_pole, '
                         'Longitude_of_southern_pole, or rlon / rlat with rotated_pole:grid_north_pole_latitude / '
                         '_longitude')
    # the south pole of the rotated grid lies opposite its north pole
    sp_lat = -float(np_lat)
    sp_lon = float(np_lon) - 180.0
    if sp_lon < -180.0:
        sp_lon += 360.0
    return {'Lo1': float(rlon[0]), 'La1': float(rlat[0]), 'Lo2': float(rlon[-1]), 'La2': float(rlat[-1]),
            'Latitude_of_southern_pole': sp_lat, 'Longitude_of_southern_pole': sp_lon}


def derive(raw, want_2mom, want_refractivity):
    """Raw model output {P, T, QV, Q*, [QN*], U, V, W} (float32 [nz, ny, nx]) -> the derived variables of
    BASE_VARIABLES (float32), see the module docstring."""
    f64 = lambda k: raw[k].astype(np.float64)                                     # noqa: E731
    T, P, QV = f64('T'), f64('P'), f64('QV')
    load = sum(f64(k) for k in ('QC', 'QR', 'QS', 'QG', 'QI') if k in raw)
    rho = P / (R_D * T * (1.0 + (R_V / R_D - 1.0) * QV - load))
    out = {'U': raw['U'], 'V': raw['V'], 'W': raw['W'], 'T': raw['T'], 'RHO': rho.astype(np.float32)}
    for k in ('QR', 'QS', 'QG', 'QI'):
        out[k + '_v'] = (f64(k) * rho).astype(np.float32)
    if want_2mom:
        for k in ('QH', 'QNH', 'QNR', 'QNS', 'QNG', 'QNI'):
            out[k + '_v'] = ((f64(k) * rho) if k in raw else np.zeros_like(rho)).astype(np.float32)
    if want_refractivity:
        e = QV * (P / 100.0) / (0.622 + 0.378 * QV)
        out['N'] = (77.6 / T * (P / 100.0 + 4810.0 * e / T)).astype(np.float32)
    return out


def read_model_file(filename, cfilename=None, want_refractivity=False):
    """-> dict(data={name: [nz, ny, nx] float32}, zlevels [nz, ny, nx] float32, proj_info, resolution (dlon, dlat),
    time, scheme '1mom' | '2mom', derived_from_raw bool).  Raises ValueError like the reference when a necessary
    variable is missing (radar_operator.py:264-275)."""
    src = _open(filename)
    csrc = None
    try:
        names = src.names()
        have_derived = all(k in names for k in BASE_VARIABLES)
        have_raw = all(k in names for k in RAW_BASE)
        if not (have_derived or have_raw):
            need = BASE_VARIABLES if any(k.endswith('_v') for k in names) else RAW_BASE
            raise ValueError('Not all necessary variables could be found in file: missing %s'
                             % sorted(k for k in need if k not in names))
        if have_derived:
            two_mom = all(k in names for k in BASE_VARIABLES_2MOM)
            keys = BASE_VARIABLES + (BASE_VARIABLES_2MOM if two_mom else []) + (['N'] if 'N' in names else [])
            data = {k: _levels_first(src.get(k)) for k in keys}
        else:
            two_mom = all(k in names for k in RAW_2MOM)
            keys = RAW_BASE + ([k for k in RAW_2MOM + ['QNI'] if k in names] if two_mom else [])
            raw = {k: _levels_first(src.get(k)) for k in keys}
            nz = raw['T'].shape[0]
            if raw['W'].shape[0] == nz + 1:                      # W on half levels
                raw['W'] = (0.5 * (raw['W'][:-1].astype(np.float64) + raw['W'][1:])).astype(np.float32)
            data = derive(raw, two_mom, want_refractivity)
        nz, ny, nx = data['T'].shape
        for k, v in data.items():
            if v.shape != (nz, ny, nx):
                raise ValueError('variable %s has shape %s, T has %s' % (k, v.shape, (nz, ny, nx)))
        # heights: this file, else the c-file
        zl = None
        for s in (src, None):
            if s is None:
                if cfilename is None:
                    break
                csrc = s = _open(cfilename)
            n = s.names()
            if 'z-levels' in n:
                zl = _levels_first(s.get('z-levels'))
            elif 'HHL' in n:
                hhl = _levels_first(s.get('HHL')).astype(np.float64)
                zl = (0.5 * (hhl[:-1] + hhl[1:])).astype(np.float32) if hhl.shape[0] == nz + 1 else hhl.astype(np.float32)
            if zl is not None:
                break
        if zl is None:
            raise ValueError('no level heights: the model file or the c-file (cfilename) must hold HHL '
                             '[nz + 1, rlat, rlon] or z-levels [nz, rlat, rlon]')
        if zl.shape != (nz, ny, nx):
            raise ValueError('level heights have shape %s, the variables %s' % (zl.shape, (nz, ny, nx)))
        if zl[0].mean() < zl[-1].mean():
            raise ValueError('level 0 must be the model top (heights decreasing with the level index)')
        geo = src if 'rlon' in names or src.attr(None, 'Lo1') is not None else (csrc or src)
        gn = geo.names()
        rlon = np.asarray(geo.get('rlon'), dtype=np.float64).reshape(-1) if 'rlon' in gn else None
        rlat = np.asarray(geo.get('rlat'), dtype=np.float64).reshape(-1) if 'rlat' in gn else None
        proj = _proj_info(geo, rlon, rlat)
        res = geo.attr(None, 'resolution_lon'), geo.attr(None, 'resolution_lat')
        if res[0] is None or res[1] is None:
            res = ((proj['Lo2'] - proj['Lo1']) / max(nx - 1, 1), (proj['La2'] - proj['La1']) / max(ny - 1, 1))
        time = None
        if 'time' in names:
            t = np.asarray(src.get('time')).reshape(-1)
            units = src.attr('time', 'units')
            time = ('%s %s' % (t[0], units) if units else (str(t[0]) if t.dtype.kind in 'US' else float(t[0])))
        elif src.attr(None, 'time') is not None:
            time = src.attr(None, 'time')
        return {'data': data, 'zlevels': zl, 'proj_info': proj, 'resolution': (float(res[0]), float(res[1])),
                'time': time, 'scheme': '2mom' if two_mom else '1mom', 'derived_from_raw': not have_derived}
    finally:
        src.close()
        if csrc is not None:
            csrc.close()


# ---------------------------------------------------------------- writers (tools, tests)
def write_npz(path, data, zlevels=None, hhl=None, proj_info=None, rlon=None, rlat=None, north_pole=None, time=None):
    """The .npz layout read_model_file reads."""
    out = {k: np.asarray(v) for k, v in data.items()}
    if zlevels is not None:
        out['z-levels'] = np.asarray(zlevels)
    if hhl is not None:
        out['HHL'] = np.asarray(hhl)
    for k, v in (proj_info or {}).items():
        out[k] = np.float64(v)
    if rlon is not None:
        out['rlon'], out['rlat'] = np.asarray(rlon, dtype=np.float64), np.asarray(rlat, dtype=np.float64)
    if north_pole is not None:
        out['grid_north_pole_latitude'], out['grid_north_pole_longitude'] = np.float64(north_pole[0]), np.float64(north_pole[1])
    if time is not None:
        out['time'] = np.array(time)
    np.savez(path, **out)


def write_netcdf(path, data, rlon, rlat, north_pole, hhl=None, time_hours=0.0):
    """A NetCDF classic file in COSMO's conventions (time, level, rlat, rlon; rotated_pole; HHL on level1)."""
    from scipy.io import netcdf_file
    any_v = next(iter(data.values()))
    nz = min(np.asarray(v).shape[0] for v in data.values())
    f = netcdf_file(path, 'w')
    try:
        f.createDimension('time', 1)
        f.createDimension('level', nz)
        f.createDimension('level1', nz + 1)
        f.createDimension('rlat', any_v.shape[1])
        f.createDimension('rlon', any_v.shape[2])
        v = f.createVariable('time', 'f8', ('time',))
        v[:] = [time_hours]
        v.units = 'hours since 2014-08-13 00:00:00'
        for name, vals in (('rlon', rlon), ('rlat', rlat)):
            v = f.createVariable(name, 'f8', (name,))
            v[:] = np.asarray(vals, dtype=np.float64)
            v.units = 'degrees'
        v = f.createVariable('rotated_pole', 'c', ())
        v.grid_mapping_name = 'rotated_latitude_longitude'
        v.grid_north_pole_latitude = float(north_pole[0])
        v.grid_north_pole_longitude = float(north_pole[1])
        for name, vals in data.items():
            vals = np.asarray(vals, dtype=np.float32)
            lev = 'level1' if vals.shape[0] == nz + 1 else 'level'
            v = f.createVariable(name, 'f4', ('time', lev, 'rlat', 'rlon'))
            v[0] = vals
            v.grid_mapping = 'rotated_pole'
        if hhl is not None:
            v = f.createVariable('HHL', 'f4', ('time', 'level1', 'rlat', 'rlon'))
            v[0] = np.asarray(hhl, dtype=np.float32)
    finally:
        f.close()
