"""Py-ART packaging of a simulated scan.

The reference returns `PyartRadop(scan_type, scan)`, a subclass of `pyart.core.Radar`
(cosmo_pol/radar/pyart_wrapper.py:186-342), from get_PPI / get_RHI.  Py-ART is not
installable in the build image, so `RadarScan` (radar_operator.py) is the default
container; when `pyart` imports, `as_pyart_radar(scan)` builds the same object the
reference would: same positional arguments of `pyart.core.Radar.__init__`, the field
metadata tables of the reference (units, long names, display ranges), ZH / ZV / ZDR in dB
with 0 -> NaN, masked where NaN, `Latitude` / `Longitude` / `rangearray` fields and the
Doppler velocity bins under `instrument_parameters['varray']`.
"""
import numpy as np

# field metadata of the reference (pyart_wrapper.py:26-104): data tables
UNITS_SIMUL = {'ZH': 'dBZ', 'KDP': 'deg/km', 'PHIDP': 'deg', 'RHOHV': '-', 'ZDR': 'dB', 'RVEL': 'm/s',
               'DSPECTRUM': 'dBZ', 'ZV': 'dBZ', 'U': 'm/s', 'V': 'm/s', 'W': 'm/s', 'T': 'K',
               'RHO': 'kg/m3', 'QR_v': 'kg/m3', 'QS_v': 'kg/m3', 'QG_v': 'kg/m3', 'QH_v': 'kg/m3',
               'ATT_H': 'dBZ', 'ATT_V': 'dBZ'}
VAR_LABELS_SIMUL = {'ZH': 'Reflectivity', 'KDP': 'Specific diff. phase', 'RHOHV': 'Copolar corr. coeff.',
                    'ZDR': 'Diff. reflectivity', 'RVEL': 'Mean doppler velocity',
                    'DSPECTRUM': 'Doppler spectrum', 'ZV': 'Vert. reflectivity', 'U': 'U-wind component',
                    'V': 'V-wind component', 'W': 'Vertical wind component', 'T': 'Temperature',
                    'RHO': 'Air density', 'QR_v': 'Mass density of rain', 'QS_v': 'Mass density of snow',
                    'QG_v': 'Mass density of graupel', 'QH_v': 'Mass density of hail',
                    'PHIDP': 'Diff. phase shift', 'ATT_H': 'Attenuation at hor. pol.',
                    'ATT_V': 'Attenuation at vert. pol.'}
VMIN_SIMUL = {'ZH': 0., 'KDP': 0., 'PHIDP': 0., 'RHOHV': 0.6, 'ZDR': 0., 'RVEL': -25, 'DSPECTRUM': -50,
              'ZV': 0., 'U': -30, 'V': -30, 'W': -10, 'T': 200, 'RHO': 0.5, 'QR_v': 0., 'QS_v': 0.,
              'QG_v': 0., 'QH_v': 0., 'ATT_H': 0, 'ATT_V': 0}
VMAX_SIMUL = {'ZH': 55, 'KDP': 1, 'PHIDP': 20, 'RHOHV': 1, 'ZDR': 2, 'RVEL': 25, 'DSPECTRUM': 30, 'ZV': 45,
              'U': 30, 'V': 30, 'W': 10, 'T': 300, 'RHO': 1.4, 'QR_v': 1E-3, 'QS_v': 1E-3, 'QG_v': 1E-3,
              'QH_v': 1E-2, 'ATT_H': 5, 'ATT_V': 5}


def pyart_available():
    try:
        import pyart  # noqa: F401
        return True
    except Exception:
        return False


def radar_arguments(scan, varray=None):
    """(args, kwargs) of pyart.core.Radar.__init__ as PyartRadop passes them
    (pyart_wrapper.py:335-339), from a RadarScan."""
    fields = {}
    for k, f in scan.fields.items():
        e = {'data': f['data']}
        for tab, key in ((VAR_LABELS_SIMUL, 'long_name'), (UNITS_SIMUL, 'units'),
                         (VMIN_SIMUL, 'valid_min'), (VMAX_SIMUL, 'valid_max')):
            if k in tab:
                e[key] = tab[k]
        if 'units' in f:
            e['units'] = f['units']
        fields[k] = e
    instrument_parameters = {'varray': {'data': np.asarray(varray)}} if varray is not None else {}
    args = (scan.time, scan.range, fields, {}, scan.scan_type, scan.latitude, scan.longitude,
            scan.altitude, scan.sweep_number, scan.sweep_mode, scan.fixed_angle,
            scan.sweep_start_ray_index, scan.sweep_stop_ray_index, scan.azimuth, scan.elevation)
    return args, {'instrument_parameters': instrument_parameters}


def as_pyart_radar(scan, varray=None):
    """RadarScan -> PyartRadop (a pyart.core.Radar with the reference's get_field); raises
    ImportError when Py-ART is missing."""
    from pyart import core

    class PyartRadop(core.Radar):
        """Radar-operator output as a Py-ART Radar (cosmo_pol/radar/pyart_wrapper.py:186)."""

        def get_field(self, sweep_idx, variable):
            i0 = int(self.sweep_start_ray_index['data'][sweep_idx])
            i1 = int(self.sweep_stop_ray_index['data'][sweep_idx]) + 1
            return self.fields[variable]['data'][i0:i1]

    args, kwargs = radar_arguments(scan, varray)
    return PyartRadop(*args, **kwargs)
