"""User configuration: YAML parsing, defaults, validation.

Keeps the keys, defaults and validation behaviour of the reference's
cosmo_pol/config/cfg.py (DEFAULTS :43-76, VALID_VALUES :84-118, init :162-188,
sanity_check :190-281; Range / TypeList: cosmo_pol/utilities/utilities.py:23-124):
invalid values are replaced by the default with a printed notice, a missing or
invalid mandatory key (no default, e.g. radar/coords) raises ValueError, keys
outside VALID_VALUES are kept untouched (e.g. the `attenuation:` section of the
shipped option files is silently ignored, as upstream).

Differences (documented): the configuration is an explicit object owned by a
RadarOperator, not a module-global; `radar/type` defaults to 'ground' (upstream
reads it unconditionally, global_constants.py:186, without defaulting it);
the sum-of-Gaussians antenna fit runs only for integration scheme 2 (and can
be bypassed with `integration/antenna_params`); a Nyquist file becomes a
`nyquist.NyquistTable`.
"""
import builtins
import copy
import re
from textwrap import dedent

import numpy as np
import yaml

BASIC_TYPES = [float, int, str]


class Range(object):
    """Closed interval test that is type-strict: 2.3 in Range(1.2, 5.3)."""

    def __init__(self, x0, x1):
        if type(x0) != type(x1):
            raise ValueError('range bounds are not of the same type!')
        if x1 <= x0:
            raise ValueError('Lower bound is larger than upper bound!')
        self.x0, self.x1, self.type = x0, x1, type(x0)

    def __contains__(self, z):
        return type(z) == self.type and self.x0 <= z <= self.x1

    def __str__(self):
        return 'Range of values from {:f} to {:f}'.format(self.x0, self.x1)


def generic_type(value):
    t = type(value)
    if t in (np.int64, np.int32):
        return int
    if t in (np.float64, np.float32, np.longdouble):
        return float
    if t == np.str_:
        return str
    return t


class TypeList(object):
    """Array-like of given element types and (optionally) given shape."""

    def __init__(self, types, dim=()):
        if type(types) != list:
            types = [types]
        if set(types) - set(BASIC_TYPES):
            raise ValueError('One of the specified types is invalid! Must be int, float, str')
        if any(d < 0 for d in dim):
            raise ValueError('Specified dimension is invalid (<0)!')
        self.types, self.dim = types, list(dim)

    def __eq__(self, array):
        try:
            arr = np.array(array)
            ok = True
            if len(self.dim):
                ok = all(d1 == d2 for d1, d2 in zip(arr.shape, self.dim))
            return bool(ok and all(generic_type(v) in self.types for v in arr.ravel()))
        except Exception:
            return False

    def __str__(self):
        names = [t.__name__ for t in self.types]
        if self.dim:
            return 'Array of {}, with dimensions {}'.format(names, self.dim)
        return 'Array of {}, with arbitrary dimensions'.format(names)


DEFAULTS = {
    'radar': {'range': 150000, 'radial_resolution': 500, 'PRI': 700, 'FFT_length': 256,
              'sensitivity': [-5, 10000], '3dB_beamwidth': 1., 'K_squared': 0.93,
              'antenna_speed': 0.2, 'nyquist_velocity': None, 'frequency': 9.41},
    'refraction': {'scheme': 1},
    'integration': {'scheme': 1, 'nv_GH': 9, 'nh_GH': 3, 'n_gaussians': 7,
                    'antenna_diagram': None, 'weight_threshold': 1., 'nr_GH': 7, 'na_GL': 7},
    'doppler': {'scheme': 1, 'turbulence_correction': 0, 'motion_correction': 0},
    'microphysics': {'scheme': '1mom', 'with_melting': 0, 'with_ice_crystals': 1,
                     'with_attenuation': 1, 'scattering': 'tmatrix_masc'},
}

VALID_VALUES = {
    'radar': {'coords': TypeList([float, int], [3]),
              'frequency': [2.7, 5.6, 9.41, 9.8, 13.6, 35.6],
              'range': Range(5000, 500000),
              'radial_resolution': Range(25, 5000),
              'PRI': Range(10, 3000),
              'FFT_length': Range(16, 2048),
              'sensitivity': [TypeList([float, int], [3]), TypeList([float, int], [2]), float],
              '3dB_beamwidth': Range(0.1, 10.),
              'K_squared': [float, None],
              'nyquist_velocity': [None, str],
              'antenna_speed': Range(1E-6, 10.)},
    'refraction': {'scheme': [1, 2]},
    'integration': {'scheme': [1, 2, 3, 4, 'ml'],
                    'nv_GH': range(1, 31, 2), 'nh_GH': range(1, 31, 2),
                    'n_gaussians': range(1, 13, 2),
                    'weight_threshold': Range(0.0001, 1.),
                    'nr_GH': range(1, 31, 2), 'na_GL': range(1, 31, 2)},
    'doppler': {'scheme': [1, 2, 3], 'turbulence_correction': [0, 1], 'motion_correction': [0, 1]},
    'microphysics': {'scheme': ['1mom', '2mom'], 'with_ice_crystals': [0, 1],
                     'with_melting': [0, 1], 'with_attenuation': [0, 1],
                     'scattering': ['tmatrix_masc', 'tmatrix', 'dda']},
}


def _check_validity(value, valid):
    if type(value) == list:
        # a list is valid as a whole (TypeList alternative) or element-wise
        if type(valid) == TypeList:
            return valid == value
        if type(valid) == list and any(type(v) == TypeList and v == value for v in valid):
            return True
        return all(_check_validity(i, valid) for i in value)
    if type(valid) == builtins.type:
        return type(value) == valid
    if type(valid) == list:
        return any(_check_validity(value, v) for v in valid)
    if type(valid) in (Range, range):
        return value in valid
    if type(valid) == str and valid[0:5] == '-reg-':
        return bool(re.match(valid[5:] + r'\Z', value))
    if type(valid) == TypeList:
        return valid == value
    return valid == value


def init(options_file):
    """Reads a YAML option file; falls back to DEFAULTS (with a notice) when
    the file cannot be read, as cfg.init does (cfg.py:162-188)."""
    try:
        with open(options_file, 'r') as f:
            return yaml.safe_load(f)
    except Exception as e:
        print(dedent('''
        Could not find or read {}, using default options...
        The error was:'''.format(options_file)))
        print(e)
        return copy.deepcopy(DEFAULTS)


def sanity_check(config):
    """Fills defaults / replaces invalid values; raises ValueError for missing
    mandatory keys (cfg.py:190-281)."""
    nyq_obj = None
    if config is not None and not isinstance((config.get('radar') or {}).get('nyquist_velocity'),
                                             (str, type(None))):
        nyq_obj = config['radar']['nyquist_velocity']      # already a NyquistTable (re-validation)
        config = dict(config, radar=dict(config['radar'], nyquist_velocity=None))
    config = copy.deepcopy(config) if config is not None else {}
    for section in VALID_VALUES:
        if section not in config or config[section] is None:
            config[section] = {}
        for key in VALID_VALUES[section]:
            mandatory = key not in DEFAULTS[section]
            if key not in config[section]:
                if mandatory:
                    raise ValueError(dedent('''
                        The mandatory key {:s}/{:s} is missing, please provide a valid
                        value, aborting...'''.format(section, key)))
                config[section][key] = copy.deepcopy(DEFAULTS[section][key])
            if not _check_validity(config[section][key], VALID_VALUES[section][key]):
                valid = VALID_VALUES[section][key]
                valid_str = [str(v) for v in valid] if type(valid) == list else str(valid)
                print(dedent('''
                    Invalid value entered for key: {:s}/{:s}
                    The value must be: {:s}'''.format(section, key, str(valid_str))))
                if mandatory:
                    raise ValueError(dedent('''
                        This key is mandatory, please provide a
                        valid value, aborting...'''))
                print('The default value {:s} was assigned'.format(str(DEFAULTS[section][key])))
                config[section][key] = copy.deepcopy(DEFAULTS[section][key])
    config['radar'].setdefault('type', 'ground')
    integ = config['integration']
    if integ.get('antenna_diagram') is not None and integ['scheme'] == 2 \
            and integ.get('antenna_params') is None:
        # cfg.py:241-252 fits whenever a diagram is given; only scheme 2 uses the fit
        from .quadrature import antenna_power_sq, fit_gaussians
        print('Trying to fit sum of gaussians on the provided antenna diagram...')
        angles, p2 = antenna_power_sq(integ['antenna_diagram'])
        integ['antenna_params'] = fit_gaussians(angles, 10 * np.log10(p2), integ['n_gaussians'])
        print('Fit was successful !')
    if isinstance(config['radar']['nyquist_velocity'], str):
        from .nyquist import NyquistTable
        config['radar']['nyquist_velocity'] = NyquistTable(config['radar']['nyquist_velocity'])
    if nyq_obj is not None:
        config['radar']['nyquist_velocity'] = nyq_obj
    if config['radar']['K_squared'] is None:
        from .dielectric import K_squared
        config['radar']['K_squared'] = float(K_squared(config['radar']['frequency']))
    return config
