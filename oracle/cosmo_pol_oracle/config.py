"""Oracle configuration helper (TEST INFRASTRUCTURE ONLY).

Default values restate cosmo_pol/config/cfg.py:43-76 (DEFAULTS); 'type' is not
in the reference's DEFAULTS but is read unconditionally
(global_constants.py:186), so 'ground' is supplied here.
"""
import copy

DEFAULTS = {
    'radar': {'range': 150000, 'radial_resolution': 500, 'PRI': 700, 'FFT_length': 256,
              'sensitivity': [-5, 10000], '3dB_beamwidth': 1., 'K_squared': 0.93,
              'antenna_speed': 0.2, 'nyquist_velocity': None, 'frequency': 9.41,
              'type': 'ground'},
    'refraction': {'scheme': 1},
    'integration': {'scheme': 1, 'nv_GH': 9, 'nh_GH': 3, 'n_gaussians': 7,
                    'weight_threshold': 1., 'nr_GH': 7, 'na_GL': 7},
    'doppler': {'scheme': 1, 'turbulence_correction': 0, 'motion_correction': 0},
    'microphysics': {'scheme': '1mom', 'with_melting': 0, 'with_ice_crystals': 1,
                     'with_attenuation': 1, 'scattering': 'tmatrix_masc'},
}


def make_config(overrides=None):
    conf = copy.deepcopy(DEFAULTS)
    for sec, d in (overrides or {}).items():
        conf.setdefault(sec, {})
        conf[sec].update(copy.deepcopy(d))
    if 'coords' not in conf['radar']:
        raise ValueError('radar/coords is mandatory')
    return conf


def hydrometeor_list(config):
    """Order as in doppler_scatter.py:99-106 (NOT radar_operator.py:171-177)."""
    mp = config['microphysics']
    h = ['R', 'S', 'G']
    if mp['with_melting']:
        h += ['mS', 'mG']
    if mp['scheme'] == '2mom':
        h += ['H']
    if mp['with_ice_crystals']:
        h += ['I']
    return h
