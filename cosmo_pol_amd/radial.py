"""Radial records over the batched results of the HIP path.

The reference hands lists of `Radial` objects from the scan loop to
`cut_at_sensitivity` and `PyartRadop` (cosmo_pol/interpolation/radial.py:17-54,
cosmo_pol/radar_operator.py:411-421, 445-451).  The HIP path returns one
[n_rays, n_gates] array per variable; `to_radials` re-expresses such a result as
the reference's per-radial records (row views, no copies), so reference-side code
that consumes lists of radials -- `cut_at_sensitivity(list_sweeps)`,
`PyartRadop('ppi', {... 'data': list_sweeps})` -- runs on it unchanged
(level B).
"""
import numpy as np

# keys of a simulate_rays result that are not simulated / model variables
_GEOMETRY = ('mask', 'lats', 'lons', 'dist', 'heights', 'n_sub', 'model_vars')


class Radial(object):
    """Same attributes as cosmo_pol.interpolation.radial.Radial (radial.py:17-54)."""

    def __init__(self, dic_values, mask, lats_profile, lons_profile, dist_ground_profile,
                 heights_profile, elev_profile=None, quad_pt=None, quad_weight=1):
        self.mask = mask
        self.quad_pt = [] if quad_pt is None else quad_pt
        self.quad_weight = quad_weight
        self.lats_profile = lats_profile
        self.lons_profile = lons_profile
        self.dist_profile = dist_ground_profile
        self.heights_profile = heights_profile
        self.elev_profile = [] if elev_profile is None else elev_profile
        self.values = dic_values
        # only meaningful for sub-radials inside the melting scheme (radial.py:50-53)
        self.has_melting = False
        self.mask_ml = None


def to_radials(result, azimuths=None, elevations=None, model_names=None):
    """`result`: dict of [n_rays, n_gates] arrays as returned by
    RadarOperator.simulate_rays (or one packaged sweep of RadarScan.raw, whose variables sit
    under 'fields').  Returns one Radial per ray whose `values` are row VIEWS of the batched
    arrays (so censoring them in place, as cut_at_sensitivity does, edits the batch).
    `azimuths` / `elevations` fill `quad_pt` = [azimuth, elevation] of the integrated radial's
    central sub-beam when given (doppler_scatter.py:480-489 leaves it empty)."""
    if 'fields' in result:                                  # a packaged sweep
        variables = dict(result['fields'])
        azimuths = result.get('azimuth') if azimuths is None else azimuths
        elevations = result.get('elevation') if elevations is None else elevations
    else:
        variables = {k: v for k, v in result.items() if k not in _GEOMETRY}
        if 'model_vars' in result and model_names is not None:
            for i, name in enumerate(model_names):
                variables[name] = result['model_vars'][i]
    n_rays = np.asarray(result['mask']).shape[0]
    out = []
    for r in range(n_rays):
        values = {k: v[r] for k, v in variables.items()}
        qp = []
        if azimuths is not None and elevations is not None:
            qp = [float(np.asarray(azimuths).reshape(-1)[r]), float(np.asarray(elevations).reshape(-1)[r])]
        out.append(Radial(values, result['mask'][r], result['lats'][r], result['lons'][r],
                          result['dist'][r], result['heights'][r], quad_pt=qp))
    return out
