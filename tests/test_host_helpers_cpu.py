"""Host-side helpers of RadarOperator that need no GPU."""
import numpy as np


def test_mask_from_its_one_byte_form_equals_the_reference_statements():
    """cpol_outputs.mask_sum8 (the sum of the sub-beams' mask codes, one byte per gate) -> the float64 radial mask: the two
    statements of doppler_scatter.py:472-477 (`mask /= n_sub`, values in (-1, 0] -> 0) on every possible sum."""
    from cosmo_pol_amd.radar_operator import _mask_from_sum
    rng = np.random.default_rng(3)
    for n_sub in (1, 3, 15, 49, 63):
        s = rng.integers(-n_sub, 2 * n_sub + 1, size=(7, 50)).astype(np.int8)
        s.flat[:4] = (-n_sub, 0, n_sub, 2 * n_sub)
        ref = np.zeros(s.shape)
        ref = ref + s
        ref /= float(n_sub)
        ref[np.logical_and(ref > -1, ref <= 0)] = 0
        got = _mask_from_sum(s, n_sub)()
        assert got.dtype == np.float64 and np.array_equal(got, ref), n_sub


def test_lazy_result_builds_the_mask_once_and_keeps_its_place():
    from cosmo_pol_amd.radar_operator import LazyDict, _mask_from_sum
    s = np.array([[0, 1, -1, 2]], dtype=np.int8)
    res = LazyDict()
    res['ZH'] = np.ones((1, 4), dtype=np.float32)
    res['mask_sum8'] = s
    res.add('mask', _mask_from_sum(s, 1))
    assert list(res) == ['ZH', 'mask_sum8', 'mask'] and res.pending('mask')
    m = res['mask']
    assert not res.pending('mask') and res['mask'] is m and m.tolist() == [[0.0, 1.0, -1.0, 2.0]]
    assert dict(res.items())['mask'] is m
