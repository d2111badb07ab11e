"""The persistent cache of the slow host-side staging tables (cosmo_pol_amd/tablecache.py): a warm
cache returns the same bits, a stale / damaged / foreign entry is rejected and rebuilt."""
import glob
import os

import numpy as np
import pytest


@pytest.fixture()
def cache(tmp_path, monkeypatch):
    monkeypatch.setenv('CPOL_CACHE_DIR', str(tmp_path))
    monkeypatch.setenv('CPOL_CACHE', '1')
    from cosmo_pol_amd import tablecache
    for k in tablecache.stats:
        tablecache.stats[k] = 0
    return tablecache, tmp_path


def _ice_inputs():
    from cosmo_pol_amd import hydrometeors as hyd, synthetic, config as cfg
    conf = cfg.sanity_check({'radar': {'coords': [46., 7., 500.], 'frequency': 5.6},
                             'microphysics': {'with_ice_crystals': 1}})
    vi = {v: i for i, v in enumerate(hyd.variable_list(conf))}
    return hyd, synthetic.make_lut('I', 5.6, '1mom', n_e=2, n_t=2), vi


def test_warm_cache_returns_the_same_bits_and_stale_entries_are_rebuilt(cache):
    tablecache, d = cache
    hyd, lut, vi = _ice_inputs()
    cold = hyd.build_hydro('I', '1mom', lut, vi)[4]
    assert tablecache.stats == {'hit': 0, 'miss': 1, 'stale': 0}
    files = glob.glob(str(d / 'ice_norm-*.npz'))
    assert len(files) == 1
    warm = hyd.build_hydro('I', '1mom', lut, vi)[4]
    assert tablecache.stats['hit'] == 1
    assert np.array_equal(np.asarray(cold).view(np.uint64), np.asarray(warm).view(np.uint64))
    # a file of the right name whose content belongs to something else (digest inside differs)
    with np.load(files[0]) as z:
        value = z['value']
    np.savez(files[0], digest=np.array('0' * 40), is_none=np.array(False), value=value + 1.0)
    again = hyd.build_hydro('I', '1mom', lut, vi)[4]
    assert tablecache.stats['stale'] == 1
    assert np.array_equal(np.asarray(cold).view(np.uint64), np.asarray(again).view(np.uint64))
    with np.load(files[0]) as z:                           # ... and it was replaced by a good one
        assert str(z['digest']) != '0' * 40
    # a truncated file
    with open(files[0], 'r+b') as f:
        f.truncate(100)
    again = hyd.build_hydro('I', '1mom', lut, vi)[4]
    assert tablecache.stats['stale'] == 2
    assert np.array_equal(np.asarray(cold).view(np.uint64), np.asarray(again).view(np.uint64))
    # the right digest but a damaged payload (wrong length): the verify callback rejects it
    with np.load(files[0]) as z:
        dg = str(z['digest'])
    np.savez(files[0], digest=np.array(dg), is_none=np.array(False), value=value[:-3])
    hyd.build_hydro('I', '1mom', lut, vi)
    assert tablecache.stats['stale'] == 3


def test_digest_follows_inputs_and_builder_source(cache, monkeypatch):
    tablecache, d = cache
    a = tablecache.digest_of(['x', np.arange(3.0), 1.5])
    assert a == tablecache.digest_of(['x', np.arange(3.0), 1.5])
    assert a != tablecache.digest_of(['x', np.arange(3.0) + 1e-16 * 0 + np.array([0, 0, 1e-12]), 1.5])
    assert a != tablecache.digest_of(['x', np.arange(3.0).astype(np.float32), 1.5])
    hyd, lut, vi = _ice_inputs()
    hyd.build_hydro('I', '1mom', lut, vi)
    n0 = len(os.listdir(d))
    monkeypatch.setattr(hyd, 'ICE_PANELS_PER_OCTAVE', 4)       # a different table: a different entry
    hyd.build_hydro('I', '1mom', lut, vi)
    assert len(os.listdir(d)) == n0 + 1
    monkeypatch.setenv('CPOL_CACHE', '0')                      # switched off: nothing read, nothing written
    before = dict(tablecache.stats)
    hyd.build_hydro('I', '1mom', lut, vi)
    assert tablecache.stats == before


def test_a_cache_directory_others_can_write_to_is_not_used(tmp_path, monkeypatch):
    """Advisor finding of round 3: the default directory name is predictable, so the cache only trusts a
    directory of this user that nobody else can write to (0700 when it creates it)."""
    import stat
    from cosmo_pol_amd import tablecache
    d = tmp_path / 'shared'
    monkeypatch.setenv('CPOL_CACHE_DIR', str(d))
    monkeypatch.setenv('CPOL_CACHE', '1')
    calls = []

    def compute():
        calls.append(1)
        return np.arange(4.0)
    tablecache.memo('t', ['x'], compute)
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700 and len(os.listdir(d)) == 1
    tablecache.memo('t', ['x'], compute)
    assert len(calls) == 1                                  # served from the cache
    os.chmod(d, 0o777)                                      # somebody else may write: not trusted any more
    assert tablecache.usable_dir() is None
    tablecache.memo('t', ['x'], compute)
    assert len(calls) == 2
    os.chmod(d, 0o700)
    tablecache.memo('t', ['x'], compute)
    assert len(calls) == 2
    assert tablecache.platform_parts()[0] == 'platform' and tablecache.platform_parts()[2] >= 64
