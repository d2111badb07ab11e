"""tools/make_luts.py: the on-disk table format either side of the hot path
(reference: cosmo_pol/lookup/lut.py:27-154)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import make_luts  # noqa: E402

from cosmo_pol_amd import lut as lutmod  # noqa: E402
from cosmo_pol_amd import synthetic  # noqa: E402


def test_write_then_load_all(tmp_path, capsys):
    hyds = ['R', 'S', 'mS']
    files = make_luts.write(str(tmp_path), 5.6, '1mom', hyds, 'tmatrix_masc', n_e=3, n_t=4, quiet=True)
    assert [os.path.basename(f) for f in files] == ['lut_SZ_R_5_6_1mom.lut', 'lut_SZ_S_5_6_1mom.lut',
                                                    'lut_SZ_mS_5_6_1mom.lut']
    got = lutmod.load_all_lut('1mom', hyds, 5.6, 'tmatrix_masc', lut_dir=str(tmp_path))
    for h in hyds:
        want = synthetic.make_lut(h, 5.6, '1mom', n_e=3, n_t=4)
        assert got[h].value_table.dtype == np.float64
        assert np.array_equal(got[h].value_table, want.value_table)
        assert got[h].axes_names == want.axes_names
        for a, b in zip(got[h].axes, want.axes):
            assert np.array_equal(np.asarray(a), np.asarray(b))
        e = np.array([0.3, 3.9], dtype=np.float32)
        t = np.array([want.axes[1][0] + 0.1, want.axes[1][2]], dtype=np.float32)
        kw = {'e': e, ('wc' if h == 'mS' else 't'): t}
        assert np.array_equal(got[h].lookup_line(**kw), want.lookup_line(**kw))
    make_luts.main(['info', files[2]])
    out = capsys.readouterr().out
    assert '(3, 4, 1024, 12)' in out and 'axis 2 d' in out


def test_missing_table_raises(tmp_path):
    import pytest
    with pytest.raises(IOError):
        lutmod.load_all_lut('1mom', ['R'], 5.6, 'tmatrix_masc', lut_dir=str(tmp_path))


def test_file_read_by_reference_loader(tmp_path):
    """tests/golden/lut_file.npz: our .lut files, loaded and queried by the reference's
    own load_lut / lookup_line (oracle/gen_golden.py::gen_lut_file)."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'lut_file.npz'))
    for h in ['S', 'mG']:
        path = str(tmp_path / lutmod.lut_filename(h, 9.41, '1mom'))
        lutmod.save_lut(synthetic.make_lut(h, 9.41, '1mom', n_e=3, n_t=5), path)
        L = lutmod.load_lut(path)
        assert list(L.value_table.shape) == list(g[h + '_shape'])
        kw = {'e': g[h + '_e'], ('wc' if h == 'mG' else 't'): g[h + '_t']}
        assert np.array_equal(L.lookup_line(**kw).sum(axis=1), g[h + '_sum_d'])
