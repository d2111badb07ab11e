"""Scattering lookup tables: container, file format, host-side staging helpers.

Mirrors the interface of cosmo_pol/lookup/lut.py (Lookup_table :162,
load_all_lut :27, save_lut :78, load_lut :127) so tables written by either
side load in the other: a .lut file is a tar of value_table / axes /
axes_names / axes_step / axes_limits .npy members.

Table layout (cosmo_pol/lookup/compute_lut_sz.py:60-69, 265-297):
  value_table[n_e, n_t | n_wc, n_d = 1024, 12] float64
  axes: 'e' (deg, 0..90 step 2), 't' (K) or 'wc' (wet fraction), 'd' (mm;
  [n_wc, 1024] for melting species), 'sz' (12 columns: Z11 Z12 Z21 Z22 Z33 Z34
  Z43 Z44 backscatter, Re/Im S11, Re/Im S22 forward).

The query used on the hot path (lookup_line :309-344: floor to the lower bin)
runs on the GPU; `Lookup_table.lookup_line` here is the host twin kept for API
compatibility and tests.
"""
import os
import shutil
import tarfile
import tempfile
from io import BytesIO

import numpy as np


class Error(Exception):
    """Lookup Table Error"""


class Lookup_table(object):
    def __init__(self):
        self.axes = []
        self.axes_names = {}
        self.axes_limits = []
        self.axes_step = []
        self.axes_len = []
        self.value_table = []

    def add_axis(self, name, axis_values=None):
        if name in self.axes_names:
            raise Error("Axis already exists with name: '%s'" % name)
        axis_values = np.asarray(axis_values).astype('float32')
        self.axes_names[name] = len(self.axes)
        self.axes_limits.append([np.min(axis_values), np.max(axis_values)])
        self.axes_step.append(axis_values[1] - axis_values[0])
        self.axes_len.append(axis_values.shape[-1])
        self.axes.append(axis_values)

    def set_axis_values(self, axis_name, axis_values):
        i = self.axes_names[axis_name]
        axis_values = np.asarray(axis_values).astype('float32')
        self.axes_limits[i] = [np.min(axis_values), np.max(axis_values)]
        self.axes[i] = axis_values

    def set_value_table(self, value_table):
        value_table = np.asarray(value_table)
        if value_table.shape != tuple(self.axes_len):
            raise ValueError('value_table shape %s does not match the axes %s'
                             % (value_table.shape, tuple(self.axes_len)))
        self.value_table = value_table

    def get_axis_name(self, axis_i):
        for name, i in self.axes_names.items():
            if i == axis_i:
                return name
        return None

    def bin_index(self, name, values):
        """Lower-bin index of `values` along axis `name`, clipped to the table
        (the integer the reference's lookup_line computes, lut.py:336-341)."""
        ax = self.axes_names[name]
        closest = np.floor((values - self.axes_limits[ax][0]) / self.axes_step[ax])
        closest = np.array(closest, dtype=int)
        closest[closest < 0] = 0
        closest[closest >= self.value_table.shape[ax]] = self.value_table.shape[ax] - 1
        return closest

    def lookup_line(self, **kwargs):
        I = [slice(None)] * self.value_table.ndim
        for k in kwargs:
            if k in self.axes_names:
                I[self.axes_names[k]] = self.bin_index(k, kwargs[k])
        return self.value_table[tuple(I)]


def _as_array(obj):
    try:
        return np.array(obj)
    except ValueError:
        arr = np.empty(len(obj), dtype=object)
        for i, o in enumerate(obj):
            arr[i] = o
        return arr


def save_lut(lut, filename):
    tmp_dir = tempfile.mkdtemp(prefix='cpol_lut_')
    try:
        np.save(os.path.join(tmp_dir, 'value_table'), lut.value_table)
        np.save(os.path.join(tmp_dir, 'axes'), _as_array(lut.axes), allow_pickle=True)
        np.save(os.path.join(tmp_dir, 'axes_names'), lut.axes_names, allow_pickle=True)
        np.save(os.path.join(tmp_dir, 'axes_step'), _as_array(lut.axes_step), allow_pickle=True)
        np.save(os.path.join(tmp_dir, 'axes_limits'), _as_array(lut.axes_limits), allow_pickle=True)
        with tarfile.open(filename, 'w') as tar:
            for n in sorted(os.listdir(tmp_dir)):
                tar.add(os.path.join(tmp_dir, n), arcname=n)
    finally:
        shutil.rmtree(tmp_dir)


def load_lut(filename):
    lut = Lookup_table()
    with tarfile.open(filename, 'r') as tar:
        for member in tar.getmembers():
            buf = BytesIO(tar.extractfile(member).read())
            name = member.name.replace('.npy', '')
            data = np.load(buf, allow_pickle=True, encoding='latin1')
            if name == 'axes_names':
                data = data.item()      # 0-d object array holding the dict
            setattr(lut, name, data)
    lut.axes_len = list(np.shape(lut.value_table))
    return lut


def lut_filename(hydrom, frequency, scheme):
    return 'lut_SZ_' + hydrom + '_' + str(frequency).replace('.', '_') + '_' + scheme + '.lut'


def load_all_lut(scheme, list_hydrom, frequency, scattering_method, lut_dir=None):
    """Same arguments as the reference's load_all_lut (lut.py:27-76) plus
    `lut_dir` (the reference hard-codes <package>/lookup/lut_<method>/)."""
    base = lut_dir if lut_dir is not None else os.path.join(
        os.path.dirname(os.path.realpath(__file__)), 'lookup')
    sub = {'tmatrix': 'lut_tmatrix', 'tmatrix_masc': 'lut_tmatrix_masc', 'dda': 'lut_dda'}
    folder = os.path.join(base, sub.get(scattering_method, 'lut_tmatrix_masc'))
    default = os.path.join(base, 'lut_tmatrix_masc')
    out = {}
    for h in list_hydrom:
        name = lut_filename(h, frequency, scheme)
        use = default if (scattering_method == 'dda' and h in ['R', 'H']) else folder
        path = os.path.join(use, name)
        if not os.path.exists(path) and os.path.exists(os.path.join(base, name)):
            path = os.path.join(base, name)
        if not os.path.exists(path):
            raise IOError('Could not find lookup table %s (scheme=%s, hydrometeor=%s, '
                          'frequency=%s, scattering=%s)' % (path, scheme, h, frequency,
                                                            scattering_method))
        out[h] = load_lut(path)
    return out
