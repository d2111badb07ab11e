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
import tarfile
from io import BytesIO

import numpy as np


class Error(Exception):
    """Lookup Table Error"""


class Lookup_table(object):
    """A scattering table as the `.lut` format stores it.  The five attribute names below ARE the format (each is one
    member of the tar file, read back by the reference's `load_lut`, cosmo_pol/lookup/lut.py:127-154) and are what the
    reference's query code reads (`lookup_line`, :309-344); everything else about this class is this repo's own: a table
    is built in one go from its axes and values (`from_axes`) and is not mutated afterwards."""

    __slots__ = ('axes', 'axes_names', 'axes_limits', 'axes_step', 'axes_len', 'value_table')

    def __init__(self):
        self.axes, self.axes_names, self.axes_limits, self.axes_step, self.axes_len = [], {}, [], [], []
        self.value_table = []

    @classmethod
    def from_axes(cls, axes, value_table):
        """axes: [(name, values), ...] in table order (values float32 on file; the last dimension of a 2-D axis -- the
        diameters of a melting species, one row per wet fraction -- is the table's); value_table: array of that shape."""
        t = cls()
        for position, (name, values) in enumerate(axes):
            if name in t.axes_names:
                raise Error("Axis already exists with name: '%s'" % name)
            values = np.asarray(values).astype('float32')
            t.axes_names[name] = position
            t.axes.append(values)
            t.axes_limits.append([np.min(values), np.max(values)])
            t.axes_step.append(values[1] - values[0])
            t.axes_len.append(values.shape[-1])
        value_table = np.asarray(value_table)
        if value_table.shape != tuple(t.axes_len):
            raise ValueError('value_table shape %s does not match the axes %s' % (value_table.shape, tuple(t.axes_len)))
        t.value_table = value_table
        return t

    def axis(self, name):
        return self.axes[self.axes_names[name]]

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


# the members of a .lut tar file, in the order they are written; `ragged`: a list of arrays of different shapes (the axes: one
# of them is 2-D for a melting species) goes into ONE object array, as np.save needs it
_MEMBERS = (('axes', True), ('axes_limits', True), ('axes_names', False), ('axes_step', True), ('value_table', False))


def _object_array(items):
    try:
        return np.array(items)
    except ValueError:
        out = np.empty(len(items), dtype=object)
        out[:] = list(items)
        return out


def save_lut(lut, filename):
    """Writes `lut` as the reference's `save_lut` does (lut.py:78-125: a tar of five .npy members), straight from memory."""
    with tarfile.open(filename, 'w') as tar:
        for name, ragged in _MEMBERS:
            payload = getattr(lut, name)
            buf = BytesIO()
            np.save(buf, _object_array(payload) if ragged else payload, allow_pickle=name != 'value_table')
            info = tarfile.TarInfo(name + '.npy')
            info.size = buf.tell()
            buf.seek(0)
            tar.addfile(info, buf)


def load_lut(filename):
    """Reads a .lut file written by either side."""
    lut = Lookup_table()
    wanted = {name for name, _ in _MEMBERS}
    with tarfile.open(filename, 'r') as tar:
        for member in tar.getmembers():
            name = os.path.splitext(os.path.basename(member.name))[0]
            if name not in wanted:
                continue
            data = np.load(BytesIO(tar.extractfile(member).read()), allow_pickle=True, encoding='latin1')
            setattr(lut, name, data.item() if name == 'axes_names' else data)     # (the dict sits in a 0-d object array)
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
