"""Oracle scattering lookup tables (TEST INFRASTRUCTURE ONLY).

Restates cosmo_pol/lookup/lut.py: container (:162-192), the tar-of-.npy file
format (:78-154) and the floor-bin slice query lookup_line (:309-344, quirk
Q10: floor to the lower bin, no interpolation).  Axis construction follows
add_axis (:193-213): float32 axes, limits = [min, max], step = axis[1]-axis[0].
Layout spec: cosmo_pol/lookup/compute_lut_sz.py:60-69,265-297,345-364,414-433.
"""
import os
import tarfile
import tempfile
from io import BytesIO

import numpy as np


class LookupTable(object):
    def __init__(self):
        self.axes = []
        self.axes_names = {}
        self.axes_limits = []
        self.axes_step = []
        self.value_table = []

    def add_axis(self, name, values):
        values = np.asarray(values).astype('float32')
        self.axes_names[name] = len(self.axes)
        self.axes_limits.append([np.min(values), np.max(values)])
        self.axes_step.append(values[1] - values[0])
        self.axes.append(values)

    def lookup_line(self, **kwargs):
        v = self.value_table
        dim = v.shape
        I = [slice(None)] * v.ndim
        for k in kwargs.keys():
            if k in self.axes_names.keys():
                ax = self.axes_names[k]
                closest = np.floor((kwargs[k] - self.axes_limits[ax][0]) / self.axes_step[ax])
                closest = np.array(closest, dtype=int)
                closest[closest < 0] = 0
                closest[closest >= dim[ax]] = dim[ax] - 1
                I[ax] = closest
        return v[tuple(I)]

    def bin_index(self, name, values):
        """The integer bin lookup_line would use (for index-parity tests)."""
        ax = self.axes_names[name]
        closest = np.floor((values - self.axes_limits[ax][0]) / self.axes_step[ax])
        closest = np.array(closest, dtype=int)
        closest[closest < 0] = 0
        closest[closest >= self.value_table.shape[ax]] = self.value_table.shape[ax] - 1
        return closest


def save_lut(lut, filename):
    """tar of value_table/axes/axes_names/axes_step/axes_limits .npy members;
    ragged members (melting tables) fall back to pickle inside a .npy name,
    as lut.py:104-114 does."""
    tmp_dir = tempfile.mkdtemp()
    try:
        np.save(os.path.join(tmp_dir, 'value_table'), lut.value_table)
        for name in ('axes', 'axes_step', 'axes_limits'):
            obj = getattr(lut, name)
            try:
                arr = np.array(obj)
            except ValueError:          # ragged (NumPy >= 1.24 refuses implicitly)
                arr = np.empty(len(obj), dtype=object)
                for i, o in enumerate(obj):
                    arr[i] = o
            np.save(os.path.join(tmp_dir, name), arr, allow_pickle=True)
        np.save(os.path.join(tmp_dir, 'axes_names'), lut.axes_names)
        with tarfile.open(filename, 'w') as tar:
            for n in sorted(os.listdir(tmp_dir)):
                tar.add(os.path.join(tmp_dir, n), arcname=n)
    finally:
        import shutil
        shutil.rmtree(tmp_dir)


def load_lut(filename):
    lut = LookupTable()
    with tarfile.open(filename, 'r') as tar:
        for member in tar.getmembers():
            buf = BytesIO(tar.extractfile(member).read())
            name = member.name.replace('.npy', '')
            data = np.load(buf, allow_pickle=True, encoding='latin1')
            if name == 'axes_names':
                data = data.item()      # 0-d object array holding the dict
            setattr(lut, name, data)
    return lut
