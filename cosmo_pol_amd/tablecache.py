"""Persistent cache of the host-side staging tables.

The reference's only persistence is the scattering-table file itself (`save_lut` / `load_lut`,
cosmo_pol/lookup/lut.py:78-154).  This build derives more tables from it at staging time, two kinds
of them on the host and slow: the extended-precision polynomial tables of the melting species (2 s
each) and of 1-moment ice (3 s).  They depend only on the microphysics constants and the axes of the
scattering table, so they are kept as `.npz` files under

    $CPOL_CACHE_DIR,  else  <lut_dir>/.cpol_cache  (set_default_dir),  else  <tmp>/cosmo_pol_amd_cache_<uid>

(created with mode 0700; a directory that belongs to another user or that group / others may write to is
not used: the tables are then rebuilt every time) and named by a digest of EVERYTHING the table depends on
(inputs, builder source code, NumPy version, the machine type and the precision of np.longdouble, in which the
builders compute);
the same digest is stored inside the file and checked when it
is read, a `verify` callback may recompute a sample, and any mismatch, short file or load error
makes the entry stale: it is rebuilt and replaced.  CPOL_CACHE=0 switches the cache off.
(NOT cached: the integral tables -- the GPU rebuilds the 4 GB of them in 0.2 s, faster than a disk
delivers them -- and the float32 functions of T over all float32 values in [128, 512) K: 0.3 s to
evaluate, as long as reading 128 MB back and checking it against this host's NumPy.)"""
import hashlib
import os
import tempfile

import numpy as np

_default_dir = [None]
stats = {'hit': 0, 'miss': 0, 'stale': 0}


def set_default_dir(path):
    _default_dir[0] = path


def cache_dir():
    d = os.environ.get('CPOL_CACHE_DIR') or _default_dir[0]
    if not d:
        d = os.path.join(tempfile.gettempdir(), 'cosmo_pol_amd_cache_%d' % os.getuid())
    return d


def enabled():
    return os.environ.get('CPOL_CACHE', '1') != '0'


def usable_dir(create=True):
    """The cache directory if it is safe to use, else None: it must belong to this user and be writable by
    nobody else (another local user could otherwise plant entries: the stored digest is computable from
    public inputs).  Created with mode 0700 when missing."""
    d = cache_dir()
    try:
        if create:
            os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.stat(d)
    except OSError:
        return None
    if hasattr(os, 'getuid') and st.st_uid != os.getuid():
        return None
    if st.st_mode & 0o022:
        return None
    return d


def platform_parts():
    """What the bits of a host-built table depend on besides its inputs: the machine type and the
    extended-precision format the builders compute in (x86 80-bit vs aarch64 128-bit long double)."""
    import platform
    fi = np.finfo(np.longdouble)
    return ('platform', platform.machine(), int(fi.bits), float(fi.eps))


def digest_of(parts):
    h = hashlib.blake2b(digest_size=20)
    for p in parts:
        if isinstance(p, np.ndarray):
            h.update(str((p.dtype.str, p.shape)).encode())
            h.update(np.ascontiguousarray(p).tobytes())
        elif isinstance(p, (bytes, bytearray)):
            h.update(bytes(p))
        else:
            h.update(repr(p).encode())
        h.update(b'|')
    return h.hexdigest()


def source_of(*functions):
    """Source text of the builder functions: a code change invalidates their cache entries."""
    import inspect
    return '\n'.join(inspect.getsource(f) for f in functions)


def memo(name, parts, compute, verify=None):
    """compute() -> float ndarray or None, cached on disk under a digest of `parts`.
    File layout: a float64 header [magic, 20 digest bytes as numbers, is_none] then the array."""
    if not enabled():
        return compute()
    d = usable_dir()
    if d is None:
        return compute()
    dg = digest_of(list(parts) + [platform_parts()])
    path = os.path.join(d, '%s-%s.npz' % (name, dg[:24]))
    if os.path.exists(path):
        try:
            with np.load(path, allow_pickle=False) as z:
                ok = str(z['digest']) == dg
                value = None if bool(z['is_none']) else z['value']
            if ok and (value is None or verify is None or verify(value)):
                stats['hit'] += 1
                return value
        except Exception:
            pass
        stats['stale'] += 1
    else:
        stats['miss'] += 1
    value = compute()
    try:
        tmp = '%s.%d.tmp.npz' % (path, os.getpid())
        np.savez(tmp, digest=np.array(dg), is_none=np.array(value is None),
                 value=np.zeros(0) if value is None else value)
        os.replace(tmp, path)
    except OSError:
        pass                        # a read-only location: the table is simply rebuilt next time
    return value
