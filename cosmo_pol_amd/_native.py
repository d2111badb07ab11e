"""ctypes binding of libcosmo_pol_hip.so (C ABI: include/cosmo_pol_amd.h).

There is NO CPU fallback: if the HIP library is missing or fails to load the
import raises, and every entry point raises on a non-zero status.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libcosmo_pol_hip.so')

CPOL_MAX_VARS = 24
CPOL_MAX_HYDRO = 8
CPOL_MAX_PAR = 6
N_SZ = 12

PSD_GAMMA, PSD_ICE_FIELD, PSD_MELTING = 0, 1, 2
(RULE_RAIN_1MOM, RULE_SNOW_1MOM, RULE_GRAUPEL_1MOM, RULE_TWO_MOMENT, RULE_ICE_1MOM,
 RULE_MELTING_SNOW, RULE_MELTING_GRAUPEL) = range(7)
Q_MODEL, Q_MELT_SNOW, Q_MELT_GRAUPEL = 0, 1, 2
GEOM_GROUND_43, GEOM_SPACEBORNE, GEOM_HOST_PATHS = 0, 1, 2
DEBUG_EXACT_SUBBEAMS = 1

ERR_HIP, ERR_ARG, ERR_DOMAIN, ERR_NOMEM = -1, -2, -3, -4


class HydroDesc(C.Structure):
    _fields_ = [
        ('psd_family', C.c_int32), ('rule', C.c_int32), ('q_source', C.c_int32),
        ('var_q', C.c_int32), ('var_qn', C.c_int32), ('var_t', C.c_int32),
        ('n_e', C.c_int32), ('n_t', C.c_int32), ('n_d', C.c_int32),
        ('second_axis_f64', C.c_int32),
        ('e_lo', C.c_float), ('e_step', C.c_float), ('t_lo', C.c_float), ('t_step', C.c_float),
        ('dD', C.c_double),
        ('a', C.c_double), ('b', C.c_double), ('alpha', C.c_double), ('beta', C.c_double),
        ('mu', C.c_double), ('nu', C.c_double),
        ('lambda_factor', C.c_double), ('ntot_factor', C.c_double), ('vel_factor', C.c_double),
        ('n0_fixed', C.c_double), ('x_min', C.c_double), ('x_max', C.c_double),
        ('c_n0', C.c_double), ('c_lam', C.c_double),
        ('lam_exponent', C.c_double), ('n0_exponent', C.c_double),
        ('r_a', C.c_double), ('r_b', C.c_double), ('r_alpha', C.c_double), ('r_beta', C.c_double),
        ('r_n0', C.c_double), ('r_mu', C.c_double), ('r_lambda_factor', C.c_double),
        ('r_lam_exponent', C.c_double),
        ('r_dmin', C.c_double), ('r_dmax', C.c_double), ('s_dmin', C.c_double),
        ('s_dmax', C.c_double),
        ('solid_rule', C.c_int32), ('uniform_grid', C.c_int32),
        ('numeric_intv', C.c_int32), ('tab_degree', C.c_int32),
        ('pad_', C.c_int32), ('table_id', C.c_uint64),
    ]


class SweepParams(C.Structure):
    _fields_ = [
        ('n_rays', C.c_int32), ('n_gates', C.c_int32), ('n_sub', C.c_int32),
        ('n_hnodes', C.c_int32), ('n_vnodes', C.c_int32),
        ('with_melting', C.c_int32), ('with_attenuation', C.c_int32),
        ('integrate_model', C.c_int32), ('apply_sensitivity', C.c_int32),
        ('outputs_on_device', C.c_int32), ('simulate_doppler', C.c_int32),
        ('geometry_mode', C.c_int32),
        ('radar_lat', C.c_double), ('radar_lon', C.c_double), ('radar_alt', C.c_double),
        ('range0', C.c_double), ('range_step', C.c_double),
        ('ke', C.c_double), ('re', C.c_double),
        ('sin_u1', C.c_double), ('cos_u1', C.c_double),
        ('wavelength', C.c_double), ('k_squared', C.c_double), ('radial_res', C.c_double),
        ('c_zh', C.c_double),
        ('var_u', C.c_int32), ('var_v', C.c_int32), ('var_w', C.c_int32), ('var_rho', C.c_int32),
        ('n_vbins', C.c_int32), ('debug_flags', C.c_int32), ('c_spectrum', C.c_double),
    ]


class RayTables(C.Structure):
    _fields_ = [('traj', C.c_void_p), ('geo', C.c_void_p), ('sub_h', C.c_void_p),
                ('sub_v', C.c_void_p), ('sub_w', C.c_void_p), ('sens_thr', C.c_void_p),
                ('site', C.c_void_p), ('paths', C.c_void_p), ('nyquist', C.c_void_p),
                ('version', C.c_uint64), ('varray', C.c_void_p), ('sub_smooth', C.c_void_p),
                ('ml_filter', C.c_void_p),
                ('ml_radius', C.c_int32), ('pad_', C.c_int32)]


OUTPUT_FIELDS = ['ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V',
                 'RVEL', 'mask', 'lats', 'lons', 'dist', 'heights', 'model_vars', 'sz_total',
                 'DSPECTRUM', 'mask_sum8']


class Outputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in OUTPUT_FIELDS]


class Counters(C.Structure):
    _fields_ = [('n_subbeam_gates', C.c_int64), ('n_valid_items', C.c_int64),
                ('n_gates', C.c_int64), ('n_work_units', C.c_int64),
                ('ms_traj', C.c_float), ('ms_interp', C.c_float), ('ms_classify', C.c_float),
                ('ms_bucket', C.c_float), ('ms_psd', C.c_float), ('ms_final', C.c_float),
                ('ms_total', C.c_float), ('n_table_items', C.c_int32), ('pad_', C.c_int32)]


EXPORTS = ['cpol_create', 'cpol_destroy', 'cpol_fork', 'cpol_last_error', 'cpol_set_stream',
           'cpol_get_stream',
           'cpol_synchronize', 'cpol_stage_model', 'cpol_stage_hydro', 'cpol_set_num_hydro',
           'cpol_stage_doppler_weights', 'cpol_stage_spectrum_tables', 'cpol_stage_t_function', 'cpol_prepare',
           'cpol_interp_points', 'cpol_ray_tables', 'cpol_run_sweep', 'cpol_counters',
           'cpol_spaceborne_first_gate', 'cpol_host_alloc', 'cpol_host_free', 'cpol_host_alloc_near',
           'cpol_device_pci_bus_id', 'cpol_mem_info',
           'cpol_enable_timing', 'cpol_debug_read', 'cpol_debug_math']

TRAJ_STRIDE, GEO_STRIDE, SITE_STRIDE = 4, 8, 8      # CPOL_*_STRIDE of the header
MELT_DEGREE, MELT_FUNCS = 10, 4                    # CPOL_MELT_DEGREE / CPOL_MELT_FUNCS
ICE_DEGREE, ICE_FUNCS = 10, 3                      # CPOL_ICE_DEGREE / CPOL_ICE_FUNCS
TFUN_SNOW_N0, TFUN_ICE_MOM2_A = 0, 1
TFUN_FIRST_BITS, TFUN_COUNT = 0x43000000, 1 << 24  # every float32 in [128, 512)

_lib = None
_torch_hip = None


class NativeError(RuntimeError):
    pass


def hip_runtimes_mapped():
    """Distinct libamdhip64 files mapped into this process (/proc/self/maps)."""
    found = []
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if 'libamdhip64' in os.path.basename(path):
                    real = os.path.realpath(path)
                    if real not in found:
                        found.append(real)
    except OSError:
        pass
    return found


def _share_torch_hip_runtime():
    """One HIP runtime per process.  libcosmo_pol_hip.so needs `libamdhip64.so.7`; a PyTorch-ROCm
    wheel bundles its own copy of that runtime (same SONAME) and asks for it as `libamdhip64.so`.
    If this library comes first the loader serves it from /opt/rocm and torch later maps its
    bundled copy as well: two runtimes, and the second one to initialise finds no GPU
    ("No HIP GPUs are available" in torch._C._cuda_init).  So when a torch wheel with a bundled
    runtime is installed -- found without importing torch -- that copy is mapped first; this
    library then binds to it by SONAME and a later `import torch` finds it already loaded.
    CPOL_HIP_RUNTIME=system skips this (a process that will never touch torch.cuda)."""
    import sys
    if os.environ.get('CPOL_HIP_RUNTIME', '') == 'system' or hip_runtimes_mapped():
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if not os.path.exists(cand):
        return None
    try:
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError as e:
        print('cosmo_pol_amd: could not preload %s (%s); using the system HIP runtime' % (cand, e),
              file=sys.stderr)
        return None


def load_library():
    """Loads the HIP shared library (never falls back to anything else)."""
    global _lib, _torch_hip
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            'cosmo_pol_amd: HIP extension %s is missing. Build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or `make -C cosmo_pol_amd/csrc`. '
            'There is no CPU fallback.' % LIB_PATH)
    _torch_hip = _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    if len(hip_runtimes_mapped()) > 1:
        raise NativeError(
            'cosmo_pol_amd: two HIP runtimes are mapped into this process (%s): the second one to '
            'initialise will not see the GPU.  Import cosmo_pol_amd (or torch) before any other '
            'package that loads libamdhip64, or unset CPOL_HIP_RUNTIME.' % ', '.join(hip_runtimes_mapped()))
    vp = C.c_void_p
    lib.cpol_create.restype = C.c_int
    lib.cpol_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.cpol_get_stream.restype = C.c_int
    lib.cpol_get_stream.argtypes = [vp, C.POINTER(vp)]
    lib.cpol_fork.restype = C.c_int
    lib.cpol_fork.argtypes = [vp, C.POINTER(vp)]
    lib.cpol_destroy.restype = None
    lib.cpol_destroy.argtypes = [vp]
    lib.cpol_last_error.restype = C.c_char_p
    lib.cpol_last_error.argtypes = [vp]
    lib.cpol_set_stream.restype = C.c_int
    lib.cpol_set_stream.argtypes = [vp, vp]
    lib.cpol_synchronize.restype = C.c_int
    lib.cpol_synchronize.argtypes = [vp]
    lib.cpol_stage_model.restype = C.c_int
    lib.cpol_stage_model.argtypes = [vp, C.c_int, C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int,
                                     vp, vp, vp, vp]
    lib.cpol_stage_hydro.restype = C.c_int
    lib.cpol_stage_hydro.argtypes = [vp, C.c_int, C.POINTER(HydroDesc), vp, vp, vp, vp, C.c_int]
    lib.cpol_stage_doppler_weights.restype = C.c_int
    lib.cpol_stage_doppler_weights.argtypes = [vp, C.c_int, vp]
    lib.cpol_set_num_hydro.restype = C.c_int
    lib.cpol_set_num_hydro.argtypes = [vp, C.c_int]
    lib.cpol_interp_points.restype = C.c_int
    lib.cpol_interp_points.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.cpol_ray_tables.restype = C.c_int
    lib.cpol_ray_tables.argtypes = [C.POINTER(SweepParams), vp, vp, vp, vp, vp, vp]
    lib.cpol_run_sweep.restype = C.c_int
    lib.cpol_run_sweep.argtypes = [vp, C.POINTER(SweepParams), C.POINTER(RayTables),
                                   C.POINTER(Outputs)]
    lib.cpol_counters.restype = C.c_int
    lib.cpol_counters.argtypes = [vp, C.POINTER(Counters)]
    lib.cpol_spaceborne_first_gate.restype = C.c_int
    lib.cpol_spaceborne_first_gate.argtypes = [vp, C.POINTER(SweepParams), vp, vp, vp, C.c_double, vp]
    lib.cpol_enable_timing.restype = C.c_int
    lib.cpol_enable_timing.argtypes = [vp, C.c_int]
    lib.cpol_debug_read.restype = C.c_int64
    lib.cpol_debug_read.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.cpol_stage_spectrum_tables.restype = C.c_int
    lib.cpol_stage_spectrum_tables.argtypes = [vp, C.c_int, vp, vp]
    lib.cpol_debug_math.restype = C.c_int
    lib.cpol_debug_math.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    lib.cpol_prepare.restype = C.c_int
    lib.cpol_prepare.argtypes = [vp]
    lib.cpol_stage_t_function.restype = C.c_int
    lib.cpol_stage_t_function.argtypes = [vp, C.c_int, vp]
    lib.cpol_host_alloc.restype = C.c_int
    lib.cpol_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]      # (ctx may be NULL: context-free block)
    lib.cpol_host_free.restype = C.c_int
    lib.cpol_host_free.argtypes = [vp, vp]
    lib.cpol_host_alloc_near.restype = C.c_int
    lib.cpol_host_alloc_near.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp)]
    lib.cpol_mem_info.restype = C.c_int
    lib.cpol_mem_info.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.cpol_device_pci_bus_id.restype = C.c_int
    lib.cpol_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    _lib = lib
    return lib


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> set of ints."""
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_numa_info(device=0, sysfs='/sys/bus/pci/devices', pci=None):
    """{'pci': '0000:75:00.0', 'node': 1, 'cpus': {...}} of GPU `device` (HIP ordinal); node / cpus are
    None / empty where the kernel does not tell (one-socket hosts report node -1).  `pci`: the bus id
    when it is known already (tests)."""
    if pci is None:
        lib = load_library()
        buf = C.create_string_buffer(64)
        if lib.cpol_device_pci_bus_id(int(device), buf, 64) != 0:
            raise NativeError('cpol_device_pci_bus_id(%d) failed: no such GPU' % device)
        pci = buf.value.decode()
    bdf = pci.lower()
    info = {'pci': bdf, 'node': None, 'cpus': set()}
    try:
        with open(os.path.join(sysfs, bdf, 'numa_node')) as f:
            node = int(f.read().strip())
        info['node'] = node if node >= 0 else None
        with open(os.path.join(sysfs, bdf, 'local_cpulist')) as f:
            info['cpus'] = parse_cpulist(f.read())
    except (OSError, ValueError):
        pass
    return info


def bind_to_device_numa_node(device=0, sysfs='/sys/bus/pci/devices', pci=None):
    """One process per GPU on a multi-socket host: restricts the calling thread (and every thread it
    starts afterwards) to the cores next to GPU `device`, so that the threads which build ray tables,
    fill upload buffers and read results run on the socket whose memory the GPU's copies land in
    (PinnedPool takes its blocks there, cpol_host_alloc_near).  Call it before the operator creates its
    lane threads.  Never widens the affinity the process was given; does nothing when the GPU's cores
    and the allowed cores do not intersect or the kernel reports no node.  CPOL_NUMA_BIND=0 switches
    it off.  -> the info dict of device_numa_info plus 'bound' (number of cores, 0 = unchanged)."""
    info = device_numa_info(device, sysfs, pci)
    info['bound'] = 0
    if os.environ.get('CPOL_NUMA_BIND', '1') == '0' or not info['cpus'] or not hasattr(os, 'sched_setaffinity'):
        return info
    allowed = os.sched_getaffinity(0)
    want = allowed & info['cpus']
    if want and want != allowed:
        os.sched_setaffinity(0, want)
        info['bound'] = len(want)
    return info


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class PinnedPool(object):
    """Page-locked host blocks for results handed to the user.

    `take(nbytes)` returns a uint8 ndarray over a block of context-free pinned memory
    (cpol_host_alloc(NULL, ...)).  The block belongs to that array and to every view carved from it:
    when the last of them is garbage-collected the block goes back to the pool and is re-used by a
    later `take` of the same size class -- so results stay valid for as long as anybody holds them,
    whatever happens to the operator, its lanes or its contexts in the meantime, and a steady stream
    of equal sweeps allocates nothing.  A block that was the target of a non-blocking device-to-host
    copy remembers the context that issued it (`pending`); re-use waits for that context first."""
    GRANULE = 1 << 20
    MAX_PENDING = 8             # blocks of one size class that may wait for foreign copies before the pool waits

    def __init__(self, device=None):
        import collections
        import threading
        self.lib = load_library()
        self.device = device        # blocks come from the NUMA node next to this GPU (None: the current device's)
        self.free = {}              # size class -> [(address, context or None, serial)]
        self.returned = collections.deque()     # blocks handed back by finalizers, not yet sorted into `free`
        self.lock = threading.Lock()
        self.closed = False
        self.n_alloc = 0

    def _release(self, addr, size, holder):
        # Runs as a weakref finalizer: on ANY thread, possibly inside a garbage collection triggered by an
        # allocation made while that thread holds self.lock (a result kept alive only by a reference cycle).
        # So it never takes the lock: deque.append / popleft are atomic, take() sorts the blocks in.
        self.returned.append((addr, size, holder.get('ctx'), holder.get('serial', 0)))
        if self.closed:
            self._free_returned()

    def _free_returned(self):
        while True:
            try:
                addr = self.returned.popleft()[0]
            except IndexError:
                return
            self.lib.cpol_host_free(None, C.c_void_p(addr))

    @staticmethod
    def _clean(ctx, serial):
        """No copy into the block can still be in flight."""
        return ctx is None or not getattr(ctx, 'h', None) or ctx.completed >= serial

    def take(self, nbytes, writer=None):
        """-> (uint8 array over a block of >= nbytes, holder).  `writer`: the context whose stream
        will copy into the block.  A block whose last copy (queued by context C, not yet waited for)
        may still be in flight is handed out again without waiting only to C itself -- copies on one
        stream stay in order, and nobody holds the old arrays any more; for another writer the pool
        prefers a clean block, then a new one (up to MAX_PENDING blocks in flight), then waits for C.
        After queueing a non-blocking copy set holder['ctx'], holder['serial'] = context, its
        `submitted` count."""
        import weakref
        size = max(1, -(-int(nbytes) // self.GRANULE)) * self.GRANULE
        addr = wait_for = None
        with self.lock:
            while True:                                                   # blocks handed back since the last call
                try:
                    a, sz, c, ser = self.returned.popleft()
                except IndexError:
                    break
                self.free.setdefault(sz, []).append((a, c, ser))
            lst = self.free.get(size) or []
            pick = None
            for i, (a, c, ser) in enumerate(lst):                    # oldest first
                if c is writer or self._clean(c, ser):
                    pick = i
                    break
            if pick is None and lst and len(lst) >= self.MAX_PENDING:
                pick = 0
                wait_for = lst[0][1]
            if pick is not None:
                addr = lst.pop(pick)[0]
        if wait_for is not None:
            wait_for.synchronize()  # (also surfaces a deferred domain error of that lane, once)
        if addr is None:
            h = C.c_void_p()
            if self.device is None:
                rc = self.lib.cpol_host_alloc(None, size, C.byref(h))
            else:
                rc = self.lib.cpol_host_alloc_near(int(self.device), size, C.byref(h))
            if rc != 0 or not h:
                raise MemoryError('cpol_host_alloc(%d bytes of page-locked host memory) failed' % size)
            addr = h.value
            self.n_alloc += 1
        buf = (C.c_uint8 * size).from_address(addr)
        arr = np.frombuffer(buf, dtype=np.uint8)
        holder = {'ctx': None, 'serial': 0}
        weakref.finalize(arr, self._release, addr, size, holder)
        return arr, holder

    def close(self):
        """Frees the blocks nobody holds; blocks still held are freed when their last view dies."""
        with self.lock:
            self.closed = True
            blocks = [e[0] for lst in self.free.values() for e in lst]
            self.free = {}
        for a in blocks:
            self.lib.cpol_host_free(None, C.c_void_p(a))
        self._free_returned()


class Context(object):
    """One device context (one per GPU per process)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.cpol_create(int(device), C.byref(h))
        if rc != 0 or not h:
            raise NativeError('cpol_create(device=%d) failed with status %d (no usable HIP '
                              'device?)' % (device, rc))
        self.h = h
        self.device = device
        self.n_vars = 0
        self._keep = []
        self.submitted = self.completed = 0     # sweeps queued / known to have completed (synchronize)

    def fork(self):
        """A lane: shares this context's staged model / tables, own stream and work
        buffers (cpol_fork).  Close lanes before their parent."""
        h = C.c_void_p()
        self._check(self.lib.cpol_fork(self.h, C.byref(h)), 'cpol_fork')
        lane = Context.__new__(Context)
        lane.lib, lane.h, lane.device, lane.n_vars = self.lib, h, self.device, self.n_vars
        lane._keep = []
        lane.submitted = lane.completed = 0
        lane._parent = self          # keeps the parent alive
        return lane

    def close(self):
        if getattr(self, 'h', None):
            self.lib.cpol_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == 0:
            return
        msg = self.lib.cpol_last_error(self.h)
        msg = msg.decode() if msg else ''
        if rc == ERR_DOMAIN:
            raise IndexError('ERROR: ' + msg + ': ABORTING')
        if rc == ERR_ARG:
            raise ValueError('%s: %s' % (what, msg))
        if rc == ERR_NOMEM:
            raise MemoryError('%s: %s' % (what, msg))
        raise NativeError('%s failed (%d): %s' % (what, rc, msg))

    def set_stream(self, stream_ptr):
        self._check(self.lib.cpol_set_stream(self.h, C.c_void_p(stream_ptr)), 'cpol_set_stream')

    def synchronize(self):
        """Waits for the context's stream; raises IndexError if a sweep since the last
        report left the model domain (deferred error of device / pinned-host outputs)."""
        n = self.submitted
        rc = self.lib.cpol_synchronize(self.h)
        self.completed = max(self.completed, n)          # (the stream has drained, error or not)
        self._check(rc, 'cpol_synchronize')

    def host_alloc(self, nbytes):
        """uint8 array over page-locked host memory owned by the context (the target of
        outputs_on_device = 2); valid until the context is closed."""
        h = C.c_void_p()
        self._check(self.lib.cpol_host_alloc(self.h, int(nbytes), C.byref(h)), 'cpol_host_alloc')
        buf = (C.c_uint8 * int(nbytes)).from_address(h.value)
        arr = np.frombuffer(buf, dtype=np.uint8)
        self._keep.append(buf)
        return arr

    def mem_info(self):
        """(free bytes, total bytes) of the GPU's memory, work-buffer bytes per sub-beam gate of a launch sequence."""
        f, t, g = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._check(self.lib.cpol_mem_info(self.h, C.byref(f), C.byref(t), C.byref(g)), 'cpol_mem_info')
        return int(f.value), int(t.value), int(g.value)

    def enable_timing(self, on=True):
        """True / 1: events around every stage; 2: around the PSD stage only; False: off."""
        self._check(self.lib.cpol_enable_timing(self.h, 2 if on == 2 else int(bool(on))),
                    'cpol_enable_timing')

    def stream_ptr(self):
        h = C.c_void_p()
        self._check(self.lib.cpol_get_stream(self.h, C.byref(h)), 'cpol_get_stream')
        return h.value or 0

    def enable_debug(self, on=True):
        self.lib.cpol_debug_read(self.h, b'enable' if on else b'disable', None, 0)

    def stage_model(self, arrays, zlevels, llc, urc, res, south_pole):
        arrays = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
        zlevels = np.ascontiguousarray(zlevels, dtype=np.float32)
        nz, ny, nx = zlevels.shape
        for a in arrays:
            if a.shape != (nz, ny, nx):
                raise ValueError('model variable shape %s != z-levels shape %s'
                                 % (a.shape, zlevels.shape))
        ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        llc = np.ascontiguousarray(llc, dtype=np.float32)
        urc = np.ascontiguousarray(urc, dtype=np.float32)
        res = np.ascontiguousarray(res, dtype=np.float32)
        sp = np.ascontiguousarray(south_pole, dtype=np.float64)
        rc = self.lib.cpol_stage_model(self.h, len(arrays), ptrs, _ptr(zlevels), nz, ny, nx,
                                       _ptr(llc), _ptr(urc), _ptr(res), _ptr(sp))
        self._check(rc, 'cpol_stage_model')
        self.n_vars = len(arrays)

    def stage_hydro(self, slot, desc, table, pre=None, dnu=None, aux=None):
        table = np.ascontiguousarray(table, dtype=np.float64)
        f64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        pre, dnu, aux = f64(pre), f64(dnu), f64(aux)
        rc = self.lib.cpol_stage_hydro(self.h, slot, C.byref(desc), _ptr(table), _ptr(pre),
                                       _ptr(dnu), _ptr(aux), 0 if aux is None else aux.size)
        self._check(rc, 'cpol_stage_hydro')

    def stage_doppler_weights(self, slot, weights):
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self._check(self.lib.cpol_stage_doppler_weights(self.h, slot, _ptr(w)),
                    'cpol_stage_doppler_weights')

    def stage_spectrum_tables(self, slot, rcs32, dgrid):
        rcs32 = np.ascontiguousarray(rcs32, dtype=np.float32)
        dgrid = np.ascontiguousarray(dgrid, dtype=np.float32)
        self._check(self.lib.cpol_stage_spectrum_tables(self.h, int(slot), _ptr(rcs32), _ptr(dgrid)),
                    'cpol_stage_spectrum_tables')

    def stage_t_function(self, which, table):
        table = np.ascontiguousarray(table, dtype=np.float32)
        if table.size != TFUN_COUNT:
            raise ValueError('t-function table must have %d entries' % TFUN_COUNT)
        self._check(self.lib.cpol_stage_t_function(self.h, int(which), _ptr(table)),
                    'cpol_stage_t_function')

    def prepare(self):
        """Builds the integral tables now (otherwise: at the first sweep / fork)."""
        self._check(self.lib.cpol_prepare(self.h), 'cpol_prepare')

    def set_num_hydro(self, n):
        self._check(self.lib.cpol_set_num_hydro(self.h, n), 'cpol_set_num_hydro')

    def interp_points(self, coords, heights):
        coords = np.ascontiguousarray(coords, dtype=np.float32)
        heights = np.ascontiguousarray(heights, dtype=np.float32)
        n = heights.shape[0]
        out = np.empty((self.n_vars, n), dtype=np.float32)
        rc = self.lib.cpol_interp_points(self.h, n, _ptr(coords), _ptr(heights), _ptr(out))
        self._check(rc, 'cpol_interp_points')
        return out

    def run_sweep(self, params, tables, outputs):
        self.submitted += 1
        rc = self.lib.cpol_run_sweep(self.h, C.byref(params), C.byref(tables), C.byref(outputs))
        self._check(rc, 'cpol_run_sweep')

    def spaceborne_first_gate(self, params, traj, site, n_cand, ceiling_m):
        n = params.n_rays * params.n_vnodes
        out = np.empty(n, dtype=np.int32)
        traj = np.ascontiguousarray(traj, dtype=np.float64)
        site = np.ascontiguousarray(site, dtype=np.float64)
        n_cand = np.ascontiguousarray(n_cand, dtype=np.int32)
        rc = self.lib.cpol_spaceborne_first_gate(self.h, C.byref(params), _ptr(traj), _ptr(site),
                                                 _ptr(n_cand), float(ceiling_m), _ptr(out))
        self._check(rc, 'cpol_spaceborne_first_gate')
        return out.reshape(params.n_rays, params.n_vnodes)

    def counters(self):
        c = Counters()
        self._check(self.lib.cpol_counters(self.h, C.byref(c)), 'cpol_counters')
        return c

    def itab_report(self):
        """Accuracy gate of the integral tables of the staged slots (cpol_prepare): per slot the worst
        deviation of a block's polynomial from the integrating kernel at the block's check point
        (`check`: negative = table rejected, the slot is integrated bin by bin; 0 = no table), where it
        was found (`at`), the number of (block, function) pairs above the limit (`n_bad`, 1-D tables),
        the worst deviation at the second check point alone -- u = 0.96, between the last two nodes of a
        lambda panel (`check_edge`, 1-D tables; `check` is the maximum over both points) -- and the device
        time of the build and of the check alone (`build_ms`, `check_ms`)."""
        a = self.debug_read('itab_check', (4, CPOL_MAX_HYDRO), np.float64)
        t = self.debug_read('itab_times', (CPOL_MAX_HYDRO, 2), np.float64)
        return {'check': a[0], 'at': a[1], 'n_bad': a[2], 'check_edge': a[3], 'build_ms': t[:, 0], 'check_ms': t[:, 1]}

    def itab_detail(self, slot):
        """1-D integral table of `slot`: {'log2_lo', 'ppo', 'd0', 'n_pan', 'by_fn' [15], 'by_pan'
        [n_pan], 'by_pan_edge' [n_pan], 'accepted_panels'} -- the worst deviation at the check points per function and per
        lambda panel (`by_pan_edge`: at the point near the panel edge alone) (panel p covers lambda in 2^(log2_lo + [p, p + 1] / ppo)) and the run of panels
        [lo, hi) that passed the gate (None: table rejected); None for a slot without such a table."""
        n = self.lib.cpol_debug_read(self.h, ('itab_detail%d' % slot).encode(), None, 0)
        if n == 0:
            return None
        nb = -int(n) - 1000
        if nb <= 0:
            self._check(int(n), 'cpol_debug_read(itab_detail)')
        v = self.debug_read('itab_detail%d' % slot, (nb // 8,), np.float64)
        n_pan = int(v[3])
        acc = v[19 + 2 * n_pan:19 + 2 * n_pan + 2]
        return {'log2_lo': v[0], 'ppo': int(v[1]), 'd0': v[2], 'n_pan': n_pan, 'by_fn': v[4:19],
                'by_pan': v[19:19 + n_pan], 'by_pan_edge': v[19 + n_pan:19 + 2 * n_pan],
                'accepted_panels': (int(acc[0]), int(acc[1])) if len(acc) == 2 else None}

    def debug_math(self, op, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._check(self.lib.cpol_debug_math(self.h, int(op), _ptr(x), _ptr(y), x.size),
                    'cpol_debug_math')
        return y

    FORM_NAMES = ('g1r', 'gate1_ray', 'gate1', 'interp_classify', 'rare_direct', 'subbeam_sum', 'final_inplace',
                  'poly_central', 'n_sub', 'lanes_alive', 'scan_form', 'graph_replayed')

    def launch_forms(self):
        """Which launch sequence the last cpol_run_sweep of this context took (tests / bench result checks): {name: int}."""
        v = self.debug_read('launch_forms', (len(self.FORM_NAMES),), np.int32)
        return dict(zip(self.FORM_NAMES, (int(x) for x in v)))

    def debug_read(self, name, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        n = self.lib.cpol_debug_read(self.h, name.encode(), _ptr(out), out.nbytes)
        if n < 0:
            self._check(int(n), 'cpol_debug_read(%s)' % name)
        if n != out.nbytes:
            raise NativeError('cpol_debug_read(%s): got %d bytes, expected %d'
                              % (name, n, out.nbytes))
        return out
