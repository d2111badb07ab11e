"""RadarOperator: the reference's user-facing API over the HIP hot path.

Same constructor, properties and scan methods as cosmo_pol/radar_operator.py:46
(`RadarOperator(options_file, output_variables)`, `.config` getter/setter,
`set_lut`, `load_model_file`, `get_PPI`, `get_RHI`, `get_VPROF`,
`get_GPM_swath`, `get_pos_and_time`, `close`).  The body of the reference's scan
loop (one `pool.map` task per radial, radar_operator.py:407-432) is replaced by
ONE call into libcosmo_pol_hip.so per sweep; there is no CPU path.

Keyword-only extras: `device`, `lut_dir`, `luts` (pre-built tables {h: table} or a
callable (hydrometeors, frequency, scheme) -> {h: table}),
`load_model_arrays(...)` (pycosmo / GRIB are not available here).
"""
import copy
from collections.abc import MutableMapping
import ctypes as C
import math
import os
import threading

import numpy as np

from . import _native as N
from . import config as cfg
from . import constants as K
from . import geometry as geo
from . import hydrometeors as hyd
from . import quadrature
from .lut import load_all_lut

RADAR_FIELDS = ['ZH', 'ZDR', 'ZV', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V']
DOPPLER_FIELDS = ['RVEL', 'DSPECTRUM']
_DB_FIELDS = ('ZDR', 'ZV', 'ZH')
_ITEMSIZE = {np.float32: 4, np.float64: 8, np.int8: 1}


class ModelVar(object):
    """Duck type of a pycosmo variable (what interpolation.py:547-561 reads)."""

    def __init__(self, name, data, attributes):
        self.name = name
        self.data = data
        self.attributes = attributes


class LazyDict(MutableMapping):
    """A mapping whose values are built on first access (`add(key, make)`): the containers returned by
    get_PPI / get_RHI / get_GPM_swath carry a dozen derived arrays per scan (dB fields, NaN masks,
    re-gridded beams) of which a caller typically touches a few; building them eagerly -- as the
    reference's PyartRadop / SimulatedGPM do on the host -- cost 4 - 7 x the simulation itself.
    Iteration order = insertion order; every read access goes through the builders.  A full
    MutableMapping (pop, update, setdefault, ==, copy ... see every key; round-3 advisor finding: the
    dict subclass this used to be overrode only part of the dict API); `copy()`, pickling and
    deepcopy build the pending values and give a plain dict."""

    def __init__(self):
        self._make = {}
        self._data = {}             # key -> value, or the sentinel while pending (keeps the insertion order)
        self._build = threading.Lock()      # two threads reading one pending key: built once, by the first

    _PENDING = object()

    def add(self, key, make):
        self._make[key] = make
        self._data[key] = LazyDict._PENDING

    def __getitem__(self, key):
        v = self._data[key]
        if v is LazyDict._PENDING:
            with self._build:
                v = self._data[key]
                if v is LazyDict._PENDING:
                    v = self._data[key] = self._make.pop(key)()
        return v

    def __setitem__(self, key, value):
        self._make.pop(key, None)
        self._data[key] = value

    def __delitem__(self, key):
        del self._data[key]
        self._make.pop(key, None)

    def __iter__(self):
        return iter(list(self._data))

    def __len__(self):
        return len(self._data)

    def __contains__(self, key):
        return key in self._data

    def keys(self):
        return list(self._data)

    def values(self):
        return [self[k] for k in list(self._data)]

    def items(self):
        return [(k, self[k]) for k in list(self._data)]

    def pending(self, key):
        return self._data.get(key) is LazyDict._PENDING

    def copy(self):
        return dict(self.items())

    def __reduce__(self):
        return (dict, (self.copy(),))

    def __deepcopy__(self, memo):
        return copy.deepcopy(self.copy(), memo)

    def __repr__(self):
        return 'LazyDict(%s)' % ', '.join('%r%s' % (k, ' (pending)' if self.pending(k) else '') for k in self._data)


class RadarScan(object):
    """Light stand-in for PyartRadop (cosmo_pol/radar/pyart_wrapper.py:186-342):
    same field names and conventions (ZH, ZV, ZDR in dB with 0 -> NaN, masked
    where NaN, extra 'Latitude', 'Longitude', 'rangearray' fields, one entry of
    sweep_start/stop_ray_index per sweep).  `fields` is a LazyDict: the dB conversion, the
    stacking over the sweeps and the NaN mask of a field are made when the field is first read."""

    def __init__(self, scan_type, elevations, azimuths, ranges, pos_time, sweeps):
        self.scan_type = scan_type
        self.nsweeps = len(sweeps)
        self.range = {'data': np.asarray(ranges)}
        self.latitude = {'data': np.array(pos_time['latitude'], dtype=float)}
        self.longitude = {'data': np.array(pos_time['longitude'], dtype=float)}
        self.altitude = {'data': np.array(pos_time['altitude'], dtype=float)}
        self.time = {'data': np.zeros(len(sweeps[0]['azimuth'])),
                     'units': 'seconds since ' + str(pos_time.get('time'))}
        self.raw = sweeps                       # linear-unit arrays per sweep
        counts = np.array([len(s['azimuth']) for s in sweeps], dtype=np.int64)
        stop = np.cumsum(counts)
        self.azimuth = {'data': np.concatenate([np.asarray(s['azimuth'], dtype=float) for s in sweeps])}
        self.elevation = {'data': np.concatenate([np.asarray(s['elevation'], dtype=float) for s in sweeps])}
        self.sweep_start_ray_index = {'data': (stop - counts).astype(float)}
        self.sweep_stop_ray_index = {'data': (stop - 1).astype(float)}
        self.sweep_number = {'data': np.arange(0, self.nsweeps, dtype=float)}
        self.sweep_mode = {'data': [scan_type] * self.nsweeps}
        self.fixed_angle = {'data': np.array(elevations if scan_type == 'ppi' else azimuths,
                                             dtype=float)}
        self.fields = LazyDict()

        def field(k):
            def make():
                parts = [s['fields'][k] for s in sweeps]
                stack = np.concatenate(parts, axis=0) if (len(parts) > 1 or k in _DB_FIELDS) else parts[0]
                if k in _DB_FIELDS:
                    # dB, 0 -> NaN (pyart_wrapper.py:256-258); in place on the stacked copy
                    with np.errstate(divide='ignore', invalid='ignore'):
                        stack[stack == 0] = np.nan
                        np.log10(stack, out=stack)
                        stack *= 10
                return {'data': np.ma.array(stack, mask=np.isnan(stack), copy=False)}
            return make

        def coords(src):
            def make():
                stack = np.concatenate([s[src] for s in sweeps], axis=0)
                return {'data': np.ma.array(stack, mask=np.isnan(stack)), 'units': ['degrees']}
            return make
        for k in sweeps[0]['fields'].keys():
            self.fields.add(k, field(k))
        self.fields.add('Latitude', coords('lats'))
        self.fields.add('Longitude', coords('lons'))
        self.nrays = len(self.azimuth['data'])
        self.ngates = len(self.range['data'])
        # (the builders capture locals, never `self`: a scan must not be part of a reference cycle, or the
        # page-locked block of its arrays would wait for the cyclic garbage collector instead of
        # returning to the operator's pool when the scan is dropped)
        rng, shape = self.range['data'], (self.nrays, self.ngates)
        self.fields.add('rangearray', lambda: {'data': np.broadcast_to(rng, shape)})

    def to_pyart(self, varray=None):
        """The scan as the reference's PyartRadop (a pyart.core.Radar); needs Py-ART."""
        from .pyart_wrapper import as_pyart_radar
        return as_pyart_radar(self, varray)

    def to_radials(self):
        """The scan as the reference's list of sweeps of Radial records
        (what cut_at_sensitivity / PyartRadop consume, radar_operator.py:445-451): linear
        units, row views of the per-sweep arrays."""
        from .radial import to_radials
        return [to_radials(s) for s in self.raw]

    def get_field(self, sweep_idx, variable):
        i0 = int(self.sweep_start_ray_index['data'][sweep_idx])
        i1 = int(self.sweep_stop_ray_index['data'][sweep_idx]) + 1
        return self.fields[variable]['data'][i0:i1]


def _mask_from_sum(sum8, n_sub):
    """Builder of the radial mask from cpol_outputs.mask_sum8: the two statements of doppler_scatter.py:472-477 on the
    sum of the sub-beams' mask codes (the sum itself comes from the device)."""
    def make():
        if n_sub == 1:
            return sum8.astype(np.float64)          # (one sub-beam: the codes themselves; the (-1, 0] -> 0 rule changes no integer)
        # the two statements on the 256 possible sums, then one gather (5 x faster than the statements on the whole array)
        lut = np.arange(-128, 128) / float(n_sub)
        lut[np.logical_and(lut > -1, lut <= 0)] = 0
        return np.take(lut, sum8.astype(np.intp) + 128)
    return make


_table_serial = [0]


def _table_identity():
    """64-bit identity (never 0) of one staged hydrometeor slot for the library's table caches
    (cpol_hydro_desc.table_id): a process-wide serial stored with the operator's cache entry of the table
    set -- never derived from a memory address, which a later array of other content could re-use."""
    _table_serial[0] += 1
    return (os.getpid() << 32) | _table_serial[0]


class RadarOperator(object):
    _table_serial = 0          # tags of per-ray table sets handed to the library

    def __init__(self, options_file=None, output_variables='all', *, device=0, lut_dir=None,
                 luts=None, config=None, distributed=False, gather_to=None, lanes=2, backend='hip',
                 pyart_output=False):
        if backend != 'hip':
            # by design: the product path is the HIP library or nothing (no CPU fallback)
            raise N.NativeError("backend %r: only 'hip' exists; the CPU restatement lives under "
                                "oracle/ as test infrastructure and is never used by the product"
                                % (backend,))
        print('Reading options defined in options file')
        self._ctx = N.Context(device)         # raises if the HIP library / GPU is missing
        self._affinity_before = None
        if distributed:
            # one process per GPU: run this rank's threads next to its GPU (CPOL_NUMA_BIND=0: leave the
            # affinity alone; close() restores what the thread had before); the reference's worker pool is
            # not placed at all (radar_operator.py:402)
            before = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else None
            self.numa = N.bind_to_device_numa_node(device)
            if self.numa.get('bound'):
                self._affinity_before = before
        self._pool = N.PinnedPool(device)     # page-locked blocks of the results handed to the user
        # lanes: contexts forked from _ctx (shared cube / tables, own stream + work buffers);
        # the sweeps of a volume scan are spread over them so that they overlap on the GPU
        self.lanes = max(1, int(lanes))
        self._lane_ctx = []
        self._lock = threading.RLock()
        self.device = device
        self.distributed = bool(distributed)   # shard the rays of every sweep over the ranks
        self.gather_to = gather_to             # distributed scans: None = every rank gets the scan (all-gather);
                                               # r = rank r alone does (gather; get_PPI returns None elsewhere)
        self._runner = None
        self.pyart_output = bool(pyart_output) # get_PPI / get_RHI return a pyart.core.Radar (needs Py-ART)
        self.reuse_device_tables = True        # keep per-ray tables in HBM between equal sweeps
        self.volume_in_one_sequence = True     # get_PPI / get_RHI: all sweeps of a scan in one launch sequence
        self.pipeline_single_beam_scans = True # ... unless the scan has ONE sub-beam per ray and the operator lanes: a sweep per lane, page-locked outputs, one wait
                                               # (False: sweep by sweep, spread over the lanes)
        self.sequence_memory_budget = None     # bytes of device work buffers ONE launch sequence may need (about 1.2 KB per
                                               # sub-beam gate with six species): a scan beyond it is run as several
                                               # sequences of whole sweeps.  None: a third of the memory free at the time
        self.debug_flags = 0                   # cpol_sweep_params.debug_flags (tests / tools only: N.DEBUG_EXACT_SUBBEAMS)
        self.compact_mask = False              # True: in pinned (non-blocking) host outputs the radial mask crosses PCIe as one byte
                                               # per gate (the sum of the sub-beams' codes, cpol_outputs.mask_sum8) and becomes the
                                               # reference's float64 array -- sum / n_sub, (-1, 0] -> 0: doppler_scatter.py:472-477 --
                                               # on the host when `mask` is first read (0.13 ms per 180 k gates, on the reader's
                                               # thread: worth it for a consumer that seldom reads the mask; measured by bench.py).
                                               # False (the default, and every blocking call): the float64 array from the device
        self.lut_dir = lut_dir
        if lut_dir:
            from . import tablecache
            tablecache.set_default_dir(os.path.join(lut_dir, '.cpol_cache'))   # slow host-side staging tables
        self._user_luts = luts
        self.current_microphys_scheme = '1mom'
        self.dic_vars = None
        self.N = 0
        self.lut_sz = None
        self._model_staged = False
        self._staged_serial = 0
        self._staged_hydro = None
        self.constants = None
        if output_variables in ['all', 'only_model', 'only_radar']:
            self.output_variables = output_variables
        else:
            self.output_variables = 'all'
            print("Invalid output_variables input, must be either 'all', 'only_model' or "
                  "'only_radar'")
        conf = config if config is not None else cfg.init(options_file)
        self.config = conf

    # ------------------------------------------------------------------ config
    @property
    def config(self):
        return copy.deepcopy(self.__config)

    @config.setter
    def config(self, config_dic):
        """Validates, derives constants and (re)loads the lookup tables when the
        frequency or the melting switch changed (radar_operator.py:106-159)."""
        if config_dic is None:
            self.__config = None
            return
        print('Loading new configuration...')
        checked = cfg.sanity_check(config_dic)
        old = getattr(self, '_RadarOperator__config', None)
        reload_lut = (old is None or not self.lut_sz
                      or checked['radar']['frequency'] != old['radar']['frequency']
                      or checked['microphysics']['with_melting'] != old['microphysics']['with_melting']
                      or checked['microphysics']['with_ice_crystals'] != old['microphysics']['with_ice_crystals']
                      or checked['microphysics']['scheme'] != old['microphysics']['scheme']
                      or (checked['doppler']['scheme'] == 2) != (old['doppler']['scheme'] == 2)
                      or (checked['doppler']['scheme'] == 3) != (old['doppler']['scheme'] == 3))
        self.__config = checked
        self.constants = K.DerivedConstants(checked)
        self._cache = {}                       # per-configuration host-side tables
        if reload_lut:
            if old is not None:
                print('Reloading lookup tables...')
            self.set_lut()

    def _drop_lanes(self):
        """Closes the forked contexts (before re-staging).  Each is drained first, so that queued
        sweeps complete into their result arrays (which live in the operator's pinned pool, not in
        the lane) and a deferred domain error of a dropped lane is raised, not lost."""
        lanes, self._lane_ctx = self._lane_ctx, []
        err = None
        for c in lanes:
            try:
                c.synchronize()
            except IndexError as e:
                err = e
            c.close()
        if err is not None:
            raise err

    def _lane(self, i):
        """Context of lane i (0 = the root context); lanes are forked on first use and
        dropped whenever the staged model / tables change."""
        if i == 0:
            return self._ctx
        with self._lock:
            while len(self._lane_ctx) < i:
                self._lane_ctx.append(self._ctx.fork())
            return self._lane_ctx[i - 1]

    def close(self):
        if self._ctx is not None:
            try:
                self._drop_lanes()
            except IndexError as e:           # a queued sweep left the model domain and nobody waited for it
                print('RadarOperator.close: %s' % e)
            if self._runner is not None:
                self._runner.drain()
                self._runner = None
            for _, r in self.__dict__.pop('_group_runners', {}).values():
                r.drain()
            self._ctx.close()
            self._ctx = None
            self._pool.close()                # (blocks of results still held are freed with their last view)
            if self._affinity_before is not None:
                try:
                    os.sched_setaffinity(0, self._affinity_before)
                except OSError:
                    pass
                self._affinity_before = None
        self.dic_vars = None
        self.lut_sz = None
        self.__config = None

    # ------------------------------------------------------------------ tables
    def set_lut(self):
        """Loads the scattering tables of the current configuration and stages
        them (with the per-bin PSD factors) in HBM (radar_operator.py:161-182)."""
        conf = self.__config
        scheme = conf['microphysics']['scheme']
        self.current_microphys_scheme = scheme
        hl = hyd.hydrometeor_list(conf)
        # (Doppler scheme 3 with the melting scheme, round 6: the fall speed of a melting species is inverted gate by gate
        # through the reference's linear interpolator over (V(D_k), D_k), hydrometeors.py:480-500 -- cpol_spectrum.inl)
        spectrum = conf['doppler']['scheme'] == 3
        key = (scheme, conf['radar']['frequency'], conf['microphysics']['scattering'], tuple(hl),
               conf['doppler']['scheme'] == 2, spectrum)
        cache = self.__dict__.setdefault('_lut_cache', {})
        if key in cache:
            # tables of a configuration seen before (e.g. the Ku / Ka / ground switches of
            # get_GPM_swath): no file reads, no per-bin factor rebuild -- only the H2D staging
            lut, built = cache[key]
            self.lut_sz = lut
            self._drop_lanes()
            self._stage_t_functions(hl, scheme)
            for slot, h in enumerate(hl):
                d, table, pre, dnu, aux, dw, st = built[h]
                self._ctx.stage_hydro(slot, d, table, pre, dnu, aux)
                if dw is not None:
                    self._ctx.stage_doppler_weights(slot, dw)
                if st is not None:
                    self._ctx.stage_spectrum_tables(slot, *st)
            self._ctx.set_num_hydro(len(hl))
            self._ctx.prepare()                  # integral tables of this table set (cpol_prepare)
            self._staged_hydro = hl
            if self._model_staged and self._staged_vars != hyd.variable_list(conf):
                self._stage_model()
            return
        if callable(self._user_luts):
            lut = self._user_luts(hl, conf['radar']['frequency'], scheme)
        elif self._user_luts is not None:
            missing = [h for h in hl if h not in self._user_luts]
            if missing:
                raise IOError('no lookup table supplied for hydrometeors %s' % missing)
            lut = {h: self._user_luts[h] for h in hl}
        else:
            lut = load_all_lut(scheme, hl, conf['radar']['frequency'],
                               conf['microphysics']['scattering'], lut_dir=self.lut_dir)
        self.lut_sz = lut
        self._drop_lanes()
        self._stage_t_functions(hl, scheme)
        var_index = {v: i for i, v in enumerate(hyd.variable_list(conf))}
        built = {}
        for slot, h in enumerate(hl):
            d, table, pre, dnu, aux = hyd.build_hydro(h, scheme, lut[h], var_index)
            # identity of this slot's content for the library's integral-table cache: the table
            # set is kept alive (and unchanged) by _lut_cache below
            d.table_id = _table_identity() if len(cache) < 4 else 0
            self._ctx.stage_hydro(slot, d, table, pre, dnu, aux)
            dw = None
            if conf['doppler']['scheme'] == 2:
                dw = hyd.doppler_weights(h, scheme, lut[h])
                self._ctx.stage_doppler_weights(slot, dw)
            st = None
            if spectrum:
                st = hyd.spectrum_tables(h, scheme, lut[h])
                self._ctx.stage_spectrum_tables(slot, *st)
            built[h] = (d, table, pre, dnu, aux, dw, st)
        if len(cache) < 4:                       # a few table sets at most (host memory)
            cache[key] = (lut, built)
        self._ctx.set_num_hydro(len(hl))
        self._ctx.prepare()                      # integral tables of this table set (cpol_prepare)
        self._staged_hydro = hl
        if self._model_staged and self._staged_vars != hyd.variable_list(conf):
            self._stage_model()                 # variable set changed (1mom <-> 2mom)

    def _stage_t_functions(self, hl, scheme):
        """Host-tabulated float32 functions of the temperature (1-moment snow intercept, 1-moment
        ice moment relation): staged once per context, only when the species is simulated."""
        if scheme != '1mom':
            return
        done = self.__dict__.setdefault('_tfun_staged', set())
        want = [(n, w) for n, w, h in (('snow_n0', N.TFUN_SNOW_N0, 'S'), ('ice_mom2_a', N.TFUN_ICE_MOM2_A, 'I'))
                if (h in hl or (h == 'S' and 'mS' in hl)) and n not in done]
        if not want:
            return
        tabs = hyd.t_function_tables([n for n, _ in want])
        for n, w in want:
            self._ctx.stage_t_function(w, tabs[n])
            done.add(n)

    # ------------------------------------------------------------------ model
    def load_model_file(self, filename, cfilename=None):
        """Loads the model variables from a file (cosmo_pol/radar_operator.py:217-309): NetCDF classic
        in COSMO's conventions or an .npz archive with the same names (cosmo_pol_amd/model_io.py), the
        c-file `cfilename` supplying the half-level heights HHL when the file has none.  The file may hold
        the raw model output (P, T, QV, QR, QC, QI, QS, QG, U, V, W [+ QH, QNH, QNR, QNS, QNG]), from which
        the densities, RHO and -- for refraction scheme 2 -- the refractivity N are derived as pycosmo does
        for the reference, or the derived variables themselves.  GRIB needs pycosmo and is refused with a
        pointer.  ValueError when a necessary variable is missing, as in the reference (:264-275)."""
        from . import model_io
        want_n = self.__config['refraction']['scheme'] == 2
        m = model_io.read_model_file(filename, cfilename, want_refractivity=want_n)
        if want_n and 'N' not in m['data']:
            # (radar_operator.py:237-247)
            print('Necessary variables for computation of atm. refractivity were not found in file. '
                  'Using 4/3 method instead.')
        print('Using %s scheme' % ('2-moment' if m['scheme'] == '2mom' else '1-moment'))
        print('Reading variables ', sorted(m['data']), ' from file')
        self.load_model_arrays(m['data'], m['zlevels'], m['proj_info'], m['resolution'], time=m['time'])
        print('-------done------')

    def load_model_arrays(self, data, zlevels, proj_info, resolution, time=None):
        """data: {name: [nz, ny, nx] float32} with the names of the reference
        (U, V, W, QR_v, QS_v, QG_v, QI_v, RHO, T [+ QH_v, QN*_v]); zlevels
        [nz, ny, nx] (level 0 = model top); proj_info with Lo1, La1, Lo2, La2,
        Latitude_of_southern_pole, Longitude_of_southern_pole; resolution
        (dlon, dlat)."""
        two_mom = all(k in data for k in hyd.BASE_VARIABLES_2MOM)
        missing = [k for k in hyd.BASE_VARIABLES if k not in data]
        if missing:
            raise ValueError('Not all necessary variables could be found: missing %s' % missing)
        scheme = '2mom' if two_mom else '1mom'
        attrs = {'z-levels': zlevels, 'proj_info': proj_info, 'resolution': resolution,
                 'time': time}
        self.dic_vars = {k: ModelVar(k, v, attrs) for k, v in data.items() if k != 'N'}
        self.N = ModelVar('N', data['N'], attrs) if 'N' in data else 0     # refractivity
        self._zlevels = zlevels
        self._proj = proj_info
        self._res = resolution
        if scheme != self.__config['microphysics']['scheme']:
            print('Using %s scheme' % ('2-moment' if two_mom else '1-moment'))
            conf = self.config
            conf['microphysics']['scheme'] = scheme
            self._model_staged = False
            self.config = conf
        self._stage_model()

    def _stage_model(self):
        conf = self.__config
        self._drop_lanes()
        names = hyd.variable_list(conf)
        p = self._proj
        llc = np.asarray((float(p['Lo1']), float(p['La1']))).astype('float32')
        urc = np.asarray((float(p['Lo2']), float(p['La2']))).astype('float32')
        res = np.asarray(self._res, dtype=np.float32)
        sp = [float(p['Latitude_of_southern_pole']), float(p['Longitude_of_southern_pole'])]
        self._ctx.stage_model([self.dic_vars[n].data for n in names], self._zlevels, llc, urc,
                              res, sp)
        self._staged_vars = names
        self._model_staged = True
        self._staged_serial = getattr(self, '_staged_serial', 0) + 1      # (host caches that depend on the cube)

    def get_pos_and_time(self):
        c = self.__config['radar']['coords']
        t = None
        if self.dic_vars:
            t = self.dic_vars['T'].attributes.get('time')
        return {'latitude': c[0], 'longitude': c[1], 'altitude': c[2], 'time': t}

    # ------------------------------------------------------------------ sweeps
    def _check_ready(self):
        if not self.dic_vars or not self._model_staged:
            print('No model file has been loaded! Aborting...')
            return False
        return True

    def simulate_rays(self, azimuths, elevations, on_device=False, device_outputs=None,
                      apply_sensitivity=True, paths=None, lane=0, pinned=False):
        """One batched launch sequence for the given rays (az[i], el[i]) of the
        ground radar of the configuration.  Returns a dict of [n_rays, n_gates]
        arrays (linear units, NaN = no data).
        `device_outputs`: optional {field: device pointer} (outputs stay in HBM).
        `paths`: optional float32 [n_rays, n_vnodes, 3, n_gates] host-computed ray
        paths (s, h, e_deg) replacing the 4/3-earth model (CPOL_GEOM_HOST_PATHS).
        `lane`: which forked context (stream + work buffers) runs the sweep; sweeps on
        different lanes overlap on the GPU (one host thread per lane at a time).
        Host outputs are views of one block of page-locked memory from the operator's pool,
        filled by a single device-to-host copy queued behind the kernels; the block stays theirs
        until the last of them is dropped (then it is re-used by a later sweep).
        `pinned`: do not wait -- the call returns once the sweep is queued; call `wait(lane)` before
        reading the arrays.  (The copy of one sweep then overlaps the kernels of the sweeps on the
        other lanes.)  The gate coordinates (`lats`, `lons`, `dist`, `heights`) of an unchanged
        scan geometry are read-only arrays shared between results."""
        conf = self.__config
        coords = conf['radar']['coords']
        if coords[2] > K.MAX_MODEL_HEIGHT:
            raise NotImplementedError('spaceborne geometry: use get_GPM_swath')
        rr = self.constants.RANGE_RADAR
        if conf['refraction']['scheme'] == 2 and paths is None:
            if self.N is None or isinstance(self.N, int):
                # reference: falls back to the 4/3 model with a notice
                # (interpolation.py:138-145, radar_operator.py:237-247)
                print('Refraction scheme 2 needs the refractivity N as an additional model '
                      'variable; 4/3 Earth model will be used instead...')
            else:
                from . import refraction
                sub = self._cached('sub', lambda: quadrature.subbeams(conf))
                h_col, n_col = refraction.refractivity_column(
                    self.N.data, self._zlevels, self._proj, self._res, coords,
                    conf['radar'].get('type', 'ground'))
                paths = refraction.ode_paths(rr, elevations, sub.pts_ver, coords, h_col, n_col)
        mode = N.GEOM_GROUND_43 if paths is None else N.GEOM_HOST_PATHS
        return self._run_rays(azimuths, elevations, coords, len(rr), float(rr[0]), mode,
                              device_outputs=device_outputs, apply_sensitivity=apply_sensitivity,
                              paths=paths, lane=lane, pinned=pinned)

    def wait(self, lane=0):
        """Waits for the sweeps queued on `lane` (pinned / device outputs); raises IndexError
        if one of them left the model domain."""
        self._lane(lane).synchronize()

    def _run_rays(self, azimuths, elevations, coords, n_gates, range0, mode, device_outputs=None,
                  apply_sensitivity=True, paths=None, site=None, sub=None, tables=None, lane=0,
                  pinned=False):
        conf = self.__config
        az = np.ascontiguousarray(np.asarray(azimuths, dtype=np.float64).reshape(-1))
        el = np.ascontiguousarray(np.asarray(elevations, dtype=np.float64).reshape(-1))
        if az.shape != el.shape:
            raise ValueError('azimuths and elevations must have the same length')
        n_rays = len(az)
        if sub is None:
            sub = self._cached('sub', lambda: quadrature.subbeams(conf))
        version = 0
        if tables is not None:
            traj, geo_t = tables
        else:
            # per-ray tables depend only on (site, azimuths, elevations, quadrature nodes);
            # `version` lets the library keep its device copies when they did not change
            key = ('rays', az.tobytes(), el.tobytes(), tuple(np.ravel(coords)))

            def make():
                RadarOperator._table_serial += 1
                return (RadarOperator._table_serial,) + geo.ray_tables(coords, az, el, sub)
            version, traj, geo_t = self._cached(key, make, lru=8)
            if paths is not None or site is not None or not self.reuse_device_tables:
                version = 0

        want_model = self.output_variables in ('all', 'only_model')

        def prepare(paths=paths, site=site):
            """The argument structs of cpol_run_sweep for this set of rays (and the arrays they point into)."""
            p = N.SweepParams()
            p.n_rays, p.n_gates, p.n_sub = n_rays, n_gates, sub.n_sub
            p.n_hnodes, p.n_vnodes = len(sub.pts_hor), len(sub.pts_ver)
            p.with_melting = int(conf['microphysics']['with_melting'])
            p.with_attenuation = int(conf['microphysics']['with_attenuation'])
            p.integrate_model = int(want_model)
            p.outputs_on_device = 1 if device_outputs is not None else 2     # host outputs: page-locked, one copy
            # Doppler schemes 1 (analytic mean fall speed), 2 (rcs-weighted) and 3 (the full spectrum); none for GPM
            # (doppler_scatter.py:83-87)
            doppler = (conf['doppler']['scheme'] in (1, 2, 3) and conf['radar'].get('type') != 'GPM'
                       and mode != N.GEOM_SPACEBORNE)
            spectrum = doppler and conf['doppler']['scheme'] == 3
            if spectrum and (conf['doppler']['turbulence_correction'] or conf['doppler']['motion_correction']):
                # both corrections are dead upstream: spectral_width_turb / _motion read a module-level
                # CONFIG that is still None (doppler_scatter.py:24, 737, 771), EDR is never loaded
                raise NotImplementedError('doppler/turbulence_correction and doppler/motion_correction '
                                          'raise in the reference as well; not reproduced')
            p.simulate_doppler = int(conf['doppler']['scheme']) if doppler else 0
            vi = {v: i for i, v in enumerate(self._staged_vars)}
            p.var_u, p.var_v, p.var_w = vi['U'], vi['V'], vi['W']
            p.var_rho = vi.get('RHO', -1)
            varray = None
            if spectrum:
                varray = np.ascontiguousarray(self.constants.VARRAY, dtype=np.float64)
                p.n_vbins = len(varray)
                p.c_spectrum = float(self.constants.WAVELENGTH ** 4
                                     / (np.pi ** 5 * conf['radar']['K_squared'] ** 2))
            p.geometry_mode = mode
            p.debug_flags = int(self.debug_flags)
            if site is None:
                re, ke = geo.earth_radius_for_refraction(coords)
                sin_u1, cos_u1, _ = geo.radar_site_constants(coords)
                p.radar_lat, p.radar_lon, p.radar_alt = (float(coords[0]), float(coords[1]),
                                                         float(coords[2]))
                p.ke, p.re = ke, re
                p.sin_u1, p.cos_u1 = sin_u1, cos_u1
            else:
                p.ke, p.re = 1.0, 0.0
            p.range0 = range0
            p.range_step = float(conf['radar']['radial_resolution'])
            p.wavelength = float(self.constants.WAVELENGTH)
            p.k_squared = float(conf['radar']['K_squared'])
            p.radial_res = float(conf['radar']['radial_resolution'])
            p.c_zh = float(self.constants.WAVELENGTH ** 4 / (np.pi ** 5 * conf['radar']['K_squared']))

            thr = (self._cached(('sens', n_gates),
                                lambda: geo.sensitivity_threshold(conf, self.constants, n_gates))
                   if apply_sensitivity else None)
            p.apply_sensitivity = int(thr is not None)
            t = N.RayTables()
            if paths is not None:
                paths = np.ascontiguousarray(paths, dtype=np.float32)
                if paths.shape != (n_rays, p.n_vnodes, 3, n_gates):
                    raise ValueError('paths must have shape [n_rays, n_vnodes, 3, n_gates] = %s'
                                     % ((n_rays, p.n_vnodes, 3, n_gates),))
            if site is not None:
                site = np.ascontiguousarray(site, dtype=np.float64)
            keep = [traj, geo_t, sub.sub_h, sub.sub_v, sub.sub_w, thr, paths, site]
            t.traj, t.geo = traj.ctypes.data, geo_t.ctypes.data
            t.sub_h, t.sub_v, t.sub_w = sub.sub_h.ctypes.data, sub.sub_v.ctypes.data, sub.sub_w.ctypes.data
            t.sens_thr = thr.ctypes.data if thr is not None else None
            t.site = site.ctypes.data if site is not None else None
            t.paths = paths.ctypes.data if paths is not None else None
            nyq = None
            if doppler and conf['radar']['nyquist_velocity'] is not None:
                # Nyquist velocity of every ray from its nominal elevation / azimuth
                # (doppler_scatter.py:431-437: the central sub-beam's angles)
                nyq = np.ascontiguousarray(conf['radar']['nyquist_velocity'](el, az), dtype=np.float64)
                keep.append(nyq)
            t.nyquist = nyq.ctypes.data if nyq is not None else None
            if varray is not None:
                keep.append(varray)
                t.varray = varray.ctypes.data
            if sub.sub_smooth is not None:        # integration scheme 'ml': per-gate weights
                t.sub_smooth = sub.sub_smooth.ctypes.data
                t.ml_filter = sub.ml_filter.ctypes.data
                t.ml_radius = int(sub.ml_radius)
            t.version = version
            return p, t, keep, doppler, spectrum, varray

        # the structs of an unchanged set of rays are built once (a sweep repeated, the sweeps of a volume scanned
        # again): ~25 us of attribute traffic per call otherwise, half of what a c2 sweep takes on the device
        if version and paths is None and site is None:
            p, t, keep, doppler, spectrum, varray = self._cached(
                ('prepared', version, n_gates, range0, mode, bool(apply_sensitivity), want_model, device_outputs is not None,
                 int(self.debug_flags)),
                prepare, lru=16)
        else:
            p, t, keep, doppler, spectrum, varray = prepare()

        o = N.Outputs()
        res = {}
        geom = None
        if device_outputs is not None:
            for k, ptr in device_outputs.items():
                setattr(o, k, ptr)
        else:
            shape = (n_rays, n_gates)
            spec = [(k, np.float32, shape) for k in RADAR_FIELDS]
            if doppler:
                spec.append(('RVEL', np.float64, shape))
            if spectrum:
                spec.append(('DSPECTRUM', np.float64, shape + (len(varray),)))
            # (pinned calls only: a blocking call would have to widen the bytes on the caller's thread at once -- ~1 ns per gate,
            # more than the 7 bytes per gate cost on PCIe; measured on the c5 swaths, 1.7 M gates: 8.5 -> 12.7 ms per step)
            mask8 = self.compact_mask and pinned and 2 * sub.n_sub <= 127
            spec.append(('mask_sum8', np.int8, shape) if mask8 else ('mask', np.float64, shape))
            # gate coordinates of the central sub-beam depend on the ray tables only: of an
            # unchanged table set (version tag) they are copied from the device once
            gkey = ('geom', version, n_gates) if version else None
            geom = self._cache.get(gkey) if gkey else None
            if geom is None:
                spec += [('lats', np.float64, shape), ('lons', np.float64, shape),
                         ('dist', np.float32, shape), ('heights', np.float32, shape)]
            if want_model:
                spec.append(('model_vars', np.float64, (len(self._staged_vars),) + shape))
            # every host output is a view of ONE block of page-locked memory from the operator's pool
            # (64-byte aligned arrays): the kernels write a device image of the block and a single
            # device-to-host copy, queued behind them, moves it.  The block belongs to the arrays: it
            # returns to the pool when the last of them is dropped, whatever happens to the lanes or
            # the operator in between.
            counts = [math.prod(sh) for _, _, sh in spec]
            sizes = [-(-n_el * _ITEMSIZE[dt] // 64) * 64 for n_el, (_, dt, _) in zip(counts, spec)]
            slab, holder = self._pool.take(sum(sizes), writer=self._lane(lane))
            base = slab.ctypes.data
            off = 0
            for (k, dt, sh), n_el, nb in zip(spec, counts, sizes):
                res[k] = slab[off:off + n_el * _ITEMSIZE[dt]].view(dt).reshape(sh)
                setattr(o, k, base + off)           # (= res[k].ctypes.data, without an interface object per array)
                off += nb
            del slab
        ctx = self._lane(lane)
        ctx.run_sweep(p, t, o)
        del keep
        if device_outputs is None and 'mask_sum8' in res:
            # `mask` is made from its one-byte form when it is first read (after wait(lane) for a pinned call, like every
            # other array of the result)
            lazy = LazyDict()
            for k, v in res.items():
                lazy[k] = v
            lazy.add('mask', _mask_from_sum(res['mask_sum8'], sub.n_sub))
            res = lazy
        if device_outputs is None:
            if pinned:
                holder['ctx'], holder['serial'] = ctx, ctx.submitted    # a copy into the block is in flight until wait(lane)
            else:
                ctx.synchronize()                       # blocking call: results complete, IndexError raised here
            if geom is None and gkey is not None:
                if pinned:
                    ctx.synchronize()                   # once per table set: the arrays are complete
                # (own read-only copies: shared by every later result of this table set)
                geom = {k: res[k].copy() for k in ('lats', 'lons', 'dist', 'heights')}
                for a in geom.values():
                    a.flags.writeable = False
                with self._lock:
                    old = [k for k in self._cache if isinstance(k, tuple) and k and k[0] == 'geom']
                    for k in old[:max(0, len(old) - 7)]:
                        del self._cache[k]
                    self._cache[gkey] = geom
            if geom is not None:
                res.update(geom)
        res['n_sub'] = sub.n_sub
        return res

    def _cached(self, key, make, lru=None):
        with self._lock:
            return self._cached_locked(key, make, lru)

    def _cached_locked(self, key, make, lru=None):
        c = self._cache
        if key in c:
            return c[key]
        if lru is not None:
            old = [k for k in c if isinstance(k, tuple) and k and k[0] == key[0]]
            for k in old[:max(0, len(old) - lru + 1)]:
                del c[k]
        c[key] = make()
        return c[key]

    def _simulate_sweeps(self, sweeps):
        """[(az, el), ...] -> packaged sweeps.  The reference runs the sweeps of a scan one
        after the other (radar_operator.py:429-432); here up to `lanes` of them are in
        flight together, one host thread per lane (the library calls release the GIL)."""
        if self.distributed:
            res = self._simulate_volume_sharded(sweeps)
            if res is None:
                return None                   # gather_to: this rank computed its share, another holds the scan
            return [self._package(r, az, el) for r, (az, el) in zip(res, sweeps)]
        if self.pipeline_single_beam_scans and len(sweeps) > 1 and self.lanes > 1 \
                and self._cached('sub', lambda: quadrature.subbeams(self.__config)).n_sub == 1:
            # Single-beam scans are bound by PCIe, not by their kernels (a 360 x 500 sweep: 0.15 ms on the device, 0.17 ms for its
            # 9.4 MB of results): every sweep is queued on a lane with page-locked outputs and the call waits once, at the end --
            # the copy of one sweep runs beside the kernels of the next (a 5-elevation volume with melting layer: 0.93 ms against
            # 1.37 ms as ONE launch sequence, whose one copy overlaps nothing; bench.py's c3 step / `api_ms`).  Same bits
            # (tests/test_gpu_headline.py::test_c3_step_vs_oracle_and_blocking_path).
            n_par = min(self.lanes, len(sweeps))
            res, failure = [], None
            try:
                for k, (az, el) in enumerate(sweeps):
                    res.append(self.simulate_rays(az, el, pinned=True, lane=k % n_par))
            finally:
                for i in range(n_par):
                    try:
                        self.wait(i)                  # (every lane is drained, whatever happened: no copy may outlive the call)
                    except Exception as exc:          # noqa: BLE001  (re-raised below: the first failure of the scan)
                        failure = failure or exc
            if failure is not None:
                raise failure
            return [self._package(r, az, el) for r, (az, el) in zip(res, sweeps)]
        if self.volume_in_one_sequence and len(sweeps) > 1:
            # all sweeps of the scan as ONE launch sequence (rays of different elevations / azimuths in one
            # cpol_run_sweep call): one submission, 7-12 kernel launches and one device-to-host copy per
            # volume instead of per sweep; the per-sweep results are row slices of the volume's arrays --
            # bit-identical to the sweeps run one by one (tests/test_gpu_fullsize.py).  The work buffers of a
            # sequence grow with its sub-beam gates (a 5-elevation volume with 49 sub-beams: 53 GB), so a scan
            # is cut into groups of whole sweeps that fit `sequence_memory_budget`, and a group the device
            # still has no room for is run sweep by sweep.
            out = []
            for group in self._sweep_groups(sweeps):
                try:
                    out.extend(self._simulate_group(group))
                except MemoryError:
                    if len(group) == 1:
                        raise
                    out.extend(r for sw in group for r in self._simulate_group([sw]))
            return out
        n_par = min(self.lanes, len(sweeps))
        if n_par <= 1:
            return [self._package(self._simulate_sweep(az, el), az, el) for az, el in sweeps]
        import queue
        from concurrent.futures import ThreadPoolExecutor
        free = queue.Queue()
        for i in range(n_par):
            self._lane(i)
            free.put(i)

        def one(sw):
            lane = free.get()
            try:
                return self._package(self._simulate_sweep(sw[0], sw[1], lane=lane), sw[0], sw[1])
            finally:
                free.put(lane)
        with ThreadPoolExecutor(max_workers=n_par) as pool:
            return list(pool.map(one, sweeps))

    def _sweep_groups(self, sweeps):
        """Consecutive sweeps whose launch sequence fits the memory budget (at least one sweep per group).  (`free` counts
        the work buffers the context holds already, cpol_mem_info: the same scan is grouped the same way on every call.)"""
        free, _, per_gate = self._ctx.mem_info()
        budget = self.sequence_memory_budget if self.sequence_memory_budget is not None else free // 3
        sub = self._cached('sub', lambda: quadrature.subbeams(self.__config))
        per_ray = per_gate * sub.n_sub * len(self.constants.RANGE_RADAR)
        groups, cur, used = [], [], 0
        for sw in sweeps:
            need = per_ray * len(np.asarray(sw[0]).reshape(-1))
            if cur and used + need > budget:
                groups.append(cur)
                cur, used = [], 0
            cur.append(sw)
            used += need
        if cur:
            groups.append(cur)
        return groups

    def _simulate_group(self, sweeps):
        """Sweeps as one launch sequence -> packaged per-sweep results (row slices of the group's arrays)."""
        az = np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a, _ in sweeps])
        el = np.concatenate([np.asarray(e, dtype=np.float64).reshape(-1) for _, e in sweeps])
        res = self.simulate_rays(az, el)
        out, lo = [], 0
        for a, e in sweeps:
            hi = lo + len(np.asarray(a).reshape(-1))
            part = {k: (v[:, lo:hi] if k in ('model_vars',) else v[lo:hi]) if isinstance(v, np.ndarray) else v
                    for k, v in res.items()}
            out.append(self._package(part, a, e))
            lo = hi
        return out

    def _simulate_sweep(self, az, el, lane=0):
        """All rays of a sweep on this GPU."""
        return self.simulate_rays(az, el, lane=lane)

    def _dist_runner(self, group=None):
        """The operator's ShardedVolumeRunner (device buffers, the stream of the collectives) for the default
        process group, or one of its own for another `group` (kept until close())."""
        import torch
        from . import distributed as D
        if group is None:
            if self._runner is None:
                self._runner = D.ShardedVolumeRunner(torch.device('cuda', self.device), gather_to=self.gather_to,
                                                     slots=max(2, self.lanes + 1))
            return self._runner
        runners = self.__dict__.setdefault('_group_runners', {})
        if id(group) not in runners:
            import torch.distributed as dist
            members = dist.get_process_group_ranks(group)
            if dist.get_rank() not in members:
                raise ValueError('submit_volume(group=...): rank %d is not a member of the group (ranks %s)'
                                 % (dist.get_rank(), members))
            root = self.gather_to
            if root is not None:                  # (gather_to names a rank of the default group)
                if root not in members:
                    # (round-5 advisor: never re-root silently -- the scan would land on a rank nobody named)
                    raise ValueError('gather_to=%d is not a member of the group (ranks %s): results of a rooted '
                                     'gather must land on a rank of the group that computes them' % (root, members))
                root = dist.get_group_rank(group, root)
            runners[id(group)] = (group, D.ShardedVolumeRunner(torch.device('cuda', self.device), group=group,
                                                               gather_to=root, slots=max(2, self.lanes + 1)))
        return runners[id(group)][1]

    def _scan_fields(self, fields=None):
        """(name, dtype) of the arrays a distributed scan collects: all of them, or the subset named."""
        every = ([(k, np.float32) for k in RADAR_FIELDS] + [('dist', np.float32), ('heights', np.float32),
                 ('mask', np.float64), ('lats', np.float64), ('lons', np.float64)])
        if self.__config['doppler']['scheme'] in (1, 2, 3) and self.__config['radar'].get('type') != 'GPM':
            every.append(('RVEL', np.float64))               # (the spectrum itself is not gathered)
        if fields is None:
            return every
        known = dict(every)
        for k in fields:
            if k not in known:
                raise ValueError('distributed scans collect %s; %r is not one of them' % (sorted(known), k))
        return [(k, known[k]) for k in fields]

    def submit_volume(self, sweeps, fields=None, lane=0, group=None):
        """Queues a scan [(azimuths, elevations), ...] whose rays are sharded over the ranks of the default
        torch.distributed group (or of `group`) and returns at once: this rank's rays of ALL sweeps in one launch sequence
        on lane `lane`, ONE collective behind it (cosmo_pol_amd/distributed.py: all-gather, or a gather to
        rank `gather_to`), the device-to-host copy behind that.  -> PendingVolume: `wait()` gives the list of
        per-sweep result dicts (None on a rank that is not `gather_to`).  `fields`: the arrays to collect
        (default: all).  Scans submitted on different lanes overlap: the collective and the copy of one run
        beside the kernels of the next.  Every rank must submit the same scans in the same order."""
        if self.output_variables != 'only_radar':
            raise NotImplementedError('distributed sweeps return the radar observables only '
                                      "(use output_variables='only_radar')")
        n_gates = len(self.constants.RANGE_RADAR)
        ctx = self._lane(lane)

        def run_block(a, e, ptrs):
            self.simulate_rays(a, e, device_outputs=ptrs, lane=lane)
        return self._dist_runner(group).submit(run_block, ctx.stream_ptr(), sweeps, self._scan_fields(fields), n_gates,
                                               host_block=self._pool.take)

    def _simulate_volume_sharded(self, sweeps, lane=0):
        """All sweeps of a scan, the rays of every sweep sharded over the ranks (`submit_volume`, then
        wait).  -> list of per-sweep result dicts, or None on a rank other than `gather_to`."""
        res = self.submit_volume(sweeps, lane=lane).wait()
        self._lane(lane).synchronize()                          # deferred domain error, if any
        if res is None:
            return None
        sub = self._cached('sub', lambda: quadrature.subbeams(self.__config))
        for r in res:
            r['n_sub'] = sub.n_sub
        return res

    def _swath_sharded(self, az, el, coords, n_gates, range0, site, sub, traj, geo_t, dim, lane=0):
        """A spaceborne swath with its SCAN LINES sharded over the ranks in contiguous blocks:
        a rank runs the rays of its scan lines as one launch sequence into its block of the gather buffer,
        ONE collective, rows put into swath order on the device, one copy.  Every rank (or rank `gather_to`
        alone) gets the whole swath, bitwise equal to the single-GPU one.  (The swath geometry -- angles, first gates -- is
        computed by every rank: host work of milliseconds, cached per swath.)"""
        import torch.distributed as dist
        from . import distributed as D
        if self.output_variables != 'only_radar':
            raise NotImplementedError('distributed swaths return the radar observables only '
                                      "(use output_variables='only_radar')")
        n_scans, per_scan = dim
        fields = self._scan_fields()
        row = per_scan * n_gates                      # one row of the layout = one scan line
        lo, hi, _ = D.shard_bounds(n_scans, dist.get_world_size(), dist.get_rank())
        r0, r1 = lo * per_scan, hi * per_scan
        ctx = self._lane(lane)

        def run_block(_rows, _unused, ptrs):
            self._run_rays(az[r0:r1], el[r0:r1], coords[r0:r1], n_gates, range0, N.GEOM_SPACEBORNE,
                           device_outputs=ptrs, site=site[r0:r1], sub=sub, tables=(traj[r0:r1], geo_t[r0:r1]),
                           lane=lane)
        lines = [(np.arange(n_scans, dtype=np.float64), np.zeros(n_scans))]
        res = self._dist_runner().submit(run_block, ctx.stream_ptr(), lines, fields, row,
                                         host_block=self._pool.take).wait()
        ctx.synchronize()                                       # deferred domain error, if any
        if res is None:
            return None                                         # (gather_to: another rank holds the swath)
        out = {k: v.reshape(n_scans * per_scan, n_gates) for k, v in res[0].items()}
        out['n_sub'] = sub.n_sub
        return out

    def _package(self, res, az, el):
        fields = {}
        if self.output_variables in ('all', 'only_radar'):
            for k in RADAR_FIELDS + DOPPLER_FIELDS:
                if k in res:
                    fields[k] = res[k]
        if self.output_variables in ('all', 'only_model'):
            for i, name in enumerate(self._staged_vars):
                fields[name] = res['model_vars'][i]
        return {'fields': fields, 'azimuth': np.asarray(az, dtype=float),
                'elevation': np.asarray(el, dtype=float), 'lats': res['lats'],
                'lons': res['lons'], 'mask': res['mask'], 'dist': res['dist'],
                'heights': res['heights']}

    def _finish_scan(self, scan):
        """RadarScan, or -- when `pyart_output` is set and Py-ART imports -- the reference's
        PyartRadop built from it (cosmo_pol/radar/pyart_wrapper.py:191-342)."""
        if getattr(self, 'pyart_output', False):
            from . import pyart_wrapper
            if pyart_wrapper.pyart_available():
                va = self.constants.VARRAY if self.__config['doppler']['scheme'] == 3 else None
                return pyart_wrapper.as_pyart_radar(scan, va)
            print('pyart_output requested but Py-ART does not import: returning a RadarScan')
        return scan

    def get_PPI(self, elevations, azimuths=None, az_step=None, az_start=0, az_stop=359):
        """Simulates PPI scan(s) (radar_operator.py:357-453); one sweep per
        elevation, returned as a RadarScan."""
        if not self._check_ready():
            return
        if np.isscalar(elevations):
            elevations = [elevations]
        if az_step is None:
            az_step = self.__config['radar']['3dB_beamwidth']
        if azimuths is None or np.any(np.equal(azimuths, None)):
            if az_start > az_stop:
                azimuths = np.hstack((np.arange(az_start, 360., az_step),
                                      np.arange(0, az_stop + az_step, az_step)))
            else:
                azimuths = np.arange(az_start, az_stop + az_step, az_step)
        azimuths = np.asarray(azimuths, dtype=float)
        sweeps = self._simulate_sweeps([(azimuths, np.full(len(azimuths), float(e)))
                                        for e in elevations])
        if sweeps is None:
            return None                       # distributed with gather_to: rank `gather_to` holds the scan
        return self._finish_scan(RadarScan('ppi', list(elevations), list(azimuths),
                                           self.constants.RANGE_RADAR, self.get_pos_and_time(), sweeps))

    def get_RHI(self, azimuths, elevations=None, elev_step=None, elev_start=0, elev_stop=90):
        """Simulates RHI scan(s) (radar_operator.py:455-549); one sweep per azimuth."""
        if not self._check_ready():
            return
        if np.isscalar(azimuths):
            azimuths = [azimuths]
        if elevations is None or np.any(np.equal(elevations, None)):
            if elev_step is None:
                elev_step = self.__config['radar']['3dB_beamwidth']
            elevations = np.arange(elev_start, elev_stop + elev_step, elev_step)
        elevations = np.asarray(elevations, dtype=float)
        sweeps = self._simulate_sweeps([(np.full(len(elevations), float(a)), elevations)
                                        for a in azimuths])
        if sweeps is None:
            return None
        return self._finish_scan(RadarScan('rhi', list(elevations), list(azimuths),
                                           self.constants.RANGE_RADAR, self.get_pos_and_time(), sweeps))

    def get_VPROF(self):
        """90-degree vertical profile (the reference's version is broken as
        shipped, radar_operator.py:311-354; implemented as a one-ray RHI)."""
        if not self._check_ready():
            return
        return self.get_RHI(azimuths=[0.], elevations=[90.])

    def get_GPM_swath(self, GPM_file, band='Ku'):
        """Simulates a GPM-DPR swath (radar_operator.py:551-677, intended behaviour;
        the reference's version is dead as shipped).  `GPM_file`: path
        of a DPR HDF5 file (needs h5py) or a dict with Latitude, Longitude [N, M],
        scLat, scLon, dprAlt [N], scPos [N, 3].  Returns a SimulatedGPM."""
        from . import gpm
        if not self._check_ready():
            return
        swath = gpm.read_swath(GPM_file, band)
        freq, res_m = gpm.band_settings(band)
        saved = self.config
        conf = self.config
        conf['radar']['frequency'] = freq
        conf['radar']['3dB_beamwidth'] = K.GPM_3DB_BEAMWIDTH
        conf['radar']['sensitivity'] = float(K.GPM_SENSITIVITY)
        conf['radar']['type'] = 'GPM'
        conf['radar']['radial_resolution'] = res_m
        try:
            self.config = conf                   # reloads the tables of the new frequency
            # everything below depends on the swath geometry and the band only: kept for the last few
            # swaths (a Ku and a Ka call on the same file, repeated calls on one overpass)
            import hashlib
            hk = hashlib.blake2b(digest_size=16)
            for name in ('Latitude', 'Longitude', 'scLat', 'scLon', 'dprAlt', 'scPos'):
                hk.update(np.ascontiguousarray(swath[name], dtype=np.float64).tobytes())
            hk.update(repr(sorted(self.__config['integration'].items(), key=str)).encode())
            gkey = (hk.hexdigest(), band, self._staged_serial)
            gcache = self.__dict__.setdefault('_gpm_cache', {})  # (survives the configuration switches of this call)
            cached = gcache.get(gkey)
            if cached is None:
                az, el, rng, sat = gpm.swath_angles(swath)
                dim = az.shape
                az, el, rng = az.ravel(), el.ravel(), rng.ravel()
                coords = np.repeat(sat, dim[1], axis=0)                      # one site per ray
                sub = quadrature.subbeams(self.__config)
                traj, geo_t = geo.ray_tables(coords, az, el, sub)
                n_rays = len(az)
                # candidate gates: np.arange(res/2, slant range, res) (atm_refraction.py:252)
                n_cand = np.maximum(np.ceil((rng - res_m / 2.) / res_m), 0).astype(np.int32)
                sin_u1, cos_u1, _ = geo.radar_site_constants(coords)
                site = np.zeros((n_rays, 8))
                site[:, 0], site[:, 1], site[:, 2], site[:, 3] = sin_u1, cos_u1, coords[:, 1], coords[:, 2]
                site[:, 4] = geo.get_earth_radius(coords[:, 0])              # quirk Q1 (degrees as radians)
                p = N.SweepParams()
                p.n_rays, p.n_vnodes = n_rays, len(sub.pts_ver)
                p.range0, p.range_step = res_m / 2., float(res_m)
                first = self._ctx.spaceborne_first_gate(p, traj, site, n_cand, K.MAX_MODEL_HEIGHT)
                k0 = first[:, sub.sub_v[sub.central]]
                n_kept = (n_cand - k0).astype(np.int64)
                site[:, 5], site[:, 6] = k0, n_kept
                n_gates = int(max(1, n_kept.max()))
                cached = (az, el, rng, dim, coords, sub, traj, geo_t, site, n_kept, n_gates)
                with self._lock:
                    for k in list(gcache)[:-3]:
                        del gcache[k]
                    gcache[gkey] = cached
            az, el, rng, dim, coords, sub, traj, geo_t, site, n_kept, n_gates = cached
            if self.distributed:
                res = self._swath_sharded(az, el, coords, n_gates, res_m / 2., site, sub, traj, geo_t, dim)
                if res is None:
                    return None               # gather_to: another rank holds the swath
            else:
                res = self._run_rays(az, el, coords, n_gates, res_m / 2., N.GEOM_SPACEBORNE,
                                     site=site, sub=sub, tables=(traj, geo_t))
            fields = {}
            if self.output_variables in ('all', 'only_radar'):
                for k in RADAR_FIELDS:
                    fields[k] = res[k]
            if self.output_variables in ('all', 'only_model'):
                for i, name in enumerate(self._staged_vars):
                    fields[name] = res['model_vars'][i]
            out = gpm.SimulatedGPM(fields, res['mask'], res['lats'], res['lons'], n_kept, dim, band)
            out.raw = res
            out.n_kept = n_kept.reshape(dim)
            out.azimuths, out.elevations, out.ranges = (az.reshape(dim), el.reshape(dim),
                                                        rng.reshape(dim))
            return out
        finally:
            self.config = saved
