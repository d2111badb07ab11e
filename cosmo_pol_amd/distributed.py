"""Multi-GPU scans: one process per GPU, rays sharded, ONE collective per scan (volume).

The reference farms the radials of a sweep out to a fork pool and collects
pickled Radial objects over pipes (cosmo_pol/radar_operator.py:402-432).  Here
the rays of a sweep are split into contiguous blocks, one per rank; model cube
and lookup tables are replicated in every GPU's HBM (staged once); every rank
runs its block through the C ABI and the output blocks are collected with a
single all-gather (RCCL over xGMI when the backend is "nccl", gloo on CPU for
tests).  Radials never interact (all coupling - PHIDP / attenuation scans,
sub-beam sums - is inside one radial), so the gathered result is bitwise
identical to the single-GPU result.

Block layout (what one rank contributes, `BlockLayout`): for every output field
a [per, n_gates] array of the field's dtype, the fields one after the other at
8-byte aligned offsets; `per` = ceil(n_rays / world) rows, the unused rows of
the tail ranks stay zero.

A scan of several sweeps (a PPI volume, a set of RHIs; one sweep is the special case) is
sharded as a whole (`VolumeLayout`, `ShardedVolumeRunner`): a rank takes its contiguous ray
block of EVERY sweep, runs them as one launch sequence (rays of different elevations in one
cpol_run_sweep call) whose kernels write straight into the rank's block on the device, and
ONE collective at the end of the scan collects the blocks -- an all-gather (every rank gets
the scan) or a gather to one root (`gather_to`: the other ranks receive and copy nothing).
The receiving rank puts the rows into scan order on the device and copies the scan to
page-locked host memory once.  `submit` does not wait: the collective and the copy of scan k
overlap the kernels of scan k + 1.
"""
import contextlib

import numpy as np


def shard_bounds(n_rays, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; every rank gets ceil(n/world)
    rays except the tail ranks, which may get fewer (or none)."""
    per = -(-int(n_rays) // int(world))
    lo = min(rank * per, n_rays)
    hi = min(lo + per, n_rays)
    return lo, hi, per


def pad_block(arr, per, fill):
    """Pads axis 0 of a rank's block to the common block length `per`."""
    arr = np.asarray(arr)
    if arr.shape[0] == per:
        return arr
    pad = np.full((per - arr.shape[0],) + arr.shape[1:], fill, dtype=arr.dtype)
    return np.concatenate([arr, pad], axis=0)


class BlockLayout(object):
    """Byte layout of one rank's output block: field-major, [per, n_gates] per field."""

    def __init__(self, fields, per, n_gates):
        self.fields = [(k, np.dtype(dt)) for k, dt in fields]
        self.per, self.n_gates = int(per), int(n_gates)
        self.offsets, off = {}, 0
        for k, dt in self.fields:
            off = -(-off // 8) * 8
            self.offsets[k] = off
            off += self.per * self.n_gates * dt.itemsize
        self.nbytes = -(-off // 8) * 8

    def view(self, block, k):
        """[per, n_gates] view of field k inside a uint8 block (numpy)."""
        dt = dict(self.fields)[k]
        o = self.offsets[k]
        return block[o:o + self.per * self.n_gates * dt.itemsize].view(dt).reshape(self.per, self.n_gates)

    def assemble(self, gathered, n_rays, world):
        """[world * nbytes] uint8 (numpy) -> {field: [n_rays, n_gates]} (padding trimmed)."""
        blocks = gathered.reshape(world, self.nbytes)
        out = {}
        for k, dt in self.fields:
            parts = []
            for r in range(world):
                lo, hi, _ = shard_bounds(n_rays, world, r)
                if hi > lo:
                    parts.append(self.view(blocks[r], k)[:hi - lo])
            out[k] = np.concatenate(parts, axis=0) if parts else np.empty((0, self.n_gates), dtype=dt)
        return out


def gather_blocks(local, group=None):
    """All-gathers equally sized per-rank blocks (1-D uint8 torch tensors, CPU or GPU)
    -> [world * nbytes] on every rank.  ONE collective per sweep."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, local.contiguous().view(-1), group=group)
    return flat


def simulate_sharded(simulate, azimuths, elevations, fields, n_gates, device=None, group=None):
    """Host-buffer form: runs `simulate(az_block, el_block) -> {field: [n_block, n_gates]
    array}` on this rank's contiguous block of rays and gathers all ranks' results.
    `fields`: list of (name, dtype).  Used by the gloo tests with a CPU stand-in and by
    callers whose per-rank results live on the host; the GPU product path is
    `simulate_sharded_device`."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    az = np.asarray(azimuths, dtype=np.float64)
    el = np.asarray(elevations, dtype=np.float64)
    n = len(az)
    lo, hi, per = shard_bounds(n, world, rank)
    lay = BlockLayout(fields, per, n_gates)
    block = np.zeros(lay.nbytes, dtype=np.uint8)
    if hi > lo:
        res = simulate(az[lo:hi], el[lo:hi])
        for k, dt in lay.fields:
            lay.view(block, k)[:hi - lo] = np.ascontiguousarray(res[k], dtype=dt).reshape(hi - lo, n_gates)
    t = torch.from_numpy(block)
    if device is not None:
        t = t.to(device)
    return lay.assemble(gather_blocks(t, group).cpu().numpy(), n, world)


class VolumeLayout(object):
    """Sharding of a scan of several sweeps: rank r owns rays [lo, hi) of every sweep
    (`shard_bounds` per sweep) and stores them sweep after sweep in ONE block of
    `BlockLayout(fields, per, n_gates)` rows, per = the largest row count of a rank."""

    def __init__(self, fields, rays_per_sweep, world, n_gates):
        self.world = int(world)
        self.n_rays = [int(n) for n in rays_per_sweep]
        self.bounds = [[shard_bounds(n, self.world, r)[:2] for r in range(self.world)] for n in self.n_rays]
        self.rows = [sum(hi - lo for lo, hi in (b[r] for b in self.bounds)) for r in range(self.world)]
        self.per = max(1, max(self.rows))
        self.block = BlockLayout(fields, self.per, n_gates)
        self.n_gates = int(n_gates)

    def local_rays(self, rank, sweeps):
        """Concatenated (azimuths, elevations) of `rank`: its block of sweep 0, of sweep 1, ..."""
        az = [np.asarray(a, dtype=np.float64)[lo:hi] for (a, _), (lo, hi) in
              zip(sweeps, (b[rank] for b in self.bounds))]
        el = [np.asarray(e, dtype=np.float64)[lo:hi] for (_, e), (lo, hi) in
              zip(sweeps, (b[rank] for b in self.bounds))]
        return (np.concatenate(az) if az else np.empty(0)), (np.concatenate(el) if el else np.empty(0))

    # ---- the assembled volume: field-major, inside a field sweep after sweep, rays in scan order ----
    @property
    def final(self):
        """BlockLayout of the assembled volume (sum of all rays rows per field)."""
        if '_final' not in self.__dict__:
            self._final = BlockLayout(self.block.fields, max(1, sum(self.n_rays)), self.n_gates)
        return self._final

    def source_rows(self):
        """For every row of the assembled volume (sweep after sweep, rays in scan order) the rank that
        computed it and its row inside that rank's block: two int64 arrays [sum(n_rays)]."""
        ranks, rows = [], []
        row = [0] * self.world
        for b in self.bounds:
            for r, (lo, hi) in enumerate(b):
                ranks.append(np.full(hi - lo, r, dtype=np.int64))
                rows.append(np.arange(row[r], row[r] + hi - lo, dtype=np.int64))
                row[r] += hi - lo
        cat = lambda parts: np.concatenate(parts) if parts else np.empty(0, dtype=np.int64)   # noqa: E731
        return cat(ranks), cat(rows)

    def assemble_on_device(self, gathered, final, index=None):
        """gathered: [world * nbytes] uint8 tensor (the all-gathered blocks); final: [self.final.nbytes]
        uint8 tensor on the same device, filled field by field with one row gather each (the rows of a
        rank's block -> their place in the scan).  The host then needs ONE copy and no concatenation:
        `views_of_final` slices it.  `index`: (rank, row) index tensors on that device (cached by the caller)."""
        import torch
        if index is None:
            rk, rw = self.source_rows()
            index = (torch.from_numpy(rk).to(gathered.device), torch.from_numpy(rw).to(gathered.device))
        total = sum(self.n_rays)
        if total == 0:
            return index
        blocks = gathered.view(self.world, self.block.nbytes)
        for k, dt in self.block.fields:
            rb = self.n_gates * dt.itemsize
            o, of = self.block.offsets[k], self.final.offsets[k]
            src = blocks[:, o:o + self.per * rb].unflatten(1, (self.per, rb))      # [world, per, row bytes] (a view)
            final[of:of + total * rb].view(total, rb).copy_(src[index[0], index[1]])
        return index

    def views_of_final(self, final):
        """[self.final.nbytes] uint8 (numpy) -> [{field: [n_rays_s, n_gates]} for every sweep s]: views, no copy."""
        out = [dict() for _ in self.n_rays]
        for k, dt in self.block.fields:
            v = self.final.view(final, k)
            r0 = 0
            for s, n in enumerate(self.n_rays):
                out[s][k] = v[r0:r0 + n]
                r0 += n
        return out

    def assemble(self, gathered):
        """[world * nbytes] uint8 (numpy) -> [{field: [n_rays_s, n_gates]} for every sweep s] (host copy:
        the gloo / CPU form; the device form is assemble_on_device + views_of_final)."""
        blocks = gathered.reshape(self.world, self.block.nbytes)
        out = [dict() for _ in self.n_rays]
        for k, dt in self.block.fields:
            views = [self.block.view(blocks[r], k) for r in range(self.world)]
            row = [0] * self.world
            for s, b in enumerate(self.bounds):
                parts = []
                for r, (lo, hi) in enumerate(b):
                    if hi > lo:
                        parts.append(views[r][row[r]:row[r] + hi - lo])
                        row[r] += hi - lo
                out[s][k] = (np.concatenate(parts, axis=0) if parts
                             else np.empty((0, self.n_gates), dtype=dt))
        return out


def simulate_sharded_volume(simulate, sweeps, fields, n_gates, device=None, group=None):
    """Host-buffer form of the volume sharding (gloo tests, CPU stand-ins): `simulate(az, el) ->
    {field: [n, n_gates]}` is called ONCE with this rank's rays of all sweeps; ONE all-gather."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lay = VolumeLayout(fields, [len(a) for a, _ in sweeps], world, n_gates)
    block = np.zeros(lay.block.nbytes, dtype=np.uint8)
    az, el = lay.local_rays(rank, sweeps)
    if len(az):
        res = simulate(az, el)
        for k, dt in lay.block.fields:
            lay.block.view(block, k)[:len(az)] = np.ascontiguousarray(res[k], dtype=dt).reshape(len(az), n_gates)
    t = torch.from_numpy(block)
    if device is not None:
        t = t.to(device)
    return lay.assemble(gather_blocks(t, group).cpu().numpy())


class PendingVolume(object):
    """A sharded scan in flight (`ShardedVolumeRunner.submit`).  `wait()` blocks until this rank's part of
    the collective and, on a receiving rank, the device-to-host copy are complete and returns the per-sweep
    dicts of [n_rays_s, n_gates] arrays -- views of ONE page-locked block, which lives as long as one of
    them does -- or None on a rank that is not the root of a rooted gather."""

    def __init__(self, layout, arr, event, keep=None):
        self._lay, self._arr, self._event, self._keep = layout, arr, event, keep

    def done(self):
        return self._event is None or bool(self._event.query())

    def wait(self):
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        self._keep = None
        return None if self._arr is None else self._lay.views_of_final(self._arr)

    def __del__(self):
        # a dropped, never awaited result: its block must not return to the pool under a copy in flight
        try:
            if self._event is not None:
                self._event.synchronize()
        except Exception:
            pass


class ShardedVolumeRunner(object):
    """Device-resident volume sharding, the product path of RadarOperator(distributed=True) and what
    `bench.py --workload c4` times.  Per scan (`submit`):

      run_block(az, el, {field: device pointer})   queues ONE launch sequence for this rank's rays of all
          sweeps on the library stream; the kernels write straight into this rank's block of the gather
          layout (`VolumeLayout`);
      ONE collective behind it on the runner's own stream (RCCL over xGMI with backend "nccl"):
          gather_to = None   all_gather_into_tensor -- every rank receives the whole scan;
          gather_to = r      gather to rank r -- the other ranks send their block and are done: they
                             receive nothing, assemble nothing and copy nothing to their hosts;
      on a receiving rank the gathered rows are put into scan order ON THE DEVICE (one row gather per
          field) and ONE device-to-host copy lands them in a page-locked block (`host_block`).

    Nothing waits: `submit` returns a `PendingVolume`.  The device buffers of a scan shape exist
    `slots` times and are used in turn, so that the collective and the copy of scan k (runner stream)
    overlap the kernels of scan k + 1 (library stream of another lane); a slot is handed to a new scan
    only behind the event that ends its previous use.  Only the fields named in `fields` travel.
    On CPU tensors (gloo tests) the same code runs synchronously."""

    def __init__(self, device, group=None, gather_to=None, slots=3):
        import torch
        import torch.distributed as dist
        self.device = torch.device(device)
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if gather_to is not None and not 0 <= int(gather_to) < self.world:
            raise ValueError('gather_to = %r: not a rank of this group (world size %d)' % (gather_to, self.world))
        self.root = None if gather_to is None else int(gather_to)          # rank INSIDE `group`
        # (torch's rooted collectives name their root by its rank in the default group)
        self._dst = (self.root if (group is None or self.root is None) else dist.get_global_rank(group, self.root))
        self.cuda = self.device.type == 'cuda'
        self.comm = torch.cuda.Stream(self.device) if self.cuda else None
        self.n_slots = max(1, int(slots))
        backend = dist.get_backend(group)
        # gloo has no rooted gather of device tensors (tests on a one-GPU box): all-gather there, and
        # still only the root assembles and copies
        self.rooted = self.root is not None and (backend == 'nccl' or not self.cuda)
        self.collective = ('gather(dst=%d)' % self.root if self.rooted else 'all_gather_into_tensor') + ' / ' + backend
        self.receives = self.root is None or self.rank == self.root
        self._key = None
        self._slots = []
        self._turn = 0
        self._lib = {}
        self.n_collectives = 0
        self.host_pinned = None           # device runs: whether torch / the HIP runtime see the result blocks as page-locked

    def _layout(self, sweeps, fields, n_gates):
        key = (tuple(len(a) for a, _ in sweeps), tuple((k, str(dt)) for k, dt in fields), int(n_gates))
        if key != self._key:
            import torch
            self.drain()
            lay = VolumeLayout(fields, key[0], self.world, n_gates)
            rk, rw = lay.source_rows()
            self._lay = lay
            self._index = ((torch.from_numpy(rk).to(self.device), torch.from_numpy(rw).to(self.device))
                           if self.receives else None)
            self._slots = []
            self._key = key
        return self._lay

    def _slot(self, lay):
        import torch
        i = self._turn % self.n_slots
        self._turn += 1
        while len(self._slots) <= i:
            nb = lay.block.nbytes
            gets_all = self.receives or not self.rooted       # (an all-gather delivers to every rank)
            self._slots.append({
                'block': torch.zeros(nb, dtype=torch.uint8, device=self.device),
                'gathered': torch.empty(self.world * nb, dtype=torch.uint8, device=self.device) if gets_all else None,
                'final': torch.empty(lay.final.nbytes, dtype=torch.uint8, device=self.device) if self.receives else None,
                'free': None})
            if self.cuda:
                # the zero fill runs on torch's current stream; the library stream (non-blocking: not ordered
                # against it by anything) writes the block next.  The first user of the slot waits for this event
                # exactly as it waits for the slot's previous scan.
                filled = torch.cuda.Event()
                filled.record(torch.cuda.current_stream(self.device))
                self._slots[-1]['free'] = filled
        return self._slots[i]

    def drain(self):
        """Waits for everything queued on the runner's stream."""
        if self.cuda:
            self.comm.synchronize()

    def submit(self, run_block, stream_ptr, sweeps, fields, n_gates, host_block=None):
        """-> PendingVolume.  `stream_ptr`: the library stream (hipStream_t) `run_block` queues its kernels
        on.  `host_block(nbytes) -> (uint8 array, holder)`: where the result lives (the operator's PinnedPool)."""
        import torch
        import torch.distributed as dist
        lay = self._layout(sweeps, fields, n_gates)
        slot = self._slot(lay)
        nb, nf = lay.block.nbytes, lay.final.nbytes
        block = slot['block']
        lib = None
        if self.cuda:
            lib = self._lib.get(stream_ptr)
            if lib is None:
                lib = self._lib[stream_ptr] = torch.cuda.ExternalStream(stream_ptr, device=self.device)
            if slot['free'] is not None:
                lib.wait_event(slot['free'])           # the slot's last collective / copy have read its buffers
        az, el = lay.local_rays(self.rank, sweeps)
        if len(az):
            base = block.data_ptr()
            run_block(az, el, {k: base + lay.block.offsets[k] for k, _ in lay.block.fields})
        arr = None
        if self.receives:
            arr = host_block(nf)[0][:nf] if host_block is not None else np.empty(nf, dtype=np.uint8)
        ctx = contextlib.nullcontext()
        if self.cuda:
            computed = torch.cuda.Event()
            computed.record(lib)
            self.comm.wait_event(computed)             # the collective behind the kernels
            ctx = torch.cuda.stream(self.comm)
        with ctx:
            if self.rooted:
                parts = ([slot['gathered'][r * nb:(r + 1) * nb] for r in range(self.world)]
                         if self.receives else None)
                dist.gather(block, gather_list=parts, dst=self._dst, group=self.group)
                gathered = slot['gathered']
            else:
                dist.all_gather_into_tensor(slot['gathered'], block, group=self.group)
                gathered = slot['gathered']
            self.n_collectives += 1
            if self.receives:
                # rows -> their place in the scan ON THE DEVICE, then ONE copy into the page-locked block the
                # returned arrays are views of (no host-side concatenation)
                lay.assemble_on_device(gathered, slot['final'], self._index)
                host = torch.from_numpy(arr)
                if self.cuda and self.host_pinned is None:
                    # The result block comes from the operator's PinnedPool (hipHostMalloc through the library), not from
                    # torch's caching host allocator.  torch asks the HIP runtime about the pointer (is_pinned ->
                    # hipPointerGetAttributes), and the runtime itself decides at hipMemcpyAsync time: for a block it
                    # page-locked the copy is a true asynchronous DMA on the runner's stream.  Checked once per runner and
                    # kept for the tests / bench (`host_pinned`); a pageable block (no pool given) still copies correctly,
                    # staged by the runtime.
                    self.host_pinned = bool(host.is_pinned())
                host.copy_(slot['final'], non_blocking=True)
            event = None
            if self.cuda:
                event = torch.cuda.Event()
                event.record(self.comm)
                slot['free'] = event
        return PendingVolume(lay, arr, event)


def simulate_sharded_volume_device(run_block, stream_ptr, sweeps, fields, n_gates, device, group=None,
                                   cache=None, host_block=None, gather_to=None):
    """Blocking form of `ShardedVolumeRunner.submit` (one scan, then wait).  `cache`: dict of the caller that
    keeps the runner -- its device buffers and stream -- between scans.  -> list of per-sweep dicts, or None
    on a rank other than `gather_to`."""
    cache = {} if cache is None else cache
    key = ('runner', str(device), gather_to, id(group))
    if key not in cache:
        cache.clear()
        cache[key] = ShardedVolumeRunner(device, group=group, gather_to=gather_to)
    return cache[key].submit(run_block, stream_ptr, sweeps, fields, n_gates, host_block).wait()
