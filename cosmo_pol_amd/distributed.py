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
the tail ranks stay zero.  The kernels write their outputs straight into the
block on the device (`simulate_sharded_device`), the all-gather runs on device
buffers, and ONE device-to-host copy returns the whole sweep.

A scan of several sweeps (a PPI volume, a set of RHIs) is sharded as a whole
(`VolumeLayout`, `simulate_sharded_volume_device`): a rank takes its contiguous ray block
of EVERY sweep, runs them as one launch sequence (rays of different elevations in one
cpol_run_sweep call), and a single all-gather at the end of the volume assembles all
sweeps on every rank -- 10 kernel launches and one collective per volume instead of 10
launches and one collective per sweep.
"""
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


def simulate_sharded_device(run_block, stream_ptr, azimuths, elevations, fields, n_gates, device,
                            group=None, cache=None):
    """Device-resident form (the product path of RadarOperator(distributed=True)):

      run_block(az_block, el_block, {field: device pointer})   queues the sweep of this
          rank's rays on the library stream `stream_ptr`; the kernels write every output
          field straight into this rank's block of the gather buffer layout;
      one all_gather_into_tensor of the device blocks (RCCL over xGMI), ordered behind the
          kernels with an event;
      one device-to-host copy of the gathered sweep.

    `cache`: dict owned by the caller that keeps the device block / gather buffers between
    sweeps of the same shape."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    az = np.asarray(azimuths, dtype=np.float64)
    el = np.asarray(elevations, dtype=np.float64)
    n = len(az)
    lo, hi, per = shard_bounds(n, world, rank)
    lay = BlockLayout(fields, per, n_gates)
    cache = {} if cache is None else cache
    key = ('blk', lay.nbytes, world)
    if key not in cache:
        cache.clear()
        cache[key] = (torch.zeros(lay.nbytes, dtype=torch.uint8, device=device),
                      torch.empty(world * lay.nbytes, dtype=torch.uint8, device=device),
                      torch.empty(world * lay.nbytes, dtype=torch.uint8).pin_memory()
                      if torch.cuda.is_available() else torch.empty(world * lay.nbytes, dtype=torch.uint8))
    block, gathered, host = cache[key]
    lib_stream = torch.cuda.ExternalStream(stream_ptr, device=device)
    cur = torch.cuda.current_stream(device)
    lib_stream.wait_stream(cur)                 # the previous gather has consumed the block
    if hi > lo:
        base = block.data_ptr()
        run_block(az[lo:hi], el[lo:hi], {k: base + lay.offsets[k] for k, _ in lay.fields})
    cur.wait_stream(lib_stream)                 # gather behind the kernels
    dist.all_gather_into_tensor(gathered, block, group=group)
    host.copy_(gathered, non_blocking=True)
    cur.synchronize()
    return lay.assemble(host.numpy(), n, world)


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


def simulate_sharded_volume_device(run_block, stream_ptr, sweeps, fields, n_gates, device, group=None,
                                   cache=None, host_block=None):
    """Device-resident volume sharding (the product path of RadarOperator(distributed=True)):
    `run_block(az, el, {field: device pointer})` queues ONE launch sequence for this rank's rays of
    all sweeps on the library stream; one all_gather_into_tensor of the device blocks (RCCL over
    xGMI) behind it; the gathered rows put into scan order on the device; one device-to-host copy.
    `host_block(nbytes) -> (uint8 array, holder)`: where the result lives (the operator's PinnedPool; the
    returned arrays are views of it).  -> list of per-sweep dicts."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lay = VolumeLayout(fields, [len(a) for a, _ in sweeps], world, n_gates)
    nb = lay.block.nbytes
    cache = {} if cache is None else cache
    nf = lay.final.nbytes
    key = ('blk', nb, nf, world, tuple(lay.n_rays))
    if key not in cache:
        cache.clear()
        rk, rw = lay.source_rows()
        cache[key] = (torch.zeros(nb, dtype=torch.uint8, device=device),
                      torch.empty(world * nb, dtype=torch.uint8, device=device),
                      torch.empty(nf, dtype=torch.uint8, device=device),
                      (torch.from_numpy(rk).to(device), torch.from_numpy(rw).to(device)))
    block, gathered, final, index = cache[key]
    lib_stream = torch.cuda.ExternalStream(stream_ptr, device=device)
    cur = torch.cuda.current_stream(device)
    lib_stream.wait_stream(cur)                 # the previous gather has consumed the block
    az, el = lay.local_rays(rank, sweeps)
    if len(az):
        base = block.data_ptr()
        run_block(az, el, {k: base + lay.block.offsets[k] for k, _ in lay.block.fields})
    cur.wait_stream(lib_stream)                 # gather behind the kernels
    dist.all_gather_into_tensor(gathered, block, group=group)
    # rows -> their place in the scan ON THE DEVICE (one row gather per field), then ONE copy into a
    # fresh page-locked block that the returned arrays are views of (no host-side concatenation: that was
    # 70 MB read + written per C4 volume on every rank, ten times the rank's kernel time)
    lay.assemble_on_device(gathered, final, index)
    if host_block is not None:
        arr = host_block(nf)[0][:nf]            # (the block lives as long as a view of it does)
    else:
        arr = np.empty(nf, dtype=np.uint8)
    torch.from_numpy(arr).copy_(final, non_blocking=True)
    cur.synchronize()
    return lay.views_of_final(arr)
