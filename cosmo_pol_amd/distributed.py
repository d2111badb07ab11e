"""Multi-GPU sweeps: one process per GPU, rays sharded, one collective per sweep.

The reference farms the radials of a sweep out to a fork pool and collects
pickled Radial objects over pipes (cosmo_pol/radar_operator.py:402-432).  Here
the rays of a sweep (or of a whole volume) are split into contiguous blocks,
one per rank; model cube and lookup tables are replicated in every GPU's HBM
(staged once); every rank runs its block through the C ABI and the output
slabs are collected with a single all-gather (RCCL over xGMI when the backend
is "nccl", gloo on CPU for tests).  Radials never interact (all coupling -
PHIDP / attenuation scans, sub-beam sums - is inside one radial), so the
gathered result is bitwise identical to the single-GPU result.
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


def gather_rows(local, n_rays, group=None):
    """All-gathers equally padded per-rank row blocks [per, row_bytes] (uint8
    torch tensors, CPU or GPU) and trims the padding -> [n_rays, row_bytes] on
    every rank.  ONE collective per sweep."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per, nb = local.shape
    flat = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, local.contiguous().view(-1), group=group)
    return flat.view(world * per, nb)[:n_rays]


def simulate_sharded(simulate, azimuths, elevations, fields, n_gates, device=None, group=None):
    """Runs `simulate(az_block, el_block) -> {field: [n_block, n_gates] array}`
    on this rank's contiguous block of rays and gathers all ranks' results.

    `fields`: list of (name, dtype).  All outputs of one ray are packed into
    one byte row, so the whole sweep needs a single all-gather whatever the
    mix of float32 / float64 fields.  `simulate` is RadarOperator.simulate_rays
    on a GPU rank; tests inject a CPU stand-in to exercise the sharding and the
    collective with gloo."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    az = np.asarray(azimuths, dtype=np.float64)
    el = np.asarray(elevations, dtype=np.float64)
    n = len(az)
    lo, hi, per = shard_bounds(n, world, rank)
    fields = [(k, np.dtype(dt)) for k, dt in fields]
    offs, row_bytes = [], 0
    for k, dt in fields:
        row_bytes = -(-row_bytes // 8) * 8
        offs.append(row_bytes)
        row_bytes += n_gates * dt.itemsize
    row_bytes = -(-row_bytes // 8) * 8
    rows = np.zeros((per, row_bytes), dtype=np.uint8)
    if hi > lo:
        res = simulate(az[lo:hi], el[lo:hi])
        for (k, dt), o in zip(fields, offs):
            blk = np.ascontiguousarray(res[k], dtype=dt).reshape(hi - lo, n_gates)
            rows[:hi - lo, o:o + n_gates * dt.itemsize] = blk.view(np.uint8).reshape(hi - lo, -1)
    t = torch.from_numpy(rows)
    if device is not None:
        t = t.to(device)
    full = gather_rows(t, n, group).cpu().numpy()
    out = {}
    for (k, dt), o in zip(fields, offs):
        out[k] = np.ascontiguousarray(full[:, o:o + n_gates * dt.itemsize]).view(dt).reshape(n, n_gates)
    return out
