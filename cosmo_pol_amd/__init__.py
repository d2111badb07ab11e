"""cosmo_pol_amd -- MI355X-native polarimetric radar forward operator.

Drop-in for the per-beam / per-gate hot path of wolfidan/cosmo_pol behind the
reference's RadarOperator API (cosmo_pol/radar_operator.py:46).  All per-gate
work runs in hand-written HIP kernels (cosmo_pol_amd/csrc) reached through the
C ABI declared in include/cosmo_pol_amd.h; there is no CPU fallback.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name == "RadarOperator":
        from .radar_operator import RadarOperator
        return RadarOperator
    if name == "bind_to_device_numa_node":      # one process per GPU: run it on the GPU's own socket
        from ._native import bind_to_device_numa_node
        return bind_to_device_numa_node
    raise AttributeError(name)
