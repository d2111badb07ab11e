"""Host placement of a one-process-per-GPU rank (cosmo_pol_amd/_native.py: bind_to_device_numa_node).
The PCI bus id comes from the library on a GPU box (tests/test_gpu_boundary.py); here the sysfs side."""
import os

import pytest

from cosmo_pol_amd import _native as N


def test_cpulist_parser():
    assert N.parse_cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    assert N.parse_cpulist('5') == {5}
    assert N.parse_cpulist('') == set()


def _fake_sysfs(tmp_path, bdf, node, cpulist):
    d = tmp_path / bdf
    d.mkdir()
    (d / 'numa_node').write_text('%d\n' % node)
    (d / 'local_cpulist').write_text(cpulist + '\n')
    return str(tmp_path)


@pytest.mark.skipif(not hasattr(os, 'sched_setaffinity'), reason='no affinity calls on this platform')
def test_bind_narrows_but_never_widens(tmp_path, monkeypatch):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip('one core')
    monkeypatch.delenv('CPOL_NUMA_BIND', raising=False)
    half = allowed[:len(allowed) // 2]
    outside = max(allowed) + 1
    sysfs = _fake_sysfs(tmp_path, '0000:75:00.0', 1, ','.join(map(str, half + [outside])))
    try:
        info = N.bind_to_device_numa_node(0, sysfs=sysfs, pci='0000:75:00.0')
        assert info['node'] == 1 and info['bound'] == len(half)
        assert os.sched_getaffinity(0) == set(half)             # (the core outside the allowed set is not added)
        again = N.bind_to_device_numa_node(0, sysfs=sysfs, pci='0000:75:00.0')
        assert again['bound'] == 0                               # already there
    finally:
        os.sched_setaffinity(0, allowed)


@pytest.mark.skipif(not hasattr(os, 'sched_setaffinity'), reason='no affinity calls on this platform')
def test_bind_leaves_affinity_alone_when_told_or_unknown(tmp_path, monkeypatch):
    allowed = os.sched_getaffinity(0)
    one = sorted(allowed)[0]
    sysfs = _fake_sysfs(tmp_path, '0000:05:00.0', 0, str(one))
    monkeypatch.setenv('CPOL_NUMA_BIND', '0')
    assert N.bind_to_device_numa_node(0, sysfs=sysfs, pci='0000:05:00.0')['bound'] == 0
    assert os.sched_getaffinity(0) == allowed
    monkeypatch.delenv('CPOL_NUMA_BIND')
    # no sysfs entry for the device (a VM without PCI topology), node -1 with an empty list, disjoint cores
    assert N.bind_to_device_numa_node(0, sysfs=str(tmp_path), pci='0000:99:00.0')['bound'] == 0
    sysfs2 = _fake_sysfs(tmp_path, '0000:06:00.0', -1, '')
    info = N.bind_to_device_numa_node(0, sysfs=sysfs2, pci='0000:06:00.0')
    assert info['node'] is None and info['bound'] == 0
    sysfs3 = _fake_sysfs(tmp_path, '0000:07:00.0', 1, str(max(allowed) + 5))
    assert N.bind_to_device_numa_node(0, sysfs=sysfs3, pci='0000:07:00.0')['bound'] == 0
    assert os.sched_getaffinity(0) == allowed
