"""Per-local-rank CPU affinity (cabinet_amd.ddp.set_rank_affinity): the GPU -> NUMA-node -> core-list lookup reads sysfs only
(no HIP call: it runs before the runtime starts its threads), so it is testable against a synthetic tree."""
import os

from cabinet_amd import ddp


def _fake_sysfs(root, gpus, extra_non_gpu=True):
    for i, (pci, node, cpulist) in enumerate(gpus):
        real = os.path.join(root, "pci", pci)
        os.makedirs(real)
        for name, val in (("vendor", "0x1002"), ("numa_node", str(node)), ("local_cpulist", cpulist), ("mem_info_vram_total", "1")):
            open(os.path.join(real, name), "w").write(val + "\n")
        os.makedirs(os.path.join(root, "drm", f"card{i + 1}"))
        os.symlink(real, os.path.join(root, "drm", f"card{i + 1}", "device"))
    if extra_non_gpu:  # a BMC display function: another vendor, must be skipped
        real = os.path.join(root, "pci", "0000:01:00.0")
        os.makedirs(real)
        open(os.path.join(real, "vendor"), "w").write("0x1a03\n")
        os.makedirs(os.path.join(root, "drm", "card0"))
        os.symlink(real, os.path.join(root, "drm", "card0", "device"))
    os.makedirs(os.path.join(root, "drm", "renderD128"))
    return os.path.join(root, "drm")


def test_cpulist_parser():
    assert ddp._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert ddp._parse_cpulist("") == set()


def test_gpu_to_numa_cores(tmp_path):
    # cards registered out of PCI order: devices are ordered by PCI address, as the runtime enumerates them
    sysfs = _fake_sysfs(str(tmp_path), [("0000:85:00.0", 1, "8-15,24-31"), ("0000:05:00.0", 0, "0-7,16-23")])
    assert ddp.gpu_local_cpus(0, sysfs, visible="") == (set(range(0, 8)) | set(range(16, 24)), 0)
    assert ddp.gpu_local_cpus(1, sysfs, visible="") == (set(range(8, 16)) | set(range(24, 32)), 1)
    assert ddp.gpu_local_cpus(0, sysfs, visible="1") == (set(range(8, 16)) | set(range(24, 32)), 1)  # HIP_VISIBLE_DEVICES=1
    assert ddp.gpu_local_cpus(2, sysfs, visible="") == (None, None)
    assert ddp.gpu_local_cpus(0, sysfs, visible="GPU-abcdef") == (None, None)  # UUID lists: do not guess
    assert ddp.gpu_local_cpus(0, str(tmp_path / "nowhere")) == (None, None)


def test_visibility_lists_compose_and_cuda_alias_is_honoured(tmp_path, monkeypatch):
    """ROCR_VISIBLE_DEVICES filters what the runtime sees, HIP_VISIBLE_DEVICES (alias: CUDA_VISIBLE_DEVICES) indexes into THAT
    (ADVICE r04); the chosen card's PCI address is remembered for the cross-check against the runtime."""
    gpus = [("0000:05:00.0", 0, "0-3"), ("0000:45:00.0", 0, "4-7"), ("0000:85:00.0", 1, "8-11"), ("0000:c5:00.0", 1, "12-15")]
    sysfs = _fake_sysfs(str(tmp_path), gpus)
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2,3,0")
    assert ddp.gpu_local_cpus(0, sysfs) == (set(range(8, 12)), 1)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,2")          # of (2, 3, 0): cards 3 and 0
    assert ddp.gpu_local_cpus(0, sysfs) == (set(range(12, 16)), 1) and ddp.gpu_local_cpus.pci == "0000:c5:00.0"
    assert ddp.gpu_local_cpus(1, sysfs) == (set(range(0, 4)), 0) and ddp.gpu_local_cpus.pci == "0000:05:00.0"
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "1")           # HIP honours the CUDA spelling too
    assert ddp.gpu_local_cpus(0, sysfs) == (set(range(4, 8)), 0)


def test_set_rank_affinity_pins_and_never_fails(tmp_path, monkeypatch):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        import pytest

        pytest.skip("needs two allowed cores")
    half = allowed[: len(allowed) // 2]
    sysfs = _fake_sysfs(str(tmp_path), [("0000:05:00.0", 0, ",".join(map(str, half)))])
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    try:
        info = ddp.set_rank_affinity(0, sysfs)
        assert info == {"set": True, "numa_node": 0, "cpus": len(half)} and os.sched_getaffinity(0) == set(half)
    finally:
        os.sched_setaffinity(0, allowed)
    assert ddp.set_rank_affinity(5, sysfs)["set"] is False and os.sched_getaffinity(0) == set(allowed)   # no such GPU
    monkeypatch.setenv("CABINET_NO_AFFINITY", "1")
    assert ddp.set_rank_affinity(0, sysfs) == {"set": False, "why": "disabled"}


def test_only_set_rank_affinity_records_the_card_and_an_empty_visibility_list_means_no_devices(tmp_path, monkeypatch):
    """gpu_local_cpus() is a pure lookup (the card address travels as an attribute of the function, the module global
    that check_affinity_device reads is written by set_rank_affinity alone), and an EMPTY HIP_VISIBLE_DEVICES -- "no devices" to the
    runtime -- is not treated as "unset"."""
    sysfs = _fake_sysfs(str(tmp_path), [("0000:05:00.0", 0, "0-3"), ("0001:05:00.0", 1, "4-7")])
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    before = ddp._AFFINITY_PCI
    assert ddp.gpu_local_cpus(1, sysfs) == (set(range(4, 8)), 1) and ddp.gpu_local_cpus.pci == "0001:05:00.0"
    assert ddp._AFFINITY_PCI == before   # two PCI domains, same bus byte: the address keeps its domain for the cross-check
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert ddp.gpu_local_cpus(0, sysfs) == (None, None)
