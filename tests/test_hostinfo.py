"""hostinfo: the CPU quota of a container bounds the host threads (declip_amd/hostinfo.py)."""
import builtins
import io

import torch

from declip_amd import hostinfo


def _fake_open(files):
    real = builtins.open

    def f(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise OSError(path)
            return io.StringIO(files[path])
        if str(path).startswith("/sys/fs/cgroup"):
            raise OSError(path)
        return real(path, *a, **k)
    return f


def test_cgroup_v2_quota(monkeypatch):
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert hostinfo._cgroup_quota() == 16.0
    assert hostinfo.usable_cores() <= 16


def test_cgroup_v2_unlimited_and_v1(monkeypatch):
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert hostinfo._cgroup_quota() is None
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "250000\n",
                                                       "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert hostinfo._cgroup_quota() == 2.5
    assert hostinfo.usable_cores() <= 2


def test_limit_never_raises_the_thread_count(monkeypatch):
    prev = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        assert hostinfo.limit_host_threads() == 1
        monkeypatch.setattr(hostinfo, "usable_cores", lambda physical=True: 1)
        torch.set_num_threads(prev)
        assert hostinfo.limit_host_threads() == 1 and torch.get_num_threads() == 1
    finally:
        torch.set_num_threads(prev)


def test_prefetcher_counts_caption_rows_without_torch_cpu_reductions():
    from declip_amd.prefetch import DataPrefetcher
    ids = torch.zeros(4, 7, dtype=torch.int64)
    ids[0, 2] = ids[1, 6] = ids[2, 0] = ids[3, 3] = 99
    pf = DataPrefetcher(iter([{"captions": ids}]), device="cpu")
    b = pf.next()
    assert b["captions"]._dh_rows == (b["captions"]._version, 3 + 7 + 1 + 4)
    pf.close()
