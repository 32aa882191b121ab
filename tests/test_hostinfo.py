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


def test_local_world_size_prefers_per_node_counts_over_the_global_world_size(monkeypatch):
    """dist.initialize() writes the GLOBAL world size into WORLD_SIZE, also for multi-node SLURM launches; the per-node
    counts win, and WORLD_SIZE alone is capped by the node's GPU count (no GPUs here: taken as is)."""
    from declip_amd import hostinfo
    for k in ("LOCAL_WORLD_SIZE", "SLURM_NTASKS_PER_NODE", "SLURM_TASKS_PER_NODE", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert hostinfo.local_world_size() == 1
    monkeypatch.setenv("WORLD_SIZE", "64")
    monkeypatch.setenv("SLURM_NTASKS_PER_NODE", "8(x8)")
    assert hostinfo.local_world_size() == 8
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert hostinfo.local_world_size() == 4
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    monkeypatch.delenv("SLURM_NTASKS_PER_NODE")
    monkeypatch.setattr(hostinfo, "_node_gpu_count", lambda: 0)
    assert hostinfo.local_world_size() == 64          # one-node fallback, no GPU listed by the driver: taken as is
    # the cap is the NODE's GPU count (sysfs), not what this rank may see -- one visible GPU per rank must not turn into
    # "this rank owns the node's cores" -- and the helper does not touch torch.cuda
    monkeypatch.setattr(hostinfo, "_node_gpu_count", lambda: 8)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3")
    assert hostinfo.local_world_size() == 8
    monkeypatch.setenv("SLURM_TASKS_PER_NODE", "4(x16)")
    assert hostinfo.local_world_size() == 4


def test_node_gpu_count_reads_sysfs_without_the_runtime():
    import inspect
    from declip_amd import hostinfo
    assert isinstance(hostinfo._node_gpu_count(), int) and hostinfo._node_gpu_count() >= 0
    assert "torch" not in inspect.getsource(hostinfo._node_gpu_count).replace("torch.cuda / HIP", "")
