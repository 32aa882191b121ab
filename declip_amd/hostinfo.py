"""How many host cores this process may really use.

A training process sees every core of the node (`os.cpu_count()`, the affinity mask), but a container's CPU quota (cgroup
`cpu.max` / `cpu.cfs_quota_us`) can be far smaller -- the MI355X boxes of this project: 256 hardware threads visible, a quota of
16.  torch sizes its OpenMP teams by the visible cores; a 128-thread team that spin-waits after every small CPU operation burns
the quota in a few milliseconds and the kernel THROTTLES the whole process, the thread that enqueues GPU work included
(measured: the bf16 CLIP step 23.5 -> 44-63 ms as soon as one [512, 77] argmax per step ran on the host; 49 of 362 scheduler
periods throttled; profiles/r03_pipeline_host_threads.txt).  `limit_host_threads()` clamps torch's intra-op threads to what the
quota grants; the training entry points (bench.py, solver.ClipSolver) call it before anything else.
"""
import os

__all__ = ["usable_cores", "limit_host_threads"]


def _cgroup_quota():
    """cores granted by the CPU controller of this process' cgroup (v2, then v1), or None when unlimited / unreadable."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                       # v2: "<quota|max> <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max" and int(period) > 0:
            return int(quota) / int(period)
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            quota = int(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = int(fh.read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def usable_cores(physical=True):
    """min(cores of the affinity mask [physical ones: hyper-thread siblings counted once], the cgroup CPU quota), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    if physical:
        try:
            import psutil
            phys = psutil.cpu_count(logical=False)
            if phys:
                n = min(n, int(phys))
        except Exception:
            pass
    quota = _cgroup_quota()
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def _node_gpu_count():
    """GPUs of this node as the kernel driver lists them (KFD topology nodes with SIMDs; else the DRM render nodes) -- read from sysfs,
    NOT through torch.cuda / HIP: this helper runs before worker processes are forked and must not initialise the runtime."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(f):
                if line.startswith("simd_count ") and int(line.split()[1]) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    return n if n > 0 else len(glob.glob("/dev/dri/renderD*"))


def local_world_size():
    """ranks of this job on THIS node: they share the node's quota.  torchrun's LOCAL_WORLD_SIZE, else SLURM_NTASKS_PER_NODE /
    SLURM_TASKS_PER_NODE, else WORLD_SIZE as the one-node fallback -- capped by the node's GPU count, because dist.initialize() writes
    the GLOBAL world size into WORLD_SIZE also for multi-node SLURM launches (64 ranks on 8 nodes are 8 per node, not 64).  The GPU
    count is the NODE's (sysfs), not what this rank may see: a launcher that hands every rank one visible GPU (*_VISIBLE_DEVICES)
    must not make each rank believe it owns the node's cores."""
    def _int(k):
        v = os.environ.get(k, "")
        v = v.split("(")[0].split(",")[0]           # SLURM_NTASKS_PER_NODE may read "8(x4)"
        return int(v) if v.isdigit() and int(v) > 0 else None
    for k in ("LOCAL_WORLD_SIZE", "SLURM_NTASKS_PER_NODE", "SLURM_TASKS_PER_NODE"):
        n = _int(k)
        if n:
            return n
    n = _int("WORLD_SIZE")
    if n:
        gpus = _node_gpu_count()
        return min(n, gpus) if gpus > 0 else n
    return 1


def limit_host_threads(reserve=0):
    """Clamp torch's intra-op CPU threads to this rank's share of usable_cores() - reserve (never raises them); returns the count
    now in force.  Eight ranks of one node share the node's quota: each gets an eighth of it, not all of it."""
    import torch
    n = max(1, (usable_cores() - reserve) // local_world_size())
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
