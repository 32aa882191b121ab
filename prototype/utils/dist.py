"""reference: prototype/utils/dist.py:8-126."""
import functools

from declip_amd.dist import DistModule, broadcast_object, initialize  # noqa: F401
import linklink as link


def dist_init(method="slurm", device_id=0):
    initialize("nccl")
    return link.get_rank(), link.get_world_size()


def link_dist(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        dist_init()
        try:
            return func(*args, **kwargs)
        finally:
            link.finalize()
    return wrapper
