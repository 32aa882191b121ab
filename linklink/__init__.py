"""`linklink` drop-in (reference: /root/reference/linklink/__init__.py:13-71) backed by RCCL through
torch.distributed; see declip_amd/dist.py for the MI355X-first collectives the engine itself uses."""
import functools

import torch
import torch.distributed as dist

from declip_amd.dist import barrier, get_local_rank, get_rank, get_world_size, initialize  # noqa: F401

allreduce = dist.all_reduce
allgather = dist.all_gather
broadcast = dist.broadcast
init_process_group = dist.init_process_group
allreduce_async = functools.partial(dist.all_reduce, async_op=True)


def synchronize():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
