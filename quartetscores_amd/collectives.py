"""torch.distributed collectives on device tensors, whatever the backend.

Production: backend "nccl" (= RCCL over xGMI), one rank per GPU: the calls go straight through.
Rehearsal (`QS_BENCH_BACKEND=gloo`, tests/test_gpu_bench_launcher.py): several ranks SHARE one GPU -- RCCL refuses two ranks on one
device -- and talk over gloo; device tensors are staged through the host for the call (synchronous: async_op returns None). That runs
the N > 1 control flow of bench.py and distributed.py -- both multi-GPU modes, the wire formats, the sharded scoring -- with a real
world size on the one-GPU box, where no multi-GPU node is available.
"""
from __future__ import annotations


def _staged(t, group) -> bool:
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce(t, op=None, group=None, async_op=False):
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if not _staged(t, group):
        return dist.all_reduce(t, op=op, group=group, async_op=async_op)
    h = t.cpu()
    dist.all_reduce(h, op=op, group=group)
    t.copy_(h)
    return None


def reduce_scatter_tensor(recv, send, op=None, group=None, async_op=False):
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if not _staged(send, group):
        return dist.reduce_scatter_tensor(recv, send, op=op, group=group, async_op=async_op)
    h = send.cpu()
    out = recv.cpu()
    dist.reduce_scatter_tensor(out, h, op=op, group=group)
    recv.copy_(out)
    return None


def all_gather(parts, t, group=None):
    import torch.distributed as dist
    if not _staged(t, group):
        return dist.all_gather(parts, t, group=group)
    hp = [p.cpu() for p in parts]
    dist.all_gather(hp, t.cpu(), group=group)
    for p, h in zip(parts, hp):
        p.copy_(h)
    return None
