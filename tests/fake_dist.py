"""A thread-backed stand-in for torch.distributed: P "ranks" are P threads of ONE process.

TEST INFRASTRUCTURE.  It lets the SPMD orchestration of humid_amd/sharded.py run with the real HIP
stage ops for several ranks on the single GPU of a test box (RCCL refuses two ranks on one
device).  Only the collectives sharded.py uses are provided; semantics follow torch.distributed."""
import threading

import torch


class FakeWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class FakeDist:
    class ReduceOp:
        SUM = "sum"

    def __init__(self, world: FakeWorld, rank: int):
        self.w, self.rank = world, rank

    def get_world_size(self):
        return self.w.world

    def get_rank(self):
        return self.rank

    def _exchange(self, obj):
        if obj is not None and torch.is_tensor(obj) and obj.is_cuda:
            torch.cuda.synchronize()
        self.w.slots[self.rank] = obj
        self.w.barrier.wait()
        allv = list(self.w.slots)
        self.w.barrier.wait()
        return allv

    def all_reduce(self, t, op=None):
        allv = self._exchange(t.clone())
        t.copy_(torch.stack(allv).sum(dim=0).to(t.dtype))

    def all_gather_into_tensor(self, out, inp):
        allv = self._exchange(inp.clone())
        out.copy_(torch.cat([x.reshape(-1) for x in allv]).view_as(out))

    def all_gather(self, outs, inp):
        allv = self._exchange(inp.clone())
        for o, x in zip(outs, allv):
            o.copy_(x)

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        allv = self._exchange((inp.clone(), list(input_split_sizes)))
        if inp.is_cuda:
            torch.cuda.synchronize()
        parts = []
        for src, (t, splits) in enumerate(allv):
            off = sum(splits[:self.rank])
            parts.append(t[off:off + splits[self.rank]])
            assert splits[self.rank] == output_split_sizes[src]
        out.copy_(torch.cat(parts))

    def reduce_scatter_tensor(self, out, inp, op=None):
        allv = self._exchange(inp.clone())
        tot = torch.stack([x.to(torch.int64) for x in allv]).sum(dim=0)
        n = out.numel()
        out.copy_(tot[self.rank * n:(self.rank + 1) * n].to(out.dtype))
