"""gloo transport for the CPU multi-process tests: the duck-typed communicator surface of
evo_amd.utils.parallel (rank, size, allreduce, allreduce_array, bcast, Barrier) over an initialised
torch.distributed group.  Test infrastructure only: the product path (evo_amd/) never imports PyTorch."""
import numpy as np


class TorchDistComm:
    """Sums over an initialised torch.distributed group.  Host tensors (gloo)."""
    device_reduces = False

    def __init__(self, group=None):
        import torch.distributed as dist  # lazy: PyTorch is plumbing for this transport only
        self._dist = dist
        self._group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def allreduce_array(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        return t.numpy()

    def allreduce(self, value, op=None):
        if isinstance(value, np.ndarray):
            return self.allreduce_array(value).reshape(value.shape)
        out = self.allreduce_array(np.array([value], dtype=np.float64))[0]
        return type(value)(out) if isinstance(value, (int, np.integer)) else float(out)

    def bcast(self, value, root=0):
        box = [value]
        self._dist.broadcast_object_list(box, src=root, group=self._group)
        return box[0]

    def Barrier(self):
        self._dist.barrier(group=self._group)
