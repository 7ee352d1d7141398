"""One process per GPU; gradient exchange over RCCL (torch.distributed backend 'nccl' on ROCm) / gloo on CPU.

The reference is single-process single-GPU (utils/trainer.py:273-278); utterance-level data parallelism is the
build's addition: rank r trains on its own minibatch shard, weights are broadcast from rank 0 at
start, and one all-reduce of the flat gradient buffer per step keeps replicas identical.
"""
import os

import torch
import torch.distributed as td


class Dist(object):
    def __init__(self, backend=None):
        self.world_size = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.enabled = self.world_size > 1
        if self.enabled:
            fresh = not td.is_initialized()
            if fresh:
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                os.environ.setdefault('MASTER_PORT', '29500')
                if backend is None:
                    # AMS_DIST_BACKEND=gloo lets several ranks share ONE GPU (plumbing tests on a 1-GPU box; RCCL refuses that)
                    backend = os.environ.get('AMS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            if torch.cuda.is_available():
                ndev = max(1, torch.cuda.device_count())
                if fresh:
                    self.local_rank %= ndev
                    torch.cuda.set_device(self.local_rank)
            if fresh:
                td.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
            if torch.cuda.is_available() and os.environ.get('AMS_LSTM_RING') is None and self._ranks_on_this_node() > ndev:
                # several ranks time-share one GPU (plumbing tests on a 1-GPU box): the ring recurrence needs every workgroup of a
                # launch resident at once, which two processes' kernels on the same CUs cannot promise each other (measured: 2 of
                # 6 two-rank runs gave up a bounded wait) -- such ranks take the per-step recurrence kernels.  Checked whether or not
                # the caller had already initialised the process group; said once.
                from . import ops
                if ops.LSTM_RING != '0' and self.rank == 0:
                    print('[ams] %d ranks share %d GPU(s) on this node: per-step recurrence kernels instead of the rings '
                          '(AMS_LSTM_RING=1 keeps the rings)' % (self._ranks_on_this_node(), ndev))
                ops.LSTM_RING = '0'

    def _ranks_on_this_node(self):
        """Ranks of this job on this host: LOCAL_WORLD_SIZE when the launcher set it (torchrun); else counted by host name over the
        process group (srun / mpirun launches, where WORLD_SIZE spans nodes); 1 -- no sharing assumed -- when neither is possible."""
        n = getattr(self, '_node_ranks', None)
        if n is not None:
            return n
        v = os.environ.get('LOCAL_WORLD_SIZE')
        if v is not None:
            n = int(v)
        else:
            n = 1
            try:
                import socket
                names = [None] * self.world_size
                td.all_gather_object(names, socket.gethostname())
                n = sum(1 for h in names if h == socket.gethostname())
            except Exception:
                n = 1
        self._node_ranks = n
        return n

    def backend(self):
        return td.get_backend() if self.enabled else 'none'

    def all_reduce_sum(self, t):
        if self.enabled:
            td.all_reduce(t, op=td.ReduceOp.SUM)
        return t

    def all_reduce_max(self, t):
        if self.enabled:
            td.all_reduce(t, op=td.ReduceOp.MAX)
        return t

    def broadcast(self, t, src=0):
        if self.enabled:
            if t.is_contiguous():
                td.broadcast(t, src)
            else:                                   # twin-interleaved BLSTM variables are strided views
                tmp = t.contiguous()
                td.broadcast(tmp, src)
                t.copy_(tmp)
        return t

    def broadcast_object(self, obj, src=0):
        """Small picklable host object (run id, paths) from rank `src` to everyone."""
        if self.enabled:
            box = [obj if self.rank == src else None]
            td.broadcast_object_list(box, src=src)
            return box[0]
        return obj

    def all_gather_object(self, obj):
        """[obj of rank 0, ..., obj of rank world - 1] (small picklable host objects)."""
        if not self.enabled:
            return [obj]
        out = [None] * self.world_size
        td.all_gather_object(out, obj)
        return out

    def barrier(self):
        if self.enabled:
            td.barrier()

    def shard(self, n):
        """Contiguous utterance shard [lo, hi) of a global batch of n for this rank."""
        per = n // self.world_size
        return self.rank * per, (self.rank + 1) * per
