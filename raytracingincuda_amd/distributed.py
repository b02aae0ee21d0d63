"""Multi-GPU plumbing: one process per GPU, the image sharded by interleaved row strips
(rtiow_set_shard), one gather of the strips to rank 0 over torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-GPU (main.cu:81 cudaSetDevice(0)); there is no exchange during a
render -- pixels are independent and the RNG streams are keyed by the GLOBAL pixel index --
so the only collective is the final gather of the disjoint strips.
"""
import numpy as np
import torch
import torch.distributed as dist

from .api import shard_rows


def max_local_rows(height, nranks, strip_rows=8):
    return max(len(shard_rows(height, r, nranks, strip_rows)) for r in range(nranks))


class StripGather:
    """Gathers every rank's [local_rows, W, 3] strips into the full [H, W, 3] image on dst.

    Buffers are sized once (ranks may own different row counts, so shards are padded to the
    largest); `row_index` is the de-interleave permutation applied on dst after the gather.
    """

    def __init__(self, width, height, rank, nranks, strip_rows=8, dtype=torch.float32, device="cpu", dst=0, group=None,
                 stage_via_cpu=False, always_collective=False):
        self.width, self.height, self.rank, self.nranks = width, height, rank, nranks
        self.strip_rows, self.dst, self.group = strip_rows, dst, group
        # rehearsal only: a gloo group cannot gather device tensors, so bounce through the host
        self.stage_via_cpu = stage_via_cpu and str(device) != "cpu"
        # always_collective: run dist.gather even in a one-rank group (exercises the backend -- RCCL for
        # device tensors -- on a one-GPU box; without it a lone rank just de-interleaves locally)
        self.always_collective = always_collective
        self.rows = shard_rows(height, rank, nranks, strip_rows)
        self.pad_rows = max_local_rows(height, nranks, strip_rows)
        self.send = torch.zeros((self.pad_rows, width, 3), dtype=dtype, device=device)
        self.recv = None
        self.full = None
        if rank == dst:
            self.recv = [torch.zeros_like(self.send) for _ in range(nranks)]
            self.full = torch.zeros((height, width, 3), dtype=dtype, device=device)
            self.index = [torch.as_tensor(shard_rows(height, r, nranks, strip_rows).astype(np.int64), device=device)
                          for r in range(nranks)]

    def local_view(self):
        """[local_rows, W, 3] view the renderer writes into (rtiow_bind_framebuffer)."""
        return self.send[: len(self.rows)]

    def gather(self):
        if self.nranks == 1 and not self.always_collective:
            self.full.index_copy_(0, self.index[0], self.send[: len(self.rows)])
            return self.full
        if self.stage_via_cpu:
            send = self.send.cpu()
            recv = [torch.empty_like(send) for _ in range(self.nranks)] if self.rank == self.dst else None
            dist.gather(send, recv, dst=self.dst, group=self.group)
            if self.rank == self.dst:
                for r in range(self.nranks):
                    self.recv[r].copy_(recv[r])
        else:
            dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        for r in range(self.nranks):
            idx = self.index[r]
            self.full.index_copy_(0, idx, self.recv[r][: idx.numel()])
        return self.full
