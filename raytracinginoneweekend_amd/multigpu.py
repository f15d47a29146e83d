"""Row-tile sharding of one frame over the ranks of a torch.distributed job (one process per GPU).

The j loop of render (raytracer.nim:36) is the natural cut: every image row is independent once
the random stream is keyed on (pixel, sample).  Rows are dealt to ranks in interleaved tiles
(tile t of `tile_rows` rows goes to rank t mod world), because sky rows cost one ray segment per
path and ground rows several: contiguous bands would leave the ranks that own the sky idle.
Each rank renders its rows on its own GPU; the only exchange is one gather of 8-bit rows to
rank 0 (RCCL when the tensors live on GPUs, gloo on CPU for tests).
"""
import numpy as np


def tile_rows(height, world_size, rank, tile=4):
    """Image rows (0 = top) owned by `rank`: interleaved tiles of `tile` rows."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    rows = np.arange(height, dtype=np.int32)
    return rows[(rows // tile) % world_size == rank]


def padded_rows(height, world_size, tile=4):
    """Rows per rank after padding every rank to the same count (all_gather needs equal shapes)."""
    return max(len(tile_rows(height, world_size, r, tile)) for r in range(world_size))


def gather_image(local_rgb, height, width, world_size, rank, tile=4, group=None, dst=0):
    """Gather per-rank row tiles into the full frame on rank `dst`.

    local_rgb: uint8 torch tensor [n_local_rows, width, 3] on this rank's device, rows in
    tile_rows(...) order.  Returns the uint8 [height, width, 3] tensor on `dst`, None elsewhere.
    One collective: all ranks pad to the same row count and all_gather (equal-size contiguous
    messages; the per-rank payload at 1200x800 over 8 GPUs is 360 KB, so the step is latency
    bound and the simplest collective is the right one).
    """
    import torch
    import torch.distributed as dist

    if world_size == 1:
        return local_rgb
    n_pad = padded_rows(height, world_size, tile)
    send = torch.zeros((n_pad, width, 3), dtype=torch.uint8, device=local_rgb.device)
    send[: local_rgb.shape[0]] = local_rgb
    recv = [torch.empty_like(send) for _ in range(world_size)]
    dist.all_gather(recv, send, group=group)
    if rank != dst:
        return None
    full = torch.empty((height, width, 3), dtype=torch.uint8, device=local_rgb.device)
    for r in range(world_size):
        rows = torch.from_numpy(tile_rows(height, world_size, r, tile).astype(np.int64)).to(local_rgb.device)
        full[rows] = recv[r][: len(rows)]
    return full


class TileGatherer:
    """gather_image with every buffer allocated once: the per-step cost is one collective and, on
    the destination rank, one row scatter.  `local` is the tensor the renderer writes into (rows of
    this rank first, padding rows after them)."""

    def __init__(self, height, width, world_size, rank, device, tile=4, group=None, dst=0, always_collective=False):
        import torch
        self.height, self.width, self.world, self.rank, self.tile = height, width, world_size, rank, tile
        self.group, self.dst = group, dst
        # a one-rank job normally skips the collective; always_collective runs it anyway (a rehearsal of the
        # multi-GPU path on one GPU: needs an initialised process group)
        self.collective = world_size > 1 or always_collective
        # The collective is chosen once, from the backend: gloo has no flat all_gather_into_tensor.
        # Errors of the chosen collective propagate - a rank that fell back to another collective
        # after a failure would leave its peers waiting in the first one.
        self.flat = True
        if self.collective:
            import torch.distributed as dist
            self.flat = dist.get_backend(group) != "gloo"
        self.rows = tile_rows(height, world_size, rank, tile)
        self.n_pad = padded_rows(height, world_size, tile)
        self.local = torch.zeros((self.n_pad, width, 3), dtype=torch.uint8, device=device)
        self.recv = torch.empty((world_size, self.n_pad, width, 3), dtype=torch.uint8, device=device)
        self.full = torch.empty((height, width, 3), dtype=torch.uint8, device=device) if rank == dst else None
        # destination row of every gathered row; padding rows point at a scratch row past the frame
        if rank == dst:
            index = torch.full((world_size, self.n_pad), height, dtype=torch.int64)
            for r in range(world_size):
                rr = tile_rows(height, world_size, r, tile)
                index[r, : len(rr)] = torch.from_numpy(rr.astype(np.int64))
            self.index = index.view(-1).to(device)
            self.scratch = torch.empty((height + 1, width, 3), dtype=torch.uint8, device=device)

    def gather(self):
        """All ranks call this after their render has finished; returns the frame on `dst`."""
        import torch.distributed as dist
        if not self.collective:
            return self.local[: len(self.rows)]
        if self.flat:
            dist.all_gather_into_tensor(self.recv.view(-1), self.local.view(-1), group=self.group)
        else:
            dist.all_gather(list(self.recv.unbind(0)), self.local, group=self.group)
        if self.rank != self.dst:
            return None
        self.scratch.index_copy_(0, self.index, self.recv.view(-1, self.width, 3))
        self.full.copy_(self.scratch[: self.height])
        return self.full
