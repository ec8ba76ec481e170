"""Multi-GPU sharding of the compress path: frames are independent, so rank r of W owns the contiguous
frame range [r*F/W, (r+1)*F/W). The only exchange is (1) an all-gather of the per-rank frame sizes — an exclusive scan of the
per-rank body totals gives every rank its base offset and every rank can build the full seek table — and (2) a variable-length
gather of the frame bodies to the root. One process per GPU; backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
No codec work happens here; tensors are byte buffers.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(nframes, rank, world):
    """Frames [lo, hi) owned by `rank`."""
    return (nframes * rank) // world, (nframes * (rank + 1)) // world


def _comm_device(t):
    """RCCL moves device tensors directly; the gloo backend (CPU tests, single-GPU dry runs) gets host staging."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else t.device


def allgather_sizes(local_sizes, group=None):
    """local_sizes: int64 tensor [n_local]. Returns the list of per-rank size tensors (rank order)."""
    world = dist.get_world_size(group)
    local_sizes = local_sizes.to(_comm_device(local_sizes))
    n = torch.tensor([local_sizes.numel()], dtype=torch.int64, device=local_sizes.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts) if counts else 0
    padded = torch.zeros(mx, dtype=torch.int64, device=local_sizes.device)
    padded[: local_sizes.numel()] = local_sizes
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [o[:c] for o, c in zip(out, counts)]


def stitch(all_sizes, uncompressed_size, frame_size):
    """All ranks' frame sizes -> (header bytes incl. CRC-32, per-rank base offsets into the body, per-rank totals)."""
    import zra_amd
    flat = torch.cat([s.cpu() for s in all_sizes]).numpy().astype(np.uint64)
    totals = np.array([int(s.sum().item()) for s in all_sizes], dtype=np.uint64)
    bases = np.concatenate([[0], np.cumsum(totals)[:-1]]).astype(np.uint64)   # exclusive scan of per-rank body sizes
    header = zra_amd.stitch_header(flat, uncompressed_size, frame_size)
    return header, bases, totals


def gather_archive(local_body, local_sizes, uncompressed_size, frame_size, root_buffer=None, group=None):
    """local_body: uint8 tensor with this rank's packed frames; local_sizes: int64 [n_local].
    Returns (archive tensor on rank 0 / None elsewhere, header bytes, bases, totals)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    all_sizes = allgather_sizes(local_sizes, group)
    header, bases, totals = stitch(all_sizes, uncompressed_size, frame_size)
    hlen = len(header)
    if rank == 0:
        need = hlen + int(totals.sum())
        if root_buffer is None or root_buffer.numel() < need:
            root_buffer = torch.empty(need, dtype=torch.uint8, device=local_body.device)
        root_buffer[:hlen] = torch.frombuffer(bytearray(header), dtype=torch.uint8).to(local_body.device)
        root_buffer[hlen: hlen + int(totals[0])] = local_body[: int(totals[0])]
        staged = _comm_device(root_buffer) != root_buffer.device
        ops = []
        for r in range(1, world):
            if int(totals[r]):
                dst = root_buffer[hlen + int(bases[r]): hlen + int(bases[r]) + int(totals[r])]
                if staged:
                    tmp = torch.empty(int(totals[r]), dtype=torch.uint8)
                    dist.recv(tmp, src=r, group=group)
                    dst.copy_(tmp)
                else:
                    ops.append(dist.P2POp(dist.irecv, dst, r, group))
        if ops:
            # one RCCL group: the 7 inbound transfers run concurrently, each on its own point-to-point xGMI link
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return root_buffer[:need], header, bases, totals
    if int(totals[rank]):
        dist.send(local_body[: int(totals[rank])].contiguous().to(_comm_device(local_body)), dst=0, group=group)
    return None, header, bases, totals
