"""Batch sharding over the GPUs of one node — one process per GPU, `torch.distributed` ("nccl" = RCCL on
ROCm, xGMI underneath; "gloo" in the CPU tests).

Images never interact (GroupNorm / LayerNorm / attention are per sample, a CFG pair lives inside one sample), so the
global batch is split into contiguous row blocks, every rank keeps a full weight replica and runs its own sampler loop.
The only data-path collective is ONE broadcast of the text conditioning from the rank that ran CLIP; finished images
can be gathered at the end.  RNG: every rank draws the *full-batch* noise from the single seed on its host generator and
keeps its rows, which reproduces the reference's single-process draw order with zero communication.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """-> (rank, world, local_rank).  Reads the torchrun environment; a plain single process is rank 0 of 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_rows(global_batch: int, rank: int, world: int) -> slice:
    """Contiguous block of images owned by `rank` (sizes differ by at most one when world does not divide the batch)."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return slice(lo, lo + base + (1 if rank < extra else 0))


def broadcast_conditioning(tensors: Sequence[Optional[torch.Tensor]], shapes: Sequence[Tuple[int, ...]], src: int = 0,
                           device=None) -> List[torch.Tensor]:
    """One collective: pack [cond, uncond, ...] (fp32) into a flat buffer on `src`, broadcast, unpack everywhere.
    Non-source ranks pass None tensors and the agreed `shapes` (token count is fixed by the prompt chunking: 77·k)."""
    if not dist.is_initialized():
        return [t for t in tensors]
    rank = dist.get_rank()                       # (a world-1 group still runs the collective: the single-GPU RCCL check relies on it)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if rank == src:
        torch.cat([t.reshape(-1).float().to(device) for t in tensors], out=flat)
    dist.broadcast(flat, src=src)
    out, o = [], 0
    for s, n in zip(shapes, sizes):
        out.append(flat[o:o + n].view(s))
        o += n
    return out


def full_batch_noise(shape_global: Sequence[int], seed: int, rows: slice) -> torch.Tensor:
    """prepare_noise (LD.py:3145-3153) for a sharded batch: same seed, full tensor, this rank's rows."""
    gen = torch.manual_seed(seed)
    return torch.randn(list(shape_global), dtype=torch.float32, generator=gen, device="cpu")[rows].contiguous()


def gather_images(images: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Collect the per-rank [b, H, W, 3] images (any dtype) on `dst` in rank order; other ranks get None."""
    if not dist.is_initialized():
        return images
    rank, world = dist.get_rank(), dist.get_world_size()
    counts = [torch.zeros(1, dtype=torch.long, device=images.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([images.shape[0]], dtype=torch.long, device=images.device))
    mx = int(max(c.item() for c in counts))
    pad = torch.zeros((mx,) + tuple(images.shape[1:]), dtype=images.dtype, device=images.device)
    pad[: images.shape[0]] = images
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    return torch.cat([b[: int(c.item())] for b, c in zip(bufs, counts)])
