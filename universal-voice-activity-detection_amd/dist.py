"""Multi-GPU: the path shards by utterance with no exchange step.  Every
utterance is an independent sequence and the weights (5.8 MB) are replicated, so ranks need no
data-path collective: rank r owns utterances {i : i mod world == r} (``shard_indices``), generates
or loads exactly those, and runs the single-GPU path.  The only communication is the optional
gather of the per-frame outputs to rank 0 (``gather_rows``; RCCL when the tensors are on the GPU,
gloo in the CPU tests) -- tiny: B/world * T * 4 bytes per rank."""
import os
from typing import List, Optional

import torch
import torch.distributed as dist


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("UVAD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """Utterance i -> rank i mod world.  Stable under any world size, so utterance i
    always gets the same synthetic signal and the same result."""
    return list(range(rank, n, world))


def shard_count(n: int, rank: int, world: int) -> int:
    return (n - rank + world - 1) // world if n > rank else 0


def gather_rows(local: torch.Tensor, n_total: int, rank: int, world: int) -> Optional[torch.Tensor]:
    """Inverse of ``shard_indices``: rows held by each rank -> full (n_total, ...) tensor on rank 0
    (None elsewhere).  Shards may differ by one row, so they are padded to the largest."""
    if world == 1:
        return local
    per = (n_total + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_total, r, world)
        out[idx] = bufs[r][: len(idx)]
    return out


def preshard_rows(full: torch.Tensor, n_total: int, world: int) -> torch.Tensor:
    """Root only, ONCE per corpus (not per step): (n_total, ...) rows in utterance order -> (world, per, ...) rank-major, row j of
    block r = utterance r + j * world (``shard_indices``), short shards zero-padded to per = ceil(n_total / world).  ``scatter_rows``
    then sends the blocks as they lie (views, no per-call gather of the root-resident corpus)."""
    per = (n_total + world - 1) // world
    tail = tuple(full.shape[1:])
    if n_total == per * world:
        return full.reshape((per, world) + tail).transpose(0, 1).contiguous()   # one strided pass
    out = torch.zeros((world, per) + tail, dtype=full.dtype, device=full.device)
    for r in range(world):
        n = shard_count(n_total, r, world)
        out[r, :n] = full[r::world]
    return out


def scatter_rows(full: Optional[torch.Tensor], n_total: int, rank: int, world: int, like: Optional[torch.Tensor] = None,
                 presharded: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Root-resident corpus mode: rank 0 holds the rows and every rank receives the rows of ``shard_indices`` --
    one RCCL scatter (root egress over all xGMI links at once; gloo in the CPU tests).  ``full`` on rank 0 is either the
    (n_total, ...) tensor in utterance order (re-ordered here: one extra pass over the corpus, fine for a one-off) or, with
    ``presharded=True``, the (world, per, ...) tensor of ``preshard_rows`` -- the per-step form: its blocks are sent as views, the
    root does no copy.  ``like`` gives non-root ranks the trailing shape / dtype / device; ``out`` (per, ...) reuses a receive
    buffer.  Shards are padded to equal length; the result is cut to this rank's count."""
    if world == 1:
        return full if not presharded else full[0][:n_total]
    per = (n_total + world - 1) // world
    ref = full if rank == 0 else like
    if ref is None:
        raise ValueError("non-root ranks must pass `like` (a tensor with the row shape, dtype and device)")
    tail = tuple(ref.shape[2:] if (presharded and rank == 0) else ref.shape[1:])
    recv = out if out is not None else torch.empty((per,) + tail, dtype=ref.dtype, device=ref.device)
    if tuple(recv.shape) != (per,) + tail:
        raise ValueError(f"receive buffer {tuple(recv.shape)} != {(per,) + tail}")
    chunks = None
    if rank == 0:
        blocks = full if presharded else preshard_rows(full, n_total, world)
        if tuple(blocks.shape[:2]) != (world, per):
            raise ValueError(f"presharded tensor {tuple(blocks.shape)} is not (world={world}, per={per}, ...)")
        chunks = list(blocks.unbind(0))
    dist.scatter(recv, chunks, src=0)
    return recv[: shard_count(n_total, rank, world)]


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    on_gpu = dist.get_backend() == "nccl"      # RCCL reduces device tensors, gloo host tensors
    dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if on_gpu else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_floats(values, device=None):
    """Every rank contributes a short list of floats; every rank gets the list of all ranks' lists (rank order).  One all_gather
    (RCCL on device tensors, gloo on host tensors); [values] without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return [list(values)]
    on_gpu = dist.get_backend() == "nccl"
    dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if on_gpu else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    return [b.cpu().tolist() for b in bufs]


def backend_name() -> str:
    if not (dist.is_available() and dist.is_initialized()):
        return "none (single process)"
    b = dist.get_backend()
    return "nccl (RCCL)" if b == "nccl" else b


def barrier():
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
