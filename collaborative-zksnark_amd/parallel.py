"""One process per GPU.  The hot path shards by independent units (MPC parties / proofs: SURVEY.md section 8e), so
the data path needs no collective; torch.distributed (RCCL on GPUs, gloo on CPU) carries only (a) the timing
reduction of bench.py and (b) the mpc-net style share exchange of an `open` (mpc-net/src/multi.rs:145-173 is an
all-gather of one equal-length buffer per party; mpc-algebra/src/channel.rs:13-21)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str | None = None):
    """Initialises the default process group from the torchrun environment (no-op for a single process)."""
    rank, world, local_rank = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def partition_units(n_units: int, world: int, rank: int) -> list[int]:
    """Contiguous block partition of independent units (party-proofs) over ranks; sizes differ by at most one."""
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def barrier(device_sync=None):
    if device_sync:
        device_sync()
    if dist.is_initialized():
        dist.barrier()
    if device_sync:
        device_sync()


def max_over_ranks(seconds: float, device=None) -> float:
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_throughput(units_local: float, seconds_local: float, device=None) -> tuple[float, float]:
    """(whole-job units/s, max-over-ranks seconds): all ranks' units divided by the slowest rank's time."""
    t = max_over_ranks(seconds_local, device)
    return sum_over_ranks(units_local, device) / t, t


def all_gather_shares(share: torch.Tensor) -> torch.Tensor:
    """mpc-net `broadcast`: every party contributes one equal-length buffer and receives all of them
    (mpc-net/src/multi.rs:145-173).  Returns a (world, *share.shape) tensor; the modular sum of an `open`
    (share/spdz.rs:166-185) is then a local, share-linear pointwise step."""
    if not dist.is_initialized():
        return share.unsqueeze(0)
    world = dist.get_world_size()
    if dist.get_backend() == "gloo":
        # CPU tests, and rigs without one GPU per party (several ranks on one device): stage through the host
        host = share.contiguous().cpu()
        out = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(out, host)
        return torch.stack(out).to(share.device)
    # RCCL: one all-gather straight into the (world, ...) result
    out = torch.empty((world,) + tuple(share.shape), dtype=share.dtype, device=share.device)
    dist.all_gather_into_tensor(out, share.contiguous())
    return out
