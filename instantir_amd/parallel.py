"""Multi-GPU side of the denoising path: one process per GPU, images sharded across ranks, no
per-step collective.  The only exchange is the one-time broadcast of the frozen weights from rank 0
over RCCL/xGMI -- done in large flat buckets so each collective is bandwidth-,
not latency-bound on the point-to-point links."""
from __future__ import annotations

import os
from typing import Dict, List

import torch
import torch.distributed as dist


def init_from_env():
    """(rank, world, local_rank, device).  backend "nccl" is RCCL on ROCm; gloo on CPU (tests)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    # IIR_DIST_BACKEND=gloo + fewer devices than ranks: rehearsal of the N > 1 flow on a one-GPU box (ranks share the card;
    # RCCL itself refuses two ranks on one device)
    backend = os.environ.get("IIR_DIST_BACKEND", "nccl" if use_gpu else "gloo")
    if use_gpu and backend != "nccl":
        local = local % torch.cuda.device_count()
    device = torch.device(f"cuda:{local}" if use_gpu else "cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local, device


def describe_ranks(rank: int, world: int, local: int, device):
    """[{rank, device, backend}] for every rank (gathered on all ranks).  Under RCCL every rank must own a distinct GPU:
    two ranks on one device mean the launcher and the visible-device list disagree -- fail loudly, do not measure it."""
    me = {"rank": rank, "device": str(device), "backend": dist.get_backend() if dist.is_initialized() else "none"}
    if not dist.is_initialized() or world == 1:
        return [me]
    out: List[dict] = [None] * world
    dist.all_gather_object(out, me)
    if me["backend"] == "nccl":
        devs = [o["device"] for o in out]
        if len(set(devs)) != len(devs):
            raise RuntimeError(f"RCCL ranks share a device: {devs}")
    return out


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous split of the image list (infer.py:151-169 forms the list; ranks take slices)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_state_dict(sd: Dict[str, torch.Tensor], src: int = 0, bucket_bytes: int = 1 << 30, algo: str = None) -> Dict[str, torch.Tensor]:
    """In-place distribution of every tensor of `sd` (same names / shapes on all ranks) from rank `src`, in flat buckets.
    algo "broadcast" (default): one `dist.broadcast` per bucket (a ring / tree out of one GPU).  "scatter_allgather"
    (`IIR_BCAST_ALGO=scatter_allgather`): the source SCATTERS one N-th of a bucket to every rank -- on an MI355X node that is seven
    different xGMI links at once, point to point -- and an all-gather completes it, so no link carries more than ~2/N of the
    bucket out of the source.  Opt-in until it has been timed on an 8-GPU node (neither form has: this pipeline's boxes have one
    GPU); both are covered by the world-size-2 gloo test.  One-time set-up traffic (the weights are generated or read once, on
    rank 0); no collective exists on the per-step path."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return sd
    import os
    algo = algo or os.environ.get("IIR_BCAST_ALGO", "broadcast")
    if algo not in ("scatter_allgather", "broadcast"):
        raise ValueError(f"unknown weight distribution algorithm {algo!r}")
    world, rank = dist.get_world_size(), dist.get_rank()
    names = sorted(sd)
    bucket: List[str] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        parts = [sd[n].reshape(-1) for n in bucket]
        total = sum(p_.numel() for p_ in parts)
        if algo == "broadcast":
            flat = torch.cat(parts)
            dist.broadcast(flat, src)
        else:
            chunk = (total + world - 1) // world
            flat = torch.empty(chunk * world, dtype=parts[0].dtype, device=parts[0].device)
            views = [flat[r * chunk:(r + 1) * chunk] for r in range(world)]
            if rank == src:
                torch.cat(parts, out=flat[:total])
                flat[total:].zero_()
            mine = torch.empty(chunk, dtype=flat.dtype, device=flat.device)
            dist.scatter(mine, scatter_list=views if rank == src else None, src=src)
            dist.all_gather(views, mine)
        off = 0
        for n in bucket:
            k = sd[n].numel()
            sd[n].copy_(flat[off:off + k].view_as(sd[n]))
            off += k
        bucket, size = [], 0

    cur_dtype = None
    for n in names:
        t = sd[n]
        if cur_dtype is not None and t.dtype != cur_dtype:
            flush()
        cur_dtype = t.dtype
        bucket.append(n)
        size += t.numel() * t.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    return sd


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(x: float, device):
    """Every rank's value of `x`, in rank order (a one-element list without a process group)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
