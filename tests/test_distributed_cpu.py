"""CPU, world_size 2 over gloo: the N > 1 path of the denoising job -- contiguous image sharding, the
one-time weight broadcast from rank 0, max-over-ranks timing.  No data-path collective exists."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from instantir_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _, dev = parallel.init_from_env()
    g = torch.Generator().manual_seed(100 + rank)          # ranks start with DIFFERENT tensors
    sd = {"a.weight": torch.randn(300, 7, generator=g).half(), "b.bias": torch.randn(11, generator=g).half(),
          "c.f32": torch.randn(5, 5, generator=g)}
    ref = torch.Generator().manual_seed(100)
    want = {"a.weight": torch.randn(300, 7, generator=ref).half(), "b.bias": torch.randn(11, generator=ref).half(),
            "c.f32": torch.randn(5, 5, generator=ref)}
    same = True
    for algo in ("scatter_allgather", "broadcast"):       # both forms of the one-time weight distribution (odd sizes: padded chunks)
        mine = {k: v.clone() for k, v in sd.items()}
        parallel.broadcast_state_dict(mine, 0, bucket_bytes=1024, algo=algo)
        same = same and all(torch.equal(mine[k], want[k]) for k in want)
    parallel.broadcast_state_dict(sd, 0, bucket_bytes=1024)
    lo, hi = parallel.shard_range(9, r, w)
    mx = parallel.max_over_ranks(1.0 + r, dev)
    per_rank = parallel.gather_floats(10.0 + r, dev)       # bench.py's ms_per_step_by_rank
    parallel.barrier()
    q.put((r, same and per_rank == [10.0, 11.0], (lo, hi), mx))
    dist.destroy_process_group()


def test_two_rank_broadcast_shard_and_max():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert out[0] == (0, True, (0, 5), 2.0) and out[1] == (1, True, (5, 9), 2.0)


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 8, 33):
        for w in (1, 2, 4, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
