"""CPU, world_size 2 over gloo: the N>1 sharding (no data-path collective) covers every edit exactly
once and the gathered order equals the serial order."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from diffusionhandles_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = [dict(idx=i, rot_angle=float(3 * i)) for i in range(7)]
    mine = parallel.shard_edits(items)
    assert parallel.rank_world() == (rank, world)
    local = [(it["idx"], it["rot_angle"] * 2) for it in mine]            # stand-in for an edit result
    allr = parallel.gather_results(local)
    # timing reduction used by bench.py: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, [it["idx"] for it in mine], allr, float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_round_trip():
    world, port = 2, 29641
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    for _, _, allr, tmax in res:
        assert [i for i, _ in allr] == list(range(7))
        assert [v for _, v in allr] == [6.0 * i for i in range(7)]
        assert tmax == 2.0


def test_shard_edits_single_process():
    from diffusionhandles_amd import parallel
    items = list(range(10))
    parts = [parallel.shard_edits(items, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == items and parts[1] == [1, 5, 9]
