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


class _FakeHandles:
    """Stands in for DiffusionHandles in the CPU test of the sharded driver: an 'edit' is a deterministic function of its
    transform and of the identity, so a wrong shard, a wrong order or a stale identity shows in the gathered results."""

    def __init__(self):
        self.batches = []

    def transform_foreground_batch(self, depth, prompt, fg_mask, bg_depth, null_text, noise, acts, transforms):
        self.batches.append(len(transforms))
        k = float(null_text.sum() + noise.sum() + sum(a.float().sum() for a in acts))
        imgs = torch.stack([torch.full((3, 4, 4), float(a) + float(t.sum()) + k) for a, _, t in transforms])
        return imgs, [torch.full((1, 1, 4, 4), float(a)) for a, _, _ in transforms]

    def transform_foreground(self, depth, prompt, fg_mask, bg_depth, null_text, noise, acts, rot_angle=None, rot_axis=None,
                             translation=None):
        imgs, disps = self.transform_foreground_batch(depth, prompt, fg_mask, bg_depth, null_text, noise, acts,
                                                      [(rot_angle, rot_axis, translation)])
        return imgs[:1], disps[0]


def _driver_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from diffusionhandles_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the identity exists on rank 0 only and is handed over once
    identity = None
    if rank == 0:
        g = torch.Generator().manual_seed(3)
        acts = [torch.randn(5, 8, 2, 2, generator=g).half().permute(0, 2, 3, 1).permute(0, 3, 1, 2) for _ in range(3)]   # channels-last views
        identity = (torch.randn(5, 1, 7, 4, generator=g), torch.randn(1, 4, 2, 2, generator=g), acts)
    identity = parallel.broadcast_identity(identity, src=0)
    assert identity[2][0].dtype == torch.float16 and identity[2][0].shape == (5, 8, 2, 2) and identity[2][0].is_contiguous()
    edits = [dict(rot_angle=float(i), rot_axis=torch.tensor([0.0, 1.0, 0.0]), translation=torch.tensor([0.1 * i, 0.0, 0.0]))
             for i in range(21)]
    fake = _FakeHandles()
    local = parallel.run_edits(fake, identity, edits, None, None, None, "p", batch=8)
    single = parallel.run_edits(_FakeHandles(), identity, edits[:3], None, None, None, "p", batch=1)
    allr = parallel.gather_results([(gi, float(im[0, 0, 0]), float(dp[0, 0, 0, 0])) for gi, im, dp in local])
    k = float(identity[0].sum() + identity[1].sum() + sum(a.float().sum() for a in identity[2]))
    q.put((rank, fake.batches, allr, k, [gi for gi, _, _ in single]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_driver_batches_and_identity():
    """parallel.run_edits + broadcast_identity + gather_results over gloo, world 2: 21 edits -> rank 0 runs 11 (batches of
    8 + 3), rank 1 runs 10 (8 + 2); every edit once, serial order restored, both ranks hold the same identity."""
    world, port = 2, 29643
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [8, 3] and res[1][1] == [8, 2]
    assert res[0][3] == res[1][3]                                     # same identity on both ranks
    k = res[0][3]
    for _, _, allr, _, single in res:
        assert [gi for gi, _, _ in allr] == list(range(21))
        for gi, v, d in allr:
            assert abs(v - (gi + 0.1 * gi + k)) < 1e-3 and d == float(gi)
    assert res[0][4] == [0, 2] and res[1][4] == [1]


def test_bench_launcher_starts_one_rank_per_gpu():
    """`python bench.py --gpus N` with no launcher in front starts N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* per child, 127.0.0.1 rendezvous) and never touches the GPU in the parent: --dry-run-launch prints the plan."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--dry-run-launch"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert len(plan) == 4
    ports = {p["env"]["MASTER_PORT"] for p in plan}
    assert len(ports) == 1 and int(ports.pop()) > 0
    for r, p in enumerate(plan):
        assert p["env"]["RANK"] == str(r) and p["env"]["LOCAL_RANK"] == str(r) and p["env"]["WORLD_SIZE"] == "4"
        assert p["env"]["MASTER_ADDR"] == "127.0.0.1"
        assert p["cmd"][1].endswith("bench.py") and "--dry-run-launch" not in p["cmd"] and p["cmd"][-4:] == ["--gpus", "4", "--steps", "3"]


def test_bench_launcher_children_see_their_rank(tmp_path):
    """The launcher really spawns the children with those environments and returns the worst return code."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("dh_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    envs = bench.rank_environments(3, 12345, base={"PATH": os.environ.get("PATH", "")})
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and all(e["WORLD_SIZE"] == "3" for e in envs)
    # a stand-in script in place of bench.py: each child writes its rank; rank 1 fails
    script = tmp_path / "child.py"
    script.write_text("import os, sys\nr = os.environ['RANK']\nopen(os.path.join(os.path.dirname(__file__), 'r' + r), 'w')"
                      ".write(os.environ['WORLD_SIZE'] + ' ' + os.environ['MASTER_PORT'])\nsys.exit(3 if r == '1' else 0)\n")
    old = bench.__file__
    try:
        bench.__file__ = str(script)
        rc = bench.launch_ranks(types.SimpleNamespace(gpus=3, dry_run_launch=False), [])
    finally:
        bench.__file__ = old
    assert rc == 3
    seen = sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("r"))
    assert seen == ["r0", "r1", "r2"]
    assert len({(tmp_path / n).read_text() for n in seen}) == 1


def test_sharded_edit_driver_launch_plan():
    """tools/run_edits_sharded.py --gpus N (BASELINE config 4's driver) starts its own ranks through the same launcher."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_edits_sharded.py"), "--gpus", "8", "--edits", "64",
                          "--dry-run-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert [p["env"]["RANK"] for p in plan] == [str(r) for r in range(8)]
    assert all(p["env"]["WORLD_SIZE"] == "8" and p["cmd"][1].endswith("run_edits_sharded.py") and "--edits" in p["cmd"] for p in plan)


def test_config4_edit_plan_is_eight_batches_of_eight():
    """BASELINE config 4 (64 edits, 8 per GPU): `run_edits_sharded.py --gpus 8 --edits 64 --batch 8 --dry-run-launch` prints, next
    to the launch plan, which edits every rank runs and in which batches -- 8 ranks x ONE batch of 8, every edit exactly once,
    the round-robin split of parallel.shard_edits; with --streams 2 --batch 4 every rank runs two batches of 4 on two lanes."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    tool = os.path.join(ROOT, "tools", "run_edits_sharded.py")
    out = subprocess.run([sys.executable, tool, "--gpus", "8", "--edits", "64", "--batch", "8", "--dry-run-launch"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert len(out.stdout.strip().splitlines()) == 1, "stdout must carry the one JSON line only"
    rep = json.loads(out.stdout)
    assert len(rep["launch"]) == 8 and rep["launch_timeout_s"] > 0
    plan = rep["plan"]
    assert [p["rank"] for p in plan] == list(range(8))
    assert all(len(p["batches"]) == 1 and len(p["batches"][0]["edits"]) == 8 and p["batches"][0]["lane"] == 0 for p in plan)
    assert sorted(e for p in plan for e in p["edits"]) == list(range(64))
    assert plan[3]["edits"] == [3 + 8 * i for i in range(8)]
    out = subprocess.run([sys.executable, tool, "--gpus", "8", "--edits", "64", "--batch", "4", "--streams", "2",
                          "--dry-run-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout)["plan"]
    assert all([b["lane"] for b in p["batches"]] == [0, 1] and [len(b["edits"]) for b in p["batches"]] == [4, 4] for p in plan)
    assert "--streams" in json.loads(out.stdout)["launch"][0]["cmd"]


def test_run_edits_with_streams_hands_the_whole_share_to_the_lanes():
    """parallel.run_edits(streams > 1) passes the rank's whole share to ONE transform_foreground_batch call with streams / batch
    (the lanes cut it into chunks of `batch`), and restores the global indices."""
    import torch
    from diffusionhandles_amd import parallel
    calls = []

    class FakeDH:
        def transform_foreground_batch(self, depth, prompt, fg_mask, bg_depth, null_text, noise, acts, tfs, streams=1, batch=None):
            calls.append((len(tfs), streams, batch))
            return torch.stack([torch.full((3, 2, 2), float(t[0])) for t in tfs]), [torch.full((1, 1, 2, 2), float(t[0])) for t in tfs]

    edits = [dict(rot_angle=float(i), rot_axis=None, translation=None) for i in range(10)]
    old = dict(os.environ)
    try:
        os.environ["RANK"], os.environ["WORLD_SIZE"] = "1", "2"
        res = parallel.run_edits(FakeDH(), (None, None, None), edits, None, None, None, "p", batch=2, streams=2)
    finally:
        os.environ.clear()
        os.environ.update(old)
    assert calls == [(5, 2, 2)]
    assert [gi for gi, _, _ in res] == [1, 3, 5, 7, 9] and all(float(im[0, 0, 0]) == gi for gi, im, _ in res)


def test_run_lanes_interleaves_jobs_one_yield_at_a_time():
    """GuidedStableDiffuser.run_lanes (the host side of the concurrent edit lanes): job i runs on lane i % n, a lane's jobs one
    after the other, every lane advances by ONE yield per round (so every lane's stream always holds work), results come back in
    job order.  Scheduling only: lanes without a stream (no GPU)."""
    from types import SimpleNamespace
    from diffusionhandles_amd.guided_stable_diffuser import GuidedStableDiffuser
    trace = []

    def make(i, steps):
        def job(lane):
            def body():
                for k in range(steps):
                    trace.append((lane.name, i, k))
                    yield None
                return f"result{i}"
            return body()
        return job

    lanes = [SimpleNamespace(name="A", _stream=None, device="cpu"), SimpleNamespace(name="B", _stream=None, device="cpu")]
    res = GuidedStableDiffuser.run_lanes(lanes, [make(0, 3), make(1, 2), make(2, 2), make(3, 1), make(4, 1)])
    assert res == [f"result{i}" for i in range(5)]
    assert {i for ln, i, _ in trace if ln == "A"} == {0, 2, 4} and {i for ln, i, _ in trace if ln == "B"} == {1, 3}
    # a lane never works on two jobs at once, and keeps their order
    for name in "AB":
        seq = [i for ln, i, _ in trace if ln == name]
        assert seq == sorted(seq)
    # the two lanes alternate while both have work: no lane runs two steps in a row before the other ran one
    both = trace[:6]                                   # (lane B has run out of work after its third step)
    assert all(both[k][0] != both[k + 1][0] for k in range(len(both) - 1))
    # one lane only: plain sequential execution
    trace.clear()
    assert GuidedStableDiffuser.run_lanes(lanes[:1], [make(0, 2), make(1, 1)]) == ["result0", "result1"]
    assert trace == [("A", 0, 0), ("A", 0, 1), ("A", 1, 0)]

