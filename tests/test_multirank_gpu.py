"""The multi-rank control flow of the two drivers on the ONE GPU of the box (round 5): `bench.py --gpus 2` and
`tools/run_edits_sharded.py --gpus 2` start their own ranks (bench.launch_ranks: the parent never touches the GPU, children
are fresh processes, nothing is exec'd), the ranks rendezvous over gloo (DH_BENCH_BACKEND=gloo folds them onto the visible
device; on an 8-GPU node the same code runs one rank per GPU over RCCL), meet the same barriers, reduce the timing with MAX and
rank 0 prints ONE JSON line.  The reference's multi-GPU shape is one process per device as well
(/root/reference/webapp/start_webapps_in_tmux.sh:21-43)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DH_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


def test_bench_two_ranks_on_one_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--batch-edits", "0", "--streams", "1", "--no-phases", "--no-res768", "--no-cpu-baseline"],
                       env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["errors"] == [] and out["roofline"]["frac"] > 0
    assert abs(out["value"] - 2 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-2 * out["value"]      # whole-job rate = 2 ranks x steps / MAX time
    # every rank's own figure is in the line (a slow rank is visible), and the headline divides by the slowest
    pr = out["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["steps_per_s"]) == 2 and pr["backend"] == "gloo"
    assert abs(max(pr["ms_per_step"]) - out["ms_per_step"]) < 0.05 * out["ms_per_step"]
    print("two ranks on one GPU over gloo:", out["value"], "steps/s in total; per rank", pr["steps_per_s"])


def _n_gpus():
    import torch
    return torch.cuda.device_count()      # (counting devices does not initialise the GPU in this process)


@pytest.mark.skipif(_n_gpus() < 2, reason="the RCCL branch needs two devices (RCCL refuses two ranks on one GPU); runs wherever >= 2 GPUs exist")
def test_bench_two_ranks_over_rccl_one_rank_per_gpu():
    """The branch an 8-GPU node runs (round 6): backend "nccl" (= RCCL), one rank per device, init_process_group(device_id),
    device-side barrier / MAX / all_gather of the timings, the batched section on every rank.  Skipped on a 1-GPU box -- there
    the same control flow runs over gloo (tests above) and this branch stays unmeasured."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--batch-edits", "2", "--streams", "1", "--no-phases", "--no-res768", "--no-cpu-baseline"],
                       env=dict(_env(), DH_BENCH_BACKEND="nccl"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["errors"] == [] and out["per_rank"]["backend"] == "nccl"
    assert len(out["per_rank"]["steps_per_s"]) == 2 and len(out["edits"]["per_rank"]["edits_per_s"]) == 2
    assert out["edits"]["edits_per_gpu"] == 2 and out["edits"]["edits_per_s"] > 0


@pytest.mark.skipif(_n_gpus() < 2, reason="the RCCL branch needs two devices")
def test_sharded_driver_two_ranks_over_rccl(tmp_path):
    out = str(tmp_path / "rccl")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_edits_sharded.py"), "--edits", "4", "--batch", "2", "--out", out,
                        "--gpus", "2"], env=dict(_env(), DH_BENCH_BACKEND="nccl"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["edits"] == 4 and rep["n_gpus"] == 2 and rep["edits_per_s"] > 0


def test_bench_reports_a_failed_secondary_section_and_exits_non_zero():
    """A broken secondary record must not pass the driver unnoticed: the line is printed WITH the failure under "errors" and
    the return code is 3 (here the failure is injected into the batched section; the headline is still measured)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch-edits", "2",
                        "--streams", "1", "--no-phases", "--no-res768", "--no-cpu-baseline"],
                       env=dict(_env(), DH_BENCH_INJECT_FAIL="batched_section"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] > 0 and out["n_gpus"] == 1
    assert [e["section"] for e in out["errors"]] == ["batched_section"] and "error" in out["edits"]


def test_sharded_driver_two_ranks_write_the_one_rank_images(tmp_path):
    """tools/run_edits_sharded.py --gpus 2 --edits 4 --batch 2 (rank 0 computes the identity and broadcasts it, each rank runs
    its round-robin share as one batch of two, results gathered on rank 0) writes byte for byte the PNGs of the one-rank run
    of the same four edits in batches of two: an edit's result depends neither on the rank that ran it nor on its batch mates."""
    outs = {}
    for gpus in (1, 2):
        out = str(tmp_path / f"g{gpus}")
        cmd = [sys.executable, os.path.join(ROOT, "tools", "run_edits_sharded.py"), "--edits", "4", "--batch", "2", "--out", out]
        if gpus > 1:
            cmd += ["--gpus", str(gpus)]
        r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-3000:]
        rep = json.loads(r.stdout.strip().splitlines()[-1])
        assert rep["edits"] == 4 and rep["n_gpus"] == gpus and rep["batch"] == 2 and rep["edits_per_s"] > 0
        assert rep["per_rank"]["edits"] == [4 // gpus] * gpus and len(rep["per_rank"]["edits_s"]) == gpus      # every rank's share and time
        outs[gpus] = out
    for i in range(4):
        for suffix in ("", "_disparity"):
            a = open(os.path.join(outs[1], f"edit_{i:03d}{suffix}.png"), "rb").read()
            b = open(os.path.join(outs[2], f"edit_{i:03d}{suffix}.png"), "rb").read()
            assert a == b, f"edit {i}{suffix}: the two-rank run wrote a different image"
