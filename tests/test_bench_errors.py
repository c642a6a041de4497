"""bench.py fails loudly (round 5): secondary sections keep the headline line alive, but their exceptions are collected into a
top-level "errors" list and the process exits non-zero after printing the line; the N-rank launcher starts a failed job again
only on evidence that the rendezvous port was taken."""
import importlib.util
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("dh_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_failed_section_is_listed_and_sets_the_exit_code():
    bench = _bench()
    sec = bench.Sections()
    assert sec.run("hbm_records", lambda: [1, 2]) == [1, 2] and sec.errors == []

    def batched_section():
        raise RuntimeError("HIP error: out of memory")
    got = sec.run("batched_section", batched_section)
    assert got == {"error": "RuntimeError: HIP error: out of memory"}
    r, w = os.pipe()
    rc = sec.finish({"metric": "m", "value": 1.0, "edits": got}, w)
    os.close(w)
    line = os.read(r, 1 << 16).decode()
    os.close(r)
    assert rc == 3 and line.endswith("\n") and line.count("\n") == 1
    out = json.loads(line)
    assert out["value"] == 1.0 and out["errors"] == [{"section": "batched_section", "error": "RuntimeError: HIP error: out of memory"}]
    # a clean run: empty list, exit code 0
    ok = bench.Sections()
    r, w = os.pipe()
    assert ok.finish({"value": 2.0}, w) == 0
    os.close(w)
    assert json.loads(os.read(r, 1 << 16).decode())["errors"] == []
    os.close(r)


def test_injected_failure_hook(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("DH_BENCH_INJECT_FAIL", "batched_section")
    sec = bench.Sections()
    ran = []
    assert "injected" in sec.run("batched_section", lambda: ran.append(1))["error"] and ran == []
    assert sec.run("hbm_records", lambda: 7) == 7 and len(sec.errors) == 1


def test_every_secondary_section_of_main_goes_through_sections():
    """No bare `except Exception` is left in main() outside Sections (the CPU baseline's own fallbacks are reported through
    its record and re-listed)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main_src = src[src.index("def main():"):src.index("def bench_768(")]
    assert main_src.count("except Exception") == 1 and "sections.note(\"hbm_records.graph_capture\"" in main_src
    for name in ("batched_section", "hbm_records", "phases.whole_edit", "res768_bf16", "cpu_baseline"):
        assert f'sections.run("{name}"' in main_src, name
    assert "raise SystemExit(rc)" in main_src and '"graph": bool(' in main_src


def test_launcher_retries_only_on_a_port_clash(tmp_path):
    bench = _bench()
    assert bench.port_clash(b"RuntimeError: The server socket has failed to listen on any local network address. "
                            b"port: 29500, useIpv6: 0, code: -98, name: EADDRINUSE, message: address already in use")
    assert not bench.port_clash(b"ModuleNotFoundError: No module named 'x'")
    # a deterministic failure of rank 0 (bad argument / import error / OOM) is final: the job is started ONCE
    script = tmp_path / "child.py"
    script.write_text("import os, sys\nd = os.path.dirname(__file__)\nopen(os.path.join(d, 'start_' + os.environ['RANK'] + '_' + "
                      "os.environ['MASTER_PORT']), 'w').close()\nsys.stderr.write('ImportError: nope\\n')\nsys.exit(2 if os.environ['RANK'] == '0' else 0)\n")
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=2, dry_run_launch=False), [], script=str(script))
    assert rc == 2 and len([p for p in tmp_path.iterdir() if p.name.startswith("start_0_")]) == 1
    # rank 0 reporting the port as taken: started again on another port, then succeeds
    script2 = tmp_path / "child2.py"
    script2.write_text("import os, sys, glob\nd = os.path.dirname(__file__)\nn = len(glob.glob(os.path.join(d, 'try_0_*')))\n"
                       "open(os.path.join(d, 'try_' + os.environ['RANK'] + '_' + os.environ['MASTER_PORT']), 'w').close()\n"
                       "if os.environ['RANK'] == '0' and n == 0:\n    sys.stderr.write('name: EADDRINUSE, message: address already in use\\n')\n    sys.exit(1)\n")
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=2, dry_run_launch=False), [], script=str(script2))
    tries = sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("try_0_"))
    assert rc == 0 and len(tries) == 2 and tries[0] != tries[1]
