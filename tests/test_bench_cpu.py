"""CPU: bench.py's own launcher (`python bench.py --gpus N` with no WORLD_SIZE in the environment starts N rank processes before
anything touches a GPU) and the PMC-traffic stamp (a committed profile is quoted only for the kernel sources it was taken from)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=120)


def test_launcher_starts_n_ranks_and_rank0_prints_one_line():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"AVF_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j == {"dryrun": True, "world": 4, "master": "127.0.0.1", "n_gpus": 4}


def test_worker_rejects_a_world_size_that_does_not_match():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0


def test_traffic_is_only_quoted_for_the_profiled_kernel_sources(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / bench.PKG_DIR / "csrc")
    (tmp_path / bench.PKG_DIR / "csrc" / "k.hip").write_text("kernel v1")
    sha = bench.kernel_source_hash()
    json.dump({"workload": "c2", "kernel_sources_sha": sha, "commit": "abc", "per_class": {"gemm_bf16_nt": 123.4}},
              open(prof / "r09_traffic.json", "w"))
    assert bench.pmc_traffic("gemm_bf16_nt", "c2")[0] == 123
    assert bench.pmc_traffic("gemm_bf16_nt", "c3")[0] is None          # no profile of that workload
    (tmp_path / bench.PKG_DIR / "csrc" / "k.hip").write_text("kernel v2")  # the kernels changed: the number is stale
    v, note = bench.pmc_traffic("gemm_bf16_nt", "c2")
    assert v is None and "other kernel sources" in note


def test_calibration_file_is_frozen_against_regressions(tmp_path):
    """tools/calibrate_bounds.py refuses to RAISE a recorded error without --allow-regress <reason>"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("calibrate_bounds", os.path.join(ROOT, "tools", "calibrate_bounds.py"))
    cb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cb)
    out = tmp_path / "lowp.json"
    json.dump({"measured": {"a": 1e-3, "b": 2e-3}}, open(out, "w"))
    run = tmp_path / "run.json"
    json.dump({"a": 2e-3, "b": 1e-3, "c": 5e-4}, open(run, "w"))
    assert cb.main([str(run), "--merge"], out_path=str(out)) == 2                     # "a" would go up: refused
    assert json.load(open(out))["measured"] == {"a": 1e-3, "b": 2e-3}                  # nothing written
    assert cb.main([str(run), "--merge", "--allow-regress"], out_path=str(out)) == 2  # a reason is mandatory
    assert cb.main([str(run), "--merge", "--allow-regress", "new kernel rounds P once more"], out_path=str(out)) == 0
    doc = json.load(open(out))
    assert doc["measured"] == {"a": 2e-3, "b": 1e-3, "c": 5e-4}
    assert doc["regress_log"][-1]["raised"] == {"a": [1e-3, 2e-3]} and "rounds P" in doc["regress_log"][-1]["reason"]
    json.dump({"b": 5e-4, "d": 1.0}, open(run, "w"))
    assert cb.main([str(run), "--merge"], out_path=str(out)) == 0                      # lower values and new tags pass freely
