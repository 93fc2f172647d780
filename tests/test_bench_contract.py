"""bench.py's contract, checked on the CPU: the wall-budget planner keeps the driver's command inside its 600 s limit, the
output builder emits every contract key, and `--gpus N` starts N ranks by itself (dry run: launch plumbing without GPU work)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_driver_command_fits_its_time_limit():
    # python3 bench.py --gpus 1 --steps 20 --warmup 5: start-up between 40 s (warm box) and 200 s (cold first import of torch
    # twice, CPU baseline, 139 GB of arenas, capture), c3 pools between 20 and 60 s
    for startup in (40, 90, 150, 200):
        for pool_s in (20, 30, 33.2, 40, 60):
            n, warm, timed, wall = bench.simulate_plan(startup, pool_s, steps=20, warmup=5)
            assert n >= 1 and timed >= 1 and warm + timed == n
            assert wall <= max(bench.DEFAULT_BUDGET_S + 0.1 * pool_s, startup + pool_s + bench.FINAL_RESERVE_S) + 1e-9
            assert wall < 600 - 60, (startup, pool_s, wall)


def test_short_requests_run_exactly_as_asked():
    n, warm, timed, wall = bench.simulate_plan(60, 30, steps=2, warmup=1)
    assert (n, warm, timed) == (3, 1, 2)
    n, warm, timed, _ = bench.simulate_plan(60, 30, steps=1, warmup=0)
    assert (n, warm, timed) == (1, 0, 1)


def test_timed_steps_take_priority_over_warmup():
    assert bench.classify_pools(9, 20, 5) == (1, 8)  # budget cut: one pool stays untimed
    assert bench.classify_pools(9, 20, 0) == (0, 9)
    assert bench.classify_pools(22, 20, 5) == (2, 20)
    assert bench.classify_pools(25, 20, 5) == (5, 20)
    assert bench.classify_pools(1, 1, 1) == (0, 1)
    assert bench.classify_pools(2, 1, 1) == (1, 1)
    assert not bench.should_continue(100.0, 25, 30.0, 20, 5, 380.0)
    assert bench.should_continue(1e9, 0, 0.0, 20, 5, 380.0)  # one pool is always played


def _fake_stats():
    names = ["simulations", "expansions", "terminal_hits", "path_edges", "sum_valid_select", "sum_valid_leaf", "transposition_links", "nodes",
             "moves", "episodes", "probes", "key_bytes", "sum_visited_select", "visited_new"]
    tot = {k: 1000.0 for k in names}
    tot.update(episodes=65536.0, expansions=7.0e8, simulations=8.0e8, waves=24000.0, path_edges=1.9e9, sum_valid_select=8e10, sum_visited_select=2e10)
    return tot


def test_output_line_carries_the_contract():
    tot = _fake_stats()
    kms = {"k_resstage16 10x10": (0.55, 700), "k_resstage32 5x5": (0.5, 700)}
    cpu = {"value": 2.8, "unit": "episodes/s", "cores": 32, "kind": "port", "sample": "x"}
    out = bench.build_output("c3", 20, 20, 32, 400, 10.01e6, 32768, 32768, 1, 2, 0, {"steps": 20, "warmup": 5}, 66.0, tot, 24000.0, 1, 32768,
                             True, True, [0.4, 0.25, 1.9, 0.12], kms, None, cpu, 0.9, 30.0, {"ranks_joined": 1})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["steps"] == 2 and out["warmup"] == 0 and abs(out["ms_per_step"] - 33000.0) < 1e-6
    assert abs(out["value"] - 65536.0 / 66.0) < 1e-9 and out["vs_baseline"] is None and out["scaling"] == "weak"
    assert "c3" in out["config"]["workload"] and "32768 concurrent games" in out["config"]["workload"] and "model" not in out["config"]
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"].startswith("k_resstage16")
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(out["cpu_baseline"])
    json.dumps(out)
    # every own MFMA kernel has its line; the standalone figure (k_resstage16 alone on all rows) is attached to THAT kernel only
    solo = {"ms_per_launch": 0.54, "leaves_per_launch": 32768, "achieved": 111.0, "frac": 0.706}
    kms = {"k_resstage16 10x10": (0.45, 700), "k_resstage32 5x5": (0.46, 700), "k_convpool32 16->32 10x10": (0.28, 700)}
    out = bench.build_output("c3", 20, 20, 32, 400, 10.01e6, 32768, 32768, 1, 2, 0, {"steps": 20, "warmup": 5}, 66.0, tot, 24000.0, 1, 32768,
                             True, True, [0.4, 0.25, 1.9, 0.12], kms, solo, cpu, 0.9, 30.0, {"ranks_joined": 1})
    assert out["roofline"]["kernel"].startswith("k_resstage32 5x5") and "standalone" not in out["roofline"]
    rk = out["roofline_kernels"]
    assert set(rk) == set(kms) and rk["k_resstage16 10x10"]["standalone"] == solo and "standalone" not in rk["k_convpool32 16->32 10x10"]
    for r in rk.values():
        assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["algorithmic_bytes_per_launch"]
    json.dumps(out)


def test_stale_pmc_traffic_is_not_reported(tmp_path, monkeypatch):
    rec, note = bench.load_pmc_traffic()
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    stored = json.load(open(path)).get("engine_sha256") if os.path.exists(path) else None
    if stored == bench.engine_hash():
        assert rec is not None and note is None
    else:
        assert rec is None and note


@pytest.mark.timeout(300)
def test_gpus_flag_starts_the_ranks_itself():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RP_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_joined"] == 2 and line["backend"] == "gloo" and line["dry_run"] is True


def test_engine_hash_ignores_comments_but_not_code(tmp_path):
    """profiles/pmc_traffic.json is tied to the engine source by bench.engine_hash(): rewording a comment keeps the PMC figures
    valid, any change of code makes them stale (traffic: null)."""
    src = open(bench.ENGINE_SRC).read()
    a = tmp_path / "a.hip"; a.write_text(src.replace("// kernels\n", "// kernels, reworded\n\n   \n"))
    b = tmp_path / "b.hip"; b.write_text(src.replace("#define SEARCH_WAVES 5", "#define SEARCH_WAVES 4"))
    assert src != a.read_text() and src != b.read_text()
    assert bench.engine_hash(str(a)) == bench.engine_hash()
    assert bench.engine_hash(str(b)) != bench.engine_hash()


def _launcher_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RP_DIST_BACKEND"] = "gloo"
    env.update(extra)
    return env


@pytest.mark.timeout(120)
def test_a_dead_rank_ends_the_whole_launch_quickly():
    """spawn_ranks polls all children: a rank that dies (here before the rendezvous, so the others block in it) makes the parent
    stop the rest and return that rank's exit code within seconds, instead of waiting for the survivors until the driver's limit."""
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=_launcher_env(RP_BENCH_FAIL_RANK="2"),
                       capture_output=True, text=True, timeout=100)
    dt = time.time() - t0
    assert p.returncode == 3, (p.returncode, p.stderr[-1000:])
    assert dt < 10.0, "parent took %.1f s to give up on a dead rank" % dt
    assert "dry_run" not in p.stdout  # no JSON line from a broken job


def test_wait_ranks_kills_survivors_that_ignore_sigterm():
    import time
    sleeper = "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(60)"
    procs = [subprocess.Popen([sys.executable, "-c", sleeper]), subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.5); sys.exit(7)"])]
    t0 = time.time()
    rc = bench.wait_ranks(procs, poll_s=0.05, grace_s=1.0)
    assert rc == 7 and time.time() - t0 < 8.0
    assert all(p.poll() is not None for p in procs)
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(3)]
    assert bench.wait_ranks(ok, poll_s=0.05) == 0


@pytest.mark.timeout(300)
def test_eight_rank_dry_run_on_gloo():
    """The driver's N = 8 launch shape, rehearsed on the CPU: eight ranks rendezvous on 127.0.0.1, agree through a MAX and a SUM
    all-reduce and rank 0 prints one line with n_gpus = ranks_joined = 8."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], env=_launcher_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["ranks_joined"] == 8 and line["backend"] == "gloo" and line["requested"]["gpus"] == 8
