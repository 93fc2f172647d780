#!/usr/bin/env python3
"""Benchmark of the self-play hot path (BASELINE.json metric: MCTS node expansions/s + self-play episodes/s, 20x20 bin).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One STEP = one pass of the hot path over one batch of synthetic instances: a pool of `--pool` guillotine-split
20x20 / 32-item instances (BASELINE.json configs[2]: 20x20 bin, 32 items, 400 MCTS sims per move) is played to the end
through `--games` concurrent game slots per GPU (search kernels + FP32 CNN evaluator, inputs resident in HBM).  Every
rank plays its own pool (weak scaling, no data-path collective); value = episodes all ranks finished / max-over-ranks time.

Wall budget.  A c3 pool is 32 768 whole episodes (~30 s), so `--steps 20 --warmup 5` would be ~14 minutes.  The run
therefore works against a wall budget (`--budget`, default 380 s from process start, covering the CPU baseline, start-up,
graph capture, a short warm-up pass and the pools): it plays whole pools back to back until K + W are done or the next pool
would not fit, then counts the LAST min(K, done) pools as the timed steps (earlier ones as warm-up) and reports the ACTUAL
`steps`, `warmup` and `ms_per_step`.  The named configuration never shrinks.  The JSON line is printed after every
completed pool (last line wins), so a run that is cut short still leaves a parsable line.

`--gpus N` without WORLD_SIZE in the environment starts the N ranks itself (children are spawned before anything touches
the GPU; the parent never does).

The JSON line also carries
  roofline        the dominant kernel (FP32 MFMA bound) from HIP-event timings taken inside the timed region
  roofline_tree   the hand-written tree-walk kernels (k_search + k_commit, HBM bound): algorithmic bytes / event time
  cpu_baseline    the C oracle (oracle/rp_oracle.c) + the same CNN through PyTorch CPU, batch 1 per leaf as the
                  reference does, on a bounded sample of the same workload, rank 0 at N = 1 only
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

T_PROCESS_START = time.time()

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {  # name: W, H, N, sims, FLOPs per evaluator forward, default concurrent games per GPU (sized to HBM),
    # legal-move arena = node arena x this (measured peaks with level reclaim: 1.3 MB per slot at c3, 85 MB at c5)
    "c1": (10, 10, 8, 25, 2.18e6, 4096, 24),
    "c2": (10, 10, 8, 100, 2.18e6, 4096, 24),
    "c3": (20, 20, 32, 400, 10.01e6, 32768, 24),
    "c4": (20, 20, 32, 100, 10.01e6, 32768, 24),
    "c5": (50, 50, 128, 800, 133.4e6, 768, 224),  # 221 MB per slot: 768 games = 170 GB of the 288 GB (legal-move arena peak of a whole pool:
    # 102 MB per slot, profiles/r03_b_c5_full_pool_768games.json; 1 024 games x 160 overflowed it 100 k waves in)
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak
DEFAULT_BUDGET_S = 380.0       # the driver kills the command at 600 s
FINAL_RESERVE_S = 10.0         # standalone kernel timing + final line after the last pool
ENGINE_SRC = os.path.join(ROOT, "resource_packing_self_play_amd", "csrc", "rp_engine.hip")


# ---------------------------------------------------------------------------------------------------------------------
# budget planner (pure functions: tests/test_bench_contract.py runs them on the CPU)
# ---------------------------------------------------------------------------------------------------------------------
def classify_pools(n_done, steps, warmup):
    """(warm-up pools, timed pools) among n_done completed pools: the LAST pools are the timed steps, up to `steps`; what
    ran before them counts as warm-up, at most `warmup`.  When the budget cut the run short the first pool still stays
    untimed (if a warm-up was asked for and a second pool exists), so a timed step never includes first-touch effects."""
    floor = 1 if (int(warmup) > 0 and n_done >= 2) else 0
    warm = min(int(warmup), max(floor, n_done - int(steps)))
    return warm, n_done - warm


def should_continue(elapsed_s, n_done, longest_pool_s, steps, warmup, budget_s, reserve_s=FINAL_RESERVE_S):
    """Start another pool?  Always play one; stop at steps + warmup; otherwise only if a pool 10 % longer than the longest
    so far still ends inside the budget with the finalisation reserve left."""
    if n_done == 0:
        return True
    if n_done >= int(steps) + int(warmup):
        return False
    return elapsed_s + 1.1 * longest_pool_s + reserve_s <= budget_s


def simulate_plan(startup_s, pool_s, steps, warmup, budget_s=DEFAULT_BUDGET_S, reserve_s=FINAL_RESERVE_S):
    """Planned wall time of a run whose start-up (CPU baseline, imports, capture, warm-up pass) takes startup_s and whose
    pools take pool_s each.  Returns (pools played, warm-up pools, timed pools, wall seconds)."""
    t, n = float(startup_s), 0
    while should_continue(t, n, pool_s, steps, warmup, budget_s, reserve_s):
        t += pool_s
        n += 1
    warm, timed = classify_pools(n, steps, warmup)
    return n, warm, timed, t + reserve_s


def engine_hash(path=None):
    """SHA-256 of the engine source as the compiler sees it, near enough: comments removed, runs of white space collapsed -- a
    reworded comment does not make the committed PMC figures stale, any change of code does."""
    import re
    try:
        src = open(path or ENGINE_SRC, "r", encoding="utf-8", errors="replace").read()
    except OSError:
        return None
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"\s+", " ", src).strip()
    return hashlib.sha256(src.encode("utf-8")).hexdigest()


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (scripts/pmc_traffic.py).  PMC counters cannot be read
    from inside the run, so the file carries the SHA-256 of the rp_engine.hip it was measured on; a different engine
    source means the figures are stale and `traffic` is reported as null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_traffic.json missing"
    rec = json.load(open(path))
    if rec.get("engine_sha256") != engine_hash():
        return None, "profiles/pmc_traffic.json was measured on another rp_engine.hip (hash mismatch): stale, not reported"
    return rec, None


def make_instances(W, H, N, count, base_seed):
    """Guillotine splits of the W x H rectangle (h_gen = H, the hardest case: a perfect packing exists), one per
    generator seed base_seed + index, through the package's ItemsGenerator."""
    from resource_packing_self_play_amd.binpacking.BinPackingGame import ItemsGenerator
    gen = ItemsGenerator(W, H, N)
    state = np.random.get_state()
    wh = np.array([[it[:2] for it in gen.items_generator(base_seed + k)] for k in range(count)], dtype=np.uint8)
    np.random.set_state(state)
    return wh


def rank_buffer():
    return np.random.RandomState(12345).uniform(0.8, 1.0, 100)


class Args(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)
    __setattr__ = dict.__setitem__


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline worker (separate process, never touches the GPU)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_worker(cfg, episodes, seed0, budget_s):
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    from resource_packing_self_play_amd.binpacking.pytorch.BinpackingNNet import BinPackingNNet
    W, H, N, sims = CONFIGS[cfg][:4]

    class G:
        def getBoardSize(self): return (H, W)
        def getActionSize(self): return W * N
    torch.manual_seed(0)
    net = BinPackingNNet(G(), Args(num_items=N, num_bins=1)).eval()
    wh_all = make_instances(W, H, N, episodes, seed0)
    planes = np.zeros((1, N + 1, H, W), np.float32)
    cur = {}

    def evaluate(board, rem):  # NNetWrapper.predict on CPU, batch 1 (NNet.py:69-85)
        planes[0, 0] = board
        wh = cur["wh"]
        for i in range(N):
            planes[0, i + 1] = 0
            if rem[i]:
                planes[0, i + 1, :wh[i, 1], :wh[i, 0]] = 1
        with torch.no_grad():
            lp, v = net(torch.from_numpy(planes))
        return torch.exp(lp)[0].numpy(), v[0].numpy()

    m = orc.OracleMCTS(W, H, N, 1.0, 0.75, evaluate, None)
    buf = rank_buffer()
    t0 = time.time(); done = 0; expansions = 0; searches = 0
    for k in range(episodes):
        cur["wh"] = wh_all[k]
        m.begin_episode(wh_all[k, :, 0], wh_all[k, :, 1], W * H, buf)
        m.play_episode(sims, policy=1, seed=1, episode_id=k, want_counts=False)
        st = m.stats()
        done += 1; expansions += st["expansions"]; searches += st["searches"]
        if time.time() - t0 > budget_s:
            break
    print(json.dumps({"episodes": done, "seconds": time.time() - t0, "expansions": expansions, "simulations": searches}))


def host_cores():
    """(cores this process may run on, cgroup CPU quota in cores or None).  The baseline uses every core it is allowed: the
    affinity count, lowered to the container's CPU quota when that is smaller (more workers than the quota would only time-slice)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return aff, quota


def run_cpu_baseline(cfg, budget_s):
    """Spawns one single-threaded worker per host core available to this process (no cap); each plays whole episodes of the
    bench workload for about `budget_s` seconds.  Returns the cpu_baseline object."""
    aff, quota = host_cores()
    workers = max(1, aff if quota is None else min(aff, int(quota + 0.5) or 1))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", cfg, "--cpu-episodes", "64",
                               "--cpu-seed", str(100 + 1000 * w), "--cpu-budget", str(budget_s)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                              env=dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
             for w in range(workers)]
    outs = []
    for p in procs:
        out, _ = p.communicate()
        try:
            outs.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            pass
    if not outs:
        return None
    wall = max(o["seconds"] for o in outs)
    eps = sum(o["episodes"] for o in outs)
    exp = sum(o["expansions"] for o in outs)
    W, H, N, sims = CONFIGS[cfg][:4]
    return {"value": eps / wall, "unit": "episodes/s", "cores": len(outs), "kind": "port",
            "host_cores_affinity": aff, "host_cpu_quota": quota,
            "expansions_per_s": exp / wall, "per_core_episodes_per_s": eps / wall / len(outs),
            "sample": "%d whole episodes of the bench workload (%dx%d, %d items, %d sims/move, seeds 100+), C oracle search + "
                      "the same CNN via PyTorch CPU at batch 1 per leaf, one single-threaded process per core (%d workers = every "
                      "core this process may use), %.1f s wall" % (eps, W, H, N, sims, len(outs), wall)}


# ---------------------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the ranks ourselves, before anything in this process touches the GPU
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """One child process per rank with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Rank 0's stdout
    is this process's stdout (it prints the JSON line).  Returns the first non-zero exit code, or 0."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RP_BENCH_T0=repr(T_PROCESS_START))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    return wait_ranks(procs)


def wait_ranks(procs, poll_s=0.2, grace_s=5.0):
    """Polls ALL children: the first one that exits non-zero ends the job -- the others would sit in a collective (or the
    rendezvous) waiting for it until the driver's limit.  They get SIGTERM, then SIGKILL after `grace_s`.  Returns that exit
    code, or 0 when every rank ended cleanly."""
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                for q in live:
                    q.terminate()
                t_end = time.time() + grace_s
                for q in live:
                    try:
                        q.wait(timeout=max(0.0, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                        q.wait()
                return rc
        if live:
            time.sleep(poll_s)
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# the JSON line (pure: tests build it from synthetic statistics)
# ---------------------------------------------------------------------------------------------------------------------
def key_bytes(W, H, N):
    return (8 if W > 32 else 4) * H + (N + 31) // 32 * 4


def stage_kernel_flops(W, H):
    """Algorithmic flops per leaf of the fused evaluator kernels: 2 * 9 * Cin * Cout * pixels per convolution."""
    Hs, Ws = (H + 1) // 2, (W + 1) // 2
    Hs2, Ws2 = (Hs + 1) // 2, (Ws + 1) // 2
    Hs3, Ws3 = (Hs2 + 1) // 2, (Ws2 + 1) // 2
    return {"k_resstage16 %dx%d" % (Hs, Ws): 4 * 2 * 9 * 16 * 16 * Hs * Ws,
            "k_convpool32 16->32 %dx%d" % (Hs, Ws): 2 * 9 * 16 * 32 * Hs * Ws,
            "k_resstage32 %dx%d" % (Hs2, Ws2): 4 * 2 * 9 * 32 * 32 * Hs2 * Ws2,
            "k_convpool32 32->32 %dx%d" % (Hs2, Ws2): 2 * 9 * 32 * 32 * Hs2 * Ws2,
            "k_resstage32 %dx%d" % (Hs3, Ws3): 4 * 2 * 9 * 32 * 32 * Hs3 * Ws3}


def build_output(cfg_name, W, H, N, sims, flops_leaf, games, pool, world, steps, warmup, requested, dt, tot, waves_rank0, groups, slots_group0,
                 use_stem, compact_rows, per_wave, kernel_ms, solo, cpu_base, mean_score, mean_moves, extra, timed_rows=None):
    """tot: counters summed over ranks for the timed pools (+ 'waves'); per_wave: mean ms per wave of (search, stem/planes,
    evaluator, commit) from the event-timed waves; kernel_ms: {kernel label: mean ms per launch} from HIP events."""
    episodes = tot["episodes"]
    K, A = key_bytes(W, H, N), W * N
    # algorithmic bytes of the tree walk with THIS layout.  Per selected node: 32 B header, 6 B best unvisited
    # candidate, 22 B per visited entry (idx, N, Q, P), 6 B (child, action); 26 B per visited entry created (+ the rescan below);
    # per backed-up edge 32 B (Q, N, Ns read + write); per expansion: key write + compare 2K, one 128 B probe bucket (16 slots),
    # 32 B header write, 2 B action write per legal move (k_search) and, in k_commit, the 64 B header read-modify-write,
    # the evaluator's 4A + 4 B output, 2 B action read + 4 B prior write per legal move.
    # (round 3) the unvisited moves of a node are represented by ONE cached candidate (6 B: its pi and action); the prior run is scanned
    # (4 B per legal move) only when that candidate is visited for the first time, i.e. once per new visited entry
    mean_valid_sel = tot["sum_valid_select"] / max(tot["path_edges"], 1)
    sel_bytes = 44.0 * tot["path_edges"] + 22.0 * tot["sum_visited_select"] + (26.0 + 4.0 * mean_valid_sel) * tot["visited_new"]
    bak_bytes = 32.0 * tot["path_edges"]
    exp_bytes = tot["expansions"] * (2 * K + 128 + 32 + 64 + 4 * A + 4) + 8.0 * tot["sum_valid_leaf"]
    tree_bytes = sel_bytes + bak_bytes + exp_bytes
    tree_ms = per_wave[0] + per_wave[3]
    launches = max(tot["waves"], 1) * groups  # one k_search / evaluator / k_commit launch per group and wave
    tree_bytes_per_wave = tree_bytes / launches
    leaves_per_wave = tot["expansions"] / launches
    stem_flops = 2 * 9 * (N + 1) * 16 * H * W if use_stem else 0  # first convolution: table sums in k_leaf_stem, not matrix-core work
    # evaluator time per wave: with one slot group the waves run back to back, so it is the wall time per wave minus the three engine
    # phases (single kernels, timed by events); the eagerly launched evaluator of the event-timed waves also carries the host's
    # launch gaps between its kernels and is only the fallback for --groups > 1
    eval_ms = per_wave[2]
    if groups == 1 and waves_rank0 > 0:
        eval_ms = dt * 1e3 / waves_rank0 - (per_wave[0] + per_wave[1] + per_wave[3])
    rows = leaves_per_wave if compact_rows else slots_group0
    nn_tflops = rows * (flops_leaf - stem_flops) / (eval_ms * 1e-3) / 1e12 if eval_ms > 0 else 0.0
    kflops = stage_kernel_flops(W, H)
    Hs, Ws = (H + 1) // 2, (W + 1) // 2
    Hs2, Ws2 = (Hs + 1) // 2, (Ws + 1) // 2
    Hs3, Ws3 = (Hs2 + 1) // 2, (Ws2 + 1) // 2
    kbytes = {"k_resstage16 %dx%d" % (Hs, Ws): 2 * 4 * 16 * Hs * Ws,  # x in, result out, per leaf
              "k_resstage32 %dx%d" % (Hs2, Ws2): 2 * 4 * 32 * Hs2 * Ws2, "k_resstage32 %dx%d" % (Hs3, Ws3): 2 * 4 * 32 * Hs3 * Ws3,
              "k_convpool32 16->32 %dx%d" % (Hs, Ws): 4 * 16 * Hs * Ws + 4 * 32 * Hs2 * Ws2,
              "k_convpool32 32->32 %dx%d" % (Hs2, Ws2): 4 * 32 * Hs2 * Ws2 + 4 * 32 * Hs3 * Ws3}
    pmc, pmc_note = load_pmc_traffic()
    roof = None
    timed = {k: v for k, v in kernel_ms.items() if k in kflops}

    # per-kernel figures: numerator and denominator from the SAME launches -- the leaves waiting in the event-timed waves themselves
    # (read back with rp_leaf_count_async), not the mean over all waves
    krows = timed_rows if (timed_rows and compact_rows) else rows

    def kernel_roofline(kname):
        rows = krows
        kms, klaunches = timed[kname]
        ach = rows * kflops[kname] / (kms * 1e-3) / 1e12
        traffic = None
        if pmc is not None:
            # a kernel with several template instances in a wave (5x5 / 3x3 stages) appears once per instance in the PMC file
            # ("k_resstage32<5>"): take the instance whose bytes per leaf are nearest this launch's algorithmic bytes
            base = kname.split(" ")[0]
            cands = [r for k, r in pmc.get("kernels", {}).items() if (k == base or k.startswith(base + "<")) and r.get("leaves_per_launch")]
            if cands:  # the PMC passes cover the pool's first waves (every slot waiting): scale per leaf
                rec = min(cands, key=lambda r: abs(r["hbm_bytes_per_launch"] / r["leaves_per_launch"] - kbytes.get(kname, 0)))
                traffic = rec["hbm_bytes_per_launch"] / rec["leaves_per_launch"] * rows
        return {"kernel": kname + " (FP32 MFMA, own HIP kernel)", "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_note": pmc_note, "flops_per_launch": rows * kflops[kname],
                "ms_per_launch": kms, "leaves_per_launch": rows, "launches_timed": klaunches,
                "algorithmic_bytes_per_launch": rows * kbytes.get(kname, 0) or None}

    roof_kernels = {k: kernel_roofline(k) for k in sorted(timed)}
    if solo:  # k_resstage16 alone on ALL slot rows (random-free: the last stem output) after the timed region
        for k, r in roof_kernels.items():
            if k.startswith("k_resstage16 "):
                r["standalone"] = solo
                r["note"] = "achieved/frac: HIP events around the kernel's launches inside the timed region; standalone: the same kernel on all slot rows after the timed region"
    if timed:
        roof = roof_kernels[max(timed, key=lambda k: timed[k][0])]  # the dominant kernel by measured time
    if roof is None:
        roof = {"kernel": "CNN evaluator (all kernels of one forward over the slot batch)", "bound": "mfma", "achieved": nn_tflops,
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": nn_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": None}
    tree_traffic = None
    if pmc is not None and all(k in pmc.get("kernels", {}) for k in ("k_search", "k_commit")):
        ks = pmc["kernels"]
        tree_traffic = sum(ks[k]["hbm_bytes_per_launch"] / ks[k]["leaves_per_launch"] for k in ("k_search", "k_commit")) * rows
    tree_gbs = tree_bytes_per_wave / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
    out = {
        "metric": "self-play episodes/sec (with MCTS node expansions/sec alongside)", "value": episodes / dt, "unit": "episodes/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / max(steps, 1) * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64 PUCT / f32 Q + CNN, u32 bit-board", "data": "synthetic",
        "config": {"workload": "%s: %dx%d bin, %d items, %d MCTS sims/move, %d concurrent games per GPU, pool of %d instances per step per GPU"
                               % (cfg_name, W, H, N, sims, games, pool), "evaluator": "BinPackingNNet FP32 via PyTorch-ROCm, torch.manual_seed(0) init",
                   "move_rule": "sample ~ visit counts", "parallelism": "dp%d (episodes sharded, no data-path collective)" % world},
        "requested": requested,
        "expansions_per_s": tot["expansions"] / dt, "simulations_per_s": tot["simulations"] / dt, "episodes": episodes,
        "waves": tot["waves"], "mean_moves_per_episode": mean_moves, "mean_score": mean_score,
        "tree_stats": {"path_edges_per_sim": tot["path_edges"] / max(tot["simulations"], 1),
                       "valid_per_selected_node": tot["sum_valid_select"] / max(tot["path_edges"], 1),
                       "valid_per_leaf": tot["sum_valid_leaf"] / max(tot["expansions"], 1),
                       "expansions_per_sim": tot["expansions"] / max(tot["simulations"], 1),
                       "transposition_links": tot["transposition_links"], "nodes": tot["nodes"]},
        "phase_ms_per_launch": {"search": per_wave[0], "leaf_stem" if use_stem else "leaf_planes": per_wave[1], "evaluator": per_wave[2], "commit": per_wave[3]},
        "roofline": roof,
        "roofline_kernels": roof_kernels,  # every own MFMA kernel of the evaluator, same fields
        "roofline_evaluator": {"kernel": "whole CNN evaluator after the stem (fused MFMA kernels + heads) over the slot batch", "bound": "mfma",
                               "achieved": nn_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": nn_tflops / PEAK_FP32_MFMA_TFLOPS,
                               "traffic": None, "flops_per_leaf": flops_leaf - stem_flops, "flops_per_leaf_with_first_conv": flops_leaf,
                               "leaves_per_launch": rows, "ms_per_launch": eval_ms},
        "kernel_ms_per_launch": {k: v[0] for k, v in sorted(kernel_ms.items())},
        "roofline_tree": {"kernel": "k_search + k_commit", "bound": "hbm", "achieved": tree_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": tree_gbs / PEAK_HBM_GBS, "traffic": tree_traffic,
                          "bytes_per_launch": tree_bytes_per_wave, "bytes_per_sim": tree_bytes / max(tot["simulations"], 1),
                          "leaves_per_launch": leaves_per_wave, "slots_per_launch": slots_group0},
        "cpu_baseline": cpu_base,
    }
    out.update(extra)
    if cpu_base:
        out["speedup_vs_cpu_baseline"] = out["value"] / cpu_base["value"]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# --coach-iter: one CoachBPP iteration at the configuration's scale (BASELINE configs[3]: data-parallel self-play, RCCL all-gather of
# the replay, all-reduce of the gradients)
# ---------------------------------------------------------------------------------------------------------------------
def coach_iteration(a, W, H, N, sims, games, game, nnet, rank, world, joined, dev, allreduce, requested, t_start, cfg_name=None):
    """Plays and trains ONE CoachBPP iteration; returns the JSON line (rank 0) or None (other ranks)."""
    cfg_name = cfg_name or a.config
    import torch
    import torch.distributed as dist
    from resource_packing_self_play_amd import distributed as rdist
    from resource_packing_self_play_amd.CoachBPP import CoachBPP
    from resource_packing_self_play_amd.binpacking.BinPackingGame import ItemsGenerator
    node_cap = sims * (N + 1) + 2
    pow2 = lambda v: 1 << max(0, int(v - 1).bit_length())
    pchunk, vchunk = max(4096, pow2(W * N)), max(1024, pow2(W * N))
    edge_cap = max(node_cap * a.edge_factor, (min(sims, N) + 3) * pchunk)
    vis_cap = max(int(node_cap * a.vis_factor), (min(sims, N) + 3) * vchunk)
    n_eps = games * world
    args = nnet.args
    args.update(numIters=1, numEps=n_eps, iterStepThreshold=1 << 30, binH_min=H, binH=H, numScoresForRank=100, numItems=N,
                numItersForTrainExamplesHistory=50, maxlenOfQueue=200000, epochs=1, batch_size=64, max_train_steps_per_epoch=a.train_steps,
                checkpoint=os.environ.get("RP_BENCH_CKPT", "/tmp/rp_bench_coach_r%d" % rank), games_per_gpu=games, node_cap=node_cap, edge_cap=edge_cap,
                vis_cap=vis_cap, groups=a.groups, use_graph=not a.no_graph, sample_seed=12345)
    gen = ItemsGenerator(W, H, N)
    coach = CoachBPP(game, nnet, gen.items_generator(100), W * H, gen, args, saved_rewards_list=list(rank_buffer()))
    if rdist.collectives_on():  # more than one rank, or a forced one-rank group (RP_DIST_FORCE=1: RCCL smoke run on one GPU)
        rdist.attach(nnet)
        nnet.grad_hook.timing = []
    nnet.step_timing = []
    coach.drawIteration = lambda: (H, list(range(100, 100 + n_eps)))  # the bench's instances: seeds 100 + global episode index, full-height rectangle
    sp = coach._driver(n_eps)
    sp.prepare()  # evaluator warm-up + graph capture outside the timed iteration
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.time()
    coach.learn()
    torch.cuda.synchronize(dev)
    wall = time.time() - t0
    tm = coach.timings[-1]
    hook = nnet.grad_hook
    ar_ms = [e0.elapsed_time(e1) for e0, e1 in hook.timing] if (hook is not None and hook.timing) else []
    step_ms = [e0.elapsed_time(e1) for e0, e1 in nnet.step_timing]
    steady = step_ms[len(step_ms) // 2:]  # the first half carries MIOpen's kernel search for the training shapes
    step_med = allreduce([float(np.median(steady)) if steady else 0.0], dist.ReduceOp.MAX)[0]
    ar_steady = ar_ms[len(ar_ms) // 2:]
    play_s, exch_ms, train_s, wall_max = allreduce([tm["selfplay_s"], tm["exchange"]["ms"], tm.get("train_s", 0.0), wall], dist.ReduceOp.MAX)
    ar_mean = allreduce([float(np.median(ar_steady)) if ar_steady else 0.0], dist.ReduceOp.MAX)[0]
    c = sp.counters()
    exp_tot, sim_tot = allreduce([float(c["expansions"]), float(c["simulations"])], dist.ReduceOp.SUM)
    sp.close()
    if rank != 0:
        return None
    steps = int(tm.get("train_steps", 0))
    grad_bytes = 4 * sum(p.numel() for p in nnet.nnet.parameters())
    out = {"metric": "self-play episodes/sec of one CoachBPP iteration (with the iteration's replay exchange and training alongside)",
           "value": n_eps / play_s, "unit": "episodes/s", "n_gpus": world, "steps": 1, "warmup": 0, "ms_per_step": wall_max * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64 PUCT / f32 Q + CNN, u32 bit-board", "data": "synthetic",
           "config": {"workload": "%s: one CoachBPP iteration, %dx%d bin, %d items, %d MCTS sims/move, numEps = %d (%d concurrent games per GPU x %d ranks), "
                                  "replay all-gather + %d optimiser steps of batch 64 with gradient all-reduce" % (cfg_name, W, H, N, sims, n_eps, games, world, steps),
                      "parallelism": "dp%d (episodes sharded; all-gather of the packed replay, all-reduce of the gradients)" % world},
           "requested": requested, "ranks_joined": joined, "backend": dist.get_backend() if dist.is_initialized() else None,
           "coach": {"numEps": n_eps, "selfplay_s": play_s, "expansions_per_s": exp_tot / play_s, "simulations_per_s": sim_tot / play_s,
                     "examples": tm["examples"], "replay_bytes": tm["replay_bytes"], "replay_bytes_per_example": tm["replay_bytes"] / max(tm["examples"], 1),
                     "dense_bytes_per_example": 4 * ((N + 1) * H * W + W * N + 1),
                     "allgather_bytes_sent_per_rank": tm["exchange"]["bytes_sent"], "allgather_bytes_received_per_rank": tm["exchange"]["bytes_received"],
                     "allgather_ms": exch_ms, "train_set_examples": tm.get("train_examples"), "train_set_bytes": tm.get("train_set_bytes"),
                     "maxlenOfQueue": 200000, "train_steps": steps, "train_s": train_s, "train_steps_per_s_incl_kernel_search": steps / train_s if train_s > 0 else None,
                     "train_ms_per_step": step_med or None, "train_steps_per_s": 1e3 / step_med if step_med else None,
                     "train_note": "ms per step: median over the second half of the timed steps (HIP events around one whole step incl. the minibatch expansion and "
                                   "the gradient all-reduce); the first steps carry MIOpen's kernel search for the training shapes",
                     "grad_allreduce_bytes": grad_bytes + 8, "grad_allreduce_ms_per_step": ar_mean if ar_ms else None, "grad_allreduce_calls": len(ar_ms),
                     "iteration_wall_s": wall_max, "mean_score": float(np.mean(coach.iteration_scores[-1]))},
           "wall_s_since_start": time.time() - t_start}
    return out


# ---------------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--budget", type=float, default=float(os.environ.get("RP_BENCH_BUDGET_S", DEFAULT_BUDGET_S)),
                    help="wall budget in seconds from process start; whole pools are played while the next one fits")
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="default c3 (c4 with --coach-iter)")
    ap.add_argument("--games", type=int, default=0, help="concurrent game slots per GPU (default: the configuration's, 32 768 for c3)")
    ap.add_argument("--pool", type=int, default=0, help="instances per step per GPU (default = games: every slot plays one episode)")
    ap.add_argument("--sims", type=int, default=0)
    ap.add_argument("--edge-factor", type=int, default=0, help="legal-move arena = node arena x this (6 B per entry; default: the configuration's)")
    ap.add_argument("--vis-factor", type=float, default=1.5, help="visited-edge arena = node arena x this (32 B per entry)")
    ap.add_argument("--no-reclaim", action="store_true", help="keep dead levels' arena chunks (needs ~5x the arena)")
    ap.add_argument("--groups", type=int, default=1, help="slot groups per GPU, each with its own stream and graph; 3 co-schedules the groups' kernels "
                    "(+4 %% episodes/s) but then no kernel has the GPU to itself and per-kernel durations stop meaning anything")
    ap.add_argument("--step-cap", type=int, default=4, help="max simulations a slot runs per wave (bounds the launch tail; 3-6 measure the same, 16: -1.6 %)")
    ap.add_argument("--no-stem", action="store_true", help="feed FP32 planes to the full CNN instead of computing conv1 + pool in the engine")
    ap.add_argument("--no-fuse", action="store_true", help="leave bias / ReLU / skip / pool to PyTorch's own element-wise kernels")
    ap.add_argument("--no-dense", action="store_true", help="keep <= 3x3-image convolutions on MIOpen instead of one GEMM each")
    ap.add_argument("--no-resblock", action="store_true", help="16-channel residual blocks through MIOpen + fused element-wise kernels instead of the stage kernels")
    ap.add_argument("--nchw", action="store_true", help="keep the evaluator's activations NCHW instead of channels-last")
    ap.add_argument("--no-compact", action="store_true", help="evaluator row b = slot b (every slot costs convolution work) instead of the waiting slots only")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--warm-waves", type=int, default=96, help="waves of the short warm-up pass before the first pool")
    ap.add_argument("--event-every", type=int, default=64, help="take per-phase / per-kernel HIP-event timings every n-th wave (those waves run eagerly between events instead of as one graph launch: 16 cost 0.4 %% of the episodes/s, 64 still time ~190 launches of every kernel per pool)")
    ap.add_argument("--profile-waves", type=int, default=0, help="profiling aid: stop after this many waves per group and print no metric")
    ap.add_argument("--waves", type=int, default=0, help="bounded run for configurations whose pool takes many minutes (c5: ~100 k waves): play this many waves of "
                    "one pool and report expansions/s and simulations/s of that window (no episodes/s: no episode ends inside it)")
    ap.add_argument("--coach-iter", action="store_true", help="time ONE CoachBPP iteration (CoachBPP.py:123-176) instead of self-play pools: self-play of numEps = "
                    "games x ranks episodes, the replay all-gather, training steps with the gradient all-reduce; default configuration c4 (BASELINE configs[3])")
    ap.add_argument("--no-coach-block", action="store_true", help="with several ranks the run ends with one short CoachBPP iteration (c4's 100 sims/move, "
                    "<= 8 192 games per GPU) so that the line carries the two exchanges of BASELINE configs[3] -- replay all-gather and gradient all-reduce -- "
                    "timed over RCCL; this switches it off")
    ap.add_argument("--coach-block-s", type=float, default=100.0, help="hard deadline of that block: past it the process exits 0 and the line printed before stands")
    ap.add_argument("--train-steps", type=int, default=200, help="--coach-iter: optimiser steps that are timed (the reference's epochs x len / batch would be ~30 000)")
    ap.add_argument("--dry-run", action="store_true", help="launch plumbing only (ranks, process group, collectives, JSON line) without the GPU work")
    ap.add_argument("--cpu-worker", default=None)
    ap.add_argument("--cpu-episodes", type=int, default=1)
    ap.add_argument("--cpu-seed", type=int, default=100)
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    a.config = a.config or ("c4" if a.coach_iter else "c3")
    if a.cpu_worker:
        return cpu_worker(a.cpu_worker, a.cpu_episodes, a.cpu_seed, a.cpu_budget)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher: become one (this process never touches the GPU)
        sys.exit(spawn_ranks(a.gpus, argv))

    t_start = float(os.environ.get("RP_BENCH_T0", T_PROCESS_START))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not a.no_coach_block and a.budget == DEFAULT_BUDGET_S:
        a.budget = DEFAULT_BUDGET_S - 60.0  # room for the coach block behind the pools
    W, H, N, sims, flops_leaf, games_default, edge_factor_default = CONFIGS[a.config]
    a.edge_factor = a.edge_factor or edge_factor_default
    sims = a.sims or sims
    games = a.games or games_default
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_base = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.sims and not a.dry_run and not a.coach_iter:  # N = 1 only, before this process touches the GPU
        cpu_base = run_cpu_baseline(a.config, a.cpu_budget)
    t_cpu_done = time.time()

    if a.dry_run and os.environ.get("RP_BENCH_FAIL_RANK") == str(rank):  # test hook: a rank that dies before the rendezvous
        print("rank %d: failing on purpose (RP_BENCH_FAIL_RANK)" % rank, file=sys.stderr)
        sys.exit(3)
    import torch
    import torch.distributed as dist
    from resource_packing_self_play_amd import distributed as rdist
    rank, world, local = rdist.init_from_env()
    joined = dist.get_world_size() if dist.is_initialized() else 1
    host_collectives = dist.is_initialized() and dist.get_backend() == "gloo"
    requested = {"gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "budget_s": a.budget}

    def allreduce(vals, op):
        dev_ = torch.device("cpu") if (a.dry_run or host_collectives) else torch.device("cuda", local)
        t = torch.tensor(list(vals), dtype=torch.float64, device=dev_)
        if world > 1:
            dist.all_reduce(t, op=op)
        return t.tolist()

    if a.dry_run:
        if world > 1:
            dist.barrier()
        dt = allreduce([1.0 + 0.01 * rank], dist.ReduceOp.MAX)[0]
        ranks = allreduce([1.0], dist.ReduceOp.SUM)[0]
        if rank == 0:
            print(json.dumps({"metric": "self-play episodes/sec (with MCTS node expansions/sec alongside)", "value": None, "unit": "episodes/s",
                              "n_gpus": joined, "ranks_joined": int(ranks), "steps": 0, "warmup": 0, "ms_per_step": dt * 1e3, "dry_run": True,
                              "requested": requested, "backend": dist.get_backend() if dist.is_initialized() else None}), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    from resource_packing_self_play_amd import _lib
    from resource_packing_self_play_amd.binpacking.BinPackingGame import BinPackingGame
    from resource_packing_self_play_amd.binpacking.pytorch.NNet import NNetWrapper
    from resource_packing_self_play_amd.selfplay import BatchedSelfPlay

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    game = BinPackingGame(W, H, N, 1)
    args = Args(numMCTSSims=sims, cpuct=1, alpha=0.75, cuda=True, num_items=N, num_bins=1, epochs=1, batch_size=64)
    torch.manual_seed(0)
    nnet = NNetWrapper(game, args)
    if a.coach_iter:
        line = coach_iteration(a, W, H, N, sims, games, game, nnet, rank, world, joined, dev, allreduce, requested, t_start)
        if line is not None:
            print(json.dumps(line), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    pool = a.pool or games
    node_cap = sims * (N + 1) + 2
    edge_cap, vis_cap = node_cap * a.edge_factor, int(node_cap * a.vis_factor)
    # every reachable level keeps one partly filled chunk open (level arenas): small or very deep configurations
    # are bounded by chunks per level, not by the entry count.  Chunk sizes as rp_create picks them: a chunk holds the largest run.
    pow2 = lambda v: 1 << max(0, int(v - 1).bit_length())
    pchunk, vchunk = max(4096, pow2(W * N)), max(1024, pow2(W * N))
    edge_cap, vis_cap = max(edge_cap, (min(sims, N) + 3) * pchunk), max(vis_cap, (min(sims, N) + 3) * vchunk)
    sp = BatchedSelfPlay(game, nnet, args, games=games, move_rule=_lib.MOVE_SAMPLE, seed=7 + rank, node_cap=node_cap,
                         edge_cap=edge_cap, use_graph=not a.no_graph, groups=a.groups, step_cap=a.step_cap, use_stem=not a.no_stem, fuse_elementwise=not a.no_fuse, dense_small_convs=not a.no_dense,
                         reclaim=not a.no_reclaim, vis_cap=vis_cap, compact_rows=not a.no_compact, channels_last=not a.nchw, resblock_kernel=not a.no_resblock)
    sp.prepare()  # evaluator warm-up + capture of the whole wave into one HIP graph, outside every timed region
    buf = rank_buffer()
    ev_every = max(1, a.event_every)
    phase_ms = np.zeros(4)  # search, planes, evaluator, commit
    phase_n = 0
    kev, kernel_ms = [], {}
    cnt_host = torch.zeros(1 << 16, dtype=torch.int32).pin_memory()  # waiting leaves of every event-timed wave (ring)
    cnt_used = [0]

    def play_pool(pool_idx, timed, max_waves=0):
        nonlocal phase_ms, phase_n
        # instances = ItemsGenerator.items_generator(seed), seeds 100 + index as in main_bpp.py:33, generated on the device
        # (bit-identical to the host generator; 32 768 instances through the Python generator would cost seconds of host time)
        sp.start_from_seeds(np.arange(pool, dtype=np.uint32) + np.uint32(100 + (pool_idx * world + rank) * pool), buf, first_id=0)
        waves = 0
        pending = []
        g0 = sp.groups[0]
        while True:
            for _ in range(32):
                if timed and waves % ev_every == 0:  # this wave of group 0 runs eagerly between HIP events on its stream
                    with torch.cuda.stream(g0.stream):
                        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                        ev[0].record(); g0.eng.search_step(sync=False)
                        if sp.compact_rows:
                            g0.eng.leaf_count_async(cnt_host, cnt_used[0] % cnt_host.numel()); cnt_used[0] += 1
                        ev[1].record(); g0.leaf_inputs()
                        g0.eng.kernel_events = kev
                        ev[2].record(); pi, v = g0.forward(sp.nnet)
                        ev[3].record(); g0.eng.kernel_events = None; g0.commit(pi, v)
                        ev[4].record()
                    for g in sp.groups[1:]:
                        sp.step_group(g)
                    sp.steps += 1
                    pending.append(ev)  # (pi, v) live on the group's stream: the caching allocator may reuse them once the commit is enqueued
                else:
                    sp.step()
                waves += 1
            if max_waves and waves >= max_waves:
                torch.cuda.synchronize(dev)
                for ev in pending:
                    phase_ms += [ev[k].elapsed_time(ev[k + 1]) for k in range(4)]
                    phase_n += 1
                for name, e0, e1 in kev:
                    kernel_ms.setdefault(name, []).append(e0.elapsed_time(e1))
                del kev[:]
                return waves, 0.0, 0.0
            if sp.active() == 0:
                break
            if a.profile_waves and waves >= a.profile_waves:
                torch.cuda.synchronize(dev)
                print("profile run: stopped after %d waves per group" % waves, file=sys.stderr)
                sys.exit(0)
        torch.cuda.synchronize(dev)
        for ev in pending:
            phase_ms += [ev[k].elapsed_time(ev[k + 1]) for k in range(4)]
            phase_n += 1
        for name, e0, e1 in kev:
            kernel_ms.setdefault(name, []).append(e0.elapsed_time(e1))
        del kev[:]
        ids, _, score, moves = sp.pop_finished()
        assert len(ids) == pool, "pool not finished: %d of %d" % (len(ids), pool)
        return waves, float(np.mean(score)), float(np.mean(moves))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if a.waves > 0:  # bounded window of one pool (c5): steady-state expansion rate, not an episodes/s figure
        play_pool(0, False, max_waves=min(64, a.waves))
        sp.counters(reset=True)
        barrier()
        t0 = time.time()
        wv, _, _ = play_pool(0, True, max_waves=a.waves)
        barrier()
        dt = allreduce([time.time() - t0], dist.ReduceOp.MAX)[0]
        c = sp.counters()
        tot_v = allreduce([float(c[k]) for k in _lib.COUNTER_NAMES], dist.ReduceOp.SUM)
        if rank == 0:
            tot = dict(zip(_lib.COUNTER_NAMES, tot_v))
            print(json.dumps({"metric": "MCTS node expansions/sec (bounded window of one pool)", "value": tot["expansions"] / dt, "unit": "expansions/s",
                              "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "f64 PUCT / f32 Q + CNN, u%d bit-board" % (64 if W > 32 else 32), "data": "synthetic",
                              "config": {"workload": "%s: %dx%d bin, %d items, %d MCTS sims/move, %d concurrent games per GPU, first %d waves of a pool"
                                                     % (a.config, W, H, N, sims, games, a.waves)},
                              "simulations_per_s": tot["simulations"] / dt, "waves": wv, "ms_per_wave": dt / wv * 1e3,
                              "leaves_per_wave": tot["expansions"] / wv, "valid_per_leaf": tot["sum_valid_leaf"] / max(tot["expansions"], 1),
                              "path_edges_per_sim": tot["path_edges"] / max(tot["simulations"], 1),
                              "phase_ms_per_launch": dict(zip(("search", "leaf_stem", "evaluator", "commit"), (phase_ms / max(phase_n, 1)).tolist())),
                              "device_bytes": sp.device_bytes, "arena_peak_per_slot": sp.arena_peak()}), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    # short warm-up pass: the first waves of a pool (every slot waits for the evaluator: the heaviest waves) through the captured
    # graph and through the event-timed eager path, then the pool is abandoned (rp_begin_pool restarts every slot)
    if a.warm_waves > 0 and not a.profile_waves:
        play_pool(0, True, max_waves=a.warm_waves)
        sp.pop_finished()
        phase_ms[:] = 0; phase_n = 0; del kev[:]; kernel_ms.clear(); cnt_used[0] = 0
    sp.counters(reset=True)
    names = list(_lib.COUNTER_NAMES)
    barrier()
    bounds = [time.time()]       # pool boundaries on this rank's clock (after barrier + synchronize)
    snaps = [np.zeros(len(names) + 1)]  # cumulative counters (+ waves) of this rank at every boundary
    scores = []
    n_done, cum_waves = 0, 0
    solo = None
    while True:
        wv, mean_score, mean_moves = play_pool(n_done, True)
        barrier()
        bounds.append(time.time())
        c = sp.counters()
        cum_waves += wv
        snaps.append(np.array([float(c[k]) for k in names] + [float(cum_waves)]))
        scores.append((mean_score, mean_moves))
        n_done += 1
        longest = max(b1 - b0 for b0, b1 in zip(bounds[:-1], bounds[1:]))
        go = should_continue(time.time() - t_start, n_done, longest, a.steps, a.warmup, a.budget)
        go = allreduce([1.0 if go else 0.0], dist.ReduceOp.MIN)[0] > 0.5 if world > 1 else go  # every rank must agree
        if not go and n_done > 0 and solo is None and "stagefrag:0" in getattr(sp.nnet.nnet, "_dense", {}) and (W + 1) // 2 * ((H + 1) // 2) <= 128:
            # the dominant stage kernel on ALL slot rows with the GPU to itself (after the timed region)
            g0 = sp.groups[0]
            dn = sp.nnet.nnet._dense
            g0.eng.set_compact_rows(False)  # the row limit would read 0 after the last episode
            with torch.cuda.stream(g0.stream):
                o = torch.empty_like(g0.stem)
                g0.eng.kernel_events = solo_ev = []
                for _ in range(24):
                    g0.eng.nn_resstage16(g0.stem, dn["stagefrag:0"], dn["stagebias:0"], o, None)
                g0.eng.kernel_events = None
            torch.cuda.synchronize(dev)
            solo_ms = float(np.mean([e0.elapsed_time(e1) for _, e0, e1 in solo_ev[4:]]))
            g0.eng.set_compact_rows(sp.compact_rows)
            solo_fl = g0.G * stage_kernel_flops(W, H)["k_resstage16 %dx%d" % ((H + 1) // 2, (W + 1) // 2)]
            solo = {"ms_per_launch": solo_ms, "leaves_per_launch": g0.G, "achieved": solo_fl / (solo_ms * 1e-3) / 1e12,
                    "frac": solo_fl / (solo_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        # ---- the JSON line for the pools played so far (last line wins) ----
        warm, timed = classify_pools(n_done, a.steps, a.warmup)
        dt_local = bounds[-1] - bounds[warm]
        delta = snaps[-1] - snaps[warm]
        dt = allreduce([dt_local], dist.ReduceOp.MAX)[0]
        tot_v = allreduce(delta.tolist(), dist.ReduceOp.SUM)
        if rank == 0:
            tot = dict(zip(names + ["waves"], tot_v))
            per_wave = phase_ms / max(phase_n, 1)
            kms = {k: (float(np.mean(v)), len(v)) for k, v in kernel_ms.items()}
            sc = scores[warm:]
            cut = (not go) and n_done < a.steps + a.warmup
            extra = {"ranks_joined": joined, "pools_played": n_done, "pools_planned": a.steps + a.warmup, "budget_cut": bool(cut),
                     "steps_note": ("one step is a whole pool of %d episodes (%.1f s here): the requested %d + %d pools do not fit the %.0f s wall "
                                    "budget that keeps this command under the driver's limit, so %d pools were played and the last %d are timed"
                                    % (pool, longest, a.steps, a.warmup, a.budget, n_done, timed)) if cut else None,
                     "pool_seconds": [b1 - b0 for b0, b1 in zip(bounds[:-1], bounds[1:])],
                     "warmup_pass_waves": a.warm_waves, "wall_s_since_start": time.time() - t_start, "startup_s": bounds[0] - t_start,
                     "cpu_baseline_s": t_cpu_done - t_start, "device_bytes": sp.device_bytes, "arena_peak_per_slot": sp.arena_peak(),
                     "final": not go}
            out = build_output(a.config, W, H, N, sims, flops_leaf, games, pool, world, timed, warm, requested, dt, tot, float(delta[-1]),
                               len(sp.groups), sp.groups[0].G, sp.use_stem, sp.compact_rows, per_wave, kms, solo, cpu_base,
                               float(np.mean([s[0] for s in sc])), float(np.mean([s[1] for s in sc])), extra,
                               timed_rows=float(cnt_host[:min(cnt_used[0], cnt_host.numel())].double().mean()) if cnt_used[0] else None)
            print(json.dumps(out), flush=True)
        if not go:
            break
    # ---- several ranks: one short CoachBPP iteration for the configuration's two exchanges (BASELINE configs[3]) ----
    # Supplementary: the line above is final and already printed.  The block runs under a hard deadline -- a rank that fails or hangs in
    # a collective makes every rank leave with exit code 0 through its own watchdog, and the printed line stands.
    if rdist.collectives_on() and not a.no_coach_block and not a.profile_waves:
        import threading
        threading.Timer(a.coach_block_s, lambda: os._exit(0)).start()
        try:
            last_out = out if rank == 0 else None
            sp.close()  # engines, captured graphs and evaluator buffers of the pools: 152 GB back before the block allocates
            torch.cuda.empty_cache()
            a.train_steps = min(a.train_steps, 100)
            torch.manual_seed(0)
            nnet2 = NNetWrapper(game, Args(numMCTSSims=100, cpuct=1, alpha=0.75, cuda=True, num_items=N, num_bins=1, epochs=1, batch_size=64))
            line = coach_iteration(a, W, H, N, 100, min(games, 8192), game, nnet2, rank, world, joined, dev, allreduce, requested, t_start, cfg_name="c4")
            if rank == 0 and line is not None and last_out is not None:
                last_out["coach"] = dict(line["coach"], workload=line["config"]["workload"], backend=line["backend"])
                last_out["wall_s_since_start"] = time.time() - t_start
                print(json.dumps(last_out), flush=True)
        except BaseException as exc:  # the pools' line stands
            print("coach block failed on rank %d: %r" % (rank, exc), file=sys.stderr, flush=True)
            os._exit(0)
        os._exit(0)  # skip process-group teardown: nothing may turn a finished run into a hang
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
