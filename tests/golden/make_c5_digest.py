#!/usr/bin/env python3
"""Digest fixture for BASELINE configs[4] at its own depth: ONE 50x50 / 128-item game at 800 MCTS sims per move (MCTS_bpp.py:37-38)
played by the pinned C oracle (oracle/rp_oracle.c, itself checked bit for bit against the fixtures make_golden.py writes from
the imported reference) with the 'peaked' table evaluator and sampled moves.  The tree (~10^5 states, ~5 x 10^7 legal moves) is too
large to rebuild with the oracle inside a GPU test's time limit, so the test compares the engine's tree with this digest.

    python tests/golden/make_c5_digest.py            (build container, ~15 min, ~10 GB of host memory)

Writes tests/golden/mcts_c5_800_digest.json: the instance, R2 buffer, seeds, every action, outcome, score, counters, node count, the
SHA-256 of the whole tree (tests/engine_util.py: tree_digest) and one digest per level (items placed)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import evaluators as ev  # noqa: E402
import oracle_lib as orc  # noqa: E402
from engine_util import tree_digest  # noqa: E402

W, H, N, SIMS = 50, 50, 128, 800
KIND, SALT, SEED, EPISODE = "peaked", 5, 9, 70


def gen_items(rng, W, H, N):  # the same splitter as tests/test_gpu_mcts.py
    items = [(W, H)]
    while len(items) < N:
        k = int(rng.integers(len(items))); w, h = items[k]
        if rng.integers(2) == 0:
            if w == 1: continue
            c = int(rng.integers(1, w)); items.pop(k); items += [(c, h), (w - c, h)]
        else:
            if h == 1: continue
            c = int(rng.integers(1, h)); items.pop(k); items += [(w, c), (w, h - c)]
    return np.array(items, np.uint8)


def main(sims=SIMS, out_name="mcts_c5_800_digest.json"):
    rng = np.random.default_rng(2020 + sims)
    wh = gen_items(rng, W, H, N)
    buf = rng.uniform(0.8, 1.0, 100)
    A = W * N
    m = orc.OracleMCTS(W, H, N, 1.0, 0.75, lambda b, r: ev.table_eval(KIND, ev.pack_board(b), r, A, SALT),
                       lambda b, r: ev.tie_value(ev.pack_board(b), r, SALT))
    m.begin_episode(wh[:, 0], wh[:, 1], W * H, buf)
    t0 = time.time()
    actions, _, outcome, score = m.play_episode(sims, policy=1, seed=SEED, episode_id=EPISODE, want_counts=False)
    t1 = time.time()
    tree = m.dump()
    digest, levels = tree_digest(tree, N)
    rec = {"generator": "tests/golden/make_c5_digest.py", "numpy": np.__version__, "W": W, "H": H, "N": N, "sims": sims, "kind": KIND,
           "salt": SALT, "seed": SEED, "episode_id": EPISODE, "cpuct": 1.0, "alpha": 0.75, "item_wh": wh.tolist(), "buf": buf.tolist(),
           "actions": [int(a) for a in actions], "outcome": int(outcome), "score": float(score), "stats": m.stats(),
           "n_nodes": len(tree), "tree_sha256": digest, "levels": {str(k): [v[0], v[1]] for k, v in levels.items()},
           "oracle_seconds": t1 - t0}
    with open(os.path.join(HERE, out_name), "w") as f:
        json.dump(rec, f, separators=(",", ":"))
    print("wrote %s: %d moves, %d nodes, outcome %d score %.6f, oracle %.0f s, dump+digest %.0f s" % (out_name, len(actions), len(tree), outcome, score, t1 - t0, time.time() - t1))
    m.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(int(sys.argv[1]), "/tmp/mcts_c5_%s_digest.json" % sys.argv[1])
    else:
        main()
