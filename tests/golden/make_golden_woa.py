#!/usr/bin/env python3
"""Golden fixtures for the ES-WOA fine-tuner, produced by RUNNING THE REFERENCE.

Build-container only (needs /root/reference; nothing of it is copied).  The reference class
``src/baselines/WOA.py::ESWOA`` draws from numpy's global generator; for the duration of a run
``np.random.random / randint / choice`` are routed to ``oracle.woa.DrawStream`` so that the run is a pure
function of (inputs, seed).  Each case stores inputs and the reference's outputs in woa_cases.json, after
asserting that the restatement in oracle/woa.py reproduces them exactly.

    python tests/golden/make_golden_woa.py
"""
import contextlib
import copy
import json
import os
import signal
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import woa as owoa          # noqa: E402


@contextlib.contextmanager
def routed_numpy_random(stream):
    saved = np.random.random, np.random.randint, np.random.choice
    np.random.random = lambda: stream.uniform()
    np.random.randint = lambda lo, hi: lo + stream.below(hi - lo)
    np.random.choice = lambda seq: seq[stream.below(len(seq))]
    try:
        yield
    finally:
        np.random.random, np.random.randint, np.random.choice = saved


def make_case(name, T, k_range, pop, iters, seed, with_solution=True, foreign_pick=False, tight=False):
    g = np.random.default_rng(seed)
    services = []
    for _ in range(T):
        n = int(g.integers(k_range[0], k_range[1] + 1))
        services.append([tuple(float(v) for v in np.r_[g.random(2), 0.9 + 0.1 * g.random(2)]) for _ in range(n)])
    lo = 0.9 ** T * (1.3 if tight else 0.6)
    constraints = [[[float(lo), 1.0]], [[float(lo), 1.0]]]
    solution = None
    if with_solution:
        solution = [list(cat[int(g.integers(0, len(cat)))]) for cat in services]
        if foreign_pick:               # a pick that is not among the candidates: WOA.py:62-69 appends it
            solution[T // 2] = [float(v) for v in np.r_[g.random(2), 0.9 + 0.1 * g.random(2)]]
    return {"name": name, "services": [[list(s) for s in cat] for cat in services], "constraints": constraints,
            "solution": solution, "pop_size": pop, "max_iter": iters, "seed": seed}


def run_reference(WOA, case):
    stream = owoa.DrawStream(case["seed"])
    services = [[tuple(s) for s in cat] for cat in case["services"]]
    solution = copy.deepcopy(case["solution"])
    with routed_numpy_random(stream):
        if solution is None:
            model = WOA.ESWOA(services, copy.deepcopy(case["constraints"]), popSize=case["pop_size"], MAX_Iter=case["max_iter"])
        else:
            model = WOA.ESWOA(services, copy.deepcopy(case["constraints"]), solution, popSize=case["pop_size"],
                              MAX_Iter=case["max_iter"])
        q, best = model.start()
    return {"best_fitness": float(q), "best_rows": [[float(v) for v in r[:4]] for r in best],
            "best_pos": [int(v) for v in model.bestPops], "history": [float(v) for v in model.bestFitnesses],
            "draws": stream.count}


def main():
    signal.alarm(600)
    sys.path.insert(0, REF)
    from src.baselines import WOA
    cases = [
        make_case("tiny", T=4, k_range=(2, 4), pop=6, iters=12, seed=1),
        make_case("no_solution", T=6, k_range=(2, 5), pop=10, iters=20, seed=2, with_solution=False),
        make_case("foreign_pick", T=7, k_range=(1, 5), pop=12, iters=25, seed=3, foreign_pick=True),
        make_case("violated", T=9, k_range=(2, 6), pop=15, iters=30, seed=4, tight=True),
        make_case("qws_like", T=47, k_range=(3, 9), pop=100, iters=60, seed=5),
        make_case("normal_like", T=50, k_range=(5, 14), pop=100, iters=40, seed=6, foreign_pick=True),
    ]
    out = []
    for case in cases:
        want = run_reference(WOA, case)
        got = owoa.eswoa([[tuple(s) for s in cat] for cat in case["services"]], case["constraints"],
                         copy.deepcopy(case["solution"]), case["pop_size"], case["max_iter"],
                         owoa.DrawStream(case["seed"]))
        assert got["draws"] == want["draws"], (case["name"], got["draws"], want["draws"])
        assert got["history"] == want["history"], case["name"]
        assert got["best_fitness"] == want["best_fitness"], case["name"]
        assert [list(r[:4]) for r in got["best_rows"]] == want["best_rows"], case["name"]
        assert [int(v) for v in got["best_pos"]] == want["best_pos"], case["name"]
        improved = want["history"][-1] < want["history"][0] if want["history"] else False
        print(f"{case['name']}: best {want['best_fitness']:.6f} after {len(want['history'])} iterations, "
              f"{want['draws']} draws, improved during the run: {improved}")
        case["expected"] = want
        out.append(case)
    with open(os.path.join(HERE, "woa_cases.json"), "w") as f:
        json.dump(out, f)
    print("wrote woa_cases.json", os.path.getsize(os.path.join(HERE, "woa_cases.json")), "bytes")


if __name__ == "__main__":
    main()
