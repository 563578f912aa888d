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
    from src import loadData as loadData_mod
    gen_driver(WOA, loadData_mod)


def gen_driver(WOA, loadData_mod):
    """WOA.start in its ML2PNWOATest mode (WOA.py:185-296) + loadDataOther / addS (loadData.py:155-276), both `reduct`
    settings of environment.ini, on a synthetic data set with the 1000 test problems the reference's table needs."""
    import hashlib
    import io
    import tempfile
    import gnnpn_sc_amd.synth as synth
    import gnnpn_sc_amd.loadData as mine
    T, S, P, n_t, K = 5, 30, 4000, 3, 4
    params = {"T": T, "S": S, "P": P, "seed": 17, "tasks_per_problem": n_t, "lo_range": [0.86, 0.95],
              "popSize": 6, "MAX_Iter": 8, "base_seed": 99}
    ds = synth.make_dataset(T, S, P, seed=params["seed"], tasks_per_problem=n_t, lo_range=tuple(params["lo_range"]))
    n_train = P // 4 * 3
    fx = {"params": params, "modes": {}}
    old = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            synth.write_dataset(tmp, "QWS", ds)
            os.makedirs("solutions/pretrained")
            os.makedirs("solutions/WOA/QWS")
            with contextlib.redirect_stdout(io.StringIO()):
                lists0, cons0, _ = loadData_mod.loadDataOther("QWS", False)
            g = np.random.default_rng(23)
            picks, foreign, allActions = [], {}, [[[0.0, 1.0, 1.0, 1.0, 0, 0, 0, 0] for _ in range(P - n_train)] for _ in range(T)]
            for b, nodes in enumerate(ds["nodefeatures"][n_train:]):
                cats = [n[:-6].index(1) - 1 for n in nodes][1:]
                assert cats == sorted(cats) and len(lists0[b]) == len(cats), "data set must keep node order = category order"
                row = []
                for l, c in enumerate(cats):
                    k = int(g.integers(0, len(lists0[b][l])))
                    q = list(lists0[b][l][k])
                    if g.random() < 0.05:                      # a pick that is not among the candidates
                        q = [float(v) for v in np.r_[g.random(2), 0.9 + 0.1 * g.random(2)]]
                        foreign[f"{b},{l}"] = q
                        k = -1
                    row.append(k)
                    allActions[c][b] = q + [0, 0, 0, 0]
                picks.append(row)
            with open("solutions/pretrained/QWS-PNHigh.txt", "w") as f:
                json.dump(allActions, f)
            fx["picks"], fx["foreign"] = picks, foreign

            class Seeded(WOA.ESWOA):                           # problem number idx runs on stream base_seed + idx
                nxt = [n_train]

                def __init__(self, *a, **k):
                    self._stream = owoa.DrawStream(params["base_seed"] + Seeded.nxt[0])
                    Seeded.nxt[0] += 1
                    with routed_numpy_random(self._stream):
                        super().__init__(*a, **k)

                def start(self):
                    with routed_numpy_random(self._stream):
                        return super().start()
            real = WOA.ESWOA
            WOA.ESWOA = Seeded
            try:
                for reduct in (0, 0.55):
                    Seeded.nxt[0] = n_train
                    drv = WOA.WOA("QWS", T, 0, 1, 0, 0, K, reduct, -1, params["MAX_Iter"], params["popSize"])
                    with contextlib.redirect_stdout(io.StringIO()):
                        drv.start()
                        lists, cons, mins = loadData_mod.loadDataOther("QWS", reduct, sSetList=drv.sSetList, train=False)
                    with open("./solutions/WOA/QWS//ML+2PN+WOA.txt") as f:
                        res = json.load(f)
                    # the mirror's host functions against the reference's
                    m_lists, m_cons, m_mins = mine.loadDataOther("QWS", reduct, sSetList=drv.sSetList, train=False)
                    assert json.dumps(m_lists) == json.dumps(lists) and m_cons == cons and m_mins == mins, reduct
                    # the oracle on every problem with the per-problem streams: the reference's qualities
                    sols = [[allActions[c][b][:4] for c in range(T) if sum(allActions[c][b][:4]) != 3] for b in range(P - n_train)]
                    for b in range(P - n_train):
                        r = owoa.eswoa(lists[b], cons[b], copy.deepcopy(sols[b]), params["popSize"], params["MAX_Iter"],
                                       owoa.DrawStream(params["base_seed"] + n_train + b))
                        assert mins[n_train + b] / r["best_fitness"] == res["quality"][b], (reduct, b)
                    digest = hashlib.sha256(json.dumps([lists, cons]).encode()).hexdigest()
                    fx["modes"][str(reduct)] = {"quality": res["quality"], "averageQ": res["averageQ"], "lists_sha256": digest,
                                                "lists_head": lists[:3], "constraints_head": cons[:3],
                                                "kept_per_problem": [sum(len(l) for l in p) for p in lists]}
                    print(f"driver reduct={reduct}: averageQ {res['averageQ']:.6f}, "
                          f"candidates kept per problem {np.mean(fx['modes'][str(reduct)]['kept_per_problem']):.1f}")
            finally:
                WOA.ESWOA = real
        finally:
            os.chdir(old)
    with open(os.path.join(HERE, "woa_driver.json"), "w") as f:
        json.dump(fx, f)
    print("wrote woa_driver.json", os.path.getsize(os.path.join(HERE, "woa_driver.json")), "bytes")
    sys.path.insert(0, HERE)
    import manifest
    manifest.update()


if __name__ == "__main__":
    main()
