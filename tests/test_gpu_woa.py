"""-m gpu: the ES-WOA fine-tuner (gnnpn_eswoa_f64 through gnnpn_sc_amd.WOA) against the golden runs of the reference's
own ``ESWOA`` class (tests/golden/woa_cases.json, tests/golden/make_golden_woa.py) and against the oracle run live."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import woa as owoa

pytestmark = pytest.mark.gpu


def _cases():
    with open(os.path.join(GOLDEN, "woa_cases.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["tiny", "no_solution", "foreign_pick", "violated", "qws_like", "normal_like"])
def test_eswoa_golden(dev, name):
    """Bit-exact: same draw count, same best-fitness history (float64), same final positions and rows as the reference."""
    from gnnpn_sc_amd import WOA
    c = next(x for x in _cases() if x["name"] == name)
    m = WOA.ESWOA([[tuple(s) for s in cat] for cat in c["services"]], c["constraints"], copy.deepcopy(c["solution"]),
                  popSize=c["pop_size"], MAX_Iter=c["max_iter"], seed=c["seed"], device=dev)
    q, rows = m.start()
    want = c["expected"]
    assert q == want["best_fitness"]
    assert m.bestFitnesses == want["history"]
    assert [int(v) for v in m.bestPops] == want["best_pos"]
    assert [list(r) for r in rows] == want["best_rows"]


def test_eswoa_batch_vs_live_oracle(dev):
    """Many problems in one launch (different seeds, seeded and unseeded starts) against oracle/woa.py run live."""
    from gnnpn_sc_amd import WOA
    g = np.random.default_rng(3)
    T, problems, seeds = 13, [], []
    for p in range(24):
        services = [[tuple(float(v) for v in np.r_[g.random(2), 0.9 + 0.1 * g.random(2)]) for _ in range(int(g.integers(1, 8)))]
                    for _ in range(T)]
        lo = 0.9 ** T * float(g.choice([0.6, 1.2]))
        cons = [[[lo, 1.0]], [[lo, 1.0]]]
        sol = None if p % 5 == 4 else [list(cat[int(g.integers(0, len(cat)))]) for cat in services]
        if sol is not None and p % 3 == 0:
            sol[p % T] = [float(v) for v in np.r_[g.random(2), 0.9 + 0.1 * g.random(2)]]     # foreign pick
        problems.append((services, cons, sol))
        seeds.append(1000 + p)
    got = WOA.fine_tune(problems, popSize=30, MAX_Iter=40, seeds=seeds, device=dev)
    for p, (services, cons, sol) in enumerate(problems):
        want = owoa.eswoa(services, cons, copy.deepcopy(sol), 30, 40, owoa.DrawStream(seeds[p]))
        assert got[p]["draws"] == want["draws"], p
        assert got[p]["bestFitnesses"] == want["history"], p
        assert got[p]["bestFitness"] == want["best_fitness"], p
        assert [int(v) for v in got[p]["bestPops"]] == [int(v) for v in want["best_pos"]], p
        assert [tuple(r) for r in got[p]["bestSolutions"]] == [tuple(r[:4]) for r in want["best_rows"]], p


def _random_problems(g, T, n, max_cand=8):
    problems = []
    for p in range(n):
        services = [[tuple(float(v) for v in np.r_[g.random(2), 1.0 - g.random(2) * 0.2 / T]) for _ in range(int(g.integers(1, max_cand)))]
                    for _ in range(T)]
        lo = float(g.choice([0.5, 0.95]))                     # products of T factors in (1 - 0.2/T, 1]: within [0.8, 1]
        cons = [[[lo, 1.0]], [[lo, 1.0]]]
        sol = None if p % 4 == 3 else [list(cat[int(g.integers(0, len(cat)))]) for cat in services]
        if sol is not None and p % 3 == 0:
            sol[p % T] = [float(v) for v in np.r_[g.random(2), 1.0 - g.random(2) * 0.2 / T]]           # foreign pick
        problems.append((services, cons, sol))
    return problems


def _assert_run_equals_oracle(got, problems, seeds, pop, iters):
    for p, (services, cons, sol) in enumerate(problems):
        want = owoa.eswoa(services, cons, copy.deepcopy(sol), pop, iters, owoa.DrawStream(seeds[p]))
        assert got[p]["draws"] == want["draws"], p
        assert got[p]["bestFitnesses"] == want["history"], p
        assert got[p]["bestFitness"] == want["best_fitness"], p
        assert [int(v) for v in got[p]["bestPops"]] == [int(v) for v in want["best_pos"]], p
        assert [tuple(r) for r in got[p]["bestSolutions"]] == [tuple(r[:4]) for r in want["best_rows"]], p


@pytest.mark.parametrize("T,n,pop,iters", [(65, 6, 12, 10), (127, 4, 10, 8), (128, 3, 8, 6), (129, 3, 8, 6), (300, 4, 10, 8),
                                          (1000, 3, 8, 5), (2000, 2, 6, 4)])
def test_eswoa_wide_vs_live_oracle(dev, T, n, pop, iters):
    """More than 64 categories (the 1000- and 2000-task configurations): the workgroup-per-problem kernel against the oracle
    run live — same draws, same float64 history bit for bit (np.sum's pairwise recursion sets in above 128 terms: 129, 300,
    1000 and 2000 cut the column differently), same final composition."""
    from gnnpn_sc_amd import WOA
    g = np.random.default_rng(100 + T)
    problems = _random_problems(g, T, n)
    seeds = [5000 + 7 * p + T for p in range(n)]
    got = WOA.fine_tune(problems, popSize=pop, MAX_Iter=iters, seeds=seeds, device=dev)
    _assert_run_equals_oracle(got, problems, seeds, pop, iters)


def test_eswoa_wide_equals_the_lane_per_category_form(dev):
    """Where both kernels apply (T <= 64) they produce the same run, and the reference's golden runs hold for the wide one."""
    from gnnpn_sc_amd import WOA
    g = np.random.default_rng(9)
    for T in (1, 7, 8, 33, 64):
        problems = _random_problems(g, T, 5)
        seeds = [77 + p for p in range(5)]
        a = WOA.fine_tune(problems, popSize=20, MAX_Iter=25, seeds=seeds, device=dev)
        b = WOA.fine_tune(problems, popSize=20, MAX_Iter=25, seeds=seeds, device=dev, wide=True)
        assert a == b, T
    for c in _cases():
        got = WOA.fine_tune([([[tuple(s) for s in cat] for cat in c["services"]], c["constraints"], copy.deepcopy(c["solution"]))],
                            popSize=c["pop_size"], MAX_Iter=c["max_iter"], seeds=[c["seed"]], device=dev, wide=True)[0]
        assert got["bestFitness"] == c["expected"]["best_fitness"], c["name"]
        assert got["bestFitnesses"] == c["expected"]["history"], c["name"]
        assert [int(v) for v in got["bestPops"]] == c["expected"]["best_pos"], c["name"]


def test_eswoa_rejects_what_it_does_not_implement(dev):
    from gnnpn_sc_amd import WOA, ops
    services = [[(0.5, 0.5, 0.95, 0.95)] for _ in range(7000)]
    with pytest.raises(ops.GnnpnError):                       # three float64 columns of 7000 categories exceed a CU's LDS
        WOA.fine_tune([(services, [[[0.0, 1.0]], [[0.0, 1.0]]], None)], popSize=4, MAX_Iter=2, seeds=[1], device=dev)
    with pytest.raises(ops.GnnpnError):                       # two pairs for one product constraint
        WOA.fine_tune([(services[:3], [[[0.0, 1.0], [0.1, 1.0]], [[0.0, 1.0]]], None)], popSize=4, MAX_Iter=2, seeds=[1], device=dev)
    with pytest.raises(ops.GnnpnError):                       # host tensors: no CPU path
        ops.eswoa(torch.zeros(4, dtype=torch.int32), torch.zeros(3, dtype=torch.int32), torch.zeros(3, 4, dtype=torch.float64),
                  torch.zeros(1, 4, dtype=torch.float64), torch.zeros(3, dtype=torch.int32), 4, 2, torch.zeros(1, dtype=torch.int64), 3)


def test_woa_driver_end_to_end(dev, tmp_path, monkeypatch):
    """``WOA(...).start()`` in its ML2PNWOATest mode from the artefact files to ``ML+2PN+WOA.txt``: the 1000 qualities
    the reference itself produced (per-problem streams, both `reduct` settings), to the last bit."""
    from test_host_logic import _woa_driver_setup
    from gnnpn_sc_amd import WOA
    monkeypatch.chdir(tmp_path)
    fx, _actions, _n_train = _woa_driver_setup(str(tmp_path))
    p = fx["params"]
    for reduct in (0, 0.55):
        drv = WOA.WOA("QWS", p["T"], 0, 1, 0, 0, 4, reduct, -1, p["MAX_Iter"], p["popSize"], seed=p["base_seed"], device=dev)
        out = drv.start()
        with open("./solutions/WOA/QWS//ML+2PN+WOA.txt") as f:
            written = json.load(f)
        want = fx["modes"][str(reduct)]
        assert written["quality"] == out["quality"] == want["quality"]
        assert out["averageQ"] == want["averageQ"]
    with pytest.raises(NotImplementedError):
        WOA.WOA("QWS", p["T"], 1, 0, 0, 0, 4, 0, -1, 4, 4).start()


def test_main_cli_woa(dev, tmp_path, monkeypatch):
    """``python main.py QWS WOA --seed N`` (reference main.py:86-94) drives the same run as the class."""
    import contextlib
    import io
    from test_host_logic import _woa_driver_setup
    monkeypatch.chdir(tmp_path)
    fx, _a, _n = _woa_driver_setup(str(tmp_path))
    p = fx["params"]
    (tmp_path / "environment.ini").write_text(
        f"[QWS-WOA]\nserCategory = {p['T']}\nMLESWOAtest = 0\nML2PNWOATest = 1\nMLWOATest = 0\nESWOAtest = 0\n"
        f"serviceNumber = 4\nreduct = 0\nepoch = -1\nMAX_Iter = {p['MAX_Iter']}\npopSize = {p['popSize']}\n")
    import main as cli
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(["main.py", "QWS", "WOA", "--seed", str(p["base_seed"])]) == 0
    with open("./solutions/WOA/QWS//ML+2PN+WOA.txt") as f:
        assert json.load(f)["quality"] == fx["modes"]["0"]["quality"]
    assert len(buf.getvalue().strip().splitlines()) == 1000            # one progress line per problem, as the reference prints
