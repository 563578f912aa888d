"""ES-WOA fine-tuning of pointer-network solutions on the GPU.

Mirror of ``ESWOA`` of the reference (src/baselines/WOA.py:8-162): same constructor arguments, ``start()`` returns
``(bestFitness, bestSolutions)`` and fills ``bestFitnesses``.  The search itself runs in ``gnnpn_eswoa_f64`` (one
wavefront per problem); ``fine_tune`` runs many problems in one launch, which is how the reference's per-problem loop
(WOA.py:271-288) should be driven here.

Differences, all deliberate:
  * randomness — the reference draws from numpy's global generator (no seed, two runs never agree); here every problem
    has a ``seed`` and draw k is a pure function of (seed, k) (oracle/woa.py ``DrawStream``).  ``seed=None`` takes a
    fresh one from the OS, which is the reference's behaviour;
  * the six data-set specific tuple patches of WOA.py:27-41 are not applied (pass ``patches=`` to have them);
  * one [lo, hi] pair per product constraint, as WOA.start builds them (loadData.py:268-273); more raise.
There is no CPU path: without the HIP library or a GPU the calls raise ``GnnpnError``.
"""
import os

import numpy as np
import torch

from . import ops

GnnpnError = ops.GnnpnError


def _prepare(services, constraints, solution, patches=()):
    """The reference's host-side preparation for one problem: returns (lists per category of rounded tuples incl. an
    appended foreign pick, lengths before the append, start positions or None, solution rows or None, bounds)."""
    for c in constraints:
        if len(c) != 1:
            raise GnnpnError("ESWOA: exactly one [lo, hi] pair per product constraint is supported")
    bounds = [float(constraints[0][0][-2]), float(constraints[0][0][-1]), float(constraints[1][0][-2]), float(constraints[1][0][-1])]
    cats = [[tuple(float(v) for v in s[:4]) for s in cat] for cat in services]
    if any(len(c) == 0 for c in cats):
        raise GnnpnError("ESWOA: a category without candidates")
    len0 = [len(c) for c in cats]
    start, rows = None, None
    if solution is not None:                                   # WOA.py:13-26, 55-69
        if len(solution) != len(cats):
            raise GnnpnError(f"ESWOA: solution has {len(solution)} rows for {len(cats)} categories")
        cats = [[tuple(round(v, 5) for v in s) for s in cat] for cat in cats]
        rows = [[round(float(v), 5) for v in r[:4]] for r in solution]
        for want, col, val in patches:                         # :27-41, as data
            for r in rows:
                if r == list(want):
                    r[col] = val
        start = []
        for j, r in enumerate(rows):
            key = tuple(r)
            if key not in cats[j]:
                cats[j].append(key)
            start.append(cats[j].index(key))
    return cats, len0, start, rows, bounds


def fine_tune(problems, popSize=100, MAX_Iter=500, seeds=None, device=None, patches=(), wide=None):
    """ES-WOA over many problems in ONE launch.  problems: iterable of (services, constraints, solution | None) with the
    same number of categories.  Returns a list of dicts: bestFitness, bestSolutions (rows), bestPops, bestFitnesses, draws."""
    problems = list(problems)
    if not problems:
        return []
    if seeds is None:
        seeds = np.frombuffer(os.urandom(8 * len(problems)), dtype=np.uint64)
    seeds = np.asarray(seeds, dtype=np.uint64).reshape(len(problems))
    sizes = [len(s) for s, _c, _sol in problems]
    if len(set(sizes)) > 1:                                  # one launch per distinct number of categories
        out = [None] * len(problems)
        for T in sorted(set(sizes)):
            idx = [i for i, n in enumerate(sizes) if n == T]
            for i, r in zip(idx, fine_tune([problems[i] for i in idx], popSize, MAX_Iter, seeds[idx], device, patches, wide)):
                out[i] = r
        return out
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    prep = [_prepare(s, c, sol, patches) for s, c, sol in problems]
    T = len(prep[0][0])
    P = len(prep)
    ptr, flat, len0, start, bounds = [0], [], [], [], []
    for cats, l0, st, _rows, b in prep:
        for j, cat in enumerate(cats):
            flat.extend(cat)
            ptr.append(ptr[-1] + len(cat))
        len0.extend(l0)
        start.extend(st if st is not None else [-1] * T)
        bounds.append(b)
    t = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt).to(dev)      # noqa: E731
    fit, pos, hist, draws = ops.eswoa(t(ptr, torch.int32), t(len0, torch.int32), t(flat, torch.float64).reshape(-1, 4),
                                      t(bounds, torch.float64), t(start, torch.int32), popSize, MAX_Iter,
                                      torch.from_numpy(seeds.view(np.int64).copy()).to(dev), T, wide)
    fit, pos, hist, draws = fit.cpu().tolist(), pos.cpu().tolist(), hist.cpu().tolist(), draws.cpu().tolist()
    out = []
    for p, (cats, _l0, st, _rows, _b) in enumerate(prep):
        has_best = st is not None or fit[p] < 3
        rows = [cats[j][pos[p][j]] for j in range(T)] if has_best else None         # negative positions: Python indexing
        out.append({"bestFitness": fit[p], "bestSolutions": rows, "bestPops": pos[p] if has_best else None,
                    "bestFitnesses": hist[p], "draws": draws[p]})
    return out


class ESWOA:
    """``ESWOA(services, constraints, solution=None, popSize=100, MAX_Iter=500)`` of WOA.py:8-85, plus ``seed``."""

    def __init__(self, services, constraints, solution=None, popSize=100, MAX_Iter=500, seed=None, device=None, patches=()):
        self.pe = 0.2
        self.services, self.constraints, self.solution = services, constraints, solution
        self.popSize, self.MAX_Iter, self.qosNum, self.consNum = popSize, MAX_Iter, 4, 2
        self.seed, self.device, self.patches = seed, device, patches
        self.bestFitnesses, self.bestFitness, self.bestSolutions, self.bestPops = [], None, None, None

    def start(self):                                            # WOA.py:107-162
        r = fine_tune([(self.services, self.constraints, self.solution)], self.popSize, self.MAX_Iter,
                      None if self.seed is None else [self.seed], self.device, self.patches)[0]
        self.bestFitness, self.bestSolutions, self.bestPops = r["bestFitness"], r["bestSolutions"], r["bestPops"]
        self.bestFitnesses = r["bestFitnesses"]
        return self.bestFitness, self.bestSolutions


class WOA:
    """The driver of WOA.py:165-310 for the mode the ML+2PN pipeline feeds (``ML2PNWOATest``, environment.ini
    ``[<ds>-WOA]``): pointer-network actions -> per-problem seed solutions -> ``loadDataOther`` candidate lists -> ES-WOA
    on every test problem (here: all of them in one launch per workflow size) -> ``./solutions/WOA/<ds>/ML+2PN+WOA.txt``.
    The other three modes (ES-WOA from scratch, ML+ES-WOA, ML+WOA) are baselines outside the path: NotImplementedError.
    ``seed``: problem number ``idx`` runs on stream ``seed + idx`` (None: fresh seeds, the reference's behaviour).
    The reference sizes its solution table for exactly 1000 test problems (:194); any number works here."""

    def __init__(self, dataset, serCategory, MLESWOAtest, ML2PNWOATest, MLWOATest, ESWOAtest, serviceNumber, reduct, epoch,
                 MAX_Iter, popSize, seed=None, device=None):
        self.dataset, self.serCategory = dataset + "/", serCategory
        self.MLESWOAtest, self.ML2PNWOATest, self.MLWOATest, self.ESWOAtest = MLESWOAtest, ML2PNWOATest, MLWOATest, ESWOAtest
        self.serviceNumber, self.reduct, self.epoch, self.MAX_Iter, self.popSize = serviceNumber, reduct, epoch, MAX_Iter, popSize
        self.seed, self.device = seed, device
        self.times, self.qosNum, self.train, self.sSetList = 0, 4, False, None

    def start(self):
        import json
        import time

        from .loadData import loadDataOther
        if not self.ML2PNWOATest or self.MLESWOAtest or self.MLWOATest or self.ESWOAtest:
            raise NotImplementedError("WOA: only the ML2PNWOATest mode (fine-tuning the ML+2PN solution) is on the path")
        ds = self.dataset[:-1]
        src = (f"./solutions/PNHigh/{self.dataset}/allActions{self.epoch}.txt" if self.epoch >= 0
               else f"./solutions/pretrained/{ds}-PNHigh.txt")                                   # :187-192
        with open(src) as f:
            allActions = json.load(f)
        n_test = len(allActions[0]) if allActions else 0
        solutions, self.sSetList = [], []
        for b in range(n_test):                                                                # :194-208
            rows = [allActions[c][b][: self.qosNum] for c in range(len(allActions))]
            rows = [r for r in rows if sum(r) != 3]                                            # dummy action of an absent category
            solutions.append(rows)
            self.sSetList.append({tuple(round(v, 5) for v in r) for r in rows})
        features, constraintsList, minCostList = loadDataOther(ds, self.reduct, sSetList=self.sSetList, train=self.train)
        first = len(minCostList) // 4 * 3
        n = min(len(features), len(solutions))
        problems = [(features[i], constraintsList[i], solutions[i] if solutions[i] else None) for i in range(n)]
        seeds = None if self.seed is None else [(self.seed + first + i) & 0xFFFFFFFFFFFFFFFF for i in range(n)]
        t0 = time.time()
        results = fine_tune(problems, self.popSize, self.MAX_Iter, seeds, self.device)
        per_problem = (time.time() - t0) / max(n, 1)
        out = {"quality": [], "time": [], "averageQ": 0, "averageT": 0}
        self.bestFitnesses = [[r["bestFitnesses"][i] for r in results] for i in range(self.MAX_Iter)]   # :264, 283-284
        for i, r in enumerate(results):                                                        # :286-292
            out["quality"].append(minCostList[first + i] / r["bestFitness"])
            out["time"].append(per_problem)
            out["averageQ"] = sum(out["quality"]) / (self.times + 1)
            out["averageT"] = sum(out["time"]) / (self.times + 1)
            print(first + i, out["averageQ"], out["averageT"])
            self.times += 1
        os.makedirs(f"./solutions/WOA/{self.dataset}", exist_ok=True)
        with open(f"./solutions/WOA/{self.dataset}/ML+2PN+WOA.txt", "w") as f:                 # :294-296
            json.dump(out, f)
        self.results = results
        return out
