"""Fuzz campaign for the ES-WOA kernels against the live oracle: random numbers of categories (1..64 on the lane-per-category
kernel — and, every other configuration, on the workgroup-per-problem kernel as well; 65..1500 on the latter), candidates
per category (1..20), population, iterations, seeded / unseeded / foreign picks, tight and loose constraints.
Usage: fuzz_woa.py [n_configs] [seed]"""
import sys, os, copy, random, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from gnnpn_sc_amd import WOA
from oracle import woa as owoa                      # checker only
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
fails, t0 = 0, time.time()
for c in range(n_cfg):
    T = rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 47, 50, 63, 64, 65, 100, 127, 128, 129, 136, 257, 264, 500, 1000, 1500])
    wide = True if T > 64 else (c % 2 == 1)          # numpy's pairwise sum changes shape at 8, 128 and at every halving
    pop, iters, P = rng.randint(1, 40 if T <= 64 else 12), rng.randint(0, 30 if T <= 64 else 8), rng.randint(1, 6 if T <= 64 else 3)
    g = np.random.default_rng(rng.randrange(1 << 30))
    problems, seeds = [], []
    for p in range(P):
        services = [[tuple(float(v) for v in np.r_[g.random(2), 0.9 + 0.1 * g.random(2)]) for _ in range(int(g.integers(1, 21)))]
                    for _ in range(T)]
        lo = 0.9 ** min(T, 64) * float(g.choice([0.5, 0.9, 1.2, 1.6]))
        hi = float(g.choice([1.0, 0.97 ** min(T, 64)]))
        sol = None if g.random() < 0.2 else [list(cat[int(g.integers(0, len(cat)))]) for cat in services]
        if sol is not None and g.random() < 0.3:
            sol[int(g.integers(0, T))] = [float(v) for v in np.r_[g.random(2), 0.9 + 0.1 * g.random(2)]]
        problems.append((services, [[[lo, hi]], [[lo, hi]]], sol))
        seeds.append(rng.randrange(1 << 62))
    try:
        got = WOA.fine_tune(problems, pop, iters, seeds, dev, wide=wide)
        for p, (services, cons, sol) in enumerate(problems):
            want = owoa.eswoa(services, cons, copy.deepcopy(sol), pop, iters, owoa.DrawStream(seeds[p]))
            assert got[p]["draws"] == want["draws"], "draws"
            assert got[p]["bestFitnesses"] == want["history"], "history"
            assert got[p]["bestFitness"] == want["best_fitness"], "fitness"
            if want["best_pos"] is not None:
                assert [int(v) for v in got[p]["bestPops"]] == [int(v) for v in want["best_pos"]], "positions"
    except Exception as e:          # noqa: BLE001
        fails += 1
        print(f"FAIL cfg {c}: T={T} wide={wide} pop={pop} iters={iters} P={P}: {type(e).__name__} {e}")
print(f"{n_cfg} configs, {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
