"""ES-WOA fine-tuner: GPU kernel over N QWS-shaped problems at the reference's parameters (popSize 100, MAX_Iter 500)
vs the CPU oracle on a few of them.  Usage: bench_woa.py [n_problems] [max_iter]"""
import sys, os, time, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnnpn_sc_amd import WOA, ops
from oracle import woa as owoa                       # CPU baseline / checker only
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda:0")
g = np.random.default_rng(0)
T, problems = 47, []
for p in range(N):
    services = [[tuple(float(v) for v in np.r_[g.random(2), 0.9 + 0.1 * g.random(2)]) for _ in range(int(g.integers(3, 10)))]
                for _ in range(T)]
    lo = 0.9 ** T * 0.8
    sol = [list(cat[int(g.integers(0, len(cat)))]) for cat in services]
    problems.append((services, [[[lo, 1.0]], [[lo, 1.0]]], sol))
seeds = list(range(1, N + 1))
WOA.fine_tune(problems[:8], 100, 10, seeds[:8], dev)          # warm-up (module load)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = WOA.fine_tune(problems, 100, iters, seeds, dev)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# kernel alone: repeat the launch on prepared device arrays
prep = [WOA._prepare(s, c, sol) for s, c, sol in problems]
ptr, flat, len0, start, bounds = [0], [], [], [], []
for cats, l0, st, _r, b in prep:
    for cat in cats:
        flat.extend(cat); ptr.append(ptr[-1] + len(cat))
    len0.extend(l0); start.extend(st); bounds.append(b)
tt = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt).to(dev)
args = (tt(ptr, torch.int32), tt(len0, torch.int32), tt(flat, torch.float64).reshape(-1, 4), tt(bounds, torch.float64),
        tt(start, torch.int32), 100, iters, torch.arange(1, N + 1, dtype=torch.int64, device=dev), T)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); r = ops.eswoa(*args); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
evals = float(r[3].sum().item())
print(f"GPU: {N} problems x popSize 100 x {iters} iterations: kernel {ms:.1f} ms = {N / ms * 1e3:.0f} problems/s "
      f"({evals / ms / 1e3:.1f} M draws/s); with host preparation and copies {t_all:.2f} s")
n_cpu = 2
t0 = time.perf_counter()
for p in range(n_cpu):
    w = owoa.eswoa(problems[p][0], problems[p][1], copy.deepcopy(problems[p][2]), 100, iters, owoa.DrawStream(seeds[p]))
    assert w["history"] == out[p]["bestFitnesses"] and w["best_fitness"] == out[p]["bestFitness"], p
t_cpu = (time.perf_counter() - t0) / n_cpu
print(f"CPU oracle (1 thread, numpy/Python as the reference): {t_cpu:.2f} s per problem = {1 / t_cpu:.2f} problems/s; "
      f"results identical on the {n_cpu} problems checked")
imp = np.mean([o["bestFitnesses"][0] - o["bestFitness"] for o in out])
print(f"mean improvement of the figure of merit over the run: {imp:.4f}")
