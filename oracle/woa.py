"""CPU oracle for the ES-WOA fine-tuner — TEST INFRASTRUCTURE ONLY.

Restates ``ESWOA`` of the reference (src/baselines/WOA.py:8-162): the whale-optimisation search that starts
from the pointer network's selection and tries to lower ``violate + objFunc``.  The reference draws from
numpy's global generator, so two runs of it never agree; here every draw comes from an explicit counter-based
stream (``DrawStream``) in the reference's own draw order.  tests/golden/make_golden.py runs the REAL
reference class with ``np.random.{random,randint,choice}`` routed to the same stream and stores inputs and
outputs as fixtures (tests/golden/woa_*.json); tests/test_oracle_golden.py holds this restatement to them.

Python-isms of the reference that are part of its behaviour and are restated explicitly:
  * aliasing — ``self.bestPops = self.pops[i]`` / ``self.bestSolutions = self.popServices[i]`` (WOA.py:82-83,
    115-116,159-160) bind the SAME list objects, so the global phase's in-place writes (``self.pops[i][rand] =
    randi``, :110-111) also change the recorded best; the local phase rebinds ``self.pops[i]`` to a new list
    (:155-156), which ends the sharing.  Tracked with ``alias_p`` / ``alias_s``;
  * negative positions — positions produced by the local phase lie in (-len, len) (:152-154) and index from
    the end (Python list indexing); they are stored as produced and used raw in the next distance computation;
  * ``round`` is round-half-even on a float (:146,149); ``%`` has the sign of the divisor.
Not restated: the reference's data-set specific patches of individual QoS tuples (WOA.py:27-41) — they only
fire on six literal tuples of the shipped data sets; ``patch_solution`` exists so a caller can apply them.

The figure of merit (``calc``, WOA.py:87-105) runs in float64 through numpy exactly as the reference does
(np.cumprod sequential, np.sum pairwise in blocks of 8 for n >= 8, np.min): a device kernel has to reproduce
those orders, see ``objective``.
"""
import math

import numpy as np

_M64 = (1 << 64) - 1
PE = 0.2                      # WOA.py:10


class DrawStream:
    """Counter-based uniform stream: draw k is a pure function of (seed, k) (splitmix64 finaliser)."""

    def __init__(self, seed):
        self.seed, self.count = int(seed) & _M64, 0

    def uniform(self):
        self.count += 1
        z = (self.seed + 0x9E3779B97F4A7C15 * self.count) & _M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        z ^= z >> 31
        return (z >> 11) * (1.0 / (1 << 53))

    def below(self, n):
        """Integer in [0, n): what np.random.randint(0, n) and np.random.choice(range(n)) are routed to."""
        return int(self.uniform() * n)


def objective(rows, constraints):
    """``calc`` (WOA.py:87-105): (violate, objFunc) of one composition; rows = [(q0,q1,q2,q3), ...]."""
    cols = [np.array([r[j] for r in rows]) for j in range(4)]
    prod = [np.cumprod(cols[2 + i])[-1] for i in range(2)]
    violate = 0
    for i in range(len(constraints)):
        for c in constraints[i]:
            if prod[i] < c[-2] or prod[i] > c[-1]:
                violate += 1
    n_real = sum(1 for r in rows if r[0] > 0)
    obj = float((np.sum(cols[0]) / n_real + 1 - np.min(cols[1])) / 2)
    return violate, obj


def _at(seq, i):                       # Python list indexing, negative positions from the end
    return seq[i if i >= 0 else i + len(seq)]


def patch_solution(solution, patches):
    """WOA.py:27-41 as data: ``patches`` = [(tuple4, column, new_value), ...]; applied in order."""
    for row in solution:
        for want, col, val in patches:
            if row == list(want):
                row[col] = val
    return solution


def eswoa(services, constraints, solution, pop_size, max_iter, stream):
    """One ES-WOA run.  services: per category a list of (q0,q1,q2,q3) tuples; constraints: [[[lo,hi]],[[lo,hi]]];
    solution: per category [q0,q1,q2,q3] (the pointer network's pick) or None.
    Returns dict(best_fitness, best_rows, best_pos, history (best fitness after every iteration), draws)."""
    services = [list(s) for s in services]
    T = len(services)
    if solution is not None:                                             # WOA.py:13-26: 5-decimal rounding
        services = [[tuple(round(v, 5) for v in svc[:4]) + tuple(svc[4:]) for svc in cat] for cat in services]
        solution = [[round(v, 5) for v in row[:4]] + list(row[4:]) for row in solution]

    pos = [[stream.below(len(services[j])) for j in range(T)] for _ in range(pop_size)]      # :51-52
    if solution is not None:                                             # :55-69
        v, o = objective(solution, constraints)
        best_fit, best_rows, best_pos = v + o, [tuple(r) for r in solution], []
        for j in range(T):
            key = tuple(solution[j])
            if key not in services[j]:
                services[j].append(key)                                  # :65-66 (the list grows AFTER the draws)
            best_pos.append(services[j].index(key))
    else:
        best_fit, best_rows, best_pos = 3, None, None                    # :71-74
    alias_p = alias_s = -1            # index of the individual whose lists the best record shares, or -1

    rows = []
    for i in range(pop_size):                                            # :77-85
        rows.append([services[j][pos[i][j]] for j in range(T)])
        v, o = objective(rows[i], constraints)
        if best_fit > v + o:
            best_fit, best_rows, best_pos = v + o, list(rows[i]), list(pos[i])
            alias_p = alias_s = i

    history, t = [], 0
    while t < max_iter:                                                  # :107-161
        prob = 0.2 * (1 - t / max_iter)
        for i in range(pop_size):                                        # global phase :111-123
            if stream.uniform() < prob:
                j = stream.below(T)
                k = stream.below(len(services[j]))
                pos[i][j] = k
                rows[i][j] = services[j][k]
                if alias_p == i:
                    best_pos[j] = k                                      # same list object in the reference
                if alias_s == i:
                    best_rows[j] = services[j][k]
                v, o = objective(rows[i], constraints)
                if best_fit > v + o:
                    best_fit, best_rows, best_pos = v + o, list(rows[i]), list(pos[i])
                    alias_p = alias_s = i
        if PE > stream.uniform():                                        # :125-129
            t += 1
            history.append(best_fit)
            continue
        for i in range(pop_size):                                        # local phase :131-160
            a = 2 - (2 * t / max_iter)
            r = stream.uniform()
            A = 2 * a * r - a
            C = 2 * r
            l = stream.uniform()
            p = stream.uniform()
            new = None
            if p < 0.5:
                if abs(A) < 1:
                    new = [round(b - A * (C * b - x)) for b, x in zip(best_pos, pos[i])]
            else:
                e, c = math.exp(l), math.cos(2 * math.pi * l)
                new = [round((x - b) * e * c + b) for b, x in zip(best_pos, pos[i])]
            if new is not None:
                for j in range(T):
                    if abs(new[j]) >= len(services[j]):
                        new[j] %= len(services[j])
                pos[i] = new                                             # rebinding: the best record keeps the old list
                if alias_p == i:
                    alias_p = -1
                rows[i] = [_at(services[j], new[j]) for j in range(T)]
                if alias_s == i:
                    alias_s = -1
                v, o = objective(rows[i], constraints)
                if best_fit > v + o:
                    best_fit, best_rows, best_pos = v + o, list(rows[i]), list(pos[i])
                    alias_p = alias_s = i
        t += 1
        history.append(best_fit)
    return {"best_fitness": best_fit, "best_rows": best_rows, "best_pos": best_pos, "history": history,
            "draws": stream.count}
