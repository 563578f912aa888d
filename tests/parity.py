"""Shared parity rules for the GPU tests.

Index parity: the path is chaotic in the picks (one flipped argmax changes every later decoder
input of that problem), and the HIP kernels sum in a different order than MKL (|Δlogit| ~ 1e-5 on
10*tanh(dot)).  So a problem whose every decision has a top-1/top-2 margin above TAU in the oracle
must match the oracle index for index; problems with a thinner margin are *fragile* (a flip there is
classified, not hidden): they are counted and reported, and the overall agreement must stay above
the stated floor.
"""
import torch

TAU = 5e-4          # decision margin above which indices must be bit-exact
LOGIT_ATOL = 2e-4   # |C*tanh(dot)| differences (C = 10)
# QoS score tolerance of BASELINE.json's north_star: 1e-5.  Scores are rounded to 5 decimals (modelPN.py:61), i.e. they
# are integers in units of 1e-5 stored as fp32; two results whose unrounded values straddle a rounding boundary differ
# by exactly one unit.  So the comparison is made on the integers: |round(R*1e5) - round(R_ref*1e5)| <= 1
# (assert_R_parity).  R_ATOL is the same bound read as an fp32 difference (1e-5 +- ulp(3)) for the places that compare
# R of two HIP runs with each other.
R_ATOL = 1e-5 + 2 * 2.4e-7
R_UNITS = 1         # allowed difference in 5-decimal units


def R_units(R):
    """R as integers in units of 1e-5 (NaN -> a sentinel, so NaN == NaN)."""
    import numpy as np
    r = np.asarray(torch.as_tensor(R).detach().cpu().double())
    out = np.rint(r * 1e5)
    out[np.isnan(r)] = -1e18
    return out


def assert_R_parity(got, want, what="R", mask=None):
    """|round(R*1e5) - round(R_ref*1e5)| <= 1 on the (masked) problems; returns the largest difference in units."""
    import numpy as np
    d = np.abs(R_units(got) - R_units(want))
    if mask is not None:
        d = d[np.asarray(torch.as_tensor(mask).cpu()).astype(bool)]
    worst = float(d.max()) if d.size else 0.0
    assert worst <= R_UNITS, f"{what}: differs from the reference by {worst:g} units of 1e-5"
    return int(worst)


def robust_problems(*margins):
    """[B] bool: every decision of the problem has margin > TAU (inf margins for K == 1 count)."""
    ok = None
    for m in margins:
        r = (torch.as_tensor(m) > TAU).all(dim=1)
        ok = r if ok is None else ok & r
    return ok


def assert_index_parity(got, want, robust, what, min_agree=0.9, rows=None):
    """got / want: picked positions [B,T].  With ``rows`` (the PN input [B,L,F]) two picks count as
    the same selection when their input rows are identical (duplicate candidates: dummy rows /
    cyclic padding) — see oracle.pn.decision_margin."""
    got, want = torch.as_tensor(got).long().cpu(), torch.as_tensor(want).long().cpu()
    if rows is None:
        same = (got == want).all(dim=1)
    else:
        rows = torch.as_tensor(rows).cpu()
        F = rows.shape[-1]
        a = torch.gather(rows, 1, got.unsqueeze(-1).expand(-1, -1, F))
        b = torch.gather(rows, 1, want.unsqueeze(-1).expand(-1, -1, F))
        same = (a == b).all(dim=-1).all(dim=1)
    bad = robust & ~same
    assert not bool(bad.any()), f"{what}: robust problems {bad.nonzero().flatten().tolist()} differ from the oracle"
    agree = float(same.float().mean())
    assert agree >= min_agree, f"{what}: only {agree:.3f} of problems index-exact"
    return same


TAU_DRAW = 1e-5    # sampling mode: distance of the uniform draw to the nearest cdf boundary below which a draw is fragile


def prefix_parity(got_low, got_high, fx, what, rows=None, tau_high=None):
    """Step-by-step parity for ONE batch against a reference fixture / oracle result holding idx_low, idx_high,
    margin_low, margin_high [B,T]: a problem is followed while its history equals the reference's; at its FIRST
    differing decision the reference's margin there must be <= TAU (a fragile decision: the flip is classified, and
    everything after it has a different history, so it is not comparable).  A differing decision with a margin above
    TAU fails.  With ``rows`` two picks with identical input rows count as the same decision.  Returns the measured
    agreement as a dict (what the committed agreement record holds)."""
    import numpy as np
    gl, gh = np.asarray(torch.as_tensor(got_low).cpu()).astype(np.int64), np.asarray(torch.as_tensor(got_high).cpu()).astype(np.int64)
    rl, rh = np.asarray(fx["idx_low"]).astype(np.int64), np.asarray(fx["idx_high"]).astype(np.int64)
    ml, mh = np.asarray(fx["margin_low"]), np.asarray(fx["margin_high"])
    if rows is not None:
        r = np.asarray(torch.as_tensor(rows).cpu())
        take = lambda idx: np.take_along_axis(r, idx[:, :, None], 1)   # noqa: E731
        dl, dh = (take(gl) != take(rl)).any(-1), (take(gh) != take(rh)).any(-1)
    else:
        dl, dh = gl != rl, gh != rh
    B, T = gl.shape
    diff = dl | dh
    first = np.where(diff.any(1), diff.argmax(1), T)
    flips, bad = [], []
    for b in np.nonzero(first < T)[0]:
        t = int(first[b])
        for lvl, d, m in (("low", dl, ml), ("high", dh, mh)):
            if d[b, t]:
                flips.append({"problem": int(b), "step": t, "level": lvl, "margin": float(m[b, t])})
                if not m[b, t] <= (TAU if lvl == "low" or tau_high is None else tau_high):
                    bad.append(flips[-1])
    assert not bad, f"{what}: decisions with a margin above {TAU} differ from the reference: {bad[:5]}"
    th = TAU if tau_high is None else tau_high
    robust = (ml > TAU).all(1) & (mh > th).all(1)
    frag = (ml <= TAU) | (mh <= th)
    robust_prefix = np.where(frag.any(1), frag.argmax(1), T)
    return {"problems": int(B), "steps": int(T), "identical_problems": int((first == T).sum()),
            "robust_problems": int(robust.sum()), "robust_identical": int(((first == T) & robust).sum()),
            "decisions_compared": int(2 * np.minimum(first + 1, T).sum()),
            "robust_prefix_decisions": int(2 * robust_prefix.sum()),
            "identical_decisions": int(2 * first.sum()),
            "flips": len(flips), "flip_margins": [round(f["margin"], 7) for f in flips][:64],
            "same_mask": (first == T)}


def oracle_candidate_rows(pb, table, rankings, n_per, nodes_per):
    """The candidate rows of the first len(rankings) problems of a synthetic batch by the ORACLE's reduction
    (oracle.data.reduce_candidates, pinned to the real loadDataPN — src/loadData.py:101-149 — by tests/golden/make_golden.py),
    not by the product's host helper gnnpn_sc_amd.loadData.reduce_from_ranking: the synthetic problem and table are first put into
    the reference's own formats (nodes = one-hot type + 6 floats, loadData.py:26-33; serviceFeature = {"1": [[q0..q3], ...], ...},
    :35-40).  -> float32 [n, T * n_per, 8] (column 0, the category, dropped as SCDataset does, trainPNHigh.py:23-31)."""
    import numpy as np
    from oracle import data as odata
    T = table.n_cat
    feature = {str(c + 1): [[float(v) for v in table.qos[s]] for s in range(int(table.cat_ptr[c]), int(table.cat_ptr[c + 1]))]
               for c in range(T)}
    out = []
    for i, ranking in enumerate(rankings):
        nodes = []
        for row in pb.x[i * nodes_per:(i + 1) * nodes_per]:
            onehot = [0] * (T + 1)
            onehot[int(row[0])] = 1
            nodes.append(onehot + [float(v) for v in row[1:]])
        rows, _ = odata.reduce_candidates([int(s) for s in ranking], nodes, feature, n_per)
        out.append(rows)
    return torch.tensor(np.asarray(out, dtype=np.float64), dtype=torch.float32)[:, :, 1:]
