"""Oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py): the host-side data functions on the
path, restated in plain Python/numpy from /root/reference/src/loadData.py and
/root/reference/src/ML2PN.py.  Pinned against the imported reference modules by
tests/golden/make_golden.py (fixtures tests/golden/data_*.json).

All functions take already-parsed JSON (python lists / dicts) so that tests need no files; the
reference's file names are listed in DESIGN.md section 1 and INTEGRATION.md.
"""
import numpy as np

QOS_NUM = 4


def node_rows(nodefeatures):
    """loadData.py:26-33: node = one-hot type followed by 6 floats -> [type_index, 6 floats]."""
    out = []
    for nodes in nodefeatures:
        out.append([[node[:-6].index(1)] + list(node[-6:]) for node in nodes])
    return out


def service_rows(service_feature):
    """loadData.py:35-40: dict '1'..'T' -> flat [cat, q0,q1,q2,q3] rows in numeric key order."""
    keys = sorted(int(k) for k in service_feature)
    rows = []
    for k in keys:
        for feat in service_feature[str(k)]:
            rows.append([k - keys[0]] + list(feat[-4:]))
    return rows


def service_graph(labels, n_train=3000):
    """loadData.py:42-65: co-occurrence graph over services from the first ``n_train`` label
    vectors.  Undirected pair (i<j) -> directed edges i->j, j->i in that order, scanned i-major;
    weight of the edge whose SOURCE is u is adj[u][v] / useTimes[u]."""
    lab = np.asarray(labels[:n_train], dtype=np.int64)
    lab = (lab == 1).astype(np.int64)
    use = lab.sum(0)
    adj = lab.T @ lab
    iu, ju = np.nonzero(np.triu(adj, 1))
    src, dst, w = [], [], []
    for i, j in zip(iu.tolist(), ju.tolist()):
        src += [i, j]
        dst += [j, i]
        w += [adj[i, j] / use[i], adj[j, i] / use[j]]
    return [src, dst], w


def problem_constraints(nodes, n_cat):
    """loadData.py:102-114.  Returns (cons [n_cat+1][8] indexed by 1-based category, present set).
    cons[c][0:4] = local (cost_lo, cost_hi, quality_lo, quality_hi) of category c;
    cons[c][4:8] = the request node's global (c0_lo, c0_hi, c1_lo, c1_hi)."""
    cons = {c: [0] * 8 for c in range(1, n_cat + 1)}
    present = set()
    for node in nodes:
        bounds = list(node[-5:-3]) + list(node[-2:])
        if node[0] == 1:
            for c in cons:
                cons[c][4:8] = bounds
        else:
            c = node[:-6].index(1)
            cons[c][0:4] = bounds
            present.add(c)
    return cons, present


def reduce_candidates(ranking, nodes, service_feature, n_per):
    """loadDataPN's per-problem body (loadData.py:101-149) with the candidate order DEFINED as rank
    order (the reference shuffles each category's set, loadData.py:135, after a Python-set round
    trip; its order is therefore arbitrary — DESIGN.md §divergences).  Returns [T*n_per][9] rows
    ``[cat, q0,q1,q2,q3, c0lo,c0hi,c1lo,c1hi]`` (constraint columns non-zero for category 0 only)
    and the chosen global service ids [T][<=n_per] (before padding)."""
    keys = list(service_feature.keys())
    n_cat = len(keys)
    cat_of, pos_of = [], []
    for key in keys:                                                    # :94-97
        cat_of += [int(key) - 1] * len(service_feature[key])
        pos_of += list(range(len(service_feature[key])))
    cons, present = problem_constraints(nodes, n_cat)

    picked = [[] for _ in range(n_cat)]
    for s in ranking:                                                   # :117-125
        c = cat_of[s]
        if len(picked[c]) < n_per and s not in picked[c]:
            feat = service_feature[str(c + 1)][pos_of[s]]
            cost, quality = feat[-2], feat[-1]
            lim = cons[c + 1]
            if lim[0] <= cost <= lim[1] and lim[2] <= quality <= lim[3]:
                picked[c].append(s)

    rows = []
    for c in range(n_cat):                                              # :128-148
        tail = cons[1][4:8] if c == 0 else [0, 0, 0, 0]
        if (c + 1) in present and picked[c]:
            cyc = [picked[c][i % len(picked[c])] for i in range(n_per)]  # :137-138 doubling + [:K]
            for s in cyc:
                feat = service_feature[str(c + 1)][pos_of[s]]
                rows.append([c] + [feat[k] for k in (-4, -3, -2, -1)] + tail)
        else:
            # absent category (:148).  A *present* category with no feasible service makes the
            # reference spin forever (:137); the build emits dummy rows for it instead.
            rows += [[c, 0, 1, 1, 1] + tail for _ in range(n_per)]
    return rows, picked


def load_data_pn(nodefeatures, service_feature, rankings, min_cost, n_per):
    """loadDataPN (loadData.py:72-152) over all problems: (rows per problem, labels=minCost)."""
    feats = [reduce_candidates(r, n, service_feature, n_per)[0]
             for n, r in zip(nodefeatures, rankings)]
    return feats, list(min_cost[:len(feats)])


def pn_inputs(rows):
    """SCDataset with embeddingTag=0 (trainPNHigh.py:23-31): drop column 0 -> float32 [L,8]."""
    return np.asarray(rows, dtype=np.float32)[:, 1:]


def score_calc(qos, cons):
    """ML2PN.calc (ML2PN.py:6-12), float64."""
    obj = 0.5 * (np.average(qos[0]) + 1 - np.min(qos[1]))
    for col, (lo, hi) in zip((2, 3), cons):
        prod = np.cumprod(qos[col])[-1]
        if prod < lo or prod > hi:
            obj += 1
    return obj


def check(features_k1, min_cost, all_actions, n_cat):
    """ML2PN.check (ML2PN.py:15-56) minus file IO / print: mean over the test quarter of
    minCost / calc(selected QoS).  ``features_k1`` = load_data_pn(..., n_per=1)[0] (only row 0's
    constraint columns are read, ML2PN.py:51); ``all_actions`` = [T][nTest][8]."""
    n = len(features_k1)
    n_train, n_test = n // 4 * 3, n // 4                                # :20-21
    total = 0.0
    for j in range(n_test):
        chosen = [all_actions[i][j][0:QOS_NUM] for i in range(n_cat)]   # :34-36
        chosen = [a for a in chosen if sum(a) != 3]                     # :41-43 drop dummy rows
        row0 = features_k1[n_train + j][0]
        cons = [row0[QOS_NUM + 1:][:2], row0[QOS_NUM + 1:][2:]]         # :51
        qos = [[a[i] for a in chosen] for i in range(QOS_NUM)]          # :52-54
        total += min_cost[n_train + j] / score_calc(qos, cons)          # :55
    return total / n_test                                               # :57
