"""Seeded synthetic QWS-shaped data (there is no network: the reference's datasets and weights,
README.md:5-14 of the reference, cannot be fetched).

Two levels:

* :func:`make_dataset` writes/returns the reference's five JSON artefacts
  (``data/<ds>/{nodefeatures,edge_indices,labels,serviceFeature,minCostList}.data``, formats read
  at /root/reference/src/loadData.py:17-24,75-82) for small/medium cases — used by the format
  tests, the golden generator and ``main.py``.
* :func:`make_service_table` / :func:`make_problem_batch` build the same content directly as
  arrays for benchmark-scale batches (SURVEY.md §8d "Synthetic inputs") without a JSON detour.

All values are generated as float32 and widened, so host float64 comparisons
(loadData.py:123-124) and device float32 copies agree exactly.
"""
import json
import os
from dataclasses import dataclass

import numpy as np

REQUEST_FLOATS = (1.0, 0.3, 1.0, 1.0, 0.3, 1.0)   # request node's 6 floats


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def category_sizes(n_services, n_cat):
    base, extra = divmod(n_services, n_cat)
    return [base + (1 if c < extra else 0) for c in range(n_cat)]


@dataclass
class ServiceTable:
    """Problem-independent service side.  Services of category c are the contiguous id range
    [cat_ptr[c], cat_ptr[c+1]) (the order loadData.py:35-40 / :94-97 produce)."""
    n_cat: int
    cat_ptr: np.ndarray          # int32 [T+1]
    qos: np.ndarray              # float64 [S,4]  (q0,q1,q2=cost,q3=quality)
    edge_index: np.ndarray       # int64 [2,E_s]  co-occurrence graph, directed pairs
    edge_attr: np.ndarray        # float32 [E_s]

    @property
    def n_services(self):
        return int(self.qos.shape[0])

    @property
    def x_service(self):
        """[S,5] float32 rows ``[cat, q0,q1,q2,q3]`` (loadData.py:40)."""
        cat = np.repeat(np.arange(self.n_cat), np.diff(self.cat_ptr)).astype(np.float32)
        return np.concatenate([cat[:, None], self.qos.astype(np.float32)], 1)


@dataclass
class ProblemBatch:
    """B composition requests.  Workflow graphs are stored PyG-batch style (nodes of graph b
    contiguous, edge ids already offset)."""
    x: np.ndarray                # float32 [N,7]  [type, 6 floats]  (loadData.py:31)
    edge_index: np.ndarray       # int64 [2,E]
    batch: np.ndarray            # int64 [N]
    local_bounds: np.ndarray     # float64 [B,T,4] cost_lo,cost_hi,quality_lo,quality_hi (loadData.py:113)
    present: np.ndarray          # uint8 [B,T]    category has a task node (loadData.py:114)
    global_bounds: np.ndarray    # float64 [B,4]  request node c0_lo,c0_hi,c1_lo,c1_hi (loadData.py:110)

    @property
    def n_problems(self):
        return int(self.present.shape[0])


def make_service_table(n_cat, n_services, seed=0, degree=32, graph="random"):
    """q0,q1 ~ U(0,1); q2,q3 ~ U(0.9,1); E_s = S*degree directed edges as symmetric
    pairs with uniform endpoints and U(0,1] weights.  graph: "random" (pairs in the order drawn), "scan" (the same kind of
    pairs in the reference's emission order), anything else: no edges."""
    rng = np.random.default_rng(seed)
    sizes = category_sizes(n_services, n_cat)
    cat_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    qos = np.empty((n_services, 4), np.float32)
    qos[:, :2] = rng.random((n_services, 2), dtype=np.float32)
    qos[:, 2:] = 0.9 + 0.1 * rng.random((n_services, 2), dtype=np.float32)
    if graph == "random":
        n_pairs = n_services * degree // 2
        a = rng.integers(0, n_services, n_pairs)
        b = rng.integers(0, n_services, n_pairs)
        keep = a != b
        a, b = a[keep], b[keep]
        src = np.stack([a, b], 1).reshape(-1)
        dst = np.stack([b, a], 1).reshape(-1)
        w = (1.0 - rng.random(src.shape[0], dtype=np.float32)).astype(np.float32)
        edge_index = np.stack([src, dst]).astype(np.int64)
    elif graph == "scan":
        # the same random pairs in the order the reference's co-occurrence scan emits them (src/loadData.py:56-65:
        # for i < j in lexicographic order, edge i -> j then j -> i; a pair occurs once): every row's in-edges then come
        # sorted by source, which is what the tiled aggregate (gnnpn_csr_aggregate_tiled_f32) needs
        n_pairs = n_services * degree // 2
        a = rng.integers(0, n_services, n_pairs)
        b = rng.integers(0, n_services, n_pairs)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        key = np.unique(lo[lo != hi] * np.int64(n_services) + hi[lo != hi])
        lo, hi = key // n_services, key % n_services
        src = np.stack([lo, hi], 1).reshape(-1)
        dst = np.stack([hi, lo], 1).reshape(-1)
        w = (1.0 - rng.random(src.shape[0], dtype=np.float32)).astype(np.float32)
        edge_index = np.stack([src, dst]).astype(np.int64)
    else:
        edge_index = np.zeros((2, 0), np.int64)
        w = np.zeros(0, np.float32)
    return ServiceTable(n_cat, cat_ptr, qos.astype(np.float64), edge_index, w)


def scan_order(edge_index, edge_attr):
    """The same undirected weighted graph with its edge list in the order the reference's co-occurrence scan emits it
    (src/loadData.py:56-65: for i < j in lexicographic order, i -> j then j -> i; a pair occurs once — the first weight of a
    repeated directed edge is kept): every row's in-edges come sorted by source."""
    src, dst = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    n = int(max(src.max(), dst.max())) + 1 if src.size else 0
    key = src.astype(np.int64) * n + dst
    _, first = np.unique(key, return_index=True)                 # one weight per directed edge
    w_of = dict(zip(key[first].tolist(), np.asarray(edge_attr)[first].tolist()))
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    pk = np.unique(lo[lo != hi].astype(np.int64) * n + hi[lo != hi])
    lo, hi = pk // n, pk % n
    s2 = np.stack([lo, hi], 1).reshape(-1)
    d2 = np.stack([hi, lo], 1).reshape(-1)
    w2 = np.array([w_of.get(int(a) * n + int(b), w_of.get(int(b) * n + int(a))) for a, b in zip(s2, d2)], np.float32)
    return np.stack([s2, d2]).astype(np.int64), w2


def make_problem_batch(table, n_problems, seed=1, tasks_per_problem=10, lo_range=(0.0, 0.9)):
    """SURVEY.md §8d "Problem": 1 request node + n_t task nodes, chain edges in both directions;
    request floats [1,.3,1,1,.3,1]; task floats [1,lo,hi,1,lo,hi], lo ~ U(lo_range), hi = 1."""
    rng = np.random.default_rng(seed)
    T = table.n_cat
    n_t = min(T, tasks_per_problem)
    B = n_problems
    nodes_per = n_t + 1
    x = np.zeros((B * nodes_per, 7), np.float32)
    local = np.zeros((B, T, 4), np.float64)
    present = np.zeros((B, T), np.uint8)
    glob = np.tile(_f32([REQUEST_FLOATS[1], REQUEST_FLOATS[2], REQUEST_FLOATS[4], REQUEST_FLOATS[5]]), (B, 1))
    for b in range(B):
        cats = np.sort(rng.choice(T, n_t, replace=False)) if n_t < T else np.arange(T)
        lo = (lo_range[0] + (lo_range[1] - lo_range[0]) * rng.random((n_t, 2), dtype=np.float32))
        base = b * nodes_per
        x[base, 0] = 0
        x[base, 1:] = REQUEST_FLOATS
        x[base + 1: base + nodes_per, 0] = cats + 1
        x[base + 1: base + nodes_per, 1:] = np.stack(
            [np.ones(n_t), lo[:, 0], np.ones(n_t), np.ones(n_t), lo[:, 1], np.ones(n_t)], 1)
        local[b, cats, 0] = lo[:, 0]
        local[b, cats, 1] = 1.0
        local[b, cats, 2] = lo[:, 1]
        local[b, cats, 3] = 1.0
        present[b, cats] = 1
    i = np.arange(nodes_per - 1)
    e1 = np.stack([np.stack([i, i + 1], 1).reshape(-1), np.stack([i + 1, i], 1).reshape(-1)])
    edge_index = np.concatenate([e1 + b * nodes_per for b in range(B)], 1).astype(np.int64) \
        if nodes_per > 1 else np.zeros((2, 0), np.int64)
    batch = np.repeat(np.arange(B), nodes_per).astype(np.int64)
    return ProblemBatch(x, edge_index, batch, local, present, glob)


# ----------------------------------------------------------------------------------------------
# reference-format JSON artefacts
# ----------------------------------------------------------------------------------------------

def _objective(q, glob):
    """The scoring formula of /root/reference/src/ML2PN.py:6-12 (float64) for a [n,4] selection."""
    obj = 0.5 * (np.average(q[:, 0]) + 1 - np.min(q[:, 1]))
    for col, (lo, hi) in zip((2, 3), ((glob[0], glob[1]), (glob[2], glob[3]))):
        p = np.cumprod(q[:, col])[-1]
        if p < lo or p > hi:
            obj += 1
    return float(obj)


def make_dataset(n_cat, n_services, n_problems, seed=0, tasks_per_problem=10, lo_range=(0.85, 0.97)):
    """Return the five reference artefacts as python structures (json.dump-able):

    nodefeatures  [P][N_w][T+1 one-hot + 6 floats]   (loadData.py:26-33,107-114)
    edge_indices  [P][2][E_w]
    labels        [P][S] 0/1  — one feasible service per present category
    serviceFeature {"1": [[q0,q1,q2,q3],...], ...}   (loadData.py:35-40,91-97)
    minCostList   [P] objective of the labelled composition
    """
    table = make_service_table(n_cat, n_services, seed, graph="none")
    probs = make_problem_batch(table, n_problems, seed + 1, tasks_per_problem, lo_range)
    rng = np.random.default_rng(seed + 2)
    T, S = n_cat, n_services
    service_feature = {str(c + 1): table.qos[table.cat_ptr[c]: table.cat_ptr[c + 1]].tolist()
                       for c in range(T)}
    nodes_per = probs.x.shape[0] // n_problems
    nodefeatures, edge_indices, labels, min_cost = [], [], [], []
    for b in range(n_problems):
        rows = probs.x[b * nodes_per:(b + 1) * nodes_per]
        nodes = []
        for r in rows:
            onehot = [0] * (T + 1)
            onehot[int(r[0])] = 1
            nodes.append(onehot + [float(v) for v in r[1:].astype(np.float64)])
        nodefeatures.append(nodes)
        i = np.arange(nodes_per - 1)
        edge_indices.append([np.stack([i, i + 1], 1).reshape(-1).tolist(),
                             np.stack([i + 1, i], 1).reshape(-1).tolist()])
        lab = np.zeros(S, np.int64)
        chosen = []
        for c in np.nonzero(probs.present[b])[0]:
            ids = np.arange(table.cat_ptr[c], table.cat_ptr[c + 1])
            q = table.qos[ids]
            lb = probs.local_bounds[b, c]
            ok = ids[(q[:, 2] >= lb[0]) & (q[:, 2] <= lb[1]) & (q[:, 3] >= lb[2]) & (q[:, 3] <= lb[3])]
            pick = int(rng.choice(ok)) if len(ok) else int(rng.choice(ids))
            lab[pick] = 1
            chosen.append(pick)
        labels.append(lab.tolist())
        min_cost.append(_objective(table.qos[chosen], probs.global_bounds[b]))
    return {"nodefeatures": nodefeatures, "edge_indices": edge_indices, "labels": labels,
            "serviceFeature": service_feature, "minCostList": min_cost}


def write_dataset(root, name, ds):
    """Write the artefacts where the reference expects them: ``<root>/data/<name>/*.data``."""
    d = os.path.join(root, "data", name)
    os.makedirs(d, exist_ok=True)
    for key, fn in (("nodefeatures", "nodefeatures.data"), ("edge_indices", "edge_indices.data"),
                    ("labels", "labels.data"), ("serviceFeature", "serviceFeature.data"),
                    ("minCostList", "minCostList.data")):
        with open(os.path.join(d, fn), "w") as f:
            json.dump(ds[key], f)
    return d
