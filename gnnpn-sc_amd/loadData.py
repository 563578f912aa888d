"""Host-side data functions of the path behind the reference's entry points
(/root/reference/src/loadData.py): ``loadData`` (:14-69) and ``loadDataPN`` (:72-152) with the same
arguments, CWD-relative files and return values, plus converters from those JSON structures to
the array form the device pipeline packs (``tables_from_dataset``).

These are file parsing + list building, host work in the reference as well.  The per-problem
candidate reduction also exists as a kernel (gnnpn_select_candidates) for the device-resident
pipeline; ``loadDataPN`` is the artefact-format entry point (ranking JSON in, rows out).
"""
import json

import numpy as np

from .synth import ProblemBatch, ServiceTable


def _read(path):
    with open(path, "r") as f:
        return json.load(f)


def compute_inv_propesity(labels, A, B):
    """loadData.py:6-11 (returned by loadData, unused by the model)."""
    lab = np.asarray(labels)
    n = lab.shape[0]
    freqs = np.ravel(lab.sum(axis=0))
    C = (np.log(n) - 1) * np.power(B + 1, A)
    return np.ravel(1.0 + C * np.power(freqs + B, -A))


def service_cooccurrence_graph(labels, n_train=3000):
    """loadData.py:42-65 without the O(S^2) python scan: adjacency = L^T L over the first
    ``n_train`` label vectors; pairs i<j in row-major order, each as (i->j),(j->i); the edge whose
    source is u weighs adj[u][v]/useTimes[u]."""
    lab = (np.asarray(labels[:n_train]) == 1).astype(np.int64)
    use = lab.sum(0)
    adj = lab.T @ lab
    iu, ju = np.nonzero(np.triu(adj, 1))
    cnt = adj[iu, ju]
    src = np.stack([iu, ju], 1).reshape(-1)
    dst = np.stack([ju, iu], 1).reshape(-1)
    w = np.stack([cnt / use[iu], cnt / use[ju]], 1).reshape(-1)
    return [src.tolist(), dst.tolist()], w.tolist()


def loadData(dataset=""):
    """-> (nodefeatures, serviceFeatureList, edge_indices, edge_indices_service, edge_attrs_service,
    labels, inv_propen)   — loadData.py:14-69."""
    d = f"./data/{dataset}/" if dataset != "" else "./data/"
    nodefeatures = _read(d + "nodefeatures.data")
    edge_indices = _read(d + "edge_indices.data")
    labels = _read(d + "labels.data")
    serviceFeature = _read(d + "serviceFeature.data")
    nodes = [[[n[:-6].index(1)] + n[-6:] for n in prob] for prob in nodefeatures]        # :26-33
    keys = sorted(int(k) for k in serviceFeature)                                        # :36
    services = [[k - keys[0]] + feat[-4:] for k in keys for feat in serviceFeature[str(k)]]
    eis, eas = service_cooccurrence_graph(labels)
    return nodes, services, edge_indices, eis, eas, labels, compute_inv_propesity(labels[:3000], 0.55, 1.5)


def problem_bounds(nodes, n_cat):
    """loadData.py:102-114 for one problem -> (local [T,4] f64, present [T] u8, global [4] f64)."""
    local = np.zeros((n_cat, 4), np.float64)
    present = np.zeros(n_cat, np.uint8)
    glob = np.zeros(4, np.float64)
    for node in nodes:
        b = list(node[-5:-3]) + list(node[-2:])
        if node[0] == 1:
            glob[:] = b
        else:
            c = node[:-6].index(1) - 1
            local[c] = b
            present[c] = 1
    return local, present, glob


def reduce_from_ranking(ranking, local, present, glob, cat_of, qos, n_per):
    """loadData.py:116-149 for one problem with candidate order = rank order (the reference's
    np.random.shuffle at :135 makes its order arbitrary).
    -> rows [T*n_per][9]."""
    n_cat = local.shape[0]
    rk = np.asarray(ranking, dtype=np.int64)
    c = cat_of[rk]
    ok = present[c].astype(bool) & (local[c, 0] <= qos[rk, 2]) & (qos[rk, 2] <= local[c, 1]) \
        & (local[c, 2] <= qos[rk, 3]) & (qos[rk, 3] <= local[c, 3])
    rows = []
    for cat in range(n_cat):
        tail = glob.tolist() if cat == 0 else [0, 0, 0, 0]
        pick = rk[ok & (c == cat)][:n_per]
        if len(pick):
            for i in range(n_per):
                rows.append([cat] + qos[pick[i % len(pick)]].tolist() + tail)
        else:
            rows += [[cat, 0, 1, 1, 1] + tail for _ in range(n_per)]
    return rows


def loadDataPN(epoch=7, dataset="", serviceNumber=5):
    """-> (newServiceFeatures [P][T*K][9], newlabels [P])   — loadData.py:72-152."""
    d = f"./data/{dataset}/" if dataset != "" else "./data/"
    nodefeatures = _read(d + "nodefeatures.data")
    serviceFeature = _read(d + "serviceFeature.data")
    minCostList = _read(d + "minCostList.data")
    if epoch >= 0:
        rankings = _read(f"./solutions/ML/{dataset}/testServices-epoch{epoch}.txt")
    else:
        rankings = _read(f"./solutions/pretrained/{dataset}-ML.txt")
    n_cat = len(serviceFeature)
    cat_of = np.concatenate([np.full(len(serviceFeature[k]), int(k) - 1) for k in serviceFeature])   # :94-97
    qos = np.asarray([f[-4:] for k in serviceFeature for f in serviceFeature[k]], dtype=np.float64)
    feats = []
    for nodes, ranking in zip(nodefeatures, rankings):
        local, present, glob = problem_bounds(nodes, n_cat)
        feats.append(reduce_from_ranking(ranking, local, present, glob, cat_of, qos, serviceNumber))
    return feats, list(minCostList[:len(feats)])


def tables_from_dataset(ds, first=0, last=None):
    """Reference-format JSON structures -> (ServiceTable, ProblemBatch) arrays for the device
    pipeline (problems ``first:last``).  The service graph is built from the labels exactly as
    loadData does."""
    sf = ds["serviceFeature"]
    keys = sorted(int(k) for k in sf)
    sizes = [len(sf[str(k)]) for k in keys]
    cat_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    qos = np.asarray([f[-4:] for k in keys for f in sf[str(k)]], dtype=np.float64)
    eis, eas = service_cooccurrence_graph(ds["labels"])
    table = ServiceTable(len(keys), cat_ptr, qos, np.asarray(eis, dtype=np.int64).reshape(2, -1),
                         np.asarray(eas, dtype=np.float32))
    nodes_all = ds["nodefeatures"][first:last]
    edges_all = ds["edge_indices"][first:last]
    T = len(keys)
    xs, eis_w, batch, local, present, glob = [], [], [], [], [], []
    off = 0
    for b, (nodes, edges) in enumerate(zip(nodes_all, edges_all)):
        xs += [[n[:-6].index(1)] + n[-6:] for n in nodes]
        e = np.asarray(edges, dtype=np.int64).reshape(2, -1)
        eis_w.append(e + off)
        batch += [b] * len(nodes)
        off += len(nodes)
        lo, pr, gl = problem_bounds(nodes, T)
        local.append(lo)
        present.append(pr)
        glob.append(gl)
    if not nodes_all:          # the service table alone (first == last)
        return table, ProblemBatch(np.zeros((0, 7), np.float32), np.zeros((2, 0), np.int64), np.zeros(0, np.int64),
                                   np.zeros((0, T, 4)), np.zeros((0, T), np.uint8), np.zeros((0, 4)))
    pb = ProblemBatch(np.asarray(xs, dtype=np.float32), np.concatenate(eis_w, 1) if eis_w else np.zeros((2, 0), np.int64),
                      np.asarray(batch, dtype=np.int64), np.stack(local), np.stack(present), np.stack(glob))
    return table, pb


def addS(PriS, serviceFeatures, constraints, serviceIndex, ser2idxdiv, ser2idxmod, reduct=False, sSet=None):
    """Candidate lists of one problem for the WOA fine-tuner — loadData.py:155-202, same arguments and result.

    Walks the services in ``PriS`` order and keeps, per category, those inside the problem's local bounds.  With a
    truthy ``reduct`` (a response-time threshold) each category keeps a small front instead: a newcomer replaces the
    first front member it beats on both q0 (lower) and q1 (higher) while that member's q1 is below the threshold, unless
    that member is one of the pointer network's picks (``sSet``, 5-decimal tuples), and is appended when it replaced
    nothing and either is such a pick itself or straddles the threshold (q1 > reduct > q0).  The front starts from the
    sentinel (1, 0, 1, 1), which occupies slot 0 until something replaces it — the kept list is empty until then
    (:183-186).  Returns the kept lists of the categories in ``serviceIndex`` (the problem's task nodes, in node order)."""
    n_slots = 50                                                       # :156: capacity is hard-coded in the reference
    kept = [[] for _ in range(n_slots)]
    front = [[[1], [0], [1], [1]] for _ in range(n_slots)]             # per category: columns q0, q1, cost, quality
    for s in PriS:
        cat = ser2idxdiv[s]
        row = serviceFeatures[str(cat + 1)][ser2idxmod[s]]
        entry = (row[-4], row[-3], row[-2], row[-1])
        q0, q1, cost, quality = entry
        bounds = constraints[cat + 1]
        if not (bounds[0] <= cost <= bounds[1] and bounds[2] <= quality <= bounds[3]):
            continue
        if not reduct:
            kept[cat].append(entry)
            continue
        cols, replaced = front[cat], False
        for x in range(len(cols[0])):
            member = tuple(round(cols[c][x], 5) for c in range(4))
            if sSet and member in sSet:
                continue
            if q0 < cols[0][x] and q1 > cols[1][x] and cols[1][x] < reduct:
                for c in range(4):
                    cols[c][x] = entry[c]
                if not kept[cat]:
                    kept[cat].append(entry)
                else:
                    kept[cat][x] = entry
                replaced = True
                break
            if (q0 > cols[0][x] and q1 < cols[1][x]) or q1 > reduct > q0:
                break
        if not replaced and ((sSet and tuple(round(v, 5) for v in entry) in sSet) or q1 > reduct > q0):
            for c in range(4):
                cols[c].append(entry[c])
            kept[cat].append(entry)
    return [kept[c] for c in serviceIndex]


def loadDataOther(dataset="", reduct=False, sSetList=None, train=False):
    """-> (newServiceFeatures, constraintsList, minCostList) — loadData.py:205-276: per problem of the test quarter (or of
    the whole set with ``train``) the candidate lists of ``addS`` with empty categories dropped, and the two global
    product constraints as ``[[[lo0, hi0]], [[lo1, hi1]]]``."""
    d = f"./data/{dataset}/" if dataset != "" else "./data/"
    nodefeatures = _read(d + "nodefeatures.data")
    serviceFeature = _read(d + "serviceFeature.data")
    minCostList = _read(d + "minCostList.data")
    n_cat = len(serviceFeature)
    ser2idxdiv, ser2idxmod = [], []
    for key in serviceFeature:                                         # file order, :221-224
        ser2idxdiv += [int(key) - 1] * len(serviceFeature[key])
        ser2idxmod += list(range(len(serviceFeature[key])))
    n_train = len(nodefeatures) // 4 * 3
    first = 0 if train else n_train
    every_service = list(range(len(ser2idxdiv)))
    features, constraintsList = [], []
    for number, nodes in enumerate(nodefeatures[first:], start=first):
        constraints = {c: [0] * 8 for c in range(1, n_cat + 1)}        # [local lo/hi x2 | global lo/hi x2]
        for node in nodes:
            pair = node[-5:-3] + node[-2:]
            if node[0] == 1:                                           # the request node: global constraints
                for c in constraints:
                    constraints[c][-4:] = pair
            else:
                constraints[node[:-6].index(1)][-8:-4] = pair
        serviceIndex = [node[:-6].index(1) - 1 for node in nodes][1:]  # :251-257
        sSet = sSetList[number - n_train] if sSetList and number >= n_train else None
        lists = addS(every_service, serviceFeature, constraints, serviceIndex, ser2idxdiv, ser2idxmod, reduct, sSet)
        features.append([lst for lst in lists if len(lst) > 0])
        any_cat = constraints[next(iter(constraints))]                 # :268-272: the first key's global pair
        constraintsList.append([[any_cat[-4:-2]], [any_cat[-2:]]])
    return features, constraintsList, minCostList
