"""ML+2PN scorer behind the reference's entry points (/root/reference/src/ML2PN.py): ``calc``
(:6-12) and ``check(dataset, serCategory, epoch)`` (:15-56) — same CWD-relative artefact files,
same printed line ``<epoch> <score>``.  Host code in the reference as well (json + float64 numpy
over a few thousand rows); ``infer`` is the new part: it PRODUCES the artefacts ``check`` reads by
running the device pipeline, which the reference does inside its training drivers
(trainML.py:146-149, trainPNHigh.py:131-144).
"""
import json
import os

import numpy as np

from .loadData import loadDataPN


def calc(qos, cons):
    """0.5*(mean q0 + 1 - min q1) + one penalty per global constraint whose product QoS is out of
    bounds (ML2PN.py:6-12); float64."""
    obj = 0.5 * (np.average(qos[0]) + 1 - np.min(qos[1]))
    for col, (lo, hi) in zip((2, 3), cons):
        prod = np.cumprod(qos[col])[-1]
        if prod < lo or prod > hi:
            obj += 1
    return obj


def score(features_k1, min_cost, all_actions, serCategory):
    """The arithmetic of check (ML2PN.py:20-55) on in-memory structures -> mean(minCost / calc)."""
    qosNum = 4
    n = len(features_k1)
    n_train, n_test = n // 4 * 3, n // 4
    total = 0.0
    for j in range(n_test):
        picked = [all_actions[i][j][:qosNum] for i in range(serCategory)]
        picked = [a for a in picked if sum(a) != 3]                    # dummy rows [0,1,1,1] (:41-43)
        row0 = features_k1[n_train + j][0]
        cons = [row0[qosNum + 1:][:2], row0[qosNum + 1:][2:]]          # :51
        qos = [[a[i] for a in picked] for i in range(qosNum)]
        total += min_cost[n_train + j] / calc(qos, cons)
    return total / n_test


def check(dataset, serCategory, epoch):
    features, _ = loadDataPN(epoch=-1, dataset=dataset, serviceNumber=1)          # :19
    with open(f"./data/{dataset}/minCostList.data", "r") as f:
        min_cost = json.load(f)
    url = f"./solutions/pretrained/{dataset}-PNHigh.txt" if epoch == -1 else \
        f"./solutions/PNHigh/{dataset}/allActions{epoch}.txt"                     # :25-28
    with open(url) as f:
        all_actions = json.load(f)
    result = score(features, min_cost, all_actions, serCategory)
    print(epoch, result)                                                          # :57
    return result


def infer(dataset, net, low, high, n_per, epoch=-1, device="cuda:0", batch_size=128, precision="f32"):
    """Run ML+2PN inference over ``./data/<dataset>`` on the GPU and write the two artefacts that
    ``check`` reads: the rankings of ALL problems (trainML.py:146-149 format, [P][S] ints) and the
    High-level actions of the test quarter (trainPNHigh.py:133-144 format, [T][nTest][8]).
    ``precision``: arithmetic of the recurrent products and of the large-graph GIN layers — "f32" by default here (the artefact
    files are the reference's hand-off format: the parity path), "split" = what bench.py measures (the exact
    three-piece products: same selections on every pinned problem)."""
    import torch
    from . import loadData as ld
    from .pipeline import DeviceBatch, DeviceServices, ML2PNPipeline
    d = f"./data/{dataset}/"
    ds = {k: json.load(open(d + fn)) for k, fn in (
        ("nodefeatures", "nodefeatures.data"), ("edge_indices", "edge_indices.data"), ("labels", "labels.data"),
        ("serviceFeature", "serviceFeature.data"), ("minCostList", "minCostList.data"))}
    dev = torch.device(device)
    P = len(ds["nodefeatures"])
    T = len(ds["serviceFeature"])
    pipe = ML2PNPipeline(net.to(dev).eval(), low.to(dev).eval(), high.to(dev).eval(), n_per, precision=precision)
    table, _ = ld.tables_from_dataset(ds, 0, 0)
    svc = DeviceServices.from_table(table, dev)
    rankings, actions = [], [[] for _ in range(T)]
    n_train = P // 4 * 3
    from . import ops
    for lo in range(0, P, batch_size):
        hi = min(P, lo + batch_size)
        _, pb = ld.tables_from_dataset(ds, lo, hi)
        batch = DeviceBatch.from_problems(pb, dev)
        # a timed-out inter-workgroup hand-off must never reach an artefact file: checked per batch, repeated once if it happens
        act = ops.run_checked(lambda attempt: pipe.run(svc, batch, write_through=attempt > 0)["actions"].cpu().numpy().astype(np.float64),
                              dev)                                       # [b,T,8]
        rankings += pipe.rankings(svc, batch).cpu().tolist()
        for b in range(hi - lo):
            if lo + b >= n_train:
                for t in range(T):
                    actions[t].append(act[b, t].tolist())
    return write_artifacts(dataset, epoch, rankings, actions)


def artifact_paths(dataset, epoch):
    """Where ``loadDataPN`` / ``check`` look (loadData.py:84-89, ML2PN.py:25-28): epoch -1 -> ./solutions/pretrained/."""
    if epoch == -1:
        return f"./solutions/pretrained/{dataset}-ML.txt", f"./solutions/pretrained/{dataset}-PNHigh.txt"
    return f"./solutions/ML/{dataset}/testServices-epoch{epoch}.txt", f"./solutions/PNHigh/{dataset}/allActions{epoch}.txt"


def write_artifacts(dataset, epoch, rankings, actions):
    """The two artefacts of an ML+2PN run in the reference's formats: rankings [P][S] ints as TrainML.start dumps them
    (trainML.py:146-149) and the High-level actions [T][nTest][8] floats as the PNHigh eval block dumps them
    (trainPNHigh.py:133-144)."""
    p_rank, p_act = artifact_paths(dataset, epoch)
    for p in (p_rank, p_act):
        os.makedirs(os.path.dirname(p), exist_ok=True)
    with open(p_rank, "w") as f:
        json.dump(rankings, f)
    with open(p_act, "w") as f:
        json.dump(actions, f)
    return p_rank, p_act
