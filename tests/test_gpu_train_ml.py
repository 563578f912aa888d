"""-m gpu: the training step of the GNN candidate-ranking model (gnnpn_sc_amd/trainML.py, csrc/train_ml.hip)
against fixtures the reference's own Net glue produced under autograd (tests/golden/make_golden.py::gen_ml_train: model.train(),
BCELoss, backward, Adam — stand-in convs, so "parity unpinned" at the torch_geometric boundary as for the forward)."""
import numpy as np
import pytest
import torch

from conftest import golden, record_agreement
from oracle import ml as oml
from oracle import ml_train as omt

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-4       # relative to the parameter's largest gradient entry ...
GRAD_FLOOR = 1e-5      # ... or this fraction of the step's largest gradient (exact-zero gradients: biases in front of a BatchNorm)
PARAM_ATOL = 3e-6      # weights after Adam, where |g| is clear of Adam's eps
STAT_ATOL = 2e-5


def _net(fx, dev, sd):
    from gnnpn_sc_amd.modelML import Net
    net = Net(int(fx["hidden"]), int(fx["S"]), int(fx["emb"]), int(fx["n_gin"]), int(fx["n_gcn"]),
              isServices=bool(int(fx["is_services"])) if "is_services" in fx.files else True)
    net.load_state_dict(sd, strict=True)
    return net.to(dev)


def _batch(fx, tag, dev):
    from gnnpn_sc_amd.trainML import MLBatch
    x, ei, bvec = fx[f"{tag}_x"], fx[f"{tag}_edge_index"], fx[f"{tag}_batch"]
    B, S = int(fx["B"]), int(fx["S"])
    y = torch.from_numpy(fx[f"{tag}_y"]).view(B, S)
    graphs, start = [], 0
    for b in range(B):
        n = int((bvec == b).sum())
        m = (bvec[ei[0]] == b)
        graphs.append({"x": torch.from_numpy(x[start:start + n]), "edge_index": torch.from_numpy(ei[:, m] - start), "y": y[b]})
        start += n
    service = {"x_service": torch.from_numpy(fx["x_service"]), "edge_index_service": torch.from_numpy(fx["edge_index_service"]),
               "edge_attr_service": torch.from_numpy(fx["edge_attr_service"])}
    mb = MLBatch(graphs, service, dev)
    # the batching reproduces the fixture's (torch_geometric-1.7.0-style) batch
    assert torch.equal(mb.x.cpu(), torch.from_numpy(x)) and mb.N == x.shape[0]
    return mb


def _check_step(fx, tag, net, grads, loss, what):
    keys = omt.trainable_keys(net.state_dict(), int(fx["n_gin"]), int(fx["n_gcn"]), bool(int(fx["is_services"])) if "is_services" in fx.files else True)
    assert set(grads) == set(keys)
    assert abs(float(loss) - float(fx[f"{tag}_loss"])) <= 2e-6 * max(1.0, abs(float(fx[f"{tag}_loss"])))
    gmax = max(float(np.abs(fx[f"{tag}_grad/{k}"]).max()) for k in keys)
    worst = 0.0
    for k in keys:
        want = fx[f"{tag}_grad/{k}"]
        got = grads[k].reshape(want.shape).cpu().numpy()
        tol = GRAD_RTOL * float(np.abs(want).max()) + GRAD_FLOOR * gmax
        err = float(np.abs(got - want).max())
        assert err <= tol, (what, k, err, tol)
        if float(np.abs(want).max()) > 1e-3 * gmax:      # the exactly-zero gradients (biases in front of a BatchNorm) are noise on both sides
            worst = max(worst, err / float(np.abs(want).max()))
    return worst, gmax


@pytest.mark.parametrize("name", ["tiny", "qws", "noservices"])
def test_ml_training_step_vs_reference_autograd(dev, name):
    """First step from the seeded weights: loss, every gradient, the weights after Adam, BatchNorm running statistics.
    ("noservices": the isServices=False ablation of modelML.py:157-162 — noServicesLins instead of the GCN layers; round 5.)"""
    from gnnpn_sc_amd.trainML import MLAdam, ml_forward_backward
    fx = golden(f"ml_train_{name}.npz")
    n_gin, n_gcn = int(fx["n_gin"]), int(fx["n_gcn"])
    sd = oml.make_state_dict(int(fx["hidden"]), int(fx["emb"]), n_gin, n_gcn, int(fx["seed"]) + 2)
    net = _net(fx, dev, sd).train()
    mb = _batch(fx, "first", dev)
    loss, scores, grads = ml_forward_backward(net, mb)
    assert float((scores.cpu() - torch.from_numpy(fx["first_scores"])).abs().max()) < 1e-5
    worst, gmax = _check_step(fx, "first", net, grads, loss.item(), name)
    MLAdam(net, float(fx["lr"])).step(grads)
    new = net.state_dict()
    for k in grads:
        clear = np.abs(fx[f"first_grad/{k}"]) > 1e-6
        if clear.any():
            d = np.abs(new[k].cpu().numpy() - fx[f"first_param/{k}"])[clear].max()
            assert d <= PARAM_ATOL, (k, d)
    for pre in omt.bn_prefixes(n_gin, n_gcn):
        rm, rv = fx[f"first_running_mean/{pre}"], fx[f"first_running_var/{pre}"]
        assert np.abs(new[pre + ".running_mean"].cpu().numpy() - rm).max() <= STAT_ATOL * max(1.0, np.abs(rm).max())
        assert np.abs(new[pre + ".running_var"].cpu().numpy() - rv).max() <= STAT_ATOL * max(1.0, np.abs(rv).max())
        assert int(new[pre + ".num_batches_tracked"]) == 1
    record_agreement(f"ml_train_gradients_{name}", {"worst_relative_gradient_error": worst, "largest_gradient": gmax,
                                                    "loss": float(loss.item()), "loss_reference": float(fx["first_loss"])})


def test_ml_training_later_step_with_adam_state(dev):
    """The third step of the tiny run, replayed from the reference's state before it (weights, running statistics, Adam
    moments): the bias-corrected Adam update at step 3."""
    from gnnpn_sc_amd.trainML import MLAdam, ml_forward_backward
    fx = golden("ml_train_tiny.npz")
    n_gin, n_gcn, steps = int(fx["n_gin"]), int(fx["n_gcn"]), int(fx["steps"])
    sd = oml.make_state_dict(int(fx["hidden"]), int(fx["emb"]), n_gin, n_gcn, int(fx["seed"]) + 2)
    for k in sd:
        if f"before_last/{k}" in fx.files:
            sd[k] = torch.from_numpy(fx[f"before_last/{k}"])
    net = _net(fx, dev, sd).train()
    mb = _batch(fx, "last", dev)
    loss, _, grads = ml_forward_backward(net, mb)
    _check_step(fx, "last", net, grads, loss.item(), "tiny/last")
    adam = MLAdam(net, float(fx["lr"]))
    adam.step_no = steps - 1
    for k in grads:
        adam.state[k] = (torch.from_numpy(fx[f"before_last_m/{k}"]).to(dev).contiguous(), torch.from_numpy(fx[f"before_last_v/{k}"]).to(dev).contiguous())
    adam.step(grads)
    new = net.state_dict()
    for k in grads:
        clear = np.abs(fx[f"last_grad/{k}"]) > 1e-6
        if clear.any():
            d = np.abs(new[k].cpu().numpy() - fx[f"last_param/{k}"])[clear].max()
            assert d <= PARAM_ATOL, (k, d)


def test_ml_training_vs_live_autograd_oracle_other_shape(dev):
    """A shape no fixture holds (3 GIN layers, 4 GCN layers as Normal-ML, a batch of ONE graph = the last batch of an odd
    dataset) against the live autograd oracle."""
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.trainML import MLBatch, ml_forward_backward
    T, S, hidden, emb, n_gin, n_gcn = 9, 120, 32, 12, 3, 4
    table = synth.make_service_table(T, S, seed=5, degree=6)
    pb = synth.make_problem_batch(table, 1, seed=6, tasks_per_problem=4)
    sd = oml.make_state_dict(hidden, emb, n_gin, n_gcn, 9)
    y = (torch.rand(S, generator=torch.Generator().manual_seed(3)) < 0.05).float()
    data = oml.make_data(torch.from_numpy(pb.x), torch.from_numpy(pb.edge_index), torch.from_numpy(pb.batch),
                         torch.from_numpy(table.x_service), torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr))
    ref = omt.train_step(sd, data, y, n_gin, n_gcn, S, 1e-3)
    from gnnpn_sc_amd.modelML import Net
    net = Net(hidden, S, emb, n_gin, n_gcn)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).train()
    mb = MLBatch([{"x": torch.from_numpy(pb.x), "edge_index": torch.from_numpy(pb.edge_index), "y": y}],
                 {"x_service": torch.from_numpy(table.x_service), "edge_index_service": torch.from_numpy(table.edge_index),
                  "edge_attr_service": torch.from_numpy(table.edge_attr)}, dev)
    loss, _, grads = ml_forward_backward(net, mb)
    assert abs(float(loss.item()) - float(ref["loss"])) <= 2e-6
    gmax = max(float(v.abs().max()) for v in ref["grads"].values())
    for k, want in ref["grads"].items():
        err = float((grads[k].reshape(want.shape).cpu() - want).abs().max())
        assert err <= GRAD_RTOL * float(want.abs().max()) + GRAD_FLOOR * gmax, (k, err)


def test_trainml_driver_end_to_end(dev, tmp_path, monkeypatch):
    """TrainML.start on a small synthetic dataset in the reference's file formats: two epochs, the loss falls, the
    artefacts (rankings JSON [P][S], model state) are written, the learning-rate scheduler behaves as torch's."""
    import json
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.trainML import ReduceLROnPlateau, TrainML
    monkeypatch.chdir(tmp_path)
    T, S, P = 5, 30, 13                                   # odd: the last training batch holds one graph
    synth.write_dataset(str(tmp_path), "Tiny", synth.make_dataset(T, S, P, seed=2, tasks_per_problem=3))
    torch.manual_seed(0)
    tr = TrainML("Tiny", 2, 2, 16, 8, 0.0, 0.01, 3)
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        net = tr.start()
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("Epoch:")]
    assert len(lines) == 3 and lines[0].startswith("Epoch: 000, LR: 0.01000, Loss: ")
    losses = [float(ln.split("Loss: ")[1].split(",")[0]) for ln in lines]
    assert losses[-1] < losses[0], losses
    with open("solutions/ML/Tiny/testServices-epoch1.txt") as f:
        ranks = json.load(f)
    assert len(ranks) == P and sorted(ranks[0]) == list(range(S))
    sd = torch.load("solutions/ML/Tiny/model-1.pkl", map_location="cpu")
    assert set(sd) == set(net.state_dict())

    class A:
        lr = 1.0
    a = A()
    sch = ReduceLROnPlateau(a, factor=0.5, patience=1, min_lr=0.2)
    mine = []
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=1, min_lr=0.2)
    theirs = []
    for m in [1.0, 0.9, 0.95, 0.96, 0.97, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]:
        sch.step(m)
        ref.step(m)
        mine.append(a.lr)
        theirs.append(opt.param_groups[0]["lr"])
    assert mine == theirs


def test_main_cli_ml_mode(dev, tmp_path, monkeypatch):
    """``python main.py QWS ML`` (reference main.py:20-25): the [QWS-ML] section drives TrainML."""
    import importlib.util
    import json
    import os
    import gnnpn_sc_amd.synth as synth
    monkeypatch.chdir(tmp_path)
    synth.write_dataset(str(tmp_path), "QWS", synth.make_dataset(5, 30, 8, seed=3, tasks_per_problem=3))
    (tmp_path / "environment.ini").write_text("[QWS-ML]\nnumLayersGIN = 2\nnumLayersGCN = 2\nhiddenChannels = 16\n"
                                              "embeddingChannels = 8\ndropout = 0.0\nlr = 0.001\nepochs = 1\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gnnpn_main_cli", os.path.join(root, "main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["main.py", "QWS", "ML"]) == 0
    with open("solutions/ML/QWS/testServices-epoch0.txt") as f:
        assert len(json.load(f)) == 8
