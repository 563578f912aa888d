#!/usr/bin/env python3
"""CLI of the ML+2PN mode, mirroring the reference's ``python main.py <dataset> ML+2PN [epoch]``
(reference main.py:216-231): score the artefacts under ./solutions with ``ML2PN.check``.

    python main.py QWS ML+2PN            # epoch from environment.ini ([QWS-ML+2PN] -> -1)
    python main.py Normal ML+2PN 3       # argv[3] overrides the epoch (reference main.py:219-220)
    python main.py QWS ML+2PN -1 --infer # first PRODUCE the artefacts on the GPU, then score them.  Weights: epoch -1 ->
                                         # ./solutions/pretrained/<ds>-{ML.pt,PNLow.model,PNHigh.model}; epoch n -> what
                                         # `main.py <ds> ML` / `PNHigh` wrote for that epoch.  A missing file is an error
                                         # unless --random-init asks for seeded random weights

    python main.py QWS WOA [epoch]       # ES-WOA fine-tuning of the ML+2PN solution on the GPU (reference main.py:86-104,
                                         # mode ML2PNWOATest of [<ds>-WOA]); --seed N makes the run reproducible

    python main.py QWS PNLow [epochML]   # REINFORCE training of the Low pointer net on the GPU (reference main.py:40-52)
    python main.py QWS PNHigh [epochML [epochPNLow]]   # ... of the High net against the trained Low net (:66-84); --epochs=N

    python main.py QWS ML                # training of the GNN candidate-ranking model on the GPU (reference main.py:20-32)

Every other approach of the reference (the other WOA modes, GA/DQN baselines) is out of scope and answers with the
reference's own message.
"""
import configparser
import os
import sys


def _models(cfg, ds, n_services, n_cat, epoch=-1, random_init=False):
    """The three networks of an ML+2PN inference run with the weights that belong to ``epoch``:
      epoch -1  ./solutions/pretrained/<ds>-{ML.pt, PNLow.model, PNHigh.model}   (reference loadData.py:84-89, trainPNHigh.py:237-242)
      epoch  n  ./solutions/ML/<ds>/model-{n}.pkl (trainML.py:147; a state_dict here), ./solutions/PNHigh/<ds>/epoch{n}.model
                and epoch{n}_low.model (trainPNHigh.py:118-129) — what `main.py <ds> ML` / `PNHigh` of this build write.
    A missing file raises: artefacts named after an epoch must never come from untrained weights.  ``random_init``
    (--random-init) says explicitly that seeded random weights are wanted where a file is absent (smoke runs)."""
    import torch
    from gnnpn_sc_amd.modelML import Net
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    ml, pn = cfg[f"{ds}-ML"], cfg[f"{ds}-PNHigh"]
    K, H = int(pn["serNumber"]), int(pn["hidden_size"])
    if epoch == -1:
        pre = "./solutions/pretrained/"
        p_ml, p_low, p_high = pre + f"{ds}-ML.pt", pre + f"{ds}-PNLow.model", pre + f"{ds}-PNHigh.model"
    else:
        p_ml = f"./solutions/ML/{ds}/model-{epoch}.pkl"
        p_low, p_high = f"./solutions/PNHigh/{ds}/epoch{epoch}_low.model", f"./solutions/PNHigh/{ds}/epoch{epoch}.model"
    missing = [p for p in (p_ml, p_low, p_high) if not os.path.exists(p)]
    if missing and not random_init:
        raise FileNotFoundError(f"ML+2PN --infer for epoch {epoch}: no weights at {', '.join(missing)} "
                                "(train them with `main.py <ds> ML|PNLow|PNHigh`, or pass --random-init for seeded random weights)")
    torch.manual_seed(0)
    sd_ml = _load_ml_checkpoint(p_ml) if os.path.exists(p_ml) else None
    # the vocabulary the checkpoint was trained with (trainML builds the reference's Embedding(100, c); the synthetic
    # 1000-task configurations need larger tables)
    vocab = sd_ml["nodeEncoder.embeddings.0.weight"].shape[0] if sd_ml is not None else max(100, n_cat + 1)
    net = Net(int(ml["hiddenChannels"]), n_services, int(ml["embeddingChannels"]), int(ml["numLayersGIN"]),
              int(ml["numLayersGCN"]), vocab=vocab)
    if sd_ml is not None:
        net.load_state_dict(sd_ml)
    mk = lambda lvl: CombinatorialRL(0, H, n_cat * K, int(pn["n_glimpses"]), float(pn["tanh_exploration"]),  # noqa: E731
                                     int(pn["use_tanh"]), reward, "Dot", K, n_cat, level=lvl)
    low, high = mk("Low"), mk("High")
    for m, path in ((low, p_low), (high, p_high)):
        if os.path.exists(path):            # checkpoint format of trainPNLow.py:112-117 / trainPNHigh.py:118-129
            m.load_state_dict(torch.load(path, map_location="cpu")["model"])
    return net, low, high, K


def _load_ml_checkpoint(path):
    """The GNN ranker's weights as a ``state_dict``.  This build's `main.py <ds> ML` saves the state_dict; the reference
    pickles the WHOLE module (``torch.save(self.model, ...)``, trainML.py:147), which unpickles only where its classes
    (src.models.modelML.Net, torch_geometric 1.7.0's GINConv / GCNConv) are importable — then its state_dict is taken;
    anything else is refused with a message that names the expected format."""
    import torch
    try:
        try:
            obj = torch.load(path, map_location="cpu", weights_only=True)     # a state_dict: no code is executed
        except Exception:        # noqa: BLE001
            # a whole-module pickle executes arbitrary code when it is loaded: only for checkpoints the user vouches for
            if os.environ.get("GNNPN_TRUST_CHECKPOINT") != "1":
                raise
            obj = torch.load(path, map_location="cpu", weights_only=False)
    except Exception as e:        # noqa: BLE001  (unpickling a whole-module checkpoint without its classes raises many kinds)
        raise RuntimeError(f"{path}: not loadable here ({type(e).__name__}: {e}).  Expected a state_dict of Net "
                           "(keys 'nodeEncoder.embeddings.0.weight', ...) as `main.py <ds> ML` of this build writes it; a "
                           "whole-module pickle of the reference (trainML.py:147) needs src.models.modelML and "
                           "torch_geometric==1.7.0 importable AND GNNPN_TRUST_CHECKPOINT=1 (unpickling a module runs its code) — or re-save it there "
                           "with torch.save(model.state_dict(), path)") from e
    if isinstance(obj, torch.nn.Module):
        obj = obj.state_dict()
    if not (isinstance(obj, dict) and "nodeEncoder.embeddings.0.weight" in obj):
        raise RuntimeError(f"{path}: expected a state_dict of Net (key 'nodeEncoder.embeddings.0.weight' ...), got {type(obj).__name__}")
    return obj


def main(argv):
    if len(argv) < 3:
        print("Please check the parameters!")
        return 1
    dataset, approach = argv[1], argv[2]
    ds = {"QWS": "QWS", "qws": "QWS", "Normal": "Normal"}.get(dataset)
    if ds is None or approach not in ("ML+2PN", "WOA", "PNLow", "PNHigh", "ML"):
        print("Please check the parameters!")                       # reference main.py:231
        return 1
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = configparser.RawConfigParser()
    cfg.optionxform = str
    cfg.read([os.path.join(here, "environment.ini"), "environment.ini"])
    if approach == "WOA":                                           # reference main.py:86-104
        if here not in sys.path:
            sys.path.insert(0, here)
        from gnnpn_sc_amd import WOA
        sec = cfg[f"{ds}-WOA"]
        args = argv[3:]
        seed = int(args[args.index("--seed") + 1]) if "--seed" in args else None
        pos = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] != "--seed")]
        epoch = int(pos[0]) if pos else int(sec["epoch"])
        reduct = float(sec["reduct"]) if ds == "Normal" else int(sec["reduct"])
        WOA.WOA(ds, int(sec["serCategory"]), int(sec["MLESWOAtest"]), int(sec["ML2PNWOATest"]), int(sec["MLWOATest"]),
                int(sec["ESWOAtest"]), int(sec["serviceNumber"]), reduct, epoch, int(sec["MAX_Iter"]), int(sec["popSize"]),
                seed=seed).start()
        return 0
    if approach == "ML":                                            # reference main.py:20-32: GNN training on the GPU
        if here not in sys.path:
            sys.path.insert(0, here)
        from gnnpn_sc_amd import trainML
        sec = cfg[f"{ds}-ML"]
        n_epochs = next((int(a.split("=")[1]) for a in argv[3:] if a.startswith("--epochs=")), None)
        trainML.TrainML(ds, int(sec["numLayersGIN"]), int(sec["numLayersGCN"]), int(sec["hiddenChannels"]),
                        int(sec["embeddingChannels"]), float(sec["dropout"]), float(sec["lr"]),
                        n_epochs or int(sec["epochs"])).start()
        return 0
    if approach in ("PNLow", "PNHigh"):                             # reference main.py:40-84: REINFORCE training on the GPU
        if here not in sys.path:
            sys.path.insert(0, here)
        from gnnpn_sc_amd import trainPNHigh
        sec = cfg[f"{ds}-{approach}"]
        pos = [a for a in argv[3:] if not a.startswith("--")]
        n_epochs = next((int(a.split("=")[1]) for a in argv[3:] if a.startswith("--epochs=")), None)
        common = (ds, int(sec["embeddingTag"]), int(sec["USE_CUDA"]), int(sec["serCategory"]), int(sec["epochDiv"]),
                  int(sec["serNumber"]), int(sec["hidden_size"]), int(sec["n_glimpses"]), float(sec["tanh_exploration"]),
                  int(sec["use_tanh"]), float(sec["beta"]), float(sec["max_grad_norm"]), float(sec["lr"]))
        epoch_ml = int(pos[0]) if pos else int(sec["epochML"])      # argv[3] overrides epochML (reference main.py:37-38,59-60)
        if approach == "PNLow":
            drv = trainPNHigh.PNLow(*common, epoch_ml)
            drv.start(**({"n_epochs": n_epochs} if n_epochs else {}))
        else:
            epoch_low = int(pos[1]) if len(pos) > 1 and epoch_ml != -1 else int(sec["epochPNLow"])   # :61-64
            drv = trainPNHigh.PNHigh(*common, epoch_ml, epoch_low)
            drv.start(**({"n_epochs": n_epochs} if n_epochs else {}))
        return 0
    sec = cfg[f"{ds}-ML+2PN"]
    flags = [a for a in argv[3:] if a.startswith("--")]
    pos = [a for a in argv[3:] if not a.startswith("--")]
    epoch = int(pos[0]) if pos else int(sec["epoch"])
    n_cat = int(sec["serviceCategory"])
    if here not in sys.path:
        sys.path.insert(0, here)
    from gnnpn_sc_amd import ML2PN
    if "--infer" in flags:
        import json
        with open(f"./data/{ds}/serviceFeature.data") as f:
            sf = json.load(f)
        n_services = sum(len(v) for v in sf.values())
        net, low, high, K = _models(cfg, ds, n_services, len(sf), epoch, "--random-init" in flags)
        ML2PN.infer(ds, net, low, high, K, epoch)
        n_cat = len(sf)
    ML2PN.check(ds, n_cat, epoch)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
