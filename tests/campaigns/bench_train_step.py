"""Time of one REINFORCE training step of the High pointer network (trainPNHigh.py:81-110) at the reference's batch size
(128, :248) on the QWS / Normal shapes, with the breakdown sampled-forward / backward / optimiser, next to the CPU oracle
(torch autograd restatement of the same step).   python tests/campaigns/bench_train_step.py [--cpu]   (under tests/: it times the oracle beside the HIP path)"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from pn_inputs import pn_inputs
from oracle import pn as opn, pn_train as optr
from gnnpn_sc_amd.modelPN import CombinatorialRL, reward, two_level_greedy
from gnnpn_sc_amd.trainPNHigh import TrainModel, actor_gradients
dev = torch.device("cuda:0")
for name, T, K in (("qws", 47, 5), ("normal", 50, 10)):
    B, H = 128, 256
    nets = []
    for lvl, seed in (("Low", 1), ("High", 2)):
        m = CombinatorialRL(0, H, T * K, 0, 10, 1, reward, "Dot", K, T, level=lvl)
        m.load_state_dict(opn.make_state_dict(H, seed))
        nets.append(m.to(dev))
    low, high = nets
    x = pn_inputs(B, T, K, 3)
    tm = TrainModel(high, None, None, 1, 0.9, True, "QWS", T, batch_size=B, low_model=low, device=str(dev))
    for _ in range(2):
        tm.train_step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        tm.train_step(x)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / n * 1e3
    xd = x.to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = two_level_greedy(low, high, xd, sample_high_seed=5)
    torch.cuda.synchronize(); fwd_ms = (time.perf_counter() - t0) / n * 1e3
    g = (out["R"] / B).contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        actor_gradients(high.actor, xd, out["idx_high"], out["win_low"], g)
    torch.cuda.synchronize(); bwd_ms = (time.perf_counter() - t0) / n * 1e3
    rec = {"shape": name, "B": B, "T": T, "K": K, "train_step_ms": round(step_ms, 2), "sampled_forward_ms": round(fwd_ms, 2),
           "teacher_forced_forward_plus_backward_ms": round(bwd_ms, 2), "problems_per_s": round(B / step_ms * 1e3, 1)}
    if "--cpu" in sys.argv:
        torch.set_num_threads(16)
        sd_low, sd_high = opn.make_state_dict(H, 1), opn.make_state_dict(H, 2)
        t0 = time.perf_counter()
        optr.train_step(sd_low, sd_high, x, T, K, 5)
        rec["cpu_oracle_step_ms_16_threads"] = round((time.perf_counter() - t0) * 1e3, 1)
    print(json.dumps(rec), flush=True)
