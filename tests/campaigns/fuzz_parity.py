"""Fuzz campaign: random shapes / seeds, HIP two-level decode (all precisions and decoder forms) against the CPU
oracle run live, judged by the parity rule of tests/parity.py.  Usage: fuzz_parity.py [n_configs] [seed]"""
import sys, os, random, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import pn as opn                      # checker use only
from parity import LOGIT_ATOL, R_ATOL, assert_index_parity, robust_problems
from gnnpn_sc_amd import ops
from gnnpn_sc_amd.modelPN import CombinatorialRL, reward, two_level_greedy
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
torch.set_num_threads(16)
fails, t0 = 0, time.time()
for c in range(n_cfg):
    B, T, K = rng.choice([1, 2, 15, 16, 17, 31, 33, 48]), rng.randint(1, 14), rng.choice([1, 2, 3, 5, 8, 10, 16])
    sl, sh, sx = rng.randrange(1 << 20), rng.randrange(1 << 20), rng.randrange(1 << 20)
    nets = []
    for level, seed in (("Low", sl), ("High", sh)):
        m = CombinatorialRL(0, 256, T * K, 0, 10, 1, reward, "Dot", K, T, use_cuda=True, level=level)
        m.load_state_dict(opn.make_state_dict(256, seed), strict=True)
        nets.append(m.to(dev).eval())
    low, high = nets
    g = torch.Generator().manual_seed(sx)
    x = torch.rand(B, T * K, 8, generator=g)
    x[:, :, 2:4] = 0.9 + 0.1 * x[:, :, 2:4]
    x[:, K:, 4:] = 0
    if rng.random() < 0.3 and T > 1:              # dummy rows for an absent category (loadData.py:148)
        j = rng.randrange(T)
        x[:, j * K:(j + 1) * K, :4] = torch.tensor([0., 1., 1., 1.])
    ref = opn.two_level_greedy(opn.make_state_dict(256, sl), opn.make_state_dict(256, sh), x, T, K)
    robust = robust_problems(ref["margin_low"], ref["margin_high"])
    for prec in ("f32", "split"):
        for impl in (2, 3, 4):
            try:
                out = two_level_greedy(low, high, x.to(dev), precision=prec, decode_impl=impl)
                ops.check_status(dev)
                s = assert_index_parity(out["idx_high"], ref["idx_high"], robust, "high", 0.0, x) & \
                    assert_index_parity(out["idx_low"], ref["idx_low"], robust, "low", 0.0, x)
                if bool(s.any()):
                    sn = s.numpy()
                    assert float((out["win_low"].cpu()[s] - ref["win_low"][s]).abs().max()) < LOGIT_ATOL
                    assert float((out["R"].cpu()[s] - ref["R"][s]).abs().max()) <= R_ATOL, "R"
            except Exception as e:                # noqa: BLE001
                fails += 1
                print(f"FAIL B={B} T={T} K={K} seeds=({sl},{sh},{sx}) precision={prec} decode_impl={impl}: {e}")
    print(f"cfg {c}: B={B} T={T} K={K} robust {int(robust.sum())}/{B} ok", flush=True)
print(f"{n_cfg} configs x 6 variants, {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
