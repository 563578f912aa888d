#!/usr/bin/env python3
"""Reference fixtures at BASELINE.json's sizes and shapes, produced by RUNNING THE REFERENCE ITSELF
(/root/reference/src/models/modelPN.py, imported from where it lies; build container only).

    python tests/golden/make_golden_big.py [name ...]      # default: all of CONFIGS

Each fixture pins the two-level greedy decode (trainPNHigh.py:138-139) of one BASELINE config:

  pn_big_qws512      configs[1]  T=47   K=5   B=512   (two batches of the bench shape B=256: >= 150 robust problems)
  pn_big_normal1024  configs[2]  T=50   K=10  B=1024
  pn_big_synth4      configs[3]  T=1000 K=5   L=5000   B=64 in chunks of 16 (the reference needs ~60 s per chunk here)
  pn_big_synth5      configs[4]  T=2000 K=10  L=20000  B=16 in chunks of 4 (its T x [B, L] lists take GBs per chunk)

Inputs and weights are NOT stored: they are regenerated from seeds (tests/golden/pn_inputs.py,
oracle.pn.make_state_dict).  Stored: picks of both levels, R, the decision margins (so a test can tell a
robust problem from a fragile one without the oracle), and the window logits of the first ``n_win`` problems.
The reference is run in chunks of 128 problems (its own batch size, trainPNHigh.py:248; smaller chunks at the two
long shapes, whose first chunk is the whole fixture of rounds 2-4); the oracle is run on the same chunks and must agree
bit for bit before anything is written.
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import pn as opn                      # noqa: E402
from pn_inputs import pn_inputs_chunked           # noqa: E402
import make_golden                                # noqa: E402

CHUNK = 128
CONFIGS = {
    "qws512": dict(T=47, K=5, B=512, seed=131, n_win=64),
    "normal1024": dict(T=50, K=10, B=1024, seed=141, n_win=32),
    "synth4": dict(T=1000, K=5, B=64, seed=151, n_win=2, chunk=16),
    "synth5": dict(T=2000, K=10, B=16, seed=161, n_win=1, chunk=4),
}


def gen(modelPN, name, T, K, B, seed, n_win, H=256, chunk=CHUNK):
    sd_low, sd_high = opn.make_state_dict(H, seed), opn.make_state_dict(H, seed + 1)
    L = T * K

    def build(level, sd):
        m = modelPN.CombinatorialRL(0, H, L, 0, 10, 1, modelPN.reward, "Dot", K, T, use_cuda=False, level=level)
        m.load_state_dict(sd, strict=True)
        return m.eval()

    low, high = build("Low", sd_low), build("High", sd_high)
    x_all = pn_inputs_chunked(B, T, K, seed + 2, chunk)
    acc = {k: [] for k in ("idx_low", "idx_high", "R", "margin_low", "margin_high", "win_low", "win_high")}
    t0 = time.time()
    for lo in range(0, B, chunk):
        x = x_all[lo:lo + chunk]
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            _, _, _, idx_low, latent = low(x, None, sample="greedy", training="SL")       # trainPNHigh.py:138
            R, _, actions, idx_high, logits_high = high(x, None, latent, sample="greedy")  # :139
        ref = {"idx_low": torch.stack(idx_low, 1), "idx_high": torch.stack(idx_high, 1), "R": R,
               "win_low": torch.stack([latent[k][:, k * K:(k + 1) * K] for k in range(T)], 1),
               "win_high": torch.stack([(logits_high[k] + latent[k])[:, k * K:(k + 1) * K] for k in range(T)], 1)}
        del latent, logits_high
        orc = opn.two_level_greedy(sd_low, sd_high, x, T, K)
        for key in ("idx_low", "idx_high", "R", "win_low", "win_high"):
            assert torch.equal(ref[key], orc[key]), f"oracle != reference on {key} ({name}, chunk at {lo})"
        for key in acc:
            acc[key].append((ref[key] if key in ref else orc[key]).numpy())
        del orc
        print(f"  {name}: {min(B, lo + chunk)}/{B} problems, {time.time() - t0:.0f} s", flush=True)
    out = {k: np.concatenate(v) for k, v in acc.items()}
    idx_t = np.int16 if L < 32768 else np.int32
    robust = (out["margin_low"] > 5e-4).all(1) & (out["margin_high"] > 5e-4).all(1)
    np.savez_compressed(
        os.path.join(HERE, f"pn_big_{name}.npz"), hidden=H, n_cat=T, n_per=K, B=B, chunk=chunk, seed_low=seed,
        seed_high=seed + 1, seed_inputs=seed + 2, idx_low=out["idx_low"].astype(idx_t),
        idx_high=out["idx_high"].astype(idx_t), R=out["R"], margin_low=out["margin_low"].astype(np.float32),
        margin_high=out["margin_high"].astype(np.float32), win_low=out["win_low"][:n_win],
        win_high=out["win_high"][:n_win])
    print(f"pn_big_{name}: B={B} T={T} K={K}: {int(robust.sum())} robust problems (every margin > 5e-4), "
          f"min margin {min(float(out['margin_low'].min()), float(out['margin_high'].min())):.2e}, "
          f"{os.path.getsize(os.path.join(HERE, f'pn_big_{name}.npz')) / 1e6:.2f} MB")


def main():
    import signal
    signal.alarm(3600)
    modelPN, _, _, _ = make_golden.import_reference()
    torch.set_num_threads(8)                    # picks are pinned against the oracle on the same thread count, same chunks
    for name in (sys.argv[1:] or list(CONFIGS)):
        gen(modelPN, name, **CONFIGS[name])
    import manifest
    manifest.update()                           # sha256 + key lists of every fixture: what the CPU suite holds the committed files to


if __name__ == "__main__":
    main()
