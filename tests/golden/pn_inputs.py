"""Seeded pointer-network inputs shared by the golden generators and the tests (so that a fixture at
BASELINE size stores a seed instead of tens of MB of input rows).  numpy's PCG64 ``default_rng`` stream
is stable across numpy versions by policy, and the GPU box runs this same image."""
import numpy as np
import torch


def pn_inputs(B, T, K, seed, dummy_every=0):
    """QWS-shaped PN input [B, T*K, 8]: q0,q1~U(0,1), q2,q3~U(.9,1); rows < K carry the request's
    global constraints in cols 4..7 (loadData.py:130-133); optional dummy rows [0,1,1,1]."""
    rng = np.random.default_rng(seed)
    x = np.zeros((B, T * K, 8), np.float32)
    x[:, :, 0:2] = rng.random((B, T * K, 2), dtype=np.float32)
    x[:, :, 2:4] = 0.9 + 0.1 * rng.random((B, T * K, 2), dtype=np.float32)
    lo = np.float32(0.9) ** np.float32(T) * np.float32(1.6)
    x[:, :K, 4:8] = np.array([lo, 1.0, lo, 1.0], np.float32)
    if dummy_every:
        for c in range(0, T, dummy_every):
            if c:
                x[:, c * K:(c + 1) * K, 0:4] = np.array([0, 1, 1, 1], np.float32)
    return torch.from_numpy(x)


def pn_inputs_chunked(B, T, K, seed, chunk):
    """The inputs of a BASELINE-size fixture: ``chunk`` problems at a time from consecutive seeds (the
    reference evaluates in batches of 128, trainPNHigh.py:248; generating per chunk keeps the fixture
    independent of how many chunks a test chooses to run)."""
    return torch.cat([pn_inputs(min(chunk, B - lo), T, K, seed + 1000 * (lo // chunk)) for lo in range(0, B, chunk)])
