"""CPU restatement (numpy) of the exact three-piece fp16 operand split of GNNPN_PREC_SPLIT
(gnnpn-sc_amd/csrc/coop_common.h: split3 / split_weights / split_chain).  TEST INFRASTRUCTURE ONLY.

The reference multiplies W_hh and h in fp32 (nn.LSTM, /root/reference/src/models/modelPN.py:157-158,191,205); the split
precision must give that product from operands that lose no bit:
    x 2^s = p0 + p1 / 2^11 + p2 / 2^22       (three fp16 numbers; s parks x high in fp16's range)
    x y 2^(s+t) = p0 q0 + (p0 q1 + p1 q0)/2^11 + (p1 q1 + p0 q2 + p2 q0)/2^22  + [terms <= 2^-32 |x y|, dropped]
"""
import numpy as np

SCALE = np.float32(2048.0)
H_SCALE_LOG2 = 15


def pieces(x, scale_log2=0):
    """-> (p0, p1, p2) float16 arrays of x * 2^scale_log2 (round-to-nearest-even conversions, exact fp32 residuals)."""
    xs = (np.asarray(x, dtype=np.float32) * np.float32(2.0) ** np.float32(scale_log2)).astype(np.float32)
    with np.errstate(over="ignore"):
        p0 = xs.astype(np.float16)
        r1 = ((xs - p0.astype(np.float32)) * SCALE).astype(np.float32)
        p1 = r1.astype(np.float16)
        r2 = ((r1 - p1.astype(np.float32)) * SCALE).astype(np.float32)
        p2 = r2.astype(np.float16)
    return p0, p1, p2


def recompose(p0, p1, p2):
    """p0 + p1/2^11 + p2/2^22 in float64 (exact: three 11-bit numbers whose exponents span < 53 bits)."""
    return p0.astype(np.float64) + p1.astype(np.float64) / 2048.0 + p2.astype(np.float64) / 4194304.0


def column_scale_log2(w_col):
    """s of split_weights: the column's largest |w| times 2^s lies in [2^14, 2^15); clamped to +-96; 0 for a zero column."""
    m = float(np.max(np.abs(np.asarray(w_col, dtype=np.float32))))
    if not (m > 0.0) or not np.isfinite(m):
        return 0
    _, e = np.frexp(np.float32(m))
    return int(min(96, max(-96, 15 - int(e))))


def third_piece_byte(p2):
    """What the kernels keep of the third WEIGHT piece: the upper byte of the fp16 (an e5m2 number)."""
    bits = np.asarray(p2, dtype=np.float16).view(np.uint16)
    return ((bits >> 8).astype(np.uint16) << 8).view(np.float16)


def recurrent_product(W, h):
    """gates [R, 4H] = h [R, H] . W^T as the split precision defines it, with EXACT accumulation (float64 over exact
    fp16 x fp16 products) — i.e. the part of the GPU result that does not depend on the matrix core's accumulator:
    six kept products per (row, column, k), per-column power-of-two scaling, the third weight piece as its upper byte."""
    W = np.asarray(W, dtype=np.float32)
    h = np.asarray(h, dtype=np.float32)
    a = [p.astype(np.float64) for p in pieces(h, H_SCALE_LOG2)]                    # [R, H] each
    out = np.empty((h.shape[0], W.shape[0]), dtype=np.float64)
    for col in range(W.shape[0]):
        s = column_scale_log2(W[col])
        q0, q1, q2 = pieces(W[col], s)
        q2 = third_piece_byte(q2)
        q = [q0.astype(np.float64), q1.astype(np.float64), q2.astype(np.float64)]
        a0 = a[0] @ q[0]
        a1 = a[0] @ q[1] + a[1] @ q[0]
        a2 = a[1] @ q[1] + a[0] @ q[2] + a[2] @ q[0]
        out[:, col] = (a0 + a1 / 2048.0 + a2 / 4194304.0) * 2.0 ** (-(15 + s))
    return out
