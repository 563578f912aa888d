"""Oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py): the plan of the tiled CSR aggregate, restated in numpy.

What it pins.  The batched GCN aggregate (`GCNConv.propagate` over B block-diagonal copies of the service graph,
/root/reference/src/models/modelML.py:152-155 with the batching of src/models/trainML.py:109-114) sums every destination row's
messages in EDGE ORDER (torch_scatter's scatter_add over the edge list; the edge list itself is emitted row by row by the
co-occurrence scan of src/loadData.py:42-65, self loops appended by add_remaining_self_loops).  The device kernel
gnnpn_csr_aggregate_tiled_f32 walks the SOURCE rows tile by tile, which keeps that order only for rows whose neighbour lists
visit the source tiles in non-decreasing order (a trailing self loop excepted).  Everything index-valued that decides this —
tile geometry, the per-row validity rule, the per-(row, tile) runs, the order of the rows in their units, the quads per
(unit, tile), the sliced-ELL stream itself — is integer work and is restated here independently of the HIP kernels
(csrc/graph_tiled.hip: tile_plan_rows / _scan / _fill), so that tests/test_gpu_ops.py can hold the plan the GPU builds to
BIT equality with this one.  Constants = the kernel's tiling (16 wavefronts, 16 rows per unit, 4 entries per quad)."""
import numpy as np

WAVES, TILE_ROWS_MAX, PASSES_MAX, SRC_TILES_MAX, HIST_BINS = 16, 2559, 10, 8, 64
DST_ROWS_MAX = WAVES * 16 * PASSES_MAX
NAN_BITS = 0x7FC00000


def geometry(n_rows, block_rows):
    """Tile geometry of a block-local graph (None: more than 8 source tiles per block, or empty)."""
    if n_rows <= 0 or block_rows <= 0:
        return None
    g = {"R": block_rows, "n_blocks": -(-n_rows // block_rows)}
    g["NT"] = -(-block_rows // TILE_ROWS_MAX)
    g["TR"] = -(-block_rows // g["NT"])
    g["ND"] = -(-block_rows // DST_ROWS_MAX)
    g["DR"] = (-(-block_rows // g["ND"]) + 15) // 16 * 16
    g["U"] = g["DR"] // 16
    g["passes"] = -(-g["U"] // WAVES)
    return g if g["NT"] <= SRC_TILES_MAX else None


def build(rowptr, col, w, n_rows, block_rows):
    """rowptr [n+1], col [E] int, w [E] float32 or None -> dict(valid, geometry, header [items, 2], order, selfw_bits, batches
    (uint32 words of the stream, 128 per quad), stats).  `items` run over (block, destination tile, source tile, unit) in that
    order — the order the quads lie in the stream."""
    g = geometry(n_rows, block_rows)
    if g is None:
        return {"valid": False, "geometry": None}
    rowptr, col = np.asarray(rowptr, np.int64), np.asarray(col, np.int64)
    wbits = None if w is None else np.asarray(w, np.float32).view(np.uint32)
    one = np.float32(1.0).view(np.uint32)
    NT, TR, ND, DR, U, R = g["NT"], g["TR"], g["ND"], g["DR"], g["U"], g["R"]
    n_bd = g["n_blocks"] * ND
    order = np.full((n_bd, U * 16), -1, np.int32)
    selfw = np.full((n_bd, U * 16), NAN_BITS, np.uint32)
    nq = np.zeros((n_bd, NT, U), np.int64)
    tstart = {}
    hist = np.zeros(HIST_BINS, np.int64)
    edges = invalid = stream_rows = 0
    for bd in range(n_bd):
        b, d = divmod(bd, ND)
        r0 = b * R
        Rb = min(R, n_rows - r0)
        rows_d = max(0, min(DR, Rb - d * DR))
        stream_rows += rows_d
        keys, runs_of, sw_of = [], {}, {}
        for i in range(rows_d):
            r = r0 + d * DR + i
            e0, e1 = int(rowptr[r]), int(rowptr[r + 1])
            sw = NAN_BITS
            if e1 > e0 and col[e1 - 1] == r and (r - r0) // TR != NT - 1:      # trailing self loop out of tile order: the epilogue's
                sw = int(wbits[e1 - 1]) if wbits is not None else int(one)
                e1 -= 1
            run, ok, tprev = [0] * NT, True, 0
            for e in range(e0, e1):
                c = int(col[e]) - r0
                if c < 0 or c >= Rb:
                    ok = False
                    break
                t = c // TR
                ok = ok and t >= tprev
                tprev = max(tprev, t)
                run[t] += 1
            ts = [e0]
            for t in range(NT):
                ok = ok and run[t] <= 0xFFFF
                ts.append(ts[-1] + run[t])
                hist[min(run[t], HIST_BINS - 1)] += 1
            tstart[r] = ts
            edges += sum(run)
            invalid += 0 if ok else 1
            key = 0
            for t in range(NT):
                key |= min((min(run[t], 0xFFFF) + 3) >> 2, 255) << (8 * (7 - t))
            keys.append((-key, i))
            runs_of[i], sw_of[i] = run, sw
        keys.sort()                                         # key descending, ties by row ascending
        for pos, (_, i) in enumerate(keys):
            order[bd, pos] = d * DR + i
            selfw[bd, pos] = sw_of[i]
        for u in range(U):
            ids = [keys[u * 16 + j][1] for j in range(16) if u * 16 + j < len(keys)]
            for t in range(NT):
                nq[bd, t, u] = max([(min(runs_of[i][t], 0xFFFF) + 3) >> 2 for i in ids], default=0)
    flat = nq.reshape(-1)
    first = np.concatenate([[0], np.cumsum(flat)[:-1]]).astype(np.int64)
    quads = int(flat.sum())
    header = np.stack([first, flat], 1).astype(np.int32)
    stats = {"invalid_rows": invalid, "quads": quads, "edges": edges, "slots": int(flat.sum() * 64), "rows": stream_rows,
             "run_histogram": hist.tolist()}
    out = {"valid": invalid == 0, "geometry": g, "header": header, "order": order.reshape(-1), "selfw_bits": selfw.reshape(-1),
           "stats": stats}
    if invalid:
        return out
    batches = np.zeros(((quads + 4) * 128,), np.uint32)      # 512 bytes per quad: [16 rows][4] offsets, then [16 rows][4] weights; four quads of slack
    zero_off = TR * 64
    item = 0
    for bd in range(n_bd):
        r0 = (bd // ND) * R
        for t in range(NT):
            for u in range(U):
                f, n = int(first[item]), int(flat[item])
                item += 1
                for j in range(16):
                    rl = int(order[bd, u * 16 + j])
                    e_lo, e_hi = (tstart[r0 + rl][t], tstart[r0 + rl][t + 1]) if rl >= 0 else (0, 0)
                    for q in range(n):
                        base = (f + q) * 128
                        for m in range(4):
                            e = e_lo + 4 * q + m
                            if e < e_hi:
                                batches[base + 4 * j + m] = (int(col[e]) - r0 - t * TR) * 64
                                batches[base + 64 + 4 * j + m] = int(wbits[e]) if wbits is not None else int(one)
                            else:
                                batches[base + 4 * j + m] = zero_off
    out["batches"] = batches
    return out
