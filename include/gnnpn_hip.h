/*
 * gnnpn_hip.h — C ABI of libgnnpn_hip.so: the MI355X (gfx950) implementation of the ML+2PN
 * inference hot path of wangxiaohit/GNNPN-SC.
 *
 * The reference is pure Python on stock PyTorch/PyG ops and has no FFI layer; these entry points
 * are what a binding for this path would bind.  Each one names the reference code it replaces
 * (paths relative to the reference repository root).  Conventions:
 *
 *   - every pointer is a DEVICE pointer into caller-owned, contiguous memory (borrowed for the
 *     duration of the call, never retained); sizes are element counts unless stated;
 *   - float = IEEE fp32, indices = int32 unless stated; row-major everywhere;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued
 *     on it and the call returns without synchronising;
 *   - a zero-sized batch (B, M, n_rows, n_graphs ... = 0) is valid: the call returns 0 at once, launches nothing,
 *     and the buffers of that batch may be NULL (what an empty torch tensor's data_ptr() is);
 *   - return 0 on success, a negative GNNPN_E_* code on failure (nothing was enqueued);
 *     gnnpn_last_error() returns a thread-local message for the last failure.
 *   - no internal host threads, no hidden allocation: workspaces are explicit arguments.
 */
#ifndef GNNPN_HIP_H
#define GNNPN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNPN_ABI_VERSION 9   /* 9: gnnpn_lstm_pack_split_weights_f32 / gnnpn_lstm_split_weights_bytes and the whh_split field (recurrent weights split once per model) at the END of the encoder's and the decoder's net structs, sticky_status words 1-2 (cumulative declined / off-canonical seats); 8: gnnpn_decode_attn_train_{forward,backward}_f32 (training through 'Bahdanau' attention / glimpse rounds), gnnpn_attention_logits_bahdanau_f32; 7: gnnpn_launch_opts_t.sticky_status is a block of GNNPN_STATUS_WORDS words (proof-of-work counters), the 16-member forms (impl 3) are gone; 6: gnnpn_gate_wait; 5: the exact-split GIN layer (gnnpn_pack_split_weights_f16, gnnpn_gin_layer_split); 4: the tiled aggregate (gnnpn_csr_tile_plan_*, gnnpn_csr_aggregate_tiled_f32); 3: row order of the blocks form */

#define GNNPN_OK 0
#define GNNPN_E_ARG (-1)     /* null pointer / bad size / misaligned */
#define GNNPN_E_UNSUP (-2)   /* shape outside what the kernels are built for */
#define GNNPN_E_LAUNCH (-3)  /* HIP reported a launch error */

#define GNNPN_ACT_NONE 0
#define GNNPN_ACT_RELU 1
#define GNNPN_ACT_SIGMOID 2

int gnnpn_abi_version(void);
const char* gnnpn_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Dense layer: C[M,N] = act( (A[M,K] . W[N,K]^T + bias[N]) * scale[N] + shift[N] ), fp32 MFMA.
 * bias / scale / shift may be NULL.  scale/shift carry an eval-mode BatchNorm1d folded to
 * y = x*alpha + beta (alpha = gamma/sqrt(var+eps), beta = b - mean*alpha).
 * Replaces: torch.nn.Linear / BatchNorm1d / ReLU / Sigmoid chains at
 *   src/models/modelML.py:77-90 (GIN MLP), :141,:154 (BN+ReLU), :164-165 (serviceLin, nodeLin),
 *   :173-176 (score matmul + sigmoid), torch.matmul(x, weight) inside GCNConv (:153),
 *   src/models/modelPN.py:190 (embedding2) and the LSTM input projection of :191.
 * lda/ldw/ldc are row strides in elements.
 */
int gnnpn_linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                     const float* scale, const float* shift, int act, float* C, int64_t ldc,
                     int64_t M, int N, int K, void* stream);

/* Embedding lookup + concat: out[n, 0:emb] = table[(int)x[n,0], :], out[n, emb:emb+nfeat] =
 * x[n, 1:1+nfeat].  x is [n_rows, 1+nfeat], out is [n_rows, emb+nfeat].  Category ids outside
 * [0, vocab) make the call fail on the device side by writing NaN rows (checked by callers).
 * Replaces NodeEncoder.forward + torch.cat at src/models/modelML.py:22-29,133-137,145-149. */
int gnnpn_embed_concat_f32(const float* x, const float* table, int vocab, int emb, int nfeat,
                           float* out, int64_t n_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CSR gather-aggregate (one wavefront per destination row, neighbour list read coalesced):
 *   y[i,:] = epi( sum_{e in [rowptr[i], rowptr[i+1])} w[e] * x[col[e],:]  (+ self_coef * x[i,:]) )
 * summed strictly in CSR order with separate multiply and add (the order and rounding of
 * scatter_add over materialised messages).  w == NULL means weight 1 (no multiply).
 * self_coef: pointer to ONE device float added as (1 + *self_coef) * x[i,:] AFTER the neighbour
 * sum, or NULL.  epi: + bias[C] (or NULL), then * scale + shift (or NULL), then act.
 * Replaces: GINConv.propagate + "(1+eps)*x_r" and GCNConv.propagate + bias (torch_geometric 1.7.0;
 * call sites src/models/modelML.py:140,153) fused with BatchNorm1d+ReLU (:154-155).
 */
int gnnpn_csr_aggregate_f32(const int32_t* rowptr, const int32_t* col, const float* w,
                            const float* x, int64_t ldx, const float* self_coef, const float* bias,
                            const float* scale, const float* shift, int act, float* y, int64_t ldy,
                            int32_t n_rows, int32_t C, void* stream);

/* The same aggregate for BLOCK-LOCAL graphs with the node features staged in LDS: the caller promises that every edge
 * whose destination row lies in block b = [b*block_rows, (b+1)*block_rows) has its source row in the same block — a batch of
 * B service graphs (B block-diagonal copies of the table, src/models/trainML.py:109-114, modelML.py:145-156) with
 * block_rows = S, or a single graph with block_rows = n_rows.  One workgroup per (block, channel slice) copies the block's
 * slice into LDS once and serves all gathers from there (each source element crosses L2 -> CU once instead of once per
 * neighbour); same CSR-order sums, same epilogue, bit-identical results.
 * row_order: NULL, or [n_rows] from gnnpn_csr_block_row_order — the order in which a block's rows are dealt to the
 * wavefronts (speed only: rows that advance in step then have the same number of edges; the results do not depend on it).
 * GNNPN_E_UNSUP (nothing enqueued) when a block does not fit: (block_rows + 1) * 16 B > 160 KB (block_rows > 10239), or rows
 * not 16-byte aligned; callers then use gnnpn_csr_aggregate_f32.
 * Replaces: GCNConv.propagate over the batched service graph, src/models/modelML.py:153. */
int gnnpn_csr_aggregate_blocks_f32(const int32_t* rowptr, const int32_t* col, const float* w,
                                   const float* x, int64_t ldx, const float* self_coef, const float* bias,
                                   const float* scale, const float* shift, int act, float* y, int64_t ldy,
                                   int32_t n_rows, int32_t C, int32_t block_rows, const int32_t* row_order, void* stream);

/* row_order[b*block_rows + i] = the global number of the row of block b with the i-th most edges (ties: lower row first).
 * A property of the graph (one launch per graph, reused by every layer and call); block_rows <= 16384, else GNNPN_E_UNSUP.
 * No counterpart in the reference: scheduling input of gnnpn_csr_aggregate_blocks_f32. */
int gnnpn_csr_block_row_order(const int32_t* rowptr, int32_t n_rows, int32_t block_rows, int32_t* row_order, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The aggregate for block-local graphs of ANY block size up to 20472 rows, node features staged in LDS one SOURCE TILE at a
 * time (destination tile x source tile; csrc/graph_tiled.hip).  A workgroup owns (block, destination tile of <= 2560 rows,
 * 16-channel slice), keeps its destination rows' running sums in registers and walks the block's source tiles (<= 2559 rows)
 * in order.  Valid — and bit-identical to gnnpn_csr_aggregate_f32 — for graphs in which every row's neighbour list visits
 * the source tiles in non-decreasing order, an optional trailing self loop excepted (the reference's service graph:
 * src/loadData.py:56-65 emits the pairs in lexicographic order, add_remaining_self_loops appends the loops; the check is
 * part of the plan).  The edge lists are consumed from a PLAN built once per (graph, weights): a sliced-ELL stream in the
 * order the wavefronts read it (quads of 512 B: [16 rows][4] LDS offsets + [16 rows][4] weights).
 * The kernel walks a unit's quads four at a time, or two at a time with two requests in flight for blocks of three or more source
 * tiles (short runs per tile); same bits.  GNNPN_TILED_WALK=quads|pairs in the environment overrides the choice (test hook).
 *
 *   1. gnnpn_csr_tile_plan_geometry: tile counts and the byte sizes of the plan's arrays (host only).
 *   2. gnnpn_csr_tile_plan_rows: fills header / order / tstart / selfw / meta (device).  Read meta back (stream-ordered):
 *        meta[0] = rows that break the rule above (0: the graph qualifies; otherwise use gnnpn_csr_aggregate_f32),
 *        meta[1] = quads in the stream (allocate (meta[1] + 4) * 512 bytes: the kernel's fixed-shape loads read up to four
 *        quads past the last one — an empty unit at the end of the scan order starts there), meta[2] = edges in the stream, meta[3] = (row, edge) slots the stream holds = 64 * meta[1]
 *        (efficiency = meta[2] / meta[3]), meta[4] = rows, meta[8..72) = histogram of the per-(row, source tile) run lengths
 *        (last bin: >= 63).
 *   3. gnnpn_csr_tile_plan_fill: writes the stream (w == NULL: weight 1).
 *   4. gnnpn_csr_aggregate_tiled_f32 per layer: same epilogue arguments as gnnpn_csr_aggregate_f32; C a multiple of 16.
 * GNNPN_E_UNSUP (nothing enqueued) where the form does not apply.  Replaces: GCNConv.propagate over the batched service
 * graph, src/models/modelML.py:152-155 (torch_geometric 1.7.0). */
typedef struct gnnpn_tile_plan_geom {
    int32_t n_blocks;       /* ceil(n_rows / block_rows) */
    int32_t src_tiles;      /* source tiles per block, <= 8 */
    int32_t src_tile_rows;  /* rows per source tile, <= 2559: a 16-channel slice of a tile and one all-zero row fill a CU's 160 KB of LDS */
    int32_t dst_tiles;      /* destination tiles per block */
    int32_t dst_tile_rows;  /* rows per destination tile, a multiple of 16, <= 2560 */
    int32_t units;          /* dst_tile_rows / 16: groups of 16 rows that walk their edge lists in step (one wavefront) */
    int32_t wavefronts;     /* per workgroup: 16 */
    int32_t passes;         /* units per wavefront and source tile, <= 10 */
    int64_t header_bytes;   /* int32[2] per (block, destination tile, source tile, unit): first quad, quads of 4 edges per row */
    int64_t order_bytes;    /* int32 per (block, destination tile, position): block-local destination row, -1 = none */
    int64_t tstart_bytes;   /* int32 per (row, source tile + 1): the row's first edge in every source tile */
    int64_t selfw_bytes;    /* float per (block, destination tile, position), like order: weight of the row's trailing self loop where the epilogue adds it, NaN = nothing to add */
    int64_t meta_bytes;     /* 512 */
} gnnpn_tile_plan_geom_t;
int gnnpn_csr_tile_plan_geometry(int32_t n_rows, int32_t block_rows, gnnpn_tile_plan_geom_t* out);
int gnnpn_csr_tile_plan_rows(const int32_t* rowptr, const int32_t* col, const float* w, int32_t n_rows, int32_t block_rows,
                             int32_t* header, int32_t* order, int32_t* tstart, float* selfw, int32_t* meta, void* stream);
int gnnpn_csr_tile_plan_fill(const int32_t* col, const float* w, int32_t n_rows, int32_t block_rows, const int32_t* header,
                             const int32_t* order, const int32_t* tstart, void* batches, int64_t n_quads, void* stream);
int gnnpn_csr_aggregate_tiled_f32(const int32_t* header, const int32_t* order, const float* selfw, const void* batches,
                                  const float* x, int64_t ldx, const float* self_coef, const float* bias,
                                  const float* scale, const float* shift, int act, float* y, int64_t ldy,
                                  int32_t n_rows, int32_t C, int32_t block_rows, void* stream);

/* GCN symmetric normalisation on a destination-major CSR that already contains one self-loop
 * entry per node (add_remaining_self_loops, fill 1): deg[i] = sum of w_raw over row i (CSR order),
 * dis = deg^-1/2 (inf -> 0), norm[e] = dis[src[e]] * w_raw[e] * dis[dst[e]].
 * deg_inv_sqrt is an [n] workspace/output.  Replaces gcn_norm of torch_geometric 1.7.0 invoked
 * by GCNConv.forward (call site src/models/modelML.py:153). */
int gnnpn_gcn_norm_f32(const int32_t* rowptr, const int32_t* col, const float* w_raw,
                       float* deg_inv_sqrt, float* norm, int32_t n_rows, void* stream);

/* Segment mean over contiguous segments: out[g,:] = sum_{n in [segptr[g], segptr[g+1])} x[n,:]
 * / max(count,1).  Replaces torch_scatter.scatter(reduce='mean') at src/models/modelML.py:166. */
int gnnpn_segment_mean_f32(const int32_t* segptr, const float* x, int64_t ldx, float* out,
                           int64_t ldo, int32_t n_seg, int32_t C, void* stream);

/* One GIN layer of the workflow branch for LARGE graphs in one launch: neighbour aggregate ((1 + *eps) * x_i + sum_j x_j, CSR
 * order) -> Linear(c_in -> h1) + BN + ReLU -> Linear(h1 -> h2) + BN + ReLU [-> Linear(h2 -> h3) + bias when w3 != NULL: nodeLin
 * behind the last layer]; a workgroup owns 32 rows and the [rows x h1] intermediate stays in LDS (csrc/gin_layer.hip).  Built
 * for h1 = 256, h2 = 128, h3 = 128, c_in <= 256 (GNNPN_E_UNSUP otherwise: callers use the separate kernels).  w1 / w2 / w3: the
 * weights as v_mfma_f32_32x32x2_f32 B-fragments, packed[t][kp][lane] = W[32 t + lane % 32][2 kp + lane / 32] with K zero-padded
 * to a multiple of 32 (ops.pack_mfma_b32: layout only); BN folded to scale / shift (or NULL).  Stage for stage the arithmetic of
 * gnnpn_csr_aggregate_f32 (self_coef = eps) and gnnpn_linear_f32: the result is bit-identical to calling those in sequence.
 * Replaces: GINConv + BatchNorm1d + ReLU (src/models/modelML.py:75-93,139-143) and nodeLin (:165) per layer. */
int gnnpn_gin_layer_f32(const int32_t* rowptr, const int32_t* col, const float* x, int64_t ldx, int32_t c_in, const float* eps,
                        const float* w1, const float* b1, const float* bn1_scale, const float* bn1_shift, int32_t h1,
                        const float* w2, const float* b2, const float* bn2_scale, const float* bn2_shift, int32_t h2,
                        const float* w3, const float* b3, int32_t h3, float* out, int64_t ldo, int64_t n_rows, void* stream);

/* The same layer with its dense products on the fp16 matrix cores through the EXACT SPLIT (csrc/gin_layer_split.hip; the
 * arithmetic of the "split" precision of the recurrent kernels): every fp32 operand, activation or weight, is decomposed into three
 * fp16 pieces that sum to it bit for bit and the six cross products that can reach 2^-24 of a term are accumulated in fp32; the
 * aggregate, bias, BN and ReLU are the fp32 instructions of gnnpn_gin_layer_f32.  NOT bit-identical to the fp32 layer (another
 * accumulation order of the same terms): tests/test_gpu_ops.py measures both against fp64.
 *   gnnpn_split_weights_bytes(n_out, k): size of the packed image of an [n_out x k] weight (0 for shapes not supported).
 *   gnnpn_pack_split_weights_f16: w [n_out x k] row-major fp32 (n_out a multiple of 16) -> packed (16-byte aligned) + col_inv
 *     [n_out]: per output column a power-of-two scale (column maximum in [2^14, 2^15)), records (column tile of 16, k-block of
 *     32) = {piece 0: 64 lanes x 16 B, piece 1: 64 x 16 B, piece 2: 64 x 8 B (upper bytes)} in v_mfma_f32_16x16x32_f16 B-fragment
 *     order, lane (c, kq) holding W[16 t + c][32 kk + 8 kq + j]; col_inv[c] = 2^-scale.  Once per model.
 *   gnnpn_gin_layer_split: as gnnpn_gin_layer_f32 with (w, inv) pairs from the packer; x rows of 4 k channels must be 16-byte
 *     aligned, or c_in <= 32; c_in <= 128; 16-byte aligned vectors and output rows (GNNPN_E_UNSUP otherwise).
 * Replaces the same reference lines as gnnpn_gin_layer_f32. */
int64_t gnnpn_split_weights_bytes(int32_t n_out, int32_t k);
int gnnpn_pack_split_weights_f16(const float* w, int64_t ldw, int32_t n_out, int32_t k, void* packed, float* col_inv, void* stream);
int gnnpn_gin_layer_split(const int32_t* rowptr, const int32_t* col, const float* x, int64_t ldx, int32_t c_in, const float* eps,
                          const void* w1, const float* inv1, const float* b1, const float* bn1_scale, const float* bn1_shift, int32_t h1,
                          const void* w2, const float* inv2, const float* b2, const float* bn2_scale, const float* bn2_shift, int32_t h2,
                          const void* w3, const float* inv3, const float* b3, int32_t h3, float* out, int64_t ldo, int64_t n_rows,
                          void* stream);

/* The whole workflow (GIN) branch of Net.forward in ONE launch, for batches whose workflow graphs have at most 16 nodes
 * (QWS / Normal requests: <= 11): embedding lookup + concat, n_layers x {GIN aggregate, Linear+BN+ReLU, Linear+BN+ReLU},
 * nodeLin, mean over each graph's nodes; node features stay in LDS between the stages.  Stage for stage the arithmetic of
 * gnnpn_embed_concat_f32 / gnnpn_csr_aggregate_f32 / gnnpn_linear_f32 / gnnpn_segment_mean_f32, so the result is
 * bit-identical to calling those in sequence.
 *   x [N, 1+nfeat], table [vocab, emb]; rowptr/col: destination-major CSR of the batched workflow graphs, every edge
 *   inside its graph; seg_ptr [n_graphs+1]; max_nodes = the largest graph (host value);
 *   per layer (gnnpn_gin_layer_t): weights as v_mfma_f32_16x16x4_f32 B-fragments, packed[t][k16][lane][j] =
 *   W[16t + lane%16][16*k16 + 4j + lane/16] with K zero-padded to a multiple of 16 (ops.pack_mfma_b: layout only);
 *   bias, BN folded to scale/shift as for gnnpn_linear_f32; eps: one device float.
 *   out [n_graphs, hidden].
 * GNNPN_E_UNSUP (nothing enqueued) unless hidden == 128, n_layers <= 4, emb + nfeat <= 32 and max_nodes <= 16: callers
 * then use the separate kernels.
 * Replaces: src/models/modelML.py:133-143,165-166 (NodeEncoder, GINConv x numLayersGIN with BatchNorm/ReLU, nodeLin,
 * scatter mean). */
typedef struct {
    const float* w0_packed;   /* Linear(in -> 2*hidden) of the GIN MLP (modelML.py:77-90) */
    const float* b0;
    const float* bn1_scale;
    const float* bn1_shift;
    const float* w3_packed;   /* Linear(2*hidden -> hidden) */
    const float* b3;
    const float* bn2_scale;   /* nodeBatchNorms[i] (modelML.py:141) */
    const float* bn2_shift;
    const float* eps;
} gnnpn_gin_layer_t;
int gnnpn_request_branch_f32(const float* x, int32_t nfeat, const float* table, int32_t vocab, int32_t emb,
                             const int32_t* rowptr, const int32_t* col, const int32_t* seg_ptr, int32_t n_graphs,
                             int32_t max_nodes, int32_t n_layers, const gnnpn_gin_layer_t* layers, int32_t hidden,
                             const float* lin_w_packed, const float* lin_b, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Candidate reduction: for every (problem b, category c) choose the n_per best-scored services
 * of that category that satisfy the problem's local bounds, best first (score descending, ties
 * by lowest service id), pad cyclically when fewer than n_per are feasible, and emit the
 * pointer-net input rows.
 *   scores        [B, S] fp32 (sigmoid scores of Net.forward)
 *   cat_ptr       [T+1]  services of category c are ids [cat_ptr[c], cat_ptr[c+1])
 *   qos           [S, 4] fp64 (q0,q1,q2=cost,q3=quality) — fp64 because the reference compares
 *                 the JSON doubles (src/loadData.py:123-124)
 *   local_bounds  [B, T, 4] fp64 cost_lo,cost_hi,quality_lo,quality_hi;  present [B,T] uint8
 *   global_bounds [B, 4] fp64
 *   out_rows      [B, T*n_per, 8] fp32 = q0,q1,q2,q3,c0lo,c0hi,c1lo,c1hi (constraint columns
 *                 non-zero only in category 0); absent / infeasible categories -> 0,1,1,1
 *   out_ids       [B, T*n_per] int32 chosen service id, -1 for dummy rows
 * Replaces: the per-row sort of TrainML.test (src/models/trainML.py:60-68) + loadDataPN
 * (src/loadData.py:101-149) + SCDataset's column drop (src/models/trainPNHigh.py:23-31).
 */
int gnnpn_select_candidates(const float* scores, int64_t ld_scores, const int32_t* cat_ptr,
                            const double* qos, const double* local_bounds,
                            const uint8_t* present, const double* global_bounds, float* out_rows,
                            int32_t* out_ids, int32_t B, int32_t T, int32_t n_per, void* stream);

/* Full descending ranking of every score row (ties: lowest id first): ranking [B,S] int32.
 * S <= 32768 (one LDS bitonic sort per row: 64-bit keys up to 16384, service ids with keys formed on the fly above).
 * Replaces `_x.sort(dim=0, descending=True)` of src/models/trainML.py:62 (whose tie
 * order is undefined). */
int gnnpn_rank_rows(const float* scores, int64_t ld_scores, int32_t* ranking, int32_t B, int32_t S,
                    void* stream);

/* P@k: out[b, i] = #(labels[b, ranking[b, j]] == 1 for j < ks[i]) / ks[i].  ranking [B, >=max k]
 * int32, labels [B,S] fp32 (0/1), ks [n_k] int32 (device), out [B, n_k] fp32.
 * Replaces the P@1 / P@5 loop of TrainML.test (src/models/trainML.py:63-70). */
int gnnpn_precision_at_k(const int32_t* ranking, int64_t ld_rank, const float* labels, int64_t ld_lab,
                         int32_t B, int32_t S, const int32_t* ks, int32_t n_k, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pointer-network LSTM encoder recurrence for `n_nets` independent nets in one launch
 * (Low and High encoders are independent given the inputs).  Per net (gnnpn_encode_net_t):
 *   input side, ONE of
 *     pregates [B, L, 4H] fp32 = x_t . W_ih^T + b_ih (gate order i,f,g,o) computed beforehand, or
 *     inputs [B, L, F] raw rows + w_in [4H, F] + b_in [4H]: the input projection is evaluated inside
 *     the kernel as inputs_t . w_in^T + b_in (F must be 8).  With w_in = W_ih . W_emb and
 *     b_in = W_ih . b_emb + b_ih this folds embedding2 (modelPN.py:190) and the LSTM's input
 *     projection into one [4H, 8] matrix — an exact algebraic identity, rounded differently
 *     (DESIGN.md section 6 item 7); only the cooperative form evaluates it in-kernel.
 *   whh_packed [H/4][gate][j][4] fp32  = W_hh[gate*H + j][k..k+3]   (ops.pack_lstm_weight)
 *   bhh [4H];   outputs enc_out [B, L, H], h_n / c_n [B, H]
 * H must be 256 or 32 (environment.ini:55 and the unit fixtures).  `nets` is a HOST array.
 * Two implementations with bit-identical results: cooperative (H = 256 and a workspace given:
 * groups of 8 workgroups keep W_hh in registers and exchange h every step) and streaming.
 * Replaces: nn.LSTM encoder at src/models/modelPN.py:157,191. */
typedef struct {
    const float* pregates;
    const float* inputs;
    const float* w_in;
    const float* b_in;
    const float* whh_packed;
    const float* bhh;
    float* enc_out;
    float* h_n;
    float* c_n;
    const void* whh_split;   /* NULL, or whh_packed's exact split made once per model by gnnpn_lstm_pack_split_weights_f32 (ABI 9): used by
                              * GNNPN_PREC_SPLIT instead of splitting whh_packed inside every launch — same bits, 9-13 us less per launch */
} gnnpn_encode_net_t;

/* precision: operands of the recurrent W_hh.h product (accumulation, cell, outputs are fp32 in every mode)
 *   GNNPN_PREC_F32   fp32 (the parity path)
 *   GNNPN_PREC_F16   fp16 operands, encoder only: opt-in reduced precision (BASELINE configs[4])
 *   GNNPN_PREC_SPLIT every operand as THREE fp16 pieces that reproduce the fp32 value bit for bit (x 2^s = p0 + p1/2^11 +
 *                    p2/2^22), every cross term that can reach 2^-24 of a product kept (6 products on the fp16 matrix
 *                    cores), fp32 accumulation in three magnitude classes: fp32 operands, fp32 result, no operand bit
 *                    dropped; error bound <= the fp32 fma chain's (profiles/LOG_r01_r04.md section 12; gnnpn_split3_pieces_f32 and
 *                    gnnpn_recurrent_product_f32 below expose the arithmetic to the tests)
 * F16/SPLIT need the cooperative form (H = 256 and a workspace): GNNPN_E_UNSUP otherwise. */
#define GNNPN_PREC_F32 0
#define GNNPN_PREC_F16 1
#define GNNPN_PREC_SPLIT 2

/* Per-call launch options of the two recurrent entry points (NULL = all defaults).  They select among
 * implementations of the SAME arithmetic; nothing here is process-wide state.
 *   impl            encoder: 0 auto, 1 per-workgroup streaming, 2 cooperative (8-CU groups)
 *                   decoder: 0 auto, 1 streaming, 2 cooperative (8-CU groups),
 *                            4 cooperative (8-CU groups, 256-register build that shares a CU with another launch)
 *                   (3, the 16-CU-group forms of ABI versions <= 6, measured slower everywhere and were removed: GNNPN_E_ARG)
 *   lds_kb          LDS footprint (KB per workgroup, padded with unused dynamic LDS) of the cooperative kernel, 0 = none:
 *                   placement control for two launches sharing the CUs (100 on one stream + 56 on the other: a CU
 *                   takes one workgroup of each, never two of one)
 *   write_through   non-zero: always publish hand-off granules with agent-scope (sc1) stores instead of keeping them in
 *                   the XCD's L2 when a group has verified at run time that it sits on one XCD (same results)
 *   paired_start    non-zero: the caller starts this launch TOGETHER with another cooperative launch of the same LDS footprint
 *                   (two half-batches side by side, two slots started as a pair).  LDS is handed out in contiguous ranges; a
 *                   workgroup that lands above a short-lived neighbour keeps its range in the middle of the CU and the partner
 *                   launch's workgroup for that CU can then not be placed until this kernel has finished — two launches that
 *                   do that to each other both stay short of members and end in hand-off time-outs.  With the flag, a
 *                   workgroup in such a position does not take its seat (a later arrival of the over-subscribed launch does);
 *                   without it that happens only while another launch of the process is staffing at the same moment.
 *   sticky_status   device uint32[GNNPN_STATUS_WORDS] or NULL — the caller's status block, which the library only ever ADDS to:
 *                   word GNNPN_STATUS_CODE: every failure code a cooperative kernel raises (GNNPN_COOP_*: a bounded
 *                   inter-workgroup wait timed out, the workspace was not clean: outputs invalid) is OR-ed into it as well as
 *                   into word 0 of the workspace.  The library never clears it — word 0 of the workspace is zeroed by every
 *                   launch — so one host read after any number of launches tells whether ANY of them failed.
 *                   Proof of work (ABI version 7): every cooperative launch adds the workgroup-tiles it is EXPECTED to finish
 *                   (8 members x nets x ceil(B / 16)) to word GNNPN_STATUS_ENC_EXPECTED / _DEC_EXPECTED before it starts, and
 *                   every seated workgroup adds the tiles it really took to the end to GNNPN_STATUS_ENC_FINISHED / _DEC_FINISHED
 *                   when it leaves.  After the stream has been synchronised finished == expected (modulo 2^32) for every launch
 *                   that did its work; a shortfall means that a launch left outputs it never wrote — whatever the failure codes
 *                   say (a launch whose workgroups all leave as surplus raises none).  The host reports it as
 *                   GNNPN_COOP_SHORTFALL (ops.Workspaces.check / poll, which also hold their own count of the launches they
 *                   made against the expected words).
 * In flight: a CU holds TWO cooperative workgroups, so at most two cooperative launches (on two streams, with a workspace each)
 * may be in flight on a device at a time; three or more stay half-staffed behind each other until the bounded waits give up
 * (GNNPN_COOP_ENC_TIMEOUT / _DEC_TIMEOUT — loud, and very slow).  A caller with more streams orders them with events
 * (pipeline.PipelinedRunner does: two replays in flight whatever its number of slots, runners of one process take turns). */
#define GNNPN_STATUS_WORDS 8
#define GNNPN_STATUS_CODE 0
#define GNNPN_STATUS_DECLINED_SEATS 1       /* ABI 9: early arrivals that declined their seat because of their LDS position, summed over the launches */
#define GNNPN_STATUS_OFF_CANONICAL_SEATS 2  /* ABI 9: seats the reserve took on another CU than the canonical one, summed over the launches */
#define GNNPN_STATUS_ENC_EXPECTED 4
#define GNNPN_STATUS_ENC_FINISHED 5
#define GNNPN_STATUS_DEC_EXPECTED 6
#define GNNPN_STATUS_DEC_FINISHED 7
/* failure codes (bits) of the cooperative kernels, in word 0 of a workspace and in sticky_status[GNNPN_STATUS_CODE] */
#define GNNPN_COOP_ENC_TIMEOUT 1     /* an encoder hand-off sweep gave up (~0.3 s) */
#define GNNPN_COOP_DEC_TIMEOUT 2     /* a decoder hand-off sweep gave up */
#define GNNPN_COOP_MEMBER_MISSING 4  /* a group member never showed up */
#define GNNPN_COOP_DIRTY 8           /* the status area of the workspace was not clean when the launch began */
#define GNNPN_COOP_SHORTFALL 16      /* host-side: finished != expected after a synchronisation (never set by a kernel) */
typedef struct {
    int32_t impl;
    int32_t lds_kb;
    int32_t write_through;
    int32_t paired_start;
    uint32_t* sticky_status;
} gnnpn_launch_opts_t;

/* The exact split (GNNPN_PREC_SPLIT) of a packed recurrent weight matrix [H/4][gate][H][4], H = 256, made ONCE per model instead of
 * inside every cooperative launch: per group member the three fp16 pieces of its 128 gate columns in the order the lanes load them
 * (csrc/coop_common.h: load_split_weights), the column scales included.  Written by the same device function the kernels otherwise
 * run for themselves: results are bit-identical with and without it.  `split`: gnnpn_lstm_split_weights_bytes() bytes, 16-byte aligned;
 * valid for encoder and decoder W_hh alike (gnnpn_encode_net_t.whh_split, gnnpn_decode_net_t.whh_split).  No counterpart in the
 * reference (nn.LSTM weights, src/models/modelPN.py:157,165): a load-time layout change like whh_packed.  New in ABI version 9. */
int64_t gnnpn_lstm_split_weights_bytes(void);
int gnnpn_lstm_pack_split_weights_f32(const float* whh_packed, void* split, void* stream);

int gnnpn_lstm_encode_f32(int n_nets, const gnnpn_encode_net_t* nets, int32_t B, int32_t L, int32_t H,
                          int32_t F, int32_t precision, const gnnpn_launch_opts_t* opts, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* Size of the device workspace the cooperative encoder needs (status words + hand-off buffers).
 * With workspace == NULL gnnpn_lstm_encode_f32 uses the per-workgroup streaming form instead.
 * After the stream has been synchronised, word 0 of the workspace (uint32) is non-zero iff a
 * bounded inter-workgroup wait of the LAST launch on it timed out (outputs are then invalid);
 * gnnpn_launch_opts_t.sticky_status accumulates over launches. */
int64_t gnnpn_lstm_encode_workspace_bytes(void);

/* LDS footprint of the ORDINARY kernels of this library that use LDS and run in front of a cooperative launch in a pipelined step
 * (gnnpn_linear_f32, gnnpn_request_branch_f32): every such kernel launched by the CALLING THREAD from now on is padded with unused
 * dynamic LDS to `kb` KB per workgroup (0 = no padding, the default; kernels that already use more are left alone).  Returns the
 * previous setting.  Why: LDS is handed out in contiguous ranges.  A cooperative workgroup (78 KB in the exact-split precision) that
 * lands ABOVE a small kernel's few KB keeps its range in the middle of the CU after the small kernel has gone, and the other
 * pipeline slot's cooperative workgroup for that CU then has no contiguous 78 KB until this whole launch has finished (0.5 ms at the
 * QWS shape: the "slip" of two slots out of step, measured in every 20-step round without a common start).  With the small kernels
 * at the SAME footprint their range is exactly what the next cooperative workgroup needs (profiles/r05_slip_front_lds.jsonl:
 * 436 k -> 460 k problems/s, no slow round in 270).  A placement setting like gnnpn_launch_opts_t.lds_kb — it selects no kernel
 * build and changes no result; capture a HIP graph with it set and the captured launches keep it.  New in ABI version 7. */
int gnnpn_lds_footprint_kb(int kb);

/* Proof of work, host side (ABI version 7): the workgroup-tiles the LAST call of gnnpn_lstm_encode_f32 / gnnpn_pointer_decode_f32
 * on THIS thread booked as expected in the caller's status block (gnnpn_launch_opts_t.sticky_status) — 0 if the call took a
 * streaming form, had no status block or failed before launching.  A caller adds it to its own count of the work it asked for
 * and compares that with the device's GNNPN_STATUS_*_EXPECTED / _FINISHED words after a synchronisation.  Replaces nothing in the
 * reference (PyTorch launches are not cooperative: /root/reference/src/models/modelPN.py:191,204-239 cannot half-run). */
int64_t gnnpn_last_launch_units(void);

/* Diagnostics switch (process-wide; tools/ only):
 *   "lstm_ablate"    phase stamps / ablations of the cooperative kernels; results are WRONG when non-zero.
 * Implementation choice and placement control are per-call arguments (gnnpn_launch_opts_t), not options.
 * Unknown names / out-of-range values: GNNPN_E_ARG. */
int gnnpn_set_option(const char* name, int value);

/* Failure record of the cooperative decoder (diagnosis only; host call, synchronises the device): copies the first
 * n_words (<= 512) uint32 of the record to the HOST buffer `out`, optionally clearing it.  out[0] = timed-out sweeps
 * since the last clear; then 16 words per failing wave (the first 31): group, member, tile, step k, wave, expected tag,
 * mask of members whose h granules were missing, mask of members whose partial dots were missing, missing latent
 * lanes, the launch's eight per-XCD claim counters (one byte each, two words), device realtime (lo, hi), groups per
 * XCD, workgroup id, the launch's status word. */
int gnnpn_decode_diag(uint32_t* out, int32_t n_words, int32_t clear);
/* Housekeeping after a cooperative launch reported a hand-off time-out (sticky status != 0): clears the per-device count of
 * launches that are still staffing (such a launch may never have left it); synchronises the device. */
int gnnpn_coop_reset_staffing(void);
/* Diagnostics: that count (device synchronised first); 0 whenever no cooperative launch is staffing; -1 on error. */
int gnnpn_coop_staffing_count(void);

/* Greedy pointer decode of up to two pointer networks in one call: T steps of {decoder LSTM cell;
 * dot-attention logits over the step's candidate window [k*n_per,(k+1)*n_per); C*tanh; + latent
 * window (High net); softmax; first-max argmax; next input = embedded row of the pick}.
 * Per net (gnnpn_decode_net_t, all device pointers):
 *   embedded [B,L,H] (decoder inputs = rows of it), or NULL with emb_w [H,8] / emb_b [H]: the picked
 *            row is then computed in-kernel as inputs[pick] . emb_w^T + emb_b (same k-ordered fma
 *            chain + bias as gnnpn_linear_f32, so bit-identical to the stored row; cooperative form)
 *   xw_fold [4H,8], xb_fold [4H], start_fold [4H] or NULL (all three): folded input side of the
 *            decoder cell, xw_fold = W_ih.W_emb, xb_fold = W_ih.b_emb + b_ih, start_fold =
 *            W_ih.start + b_ih; the step input is then evaluated from the raw row of the pick
 *            (cooperative form only; same exact identity as gnnpn_encode_net_t.w_in)
 *   enc_out [B,L,H]  h0,c0 [B,H]  start [H]
 *   wih_packed / whh_packed / bih / bhh : decoder LSTM (packed as for the encoder)
 *   latent_win  [B,T,n_per] or NULL : window logits of a Low net computed EARLIER, added before the
 *               argmax;   latent_from : index (< own index) of a net of THIS call whose window
 *               logits are used instead (the two-level scheme of trainPNHigh.py:138-139), or -1
 *   outputs: idx [B,T] int32; win_logits [B,T,n_per] (C*tanh(dot), before the latent is added);
 *            pick_prob [B,T] softmax prob of the pick; actions [B,T,8]; queries [B,T,H] or NULL.
 *   sample = 1: the pick of step k is DRAWN from the window softmax (the reference's probs.multinomial(1),
 *            src/models/modelPN.py:227-228; its re-draw loop :229-234 cannot trigger because the step windows are
 *            disjoint): p_r = softmax over the window, cdf_r = p_0 + ... + p_r (fp32, in order), u = draw (b*T + k) of
 *            the counter-based stream of sample_seed — splitmix64(sample_seed + (b*T + k + 1) * 0x9E3779B97F4A7C15) >> 40,
 *            times 2^-24 — and the pick is the first r with u < cdf_r (the last r with p_r > 0 if rounding leaves none).
 *            pick_prob is then the probability of the drawn candidate (CombinatorialRL.forward :297-299).  The
 *            reference draws from torch's global generator; tests route its multinomial to this stream (oracle/pn.py).
 *            Built in the streaming and the 8-CU-group cooperative forms (fp32); GNNPN_E_UNSUP elsewhere.
 * inputs [B,L,8] : rows gathered into `actions`.   nets: HOST array of n_nets (1 or 2) structs.
 * Two implementations: cooperative (H = 256, n_per <= 16, workspace given: 8-CU groups keep both
 * weight matrices in registers; both nets in one launch, the High net one step behind the Low net)
 * and per-workgroup streaming (nets decoded one after the other).  They differ only in the order
 * the 256 products of an attention dot are summed.
 * Replaces: PointerNet.forward's decode loop src/models/modelPN.py:193-241 incl. Attention 'Dot'
 * (:111-114,119-120), the window mask loop (:220-222), softmax/max (:224-226), the gathers at
 * :235 and CombinatorialRL.forward :293-299. */
typedef struct {
    const float* embedded;
    const float* enc_out;
    const float* h0;
    const float* c0;
    const float* start;
    const float* wih_packed;
    const float* whh_packed;
    const float* bih;
    const float* bhh;
    const float* latent_win;
    const float* emb_w;
    const float* emb_b;
    const float* xw_fold;
    const float* xb_fold;
    const float* start_fold;
    int32_t* idx;
    float* win_logits;
    float* pick_prob;
    float* actions;
    float* queries;
    int32_t latent_from;
    int32_t sample;          /* 0: greedy, first-max argmax (modelPN.py:226); 1: multinomial draw (:228), see below */
    uint64_t sample_seed;    /* stream of the draws of this net in this call */
    const void* whh_split;   /* NULL, or gnnpn_lstm_pack_split_weights_f32(whh_packed): as gnnpn_encode_net_t.whh_split (ABI 9) */
} gnnpn_decode_net_t;

/* precision: GNNPN_PREC_F32 or GNNPN_PREC_SPLIT (the decoder cell's W_hh.h product; cooperative, folded form). */
int gnnpn_pointer_decode_f32(int n_nets, const gnnpn_decode_net_t* nets, const float* inputs,
                             float tanh_c, int use_tanh, int32_t B, int32_t T, int32_t n_per, int32_t H,
                             int32_t precision, const gnnpn_launch_opts_t* opts, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* The attention forms the reference's configurations leave switched off:
 * replaces PointerNet.forward's per-step body (modelPN.py:204-239) with attention 'Bahdanau' (:80-90,103-109) and / or
 * n_glimpses > 0 (:208-211).  One net per call (chain Low -> High through latent_win); greedy, or with net->sample the pick of
 * every step drawn from the window softmax (:227-228) from the stream of net->sample_seed, as gnnpn_pointer_decode_f32 draws.
 *   attention    0 'Dot' | 1 'Bahdanau'
 *   n_glimpses   glimpse rounds per step (each: logits over all L positions, -inf at the positions chosen so far,
 *                softmax, query = ref' . softmax)
 *   *_wq [H,H], *_bq [H]   W_query of the pointer / glimpse Attention module          ('Bahdanau' only)
 *   *_ref [B,L,H]          W_ref (the 1x1 Conv1d) applied to enc_out, bias included — step-independent, computed by the
 *                          caller with gnnpn_linear_f32                                  ('Bahdanau' only)
 *   *_v [H]                V
 * Uses of gnnpn_decode_net_t: embedded (required), enc_out, h0, c0, start, wih_packed, whh_packed, bih, bhh, latent_win
 * (optional), idx, win_logits, pick_prob, actions, queries (optional).  H = 256 or 32; L*4 + T*4 bytes of LDS <= 120 KB. */
typedef struct {
    int32_t attention;
    int32_t n_glimpses;
    const float* pointer_wq;
    const float* pointer_bq;
    const float* pointer_ref;
    const float* pointer_v;
    const float* glimpse_wq;
    const float* glimpse_bq;
    const float* glimpse_ref;
    const float* glimpse_v;
} gnnpn_attention_t;

int gnnpn_pointer_decode_attn_f32(const gnnpn_decode_net_t* net, const gnnpn_attention_t* attn, const float* inputs,
                                  float tanh_c, int use_tanh, int32_t B, int32_t T, int32_t n_per, int32_t H, void* stream);

/* Workspace the cooperative decoder needs for this shape (status words, hand-off buffers, the
 * Low->High latent granules).  Word 0 after synchronisation: as for the encoder. */
int64_t gnnpn_pointer_decode_workspace_bytes(int32_t B, int32_t T, int32_t n_per);

/* Full-length attention logits of ONE decode step, for callers that need the reference's
 * return values verbatim (the reference returns every step's whole [B,L] logits tensor,
 * src/models/modelPN.py:239,291; only the window part feeds the decision):
 *   logits[b,l] = C*tanh(dot(enc_out[b,l,:], q[b,:]))  for all l, then -inf at the n_masked
 *   previously chosen positions masked_idx[b*ld_idx + 0..n_masked) (in-place mask of :165-173).
 * queries: row b at queries + b*ld_q.  Not on the fast path. */
int gnnpn_attention_logits_f32(const float* enc_out, const float* queries, int64_t ld_q,
                               const int32_t* masked_idx, float tanh_c, int use_tanh, float* logits,
                               int32_t B, int32_t L, int32_t H, int32_t n_masked, int32_t ld_idx,
                               void* stream);
/* The 'Bahdanau' form (src/models/modelPN.py:103-109,119-120): logits[b,l] = C*tanh(V . tanh(qp[b,:] + ref[b,l,:])) with
 * qp = W_query q + b_query [B,H] (row b at qp + b*ld_q) and ref = W_ref(enc_out) + b_ref [B,L,H] formed by the caller
 * (gnnpn_linear_f32); same mask.  Not on the fast path. */
int gnnpn_attention_logits_bahdanau_f32(const float* ref, const float* qp, int64_t ld_q, const float* v,
                                        const int32_t* masked_idx, float tanh_c, int use_tanh, float* logits,
                                        int32_t B, int32_t L, int32_t H, int32_t n_masked, int32_t ld_idx, void* stream);

/* QoS reward of decoded compositions: per problem, violate = #global constraints whose product
 * QoS (q2, q3 over the T actions, fp32 running product) falls outside [lo,hi] read from the
 * step-0 action row; obj = (sum q0 / #(q0>0) + 1 - min q1)/2; level 0 ("Low") -> violate,
 * level 1 ("High") -> round(violate + obj, 5).   actions [B,T,8] -> R [B] fp32.
 * Replaces: reward/calc at src/models/modelPN.py:15-72 (minus the print at :67). */
int gnnpn_qos_reward_f32(const float* actions, float* R, int32_t B, int32_t T, int level,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * REINFORCE training step of the High-level pointer network.  Replaces the autograd graph of
 * src/models/trainPNHigh.py:84-108 over src/models/modelPN.py:175-306: actor_loss.backward() (:103-104), clip_grad_norm_
 * (:105-106) and actor_optim.step() (:108).  The picks of the step come from the sampled forward (gnnpn_pointer_decode_f32
 * with sample = 1) and are constants of the differentiated graph.  Weights: the two FORWARD entry points take the LSTM
 * matrices TRANSPOSED ([H,4H], k-major: coalesced forward products), the two BACKWARD entry points the plain row-major
 * [4H,H] matrices of the state_dict (coalesced transposed products).  H must be 256 or 32.
 *
 * gnnpn_gemm_f32: C[m,n] = sum_k Aop[m,k]*Bop[n,k], Aop[m,k] = a_kmajor ? A[k*lda+m] : A[m*lda+k] (same for B): weight
 *   gradients dW = dG^T . X (both operands k-major, k = the B*L rows of saved activations) and dX = dG . W (B k-major).
 *   split_k > 1: C holds split_k partial matrices [split_k][M][ldc], slice s reducing its own k range; the caller adds them
 *   in slice order (gnnpn_colsum_f32 over rows = split_k): deterministic, and K = B*L >> M, N fills the chip.
 * gnnpn_lstm_train_forward_f32: the encoder recurrence from pregates [B,L,4H] (= embedded . W_ih^T + b_ih), saving the full
 *   pre-activation gates [B,L,4H] and the cell states [B,L,H] besides enc_out.
 * gnnpn_decode_train_forward_f32: the decode loop with the picks idx [B,T] GIVEN (teacher forcing), saving decoder inputs,
 *   pre-activation gates, c_k, h_k, the window logits z0 = C*tanh(dot) and probabilities, and log p(pick) [B,T].
 * gnnpn_decode_train_backward_f32: T steps in reverse.  gscale[b] = dLoss/dlog p of every step of problem b
 *   (advantage_b / B; 0 where the reference zeroes the sum, :98).  Writes d_enc_out [B,L,H] (every element exactly once:
 *   the step windows partition the L rows), the gate gradients dgates [B,T,4H], the decoder-input gradients dx [B,T,H] and
 *   the gradient wrt the initial state (dh0, dc0 [B,H]) that continues into the encoder.
 * gnnpn_lstm_train_backward_f32: L steps in reverse from (dh0, dc0) and d_enc_out; writes dgates [B,L,4H].
 * gnnpn_colsum_f32 (bias gradients, the start-input gradient), gnnpn_scatter_dx_f32 (dx_k into the embedded row the step's
 *   input was gathered from, :235), gnnpn_sumsq_f32 (accumulates into a device double: the squared gradient norm),
 * gnnpn_adam_step_f32: p -= lr * mhat / (sqrt(vhat) + eps) on g * min(1, max_grad_norm / (sqrt(*grad_sumsq) + 1e-6)) —
 *   torch.nn.utils.clip_grad_norm_ followed by torch.optim.Adam with its defaults. */
typedef struct {
    const float* embedded;   /* [B,L,H] */
    const float* enc_out;    /* [B,L,H] */
    const float* h0;         /* [B,H] encoder final h */
    const float* c0;         /* [B,H] encoder final c */
    const float* start;      /* [H] decoder_start_input */
    const float* wih;        /* decoder LSTM: forward: transposed [H,4H]; backward: row-major [4H,H] */
    const float* whh;
    const float* bih;
    const float* bhh;
    const float* latent_win; /* [B,T,K] or NULL */
    const int32_t* idx;      /* [B,T] picks (global positions) */
    float* x_all;            /* [B,T,H] */
    float* gates_pre;        /* [B,T,4H] */
    float* c_all;            /* [B,T,H] */
    float* h_all;            /* [B,T,H] */
    float* z0;               /* [B,T,K] */
    float* probs;            /* [B,T,K] */
    float* logp;             /* [B,T] */
} gnnpn_decode_train_t;
int gnnpn_gemm_f32(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
                   int64_t ldc, int64_t M, int N, int K, int split_k, void* stream);
int gnnpn_lstm_train_forward_f32(const float* pregates, const float* whh, const float* bhh, float* enc_out,
                                 float* gates_pre, float* c_all, int32_t B, int32_t L, int32_t H, void* stream);
int gnnpn_decode_train_forward_f32(const gnnpn_decode_train_t* t, int32_t B, int32_t T, int32_t n_per, int32_t H,
                                   float tanh_c, int use_tanh, void* stream);
int gnnpn_decode_train_backward_f32(const gnnpn_decode_train_t* t, const float* gscale, float* d_enc_out, float* dgates,
                                    float* dx, float* dh0, float* dc0, int32_t B, int32_t T, int32_t n_per, int32_t H,
                                    float tanh_c, int use_tanh, void* stream);
int gnnpn_lstm_train_backward_f32(const float* whh, const float* gates_pre, const float* c_all, const float* d_enc_out,
                                  const float* dh0, const float* dc0, float* dgates, int32_t B, int32_t L, int32_t H,
                                  void* stream);

/* The same two decoder passes THROUGH the attention forms the reference's configurations switch off — 'Bahdanau' attention
 * (src/models/modelPN.py:80-90,103-109) and glimpse rounds (:208-211), any combination; replaces autograd over :204-239 for them.
 * `ref` = W_ref(enc_out) + b_ref per attention module ([B,L,H], gnnpn_linear_f32 by the caller).  The forward saves the queries
 * q_0 = h_k ... q_G of every step and the glimpse softmaxes; the backward (d_enc_out is ADDED to: the caller zeroes it, as d_p_ref /
 * d_g_ref) leaves d ref, the gradients wrt the projected queries (W_query's gradient = their GEMM with the saved queries, its
 * bias' their column sum) and per-problem sums for V.  W_ref's gradient and d enc_out's share through ref are GEMMs over d ref. */
typedef struct {
    gnnpn_decode_train_t base;
    int32_t bahdanau;        /* 0 'Dot', 1 'Bahdanau' */
    int32_t n_glimpses;      /* 0..8 */
    /* 'Bahdanau' only (NULL for 'Dot'); the glimpse module's only with n_glimpses > 0 */
    const float* p_wq_t;     /* pointer.W_query.weight TRANSPOSED [H,H] (forward products) */
    const float* p_wq;       /* ... row-major [H,H] (transposed products of the backward) */
    const float* p_bq;       /* [H] */
    const float* p_v;        /* [H] */
    const float* p_ref;      /* [B,L,H] */
    const float* g_wq_t;
    const float* g_wq;
    const float* g_bq;
    const float* g_v;
    const float* g_ref;
    float* q_all;            /* [B,T,G+1,H] saved queries */
    float* a_all;            /* [B,T,G,L] saved glimpse softmaxes (NULL with G = 0) */
    /* backward outputs ('Bahdanau'; the *_ref buffers are added to) */
    float* d_p_ref;          /* [B,L,H] */
    float* d_g_ref;
    float* d_p_qp;           /* [B,T,H] */
    float* d_g_qp;           /* [B,T,G,H] */
    float* d_p_v;            /* [B,H] */
    float* d_g_v;
} gnnpn_decode_attn_train_t;
int gnnpn_decode_attn_train_forward_f32(const gnnpn_decode_attn_train_t* t, int32_t B, int32_t T, int32_t n_per, int32_t H,
                                        float tanh_c, int use_tanh, void* stream);
int gnnpn_decode_attn_train_backward_f32(const gnnpn_decode_attn_train_t* t, const float* gscale, float* d_enc_out, float* dgates,
                                         float* dx, float* dh0, float* dc0, int32_t B, int32_t T, int32_t n_per, int32_t H,
                                         float tanh_c, int use_tanh, void* stream);
int gnnpn_colsum_f32(const float* X, int64_t ld, int64_t rows, int32_t cols, float* out, void* stream);
/* first pass of a two-pass column sum over many rows: partial[c][col] = sum of rows [c*rows_per_chunk, (c+1)*rows_per_chunk) */
int gnnpn_colsum_chunks_f32(const float* X, int64_t ld, int64_t rows, int32_t cols, int64_t rows_per_chunk, float* partial,
                            void* stream);
int gnnpn_scatter_dx_f32(const float* dx, const int32_t* idx, float* d_embedded, int32_t B, int32_t T, int32_t L, int32_t H,
                         void* stream);
int gnnpn_sumsq_f32(const float* x, int64_t n, double* accum, void* stream);
int gnnpn_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, const double* grad_sumsq,
                        float max_grad_norm, float lr, float beta1, float beta2, float eps, int32_t step, void* stream);

/* Test hook: the sigmoid / tanh the LSTM cells use (hardware exp2/rcp based, |error| ~1e-7),
 * evaluated on an array so that tests can measure them against the CPU's libm-grade functions. */
int gnnpn_debug_cell_activations(const float* x, float* sig, float* th, int64_t n, void* stream);

/* A gate for a stream that the HOST opens: one wavefront polls the 32-bit word `flag` (pinned host memory, or device memory) until
 * it holds `expect` or `timeout_us` (<= 100000) have passed; work enqueued behind it on `stream` starts then.  No counterpart in
 * the reference (its training loop runs one batch at a time, trainPNHigh.py:134-139): PipelinedRunner holds the first replays of a
 * burst of two free-running slots behind one gate and opens it when both replays are enqueued, so that the slots start together. */
int gnnpn_gate_wait(const void* flag, uint32_t expect, int32_t timeout_us, void* stream);

/* Test hook: a stand-in for a collective's kernel beside the cooperative launches — n_workgroups workgroups of 256 threads hold
 * lds_bytes of LDS each for hold_us microseconds (sleeping spin, no memory traffic) on `stream`.  What an 8-rank RCCL ring kernel
 * does to the placement of the cooperative kernels cannot be seen on a one-GPU box (RCCL at world size 1 is a copy); the soak
 * tests run the two-slot pipeline beside this. */
int gnnpn_debug_lds_interferer(int32_t n_workgroups, int32_t lds_bytes, int32_t hold_us, void* stream);

/* Test hook: D = A . B + C by ONE v_mfma_f32_16x16x32_f16 (A [16][32], B [32][16] fp32 values representable in fp16; C, D
 * [16][16] fp32) — lets a test pin the accumulate model of the f16 matrix core that the a-priori error bound of the
 * exact-split recurrent product rests on (groups of 8 k, truncation below 2^-24 of the group's largest operand, one
 * round-to-nearest-even per group: DESIGN.md section 5; profiles/LOG_r01_r04.md section 12, tools/probes/mfma_accum_model.hip). */
int gnnpn_debug_mfma_f16(const float* A, const float* B, const float* C, float* D, void* stream);

/* ES-WOA fine-tuning of P compositions in one launch (one wavefront per problem).  Replaces `ESWOA.__init__` +
 * `ESWOA.start` of src/baselines/WOA.py:8-162 (the step after the ML+2PN path); the caller
 * (gnnpn-sc_amd/WOA.py) does the reference's host-side preparation: 5-decimal rounding of the QoS tuples (:13-26) and
 * appending a pick that is not among its category's candidates (:62-69).
 *   cand_ptr [P*T+1] : candidate rows of (problem p, category j) are cand[cand_ptr[p*T+j] .. cand_ptr[p*T+j+1])
 *   len_init [P*T]   : candidate counts BEFORE the append (the initial population is drawn from those, :51-52)
 *   cand     [n,4]   : q0..q3 (float64, as the reference computes)
 *   bounds   [P,4]   : lo/hi of the two product constraints
 *   start_pos [P*T]  : position of the seed solution's pick in each category, or start_pos[p*T] < 0: no seed (:70-74)
 *   seeds    [P]     : draw k of a problem is splitmix64(seed + k * 0x9E3779B97F4A7C15) >> 11, scaled to [0,1)
 *                      (the reference uses numpy's global generator; see oracle/woa.py)
 *   max_cand         : the largest cand_ptr[(p+1)*T] - cand_ptr[p*T] (sizes the LDS)
 * Outputs: best_fitness [P], best_pos [P*T] (may be negative: Python list positions), history [P*max_iter] (best
 * fitness after every iteration, WOA.py:128,161), draws [P] (stream positions consumed).
 * GNNPN_E_UNSUP: T outside 1..64 (one category per lane: use gnnpn_eswoa_wide_f64), or a problem that does not fit a CU's LDS. */
int gnnpn_eswoa_f64(int32_t P, int32_t T, const int32_t* cand_ptr, const int32_t* len_init, const double* cand,
                    const double* bounds, const int32_t* start_pos, int32_t pop, int32_t max_iter, const uint64_t* seeds,
                    int32_t max_cand, double* best_fitness, int32_t* best_pos, double* history, int64_t* draws,
                    void* stream);

/* The same search for ANY number of categories (the 1000- and 2000-task configurations): one workgroup per problem, the
 * population's positions in `workspace` (gnnpn_eswoa_wide_workspace_bytes = P * pop * T int32), the three QoS columns of
 * the composition under evaluation in LDS (24 T + 8 T bytes: T <= ~5000).  Same draws, same float64 evaluation orders
 * (np.cumprod sequential; np.sum pairwise — for n > 128 numpy's recursion: halves rounded down to a multiple of 8), so a
 * run equals gnnpn_eswoa_f64's wherever both apply and the oracle's (oracle/woa.py) everywhere.
 * GNNPN_E_UNSUP: the columns do not fit a CU's LDS. */
int64_t gnnpn_eswoa_wide_workspace_bytes(int32_t P, int32_t T, int32_t pop);
int gnnpn_eswoa_wide_f64(int32_t P, int32_t T, const int32_t* cand_ptr, const int32_t* len_init, const double* cand,
                         const double* bounds, const int32_t* start_pos, int32_t pop, int32_t max_iter, const uint64_t* seeds,
                         void* workspace, int64_t workspace_bytes, double* best_fitness, int32_t* best_pos, double* history,
                         int64_t* draws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step of the GNN candidate-ranking model.  Replaces the autograd graph of
 * src/models/trainML.py:39-45 over src/models/modelML.py:131-176 (model.train(): BatchNorm1d on batch statistics).  The
 * step is assembled on the host side (gnnpn-sc_amd/trainML.py) from the inference kernels — gnnpn_linear_f32,
 * gnnpn_csr_aggregate_f32 on the graph and on its transpose, gnnpn_embed_concat_f32 — the generic training kernels above
 * (gnnpn_gemm_f32, gnnpn_colsum_f32, gnnpn_adam_step_f32) and:
 * gnnpn_bn_train_forward_f32: y = [relu](gamma * xhat + beta), xhat = (x - mean) / sqrt(var_biased + eps) over the rows
 *   (BatchNorm1d at modelML.py:79,87,141,154 in training mode); saves xhat [rows,cols] and invstd [cols]; running_mean /
 *   running_var (optional) move by `momentum` towards the batch mean / unbiased variance.
 * gnnpn_bn_train_backward_f32: dy wrt the (post-ReLU) output -> dx, dgamma, dbeta.
 * gnnpn_bce_sigmoid_f32: loss = BCELoss(mean)(p, y) (trainML.py:28,42; log terms clamped at -100) for p = sigmoid(z), and
 *   dz = dLoss/dz as autograd forms it (BCELoss backward with its 1e-12 clamp, then sigmoid backward).
 * gnnpn_dot_f32: out = sum a_i b_i (GINConv's d eps = <d out, x>, modelML.py:91 train_eps=True).
 * gnnpn_embed_grad_f32: d table[v] = sum of dh[n][0:c] over rows with id x[n*ldx] == v, rows ascending (the Embedding
 *   backward of NodeEncoder, modelML.py:22-29). */
int gnnpn_bn_train_forward_f32(const float* x, int64_t rows, int32_t cols, const float* gamma, const float* beta, float eps,
                               float momentum, int relu, float* y, float* xhat, float* invstd, float* running_mean,
                               float* running_var, void* stream);
int gnnpn_bn_train_backward_f32(const float* dy, const float* y, const float* xhat, const float* gamma, const float* invstd,
                                int64_t rows, int32_t cols, int relu, float* dx, float* dgamma, float* dbeta, void* stream);
int gnnpn_bce_sigmoid_f32(const float* p, const float* y, int64_t n, float* dz, float* loss, void* stream);
int gnnpn_dot_f32(const float* a, const float* b, int64_t n, float* out, void* stream);
int gnnpn_embed_grad_f32(const float* dh, int64_t ldh, const float* x, int64_t ldx, int64_t rows, int32_t c, int32_t vocab,
                         float* dtable, void* stream);

/* ---- the exact-split arithmetic on its own (what GNNPN_PREC_SPLIT computes; used by tests/test_split3.py) ----
 * The reference multiplies W_hh and h in fp32 inside nn.LSTM (/root/reference/src/models/modelPN.py:157-158,191,205); these two
 * entry points run the device functions the cooperative kernels inline, nothing else:
 * gnnpn_split3_pieces_f32: the three fp16 pieces (bit patterns) of x[i] * 2^scale_log2; their sum p0 + p1/2^11 + p2/2^22 is
 *   x[i] * 2^scale_log2 exactly whenever x[i] 2^scale_log2 lies in [2^-23, 65504] in magnitude or is zero.
 * gnnpn_recurrent_product_f32: gates[16][4H] = h[16][H] . W_hh^T, H = 256, W_hh in the packed recurrent layout
 *   ([H/4][4 gates][H][4]), with the fp32 MFMA chain (GNNPN_PREC_F32) or the exact split (GNNPN_PREC_SPLIT; col_inv [4H]
 *   receives each gate column's un-scaling factor 2^-(15+s)). */
int gnnpn_split3_pieces_f32(const float* x, int64_t n, int32_t scale_log2, uint16_t* p0, uint16_t* p1, uint16_t* p2, void* stream);
int gnnpn_recurrent_product_f32(const float* whh_packed, const float* h, int32_t precision, float* gates, float* col_inv,
                                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNPN_HIP_H */
