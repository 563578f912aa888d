// PyTorch-ROCm custom operators of the ML+2PN path, registered from C++ (SURVEY.md section 8b "Custom-op layer"): the
// `gnnpn::` namespace — schemas (TORCH_LIBRARY) and the CUDA (= HIP on ROCm) implementations (TORCH_LIBRARY_IMPL) — every
// operator a thin call into the C ABI of libgnnpn_hip.so (include/gnnpn_hip.h) on the CURRENT HIP stream
// (c10::hip::getCurrentHIPStream), operands borrowed, outputs allocated here.  No CPU kernel is registered: host tensors fail
// in the dispatcher; a failed launch raises (c10::Error -> RuntimeError) with gnnpn_last_error().  Inference only (no autograd
// formulas).  Host code only — compiled with the host compiler against torch's headers by build.py into libgnnpn_torch.so,
// loaded by custom_ops.py with torch.ops.load_library.  The reference is pure Python with no operator layer of its own; the
// operators stand for the torch / torch_geometric calls cited at each C entry point in gnnpn_hip.h.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <string>
#include <vector>

#include "gnnpn_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

// Every operator runs on the device of its first tensor operand: that device is made current for the call (the C library reads
// hipGetDevice for its per-device tables) and the work goes to THAT device's current stream.
#define GNNPN_ON_DEVICE_OF(t)                          \
    TORCH_CHECK((t).is_cuda(), "gnnpn: expected a CUDA tensor (the hot path has no CPU implementation)"); \
    const c10::DeviceGuard device_guard_((t).device()); \
    void* const stream_ = (void*)c10::hip::getCurrentHIPStream((t).device().index()).stream()
#define cur_stream() stream_

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == GNNPN_OK, what, " failed (", rc, "): ", gnnpn_last_error()); }

template <typename T>
const T* cptr(const Tensor& t, at::ScalarType dt, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, ": expected a CUDA tensor (the hot path has no CPU implementation)");
    TORCH_CHECK(t.scalar_type() == dt, name, ": expected dtype ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": tensor must be contiguous");
    return static_cast<const T*>(t.data_ptr());
}
template <typename T>
const T* cptr(const OptTensor& t, at::ScalarType dt, const char* name) {
    return (t.has_value() && t->defined()) ? cptr<T>(*t, dt, name) : nullptr;
}
const float* f32(const Tensor& t, const char* n) { return cptr<float>(t, at::kFloat, n); }
const float* f32(const OptTensor& t, const char* n) { return cptr<float>(t, at::kFloat, n); }
const int32_t* i32(const Tensor& t, const char* n) { return cptr<int32_t>(t, at::kInt, n); }
const int32_t* i32(const OptTensor& t, const char* n) { return cptr<int32_t>(t, at::kInt, n); }
float* out_f32(Tensor& t) { return static_cast<float*>(t.data_ptr()); }
Tensor new_f32(const Tensor& like, at::IntArrayRef shape) { return at::empty(shape, like.options().dtype(at::kFloat)); }
Tensor new_i32(const Tensor& like, at::IntArrayRef shape) { return at::empty(shape, like.options().dtype(at::kInt)); }
const Tensor& rows2d(const Tensor& t, const char* name) {
    TORCH_CHECK(t.dim() == 2, name, ": expected a 2-D tensor, got ", t.dim(), " dimensions");
    return t;
}

// ---- dense / graph ---------------------------------------------------------------------------------------------------
Tensor linear(const Tensor& a, const Tensor& weight, const OptTensor& bias, const OptTensor& scale, const OptTensor& shift, int64_t act) {
    GNNPN_ON_DEVICE_OF(a);
    rows2d(a, "linear.a"), rows2d(weight, "linear.weight");
    const int64_t M = a.size(0), K = a.size(1), N = weight.size(0);
    TORCH_CHECK(weight.size(1) == K, "linear: K mismatch ", a.sizes(), " x ", weight.sizes());
    Tensor out = new_f32(a, {M, N});
    check_rc(gnnpn_linear_f32(f32(a, "a"), K, f32(weight, "weight"), K, f32(bias, "bias"), f32(scale, "scale"), f32(shift, "shift"),
                              (int)act, out_f32(out), N, M, (int)N, (int)K, cur_stream()), "gnnpn_linear_f32");
    return out;
}

Tensor embed_concat(const Tensor& x, const Tensor& table) {
    GNNPN_ON_DEVICE_OF(x);
    rows2d(x, "embed_concat.x"), rows2d(table, "embed_concat.table");
    const int64_t n = x.size(0), nfeat = x.size(1) - 1, vocab = table.size(0), emb = table.size(1);
    Tensor out = new_f32(x, {n, emb + nfeat});
    check_rc(gnnpn_embed_concat_f32(f32(x, "x"), f32(table, "table"), (int)vocab, (int)emb, (int)nfeat, out_f32(out), n, cur_stream()),
             "gnnpn_embed_concat_f32");
    return out;
}

// the one-wave-per-row gather form (any graph)
Tensor csr_aggregate(const Tensor& rowptr, const Tensor& col, const OptTensor& w, const Tensor& x, const OptTensor& self_coef,
                     const OptTensor& bias, const OptTensor& scale, const OptTensor& shift, int64_t act) {
    GNNPN_ON_DEVICE_OF(x);
    rows2d(x, "csr_aggregate.x");
    const int64_t n = rowptr.numel() - 1, C = x.size(1);
    Tensor y = new_f32(x, {n, C});
    check_rc(gnnpn_csr_aggregate_f32(i32(rowptr, "rowptr"), i32(col, "col"), f32(w, "w"), f32(x, "x"), C, f32(self_coef, "self_coef"),
                                     f32(bias, "bias"), f32(scale, "scale"), f32(shift, "shift"), (int)act, out_f32(y), C, (int32_t)n,
                                     (int32_t)C, cur_stream()), "gnnpn_csr_aggregate_f32");
    return y;
}
// block-local graphs, a block's channel slice staged whole in LDS
Tensor csr_aggregate_blocks(const Tensor& rowptr, const Tensor& col, const OptTensor& w, const Tensor& x, const OptTensor& self_coef,
                            const OptTensor& bias, const OptTensor& scale, const OptTensor& shift, int64_t act, int64_t block_rows,
                            const OptTensor& row_order) {
    GNNPN_ON_DEVICE_OF(x);
    rows2d(x, "csr_aggregate_blocks.x");
    const int64_t n = rowptr.numel() - 1, C = x.size(1);
    Tensor y = new_f32(x, {n, C});
    check_rc(gnnpn_csr_aggregate_blocks_f32(i32(rowptr, "rowptr"), i32(col, "col"), f32(w, "w"), f32(x, "x"), C,
                                            f32(self_coef, "self_coef"), f32(bias, "bias"), f32(scale, "scale"), f32(shift, "shift"),
                                            (int)act, out_f32(y), C, (int32_t)n, (int32_t)C, (int32_t)block_rows, i32(row_order, "row_order"),
                                            cur_stream()), "gnnpn_csr_aggregate_blocks_f32");
    return y;
}
// destination tile x source tile from a plan built by gnnpn_csr_tile_plan_* (ops.TilePlan)
Tensor csr_aggregate_tiled(const Tensor& header, const Tensor& order, const Tensor& selfw, const Tensor& batches, const Tensor& x,
                           const OptTensor& self_coef, const OptTensor& bias, const OptTensor& scale, const OptTensor& shift, int64_t act,
                           int64_t n_rows, int64_t block_rows) {
    GNNPN_ON_DEVICE_OF(x);
    rows2d(x, "csr_aggregate_tiled.x");
    const int64_t C = x.size(1);
    Tensor y = new_f32(x, {n_rows, C});
    check_rc(gnnpn_csr_aggregate_tiled_f32(i32(header, "header"), i32(order, "order"), f32(selfw, "selfw"),
                                           cptr<uint8_t>(batches, at::kByte, "batches"), f32(x, "x"), C, f32(self_coef, "self_coef"),
                                           f32(bias, "bias"), f32(scale, "scale"), f32(shift, "shift"), (int)act, out_f32(y), C,
                                           (int32_t)n_rows, (int32_t)C, (int32_t)block_rows, cur_stream()), "gnnpn_csr_aggregate_tiled_f32");
    return y;
}

Tensor gcn_norm(const Tensor& rowptr, const Tensor& col, const Tensor& w_raw) {
    GNNPN_ON_DEVICE_OF(w_raw);
    const int64_t n = rowptr.numel() - 1;
    Tensor dis = new_f32(w_raw, {n}), norm = at::empty_like(w_raw);
    check_rc(gnnpn_gcn_norm_f32(i32(rowptr, "rowptr"), i32(col, "col"), f32(w_raw, "w_raw"), out_f32(dis), out_f32(norm), (int32_t)n,
                                cur_stream()), "gnnpn_gcn_norm_f32");
    return norm;
}

Tensor segment_mean(const Tensor& segptr, const Tensor& x) {
    GNNPN_ON_DEVICE_OF(x);
    rows2d(x, "segment_mean.x");
    const int64_t n_seg = segptr.numel() - 1, C = x.size(1);
    Tensor out = new_f32(x, {n_seg, C});
    check_rc(gnnpn_segment_mean_f32(i32(segptr, "segptr"), f32(x, "x"), C, out_f32(out), C, (int32_t)n_seg, (int32_t)C, cur_stream()),
             "gnnpn_segment_mean_f32");
    return out;
}

// layer_tensors: per layer w0p, b0, a1, s1, w3p, b3, a2, s2, eps (custom_ops.REQUEST_LAYER_KEYS)
Tensor request_branch(const Tensor& x, const Tensor& table, const Tensor& rowptr, const Tensor& col, const Tensor& seg_ptr, int64_t max_nodes,
                      at::TensorList layer_tensors, const Tensor& lin_w_packed, const Tensor& lin_b, int64_t hidden) {
    GNNPN_ON_DEVICE_OF(x);
    rows2d(x, "request_branch.x");
    TORCH_CHECK(layer_tensors.size() % 9 == 0, "request_branch: nine tensors per layer");
    const size_t n_layers = layer_tensors.size() / 9;
    std::vector<gnnpn_gin_layer_t> layers(n_layers);
    for (size_t i = 0; i < n_layers; ++i) {
        const float* p[9];
        for (int j = 0; j < 9; ++j) p[j] = f32(layer_tensors[i * 9 + j], "request_branch.layer_tensors");
        layers[i] = gnnpn_gin_layer_t{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]};
    }
    const int64_t B = seg_ptr.numel() - 1;
    Tensor out = new_f32(x, {B, hidden});
    check_rc(gnnpn_request_branch_f32(f32(x, "x"), (int32_t)(x.size(1) - 1), f32(table, "table"), (int32_t)table.size(0),
                                      (int32_t)table.size(1), i32(rowptr, "rowptr"), i32(col, "col"), i32(seg_ptr, "seg_ptr"), (int32_t)B,
                                      (int32_t)max_nodes, (int32_t)n_layers, layers.data(), (int32_t)hidden, f32(lin_w_packed, "lin_w_packed"),
                                      f32(lin_b, "lin_b"), out_f32(out), cur_stream()), "gnnpn_request_branch_f32");
    return out;
}

Tensor gin_layer(const Tensor& rowptr, const Tensor& col, const Tensor& x, const Tensor& eps, const Tensor& w1, const OptTensor& b1,
                 const OptTensor& a1, const OptTensor& s1, const Tensor& w2, const OptTensor& b2, const OptTensor& a2, const OptTensor& s2,
                 const OptTensor& w3, const OptTensor& b3) {
    GNNPN_ON_DEVICE_OF(x);
    rows2d(x, "gin_layer.x");
    const int64_t n = x.size(0), c_in = x.size(1);
    const bool lin3 = w3.has_value() && w3->defined();
    TORCH_CHECK(w1.dim() == 3 && w2.dim() == 3 && (!lin3 || w3->dim() == 3) && w1.size(1) * 2 == (c_in + 31) / 32 * 32,
                "gin_layer: weights must be packed with ops.pack_mfma_b32 (and w1 for this many input channels)");
    const int64_t h1 = w1.size(0) * 32, h2 = w2.size(0) * 32, h3 = lin3 ? w3->size(0) * 32 : 0;
    Tensor out = new_f32(x, {n, lin3 ? h3 : h2});
    check_rc(gnnpn_gin_layer_f32(i32(rowptr, "rowptr"), i32(col, "col"), f32(x, "x"), c_in, (int32_t)c_in, f32(eps, "eps"), f32(w1, "w1"),
                                 f32(b1, "b1"), f32(a1, "a1"), f32(s1, "s1"), (int32_t)h1, f32(w2, "w2"), f32(b2, "b2"), f32(a2, "a2"),
                                 f32(s2, "s2"), (int32_t)h2, f32(w3, "w3"), f32(b3, "b3"), (int32_t)h3, out_f32(out), out.size(1), n,
                                 cur_stream()), "gnnpn_gin_layer_f32");
    return out;
}

Tensor gin_layer_split(const Tensor& rowptr, const Tensor& col, const Tensor& x, const Tensor& eps, const Tensor& w1, const Tensor& i1,
                       const OptTensor& b1, const OptTensor& a1, const OptTensor& s1, const Tensor& w2, const Tensor& i2, const OptTensor& b2,
                       const OptTensor& a2, const OptTensor& s2, const OptTensor& w3, const OptTensor& i3, const OptTensor& b3) {
    GNNPN_ON_DEVICE_OF(x);
    rows2d(x, "gin_layer_split.x");
    const int64_t n = x.size(0), c_in = x.size(1), k1 = (c_in + 31) / 32 * 32;
    const bool lin3 = w3.has_value() && w3->defined();
    const int64_t h1 = i1.numel(), h2 = i2.numel(), h3 = lin3 ? (i3.has_value() ? i3->numel() : 0) : 0;
    TORCH_CHECK(w1.numel() == gnnpn_split_weights_bytes((int32_t)h1, (int32_t)k1) && w2.numel() == gnnpn_split_weights_bytes((int32_t)h2, (int32_t)h1) &&
                    (!lin3 || w3->numel() == gnnpn_split_weights_bytes((int32_t)h3, (int32_t)h2)),
                "gin_layer_split: packed weights do not chain (c_in -> h1 -> h2 [-> h3]); pack with ops.pack_split_weights");
    Tensor out = new_f32(x, {n, lin3 ? h3 : h2});
    check_rc(gnnpn_gin_layer_split(i32(rowptr, "rowptr"), i32(col, "col"), f32(x, "x"), c_in, (int32_t)c_in, f32(eps, "eps"),
                                   cptr<uint8_t>(w1, at::kByte, "w1"), f32(i1, "i1"), f32(b1, "b1"), f32(a1, "a1"), f32(s1, "s1"), (int32_t)h1,
                                   cptr<uint8_t>(w2, at::kByte, "w2"), f32(i2, "i2"), f32(b2, "b2"), f32(a2, "a2"), f32(s2, "s2"), (int32_t)h2,
                                   cptr<uint8_t>(w3, at::kByte, "w3"), f32(i3, "i3"), f32(b3, "b3"), (int32_t)h3, out_f32(out), out.size(1), n,
                                   cur_stream()), "gnnpn_gin_layer_split");
    return out;
}

// ---- ranking / candidate reduction -------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> segment_topk_feasible(const Tensor& scores, const Tensor& cat_ptr, const Tensor& qos, const Tensor& local_bounds,
                                                 const Tensor& present, const Tensor& global_bounds, int64_t n_per) {
    GNNPN_ON_DEVICE_OF(scores);
    rows2d(scores, "select.scores");
    const int64_t B = scores.size(0), S = scores.size(1), T = cat_ptr.numel() - 1;
    TORCH_CHECK(qos.sizes() == at::IntArrayRef({S, 4}) && local_bounds.sizes() == at::IntArrayRef({B, T, 4}) &&
                    present.sizes() == at::IntArrayRef({B, T}) && global_bounds.sizes() == at::IntArrayRef({B, 4}),
                "select_candidates: inconsistent shapes");
    Tensor rows = new_f32(scores, {B, T * n_per, 8}), ids = new_i32(scores, {B, T * n_per});
    check_rc(gnnpn_select_candidates(f32(scores, "scores"), S, i32(cat_ptr, "cat_ptr"), cptr<double>(qos, at::kDouble, "qos"),
                                     cptr<double>(local_bounds, at::kDouble, "local_bounds"), cptr<uint8_t>(present, at::kByte, "present"),
                                     cptr<double>(global_bounds, at::kDouble, "global_bounds"), out_f32(rows),
                                     static_cast<int32_t*>(ids.data_ptr()), (int32_t)B, (int32_t)T, (int32_t)n_per, cur_stream()),
             "gnnpn_select_candidates");
    return {rows, ids};
}

Tensor rank_rows(const Tensor& scores) {
    GNNPN_ON_DEVICE_OF(scores);
    rows2d(scores, "rank_rows.scores");
    const int64_t B = scores.size(0), S = scores.size(1);
    Tensor ranking = new_i32(scores, {B, S});
    check_rc(gnnpn_rank_rows(f32(scores, "scores"), S, static_cast<int32_t*>(ranking.data_ptr()), (int32_t)B, (int32_t)S, cur_stream()),
             "gnnpn_rank_rows");
    return ranking;
}

Tensor precision_at_k(const Tensor& ranking, const Tensor& labels, at::IntArrayRef ks) {
    GNNPN_ON_DEVICE_OF(labels);
    const int64_t B = labels.size(0), S = labels.size(1);
    std::vector<int32_t> k32(ks.begin(), ks.end());
    Tensor kt = at::tensor(k32, at::TensorOptions().dtype(at::kInt)).to(labels.device());
    Tensor out = new_f32(labels, {B, (int64_t)ks.size()});
    check_rc(gnnpn_precision_at_k(i32(ranking, "ranking"), ranking.size(1), f32(labels, "labels"), S, (int32_t)B, (int32_t)S, i32(kt, "ks"),
                                  (int32_t)ks.size(), out_f32(out), cur_stream()), "gnnpn_precision_at_k");
    return out;
}

Tensor attention_logits(const Tensor& enc_out, const Tensor& queries, int64_t step, const Tensor& idx, double tanh_c, bool use_tanh) {
    GNNPN_ON_DEVICE_OF(enc_out);
    const int64_t B = enc_out.size(0), L = enc_out.size(1), H = enc_out.size(2), T = queries.size(1);
    f32(enc_out, "enc_out"), f32(queries, "queries");
    Tensor out = new_f32(enc_out, {B, L});
    check_rc(gnnpn_attention_logits_f32(f32(enc_out, "enc_out"), static_cast<const float*>(queries.data_ptr()) + step * H, T * H,
                                        i32(idx, "idx"), (float)tanh_c, use_tanh ? 1 : 0, out_f32(out), (int32_t)B, (int32_t)L, (int32_t)H,
                                        (int32_t)step, (int32_t)T, cur_stream()), "gnnpn_attention_logits_f32");
    return out;
}

Tensor qos_reward(const Tensor& actions, int64_t level) {
    GNNPN_ON_DEVICE_OF(actions);
    const int64_t B = actions.size(0), T = actions.size(1);
    Tensor R = new_f32(actions, {B});
    check_rc(gnnpn_qos_reward_f32(f32(actions, "actions"), out_f32(R), (int32_t)B, (int32_t)T, (int)level, cur_stream()), "gnnpn_qos_reward_f32");
    return R;
}

// ---- the recurrent kernels ---------------------------------------------------------------------------------------------
int precision_code(const std::string& p, const char* what) {
    if (p == "f32") return GNNPN_PREC_F32;
    if (p == "f16") return GNNPN_PREC_F16;
    if (p == "split") return GNNPN_PREC_SPLIT;
    TORCH_CHECK(false, what, ": unknown precision '", p, "'");
}
bool coop_supported(int64_t H, int64_t n_per, int64_t impl) { return H == 256 && n_per <= 16 && impl != 1; }
gnnpn_launch_opts_t launch_opts(int64_t impl, int64_t lds_kb, bool write_through, bool paired_start, const OptTensor& status) {
    gnnpn_launch_opts_t o{};
    o.impl = (int32_t)impl, o.lds_kb = (int32_t)lds_kb, o.write_through = write_through ? 1 : 0, o.paired_start = paired_start ? 1 : 0;
    TORCH_CHECK(!(status.has_value() && status->defined()) || status->numel() >= GNNPN_STATUS_WORDS,
                "status: a block of ", GNNPN_STATUS_WORDS, " int32 words (ops.Workspaces.status)");
    o.sticky_status = (status.has_value() && status->defined())
                          ? const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(cptr<int32_t>(*status, at::kInt, "status")))
                          : nullptr;
    return o;
}

// net_tensors: per net pregates, inputs, w_in, b_in, whh (packed), bhh, whh_split (custom_ops.ENCODE_KEYS; None = absent).
// workspace / status: ops.Workspaces.encode() and .status (required for the cooperative form, H = 256).
// -> enc_out x n, h_n x n, c_n x n
std::vector<Tensor> lstm_encode(const c10::List<OptTensor>& net_tensors, int64_t n_nets, std::string precision, int64_t impl, int64_t lds_kb,
                                bool write_through, const OptTensor& workspace, const OptTensor& status, bool paired_start) {
    constexpr size_t EK = 7;
    TORCH_CHECK(n_nets >= 1 && net_tensors.size() == EK * (size_t)n_nets, "lstm_encode: seven tensors per net (custom_ops.ENCODE_KEYS)");
    std::vector<OptTensor> t(net_tensors.size());
    for (size_t i = 0; i < t.size(); ++i) t[i] = net_tensors.get(i);
    auto has = [&](size_t i) { return t[i].has_value() && t[i]->defined(); };
    TORCH_CHECK(has(5), "lstm_encode: nets[0].bhh required");
    GNNPN_ON_DEVICE_OF(*t[5]);
    const int64_t H = t[5]->numel() / 4;
    const Tensor& first = has(0) ? *t[0] : *t[1];
    TORCH_CHECK(first.dim() == 3, "lstm_encode: pregates / inputs must be [B, L, .]");
    const int64_t B = first.size(0), L = first.size(1);
    const bool coop = coop_supported(H, 1, impl);
    const int prec = precision_code(precision, "lstm_encode");
    TORCH_CHECK(prec == GNNPN_PREC_F32 || coop, "lstm_encode: precision='", precision, "' needs the cooperative form (H = 256)");
    std::vector<gnnpn_encode_net_t> nets(n_nets);
    std::vector<Tensor> enc, hn, cn, keep;
    for (int64_t i = 0; i < n_nets; ++i) {
        const size_t o = EK * i;
        Tensor pre = has(o) ? *t[o] : Tensor();
        if (!pre.defined() && !coop) {                     // no in-kernel input projection in the streaming form: the same fma chain + bias
            TORCH_CHECK(has(o + 1) && has(o + 2) && has(o + 3), "lstm_encode: nets[", i, "] needs pregates or inputs + w_in + b_in");
            const Tensor& x = *t[o + 1];
            pre = linear(x.reshape({B * L, x.size(2)}), *t[o + 2], t[o + 3], std::nullopt, std::nullopt, 0).view({B, L, 4 * H});
            keep.push_back(pre);
        }
        enc.push_back(new_f32(first, {B, L, H})), hn.push_back(new_f32(first, {B, H})), cn.push_back(new_f32(first, {B, H}));
        gnnpn_encode_net_t& a = nets[i];
        a = gnnpn_encode_net_t{};
        a.pregates = pre.defined() ? f32(pre, "nets.pregates") : nullptr;
        if (!pre.defined()) {
            TORCH_CHECK(has(o + 1) && has(o + 2) && has(o + 3), "lstm_encode: nets[", i, "] needs pregates or inputs + w_in + b_in");
            a.inputs = f32(*t[o + 1], "nets.inputs"), a.w_in = f32(*t[o + 2], "nets.w_in"), a.b_in = f32(*t[o + 3], "nets.b_in");
        }
        TORCH_CHECK(has(o + 4) && has(o + 5), "lstm_encode: nets[", i, "].whh / bhh required");
        a.whh_packed = f32(*t[o + 4], "nets.whh"), a.bhh = f32(*t[o + 5], "nets.bhh");
        a.whh_split = has(o + 6) ? cptr<uint8_t>(*t[o + 6], at::kByte, "nets.whh_split") : nullptr;   // the exact split, made once per model
        a.enc_out = out_f32(enc.back()), a.h_n = out_f32(hn.back()), a.c_n = out_f32(cn.back());
    }
    const bool ws = coop && workspace.has_value() && workspace->defined();
    TORCH_CHECK(!coop || ws, "lstm_encode: the cooperative form (H = 256) needs a workspace (ops.Workspaces.encode())");
    const gnnpn_launch_opts_t opts = launch_opts(impl, lds_kb, write_through, paired_start, coop ? status : OptTensor());
    check_rc(gnnpn_lstm_encode_f32((int)n_nets, nets.data(), (int32_t)B, (int32_t)L, (int32_t)H, 8, prec, &opts,
                                   ws ? const_cast<uint8_t*>(cptr<uint8_t>(*workspace, at::kByte, "workspace")) : nullptr,
                                   ws ? workspace->numel() : 0, cur_stream()), "gnnpn_lstm_encode_f32");
    std::vector<Tensor> out;
    out.insert(out.end(), enc.begin(), enc.end()), out.insert(out.end(), hn.begin(), hn.end()), out.insert(out.end(), cn.begin(), cn.end());
    return out;
}

// net_tensors: per net enc_out, h0, c0, start, wih, whh, bih, bhh, embedded, emb_w, emb_b, xw_fold, xb_fold, start_fold, latent_win, whh_split
// (custom_ops.DECODE_KEYS).  -> per net idx, win_logits, pick_prob, actions, queries (empty when not wanted)
std::vector<Tensor> pointer_decode(const c10::List<OptTensor>& net_tensors, at::IntArrayRef latent_from, const Tensor& inputs, int64_t n_cat,
                                   int64_t n_per, double tanh_c, bool use_tanh, bool want_queries, std::string precision, int64_t impl,
                                   int64_t lds_kb, bool write_through, const OptTensor& workspace, const OptTensor& status,
                                   at::IntArrayRef sample_seeds, bool paired_start) {
    GNNPN_ON_DEVICE_OF(inputs);
    constexpr size_t NK = 16;
    const size_t n_nets = latent_from.size();
    TORCH_CHECK(n_nets >= 1 && n_nets <= 2 && net_tensors.size() == NK * n_nets, "pointer_decode: sixteen tensors per net (custom_ops.DECODE_KEYS), 1 or 2 nets");
    std::vector<OptTensor> t(net_tensors.size());
    for (size_t i = 0; i < t.size(); ++i) t[i] = net_tensors.get(i);
    auto has = [&](size_t i) { return t[i].has_value() && t[i]->defined(); };
    TORCH_CHECK(has(0) && t[0]->dim() == 3, "pointer_decode: nets[0].enc_out [B, L, H] required");
    const int64_t B = t[0]->size(0), L = t[0]->size(1), H = t[0]->size(2);
    const int prec = precision_code(precision, "pointer_decode");
    TORCH_CHECK(L == n_cat * n_per, "pointer_decode: seq_len ", L, " != ", n_cat, "*", n_per);
    const bool coop = coop_supported(H, n_per, impl);
    std::vector<gnnpn_decode_net_t> nets(n_nets);
    std::vector<Tensor> out, keep;
    for (size_t i = 0; i < n_nets; ++i) {
        const size_t o = NK * i;
        gnnpn_decode_net_t& a = nets[i];
        a = gnnpn_decode_net_t{};
        for (size_t j = 0; j < 8; ++j) TORCH_CHECK(has(o + j), "pointer_decode: nets[", i, "] operand ", j, " required");
        a.enc_out = f32(*t[o + 0], "enc_out"), a.h0 = f32(*t[o + 1], "h0"), a.c0 = f32(*t[o + 2], "c0"), a.start = f32(*t[o + 3], "start");
        a.wih_packed = f32(*t[o + 4], "wih"), a.whh_packed = f32(*t[o + 5], "whh"), a.bih = f32(*t[o + 6], "bih"), a.bhh = f32(*t[o + 7], "bhh");
        if (has(o + 11) && coop) {                           // folded input side (cooperative form)
            TORCH_CHECK(has(o + 12) && has(o + 13), "pointer_decode: xw_fold needs xb_fold and start_fold");
            a.xw_fold = f32(*t[o + 11], "xw_fold"), a.xb_fold = f32(*t[o + 12], "xb_fold"), a.start_fold = f32(*t[o + 13], "start_fold");
        }
        Tensor emb = has(o + 8) ? *t[o + 8] : Tensor();
        if (!emb.defined() && !coop) {                       // no in-kernel embedding there
            TORCH_CHECK(has(o + 9) && has(o + 10), "pointer_decode: nets[", i, "] needs embedded or emb_w + emb_b");
            emb = linear(inputs.reshape({B * L, inputs.size(2)}), *t[o + 9], t[o + 10], std::nullopt, std::nullopt, 0).view({B, L, H});
            keep.push_back(emb);
        }
        a.embedded = emb.defined() ? f32(emb, "embedded") : nullptr;
        if (!emb.defined()) {
            TORCH_CHECK(has(o + 9) && has(o + 10), "pointer_decode: nets[", i, "] needs embedded or emb_w + emb_b");
            a.emb_w = f32(*t[o + 9], "emb_w"), a.emb_b = f32(*t[o + 10], "emb_b");
        }
        a.latent_win = has(o + 14) ? f32(*t[o + 14], "latent_win") : nullptr;
        a.whh_split = has(o + 15) ? cptr<uint8_t>(*t[o + 15], at::kByte, "whh_split") : nullptr;
        a.latent_from = (int32_t)latent_from[i];
        const int64_t seed = i < sample_seeds.size() ? sample_seeds[i] : -1;     // >= 0: draw this net's picks from the stream of that seed
        a.sample = seed >= 0 ? 1 : 0;
        a.sample_seed = seed >= 0 ? (uint64_t)seed : 0;
        Tensor idx = new_i32(inputs, {B, n_cat}), win = new_f32(inputs, {B, n_cat, n_per}), prob = new_f32(inputs, {B, n_cat});
        Tensor actions = new_f32(inputs, {B, n_cat, 8}), q = want_queries ? new_f32(inputs, {B, n_cat, H}) : new_f32(inputs, {0});
        a.idx = static_cast<int32_t*>(idx.data_ptr()), a.win_logits = out_f32(win), a.pick_prob = out_f32(prob), a.actions = out_f32(actions);
        a.queries = want_queries ? out_f32(q) : nullptr;
        out.push_back(idx), out.push_back(win), out.push_back(prob), out.push_back(actions), out.push_back(q);
    }
    const bool ws = coop && workspace.has_value() && workspace->defined();
    TORCH_CHECK(!coop || ws, "pointer_decode: the cooperative form needs a workspace (ops.Workspaces.decode(B, T, n_per))");
    const gnnpn_launch_opts_t opts = launch_opts(impl, lds_kb, write_through, paired_start, coop ? status : OptTensor());
    check_rc(gnnpn_pointer_decode_f32((int)n_nets, nets.data(), f32(inputs, "inputs"), (float)tanh_c, use_tanh ? 1 : 0, (int32_t)B, (int32_t)n_cat,
                                      (int32_t)n_per, (int32_t)H, prec == GNNPN_PREC_SPLIT ? GNNPN_PREC_SPLIT : GNNPN_PREC_F32, &opts,
                                      ws ? const_cast<uint8_t*>(cptr<uint8_t>(*workspace, at::kByte, "workspace")) : nullptr,
                                      ws ? workspace->numel() : 0, cur_stream()), "gnnpn_pointer_decode_f32");
    return out;
}

}  // namespace

TORCH_LIBRARY(gnnpn, m) {
    m.def("linear(Tensor a, Tensor weight, Tensor? bias=None, Tensor? scale=None, Tensor? shift=None, int act=0) -> Tensor");
    m.def("embed_concat(Tensor x, Tensor table) -> Tensor");
    m.def("csr_aggregate(Tensor rowptr, Tensor col, Tensor? w, Tensor x, Tensor? self_coef=None, Tensor? bias=None, Tensor? scale=None, "
          "Tensor? shift=None, int act=0) -> Tensor");
    m.def("csr_aggregate_blocks(Tensor rowptr, Tensor col, Tensor? w, Tensor x, Tensor? self_coef, Tensor? bias, Tensor? scale, "
          "Tensor? shift, int act, int block_rows, Tensor? row_order=None) -> Tensor");
    m.def("csr_aggregate_tiled(Tensor header, Tensor order, Tensor selfw, Tensor batches, Tensor x, Tensor? self_coef, Tensor? bias, "
          "Tensor? scale, Tensor? shift, int act, int n_rows, int block_rows) -> Tensor");
    m.def("gcn_norm(Tensor rowptr, Tensor col, Tensor w_raw) -> Tensor");
    m.def("segment_mean(Tensor segptr, Tensor x) -> Tensor");
    m.def("request_branch(Tensor x, Tensor table, Tensor rowptr, Tensor col, Tensor seg_ptr, int max_nodes, Tensor[] layer_tensors, "
          "Tensor lin_w_packed, Tensor lin_b, int hidden) -> Tensor");
    m.def("gin_layer(Tensor rowptr, Tensor col, Tensor x, Tensor eps, Tensor w1, Tensor? b1, Tensor? a1, Tensor? s1, Tensor w2, Tensor? b2, "
          "Tensor? a2, Tensor? s2, Tensor? w3=None, Tensor? b3=None) -> Tensor");
    m.def("gin_layer_split(Tensor rowptr, Tensor col, Tensor x, Tensor eps, Tensor w1, Tensor i1, Tensor? b1, Tensor? a1, Tensor? s1, "
          "Tensor w2, Tensor i2, Tensor? b2, Tensor? a2, Tensor? s2, Tensor? w3=None, Tensor? i3=None, Tensor? b3=None) -> Tensor");
    m.def("segment_topk_feasible(Tensor scores, Tensor cat_ptr, Tensor qos, Tensor local_bounds, Tensor present, Tensor global_bounds, "
          "int n_per) -> (Tensor, Tensor)");
    m.def("rank_rows(Tensor scores) -> Tensor");
    m.def("precision_at_k(Tensor ranking, Tensor labels, int[] ks) -> Tensor");
    m.def("attention_logits(Tensor enc_out, Tensor queries, int step, Tensor idx, float tanh_c, bool use_tanh) -> Tensor");
    m.def("qos_reward(Tensor actions, int level) -> Tensor");
    m.def("lstm_encode(Tensor?[] net_tensors, int n_nets, str precision='f32', int impl=0, int lds_kb=0, bool write_through=False, "
          "Tensor? workspace=None, Tensor? status=None, bool paired_start=False) -> Tensor[]");
    m.def("pointer_decode(Tensor?[] net_tensors, int[] latent_from, Tensor inputs, int n_cat, int n_per, float tanh_c=10.0, "
          "bool use_tanh=True, bool want_queries=False, str precision='f32', int impl=0, int lds_kb=0, bool write_through=False, "
          "Tensor? workspace=None, Tensor? status=None, int[] sample_seeds=[], bool paired_start=False) -> Tensor[]");
}

TORCH_LIBRARY_IMPL(gnnpn, CUDA, m) {
    m.impl("linear", linear);
    m.impl("embed_concat", embed_concat);
    m.impl("csr_aggregate", csr_aggregate);
    m.impl("csr_aggregate_blocks", csr_aggregate_blocks);
    m.impl("csr_aggregate_tiled", csr_aggregate_tiled);
    m.impl("gcn_norm", gcn_norm);
    m.impl("segment_mean", segment_mean);
    m.impl("request_branch", request_branch);
    m.impl("gin_layer", gin_layer);
    m.impl("gin_layer_split", gin_layer_split);
    m.impl("segment_topk_feasible", segment_topk_feasible);
    m.impl("rank_rows", rank_rows);
    m.impl("precision_at_k", precision_at_k);
    m.impl("attention_logits", attention_logits);
    m.impl("qos_reward", qos_reward);
    m.impl("lstm_encode", lstm_encode);
    m.impl("pointer_decode", pointer_decode);
}
