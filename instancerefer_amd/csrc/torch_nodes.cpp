// torch_nodes.cpp — C++ autograd nodes over the C-ABI of libirx (include/irx.h) for the dense heads.
//
// The training step of the bf16 headline configuration is HOST-bound (every main-stream event of the
// pipelined loop completes as it is issued). A torch.autograd.Function written in Python costs 30 us (forward) to 100 us
// (backward, entered from the C++ engine through the GIL) of interpreter time per node — more than the handful of ATen
// dispatches a fused head operator replaces (the round-4 negative result of dense.MLP2Fn). The same operators behind
// torch::autograd::Function cost a few microseconds: tensors are allocated with ATen, the kernels are the C-ABI entry points
// of libirx (handed over as addresses by _lib.py — this module does not link the library, so IRX_LIB_PATH builds keep
// working), parameter gradients go straight into the optimizer's flat gradient buffer (the gradient-sink protocol of
// optim.FlatAdam: slot addresses in, one delivered flag per producer out — no AccumulateGrad nodes).
//
// This file is binding plumbing: no arithmetic happens here. Reference semantics of the operators: models/attribute_module.py:26-34,
// relation_module.py:18-27, scene_module.py:38-42 (nn.Sequential(Linear, BatchNorm1d | LayerNorm, ReLU, [Dropout], Linear)).
#include <torch/extension.h>

#include "torch_nodes.h"

#include <ATen/SequenceNumber.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

typedef size_t (*saved_floats_fn)(int, int);
typedef int (*mlp2_fwd_fn)(const float*, int, int, int, int, const float*, const float*, int, const float*, const float*, float,
                           float*, float*, float, float, unsigned long long, const float*, const float*, float*, float*, void*);
typedef int (*mlp2_bwd_fn)(const float*, const float*, int, int, int, int, const float*, int, const float*, const float*,
                           const float*, float, float*, float*, float*, float*, float*, float*, float*, float*, void*);
typedef int (*gru_fwd_fn)(const float*, const int32_t*, const float*, const float*, int, int, int, int, float*, float*, void*);
typedef int (*gru_bwd_fn)(const float*, const float*, const float*, const int32_t*, const float*, int, int, int, int, float*,
                          float*, void*);
typedef int (*gru_wgrad_fn)(const float*, const float*, const float*, const float*, int, int, int, int, int, float*, float*, float*,
                            float*, float*, float*, float*, float*, void*);
using irxn::last_error_fn;
typedef size_t (*hash_capacity_fn)(int);
typedef int (*hash_build_fn)(const uint64_t*, int, uint64_t*, int32_t*, size_t, void*);
typedef int (*kmap_s1_fn)(const int32_t*, int, int, const uint64_t*, const int32_t*, size_t, int32_t*, int, void*);
typedef int (*kmaps_multi_fn)(int, const uint64_t* const*, const int32_t* const*, const int*, const int*, uint64_t* const*, int32_t* const*,
                              const size_t*, int32_t* const*, const int*, void*);

typedef int (*kmaps_pyramid_fn)(int, const int*, const int*, const uint64_t*, const int32_t*, uint64_t*, int32_t*, size_t, const int32_t* const*,
                                const uint8_t* const*, const int32_t* const*, const int*, int32_t* const*, const int*, void*);

typedef size_t (*pairs_ws_fn)(int, int);
typedef int (*pairs_multi_fn)(int, const int32_t* const*, const int*, const int*, const int*, int32_t* const*, int32_t* const*, const int*,
                              int32_t* const*, void*, size_t, void*);
typedef int (*down_transpose_fn)(const int32_t*, const uint8_t*, int, int32_t*, int, void*);

struct Api {
  saved_floats_fn mlp2_saved_floats = nullptr;
  mlp2_fwd_fn mlp2_fwd = nullptr;
  mlp2_bwd_fn mlp2_bwd = nullptr;
  gru_fwd_fn gru_fwd = nullptr;
  gru_bwd_fn gru_bwd = nullptr;
  gru_wgrad_fn gru_wgrad = nullptr;
  last_error_fn last_error = nullptr;
  hash_capacity_fn hash_capacity = nullptr;
  hash_build_fn hash_build = nullptr;
  kmap_s1_fn kmap_s1 = nullptr;
  kmaps_multi_fn kmaps_multi = nullptr;
  kmaps_pyramid_fn kmaps_pyramid = nullptr;
  pairs_ws_fn pairs_ws = nullptr;
  pairs_multi_fn pairs_multi = nullptr;
  down_transpose_fn down_transpose = nullptr;
} g_api;

using irxn::check;
using irxn::Sink;

// the sink's keep-alive tensors (flat gradient buffer, record tensor) ride in the node's saved_data
inline void keep_alive(AutogradContext* ctx, const c10::optional<Tensor>& a, const c10::optional<Tensor>& b) {
  if (a.has_value() && a->defined()) ctx->saved_data["keep_a"] = *a;
  if (b.has_value() && b->defined()) ctx->saved_data["keep_b"] = *b;
}
inline c10::optional<Tensor> opt_at(const std::vector<Tensor>& v, size_t i) {
  return i < v.size() ? c10::optional<Tensor>(v[i]) : c10::optional<Tensor>();
}

using irxn::fp;
using irxn::fpm;

// y = W2 . D(relu(N(W1 x + b1))) + b2, irx_mlp2_fwd / irx_mlp2_bwd.  slot_ptrs: addresses of the six gradient slots
// (w1, b1, gamma, beta, w2, b2) in the optimizer's flat buffer + the Sink trailer, or empty (see Sink above); a second backward
// before the records are cleared, a backward after the optimizer was replaced, or a backward
// without slots, returns ordinary gradient tensors.
struct MLP2Node : public torch::autograd::Function<MLP2Node> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_in, const Tensor& w1, const Tensor& b1,
                        const c10::optional<Tensor>& gamma_o, const c10::optional<Tensor>& beta_o, const Tensor& w2,
                        const Tensor& b2, int64_t norm, double eps,
                        const c10::optional<Tensor>& rmean, const c10::optional<Tensor>& rvar, double momentum, double drop_p,
                        int64_t seed, int64_t stream, std::vector<int64_t> slot_ptrs, const c10::optional<Tensor>& keep_a,
                        const c10::optional<Tensor>& keep_b) {
    const Tensor x = x_in.contiguous().to(torch::kFloat32);
    const Tensor gamma = gamma_o.has_value() ? *gamma_o : Tensor(), beta = beta_o.has_value() ? *beta_o : Tensor();
    const int rows = (int)x.size(0), din = (int)x.size(1), dh = (int)w1.size(0), dout = (int)w2.size(0);
    Tensor y = torch::empty({rows, dout}, x.options());
    Tensor saved = torch::empty({(int64_t)g_api.mlp2_saved_floats(rows, dh)}, x.options());
    check(g_api.mlp2_fwd(fp(x), rows, din, dh, dout, fp(w1), fp(b1), (int)norm, fp(gamma), fp(beta), (float)eps,
                         rmean.has_value() ? fpm(*rmean) : nullptr, rvar.has_value() ? fpm(*rvar) : nullptr, (float)momentum,
                         (float)drop_p, (unsigned long long)seed, fp(w2), fp(b2), fpm(saved), fpm(y), (void*)stream),
          "irx_mlp2_fwd");
    ctx->save_for_backward({x, w1, gamma, w2, saved, (norm & 8) ? y : Tensor()});    // (+ 8: ReLU on y — the backward masks dy with it)
    ctx->saved_data["norm"] = norm;
    ctx->saved_data["drop_p"] = drop_p;
    ctx->saved_data["stream"] = stream;          // (the engine replays the node on its forward stream)
    keep_alive(ctx, keep_a, keep_b);
    ctx->saved_data["slots"] = slot_ptrs;
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    const auto sv = ctx->get_saved_variables();
    const Tensor &x = sv[0], &w1 = sv[1], &gamma = sv[2], &w2 = sv[3], &saved = sv[4];
    const int rows = (int)x.size(0), din = (int)x.size(1), dh = (int)w1.size(0), dout = (int)w2.size(0);
    const int64_t norm = ctx->saved_data["norm"].toInt();
    const double drop_p = ctx->saved_data["drop_p"].toDouble();
    void* stream = (void*)ctx->saved_data["stream"].toInt();
    const Sink sink(ctx->saved_data["slots"].toIntVector());
    const auto& slots = sink.slots;
    Tensor dy = grad_out[0].contiguous().to(torch::kFloat32);
    if (norm & 8) dy = at::threshold_backward(dy, sv[5], 0);          // the output ReLU
    const bool has_norm = (norm & 7) != 4;                            // 4: no normalisation layer, no gamma / beta
    const bool want_dx = ctx->needs_input_grad(0);
    Tensor scratch = torch::empty({(int64_t)rows * (dh + (want_dx ? din : 0))}, x.options());
    float* base = scratch.data_ptr<float>();
    float* dx_ptr = want_dx ? base + (size_t)rows * dh : nullptr;
    // slots: (w1, b1, gamma, beta, w2, b2), or (w1, b1, w2, b2) without a normalisation layer
    const bool deliver = sink.deliver(has_norm ? 6u : 4u);
    float* gp[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    variable_list out(18);
    if (deliver) {
      int q = 0;
      for (int i = 0; i < 6; ++i)
        if (has_norm || (i != 2 && i != 3)) gp[i] = (float*)slots[q++];
    } else {
      const int64_t sizes[6] = {(int64_t)dh * din, dh, has_norm ? dh : 0, has_norm ? dh : 0, (int64_t)dout * dh, dout};
      int64_t total = 0;
      for (int i = 0; i < 6; ++i) total += sizes[i];
      Tensor buf = torch::empty({total}, x.options());
      int64_t off = 0;
      for (int i = 0; i < 6; ++i) {
        if (sizes[i] == 0) continue;
        gp[i] = buf.data_ptr<float>() + off;
        Tensor v = buf.narrow(0, off, sizes[i]);
        out[1 + i] = (i == 0) ? v.view({dh, din}) : (i == 4 ? v.view({dout, dh}) : v);
        off += sizes[i];
      }
    }
    check(g_api.mlp2_bwd(fp(x), fp(dy), rows, din, dh, dout, fp(w1), (int)norm, fp(gamma), fp(w2), fp(saved),
                         drop_p > 0 ? (float)(1.0 / (1.0 - drop_p)) : 1.f, base, dx_ptr, gp[0], gp[1], gp[2], gp[3], gp[4], gp[5],
                         stream),
          "irx_mlp2_bwd");
    if (deliver) sink.delivered(stream);
    if (want_dx) out[0] = scratch.narrow(0, (int64_t)rows * dh, (int64_t)rows * din).view({rows, din});
    return out;
  }
};

Tensor mlp2(const Tensor& x, const Tensor& w1, const Tensor& b1, const Tensor& gamma, const Tensor& beta, const Tensor& w2,
            const Tensor& b2, int64_t norm, double eps, const c10::optional<Tensor>& rmean, const c10::optional<Tensor>& rvar,
            double momentum, double drop_p, int64_t seed, int64_t stream, std::vector<int64_t> slot_ptrs, std::vector<Tensor> keep) {
  TORCH_CHECK(g_api.mlp2_fwd && g_api.mlp2_bwd && g_api.mlp2_saved_floats, "irx nodes: bind() has not been called");
  return MLP2Node::apply(x, w1, b1, c10::optional<Tensor>(gamma), c10::optional<Tensor>(beta), w2, b2, norm, eps, rmean, rvar, momentum,
                         drop_p, seed, stream,
                         std::move(slot_ptrs), opt_at(keep, 0), opt_at(keep, 1));
}

// One GRU layer, both directions (reference models/lang_module.py:24-32,58-60: nn.GRU over packed sequences; dense.gru_packed).
// The recurrence and its BPTT are irx_gru_forward / irx_gru_backward; the input / hidden projections and their weight gradients
// are library GEMMs issued from here (ATen called from C++: no interpreter, no autograd recording), the four weight gradients
// one launch of irx_gru_wgrad. params = per direction
// (w_ih [3H][I], w_hh [3H][H], b_ih [3H], b_hh [3H]) — nn.GRU's own parameter tensors, so no cat / stack nodes sit between the
// parameters and this node; with slot_ptrs (4 per direction, same order) the weight gradients are written straight into the
// optimizer's flat buffer.
struct GRULayerNode : public torch::autograd::Function<GRULayerNode> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_in, const Tensor& len32, const Tensor& p0, const Tensor& p1,
                        const Tensor& p2, const Tensor& p3, const c10::optional<Tensor>& p4, const c10::optional<Tensor>& p5,
                        const c10::optional<Tensor>& p6, const c10::optional<Tensor>& p7, int64_t stream,
                        std::vector<int64_t> slot_ptrs, const c10::optional<Tensor>& keep_a, const c10::optional<Tensor>& keep_b) {
    // (a std::vector<Tensor> argument would be ONE input of the node: the parameters are separate arguments, the second
    //  direction's are optional)
    std::vector<Tensor> params = {p0, p1, p2, p3};
    if (p4.has_value() && p4->defined()) {
      params.push_back(*p4); params.push_back(*p5); params.push_back(*p6); params.push_back(*p7);
    }
    const int ndir = (int)params.size() / 4;
    const Tensor x = x_in.contiguous().to(torch::kFloat32);
    const int B = (int)x.size(0), T = (int)x.size(1), I = (int)x.size(2), H = (int)params[1].size(1);
    const Tensor x2 = x.view({(int64_t)B * T, I});
    std::vector<Tensor> wi, wh, bi, bh;
    for (int d = 0; d < ndir; ++d) {
      wi.push_back(params[4 * d]); wh.push_back(params[4 * d + 1]); bi.push_back(params[4 * d + 2]); bh.push_back(params[4 * d + 3]);
    }
    const Tensor w_ih = ndir == 1 ? wi[0] : at::cat(wi, 0);            // (ndir*3H, I)
    const Tensor b_ih = ndir == 1 ? bi[0] : at::cat(bi, 0);
    const Tensor w_hh = (ndir == 1 ? wh[0].unsqueeze(0) : at::stack(wh, 0)).contiguous();      // (ndir, 3H, H)
    const Tensor b_hh = (ndir == 1 ? bh[0].unsqueeze(0) : at::stack(bh, 0)).contiguous();
    const Tensor gi = at::addmm(b_ih, x2, w_ih.t());                  // (B*T, ndir*3H), contiguous
    Tensor out = torch::empty({B, T, (int64_t)ndir * H}, x.options());
    Tensor gates = torch::empty({B, T, ndir, (int64_t)4 * H}, x.options());
    check(g_api.gru_fwd(fp(gi), len32.data_ptr<int32_t>(), fp(w_hh), fp(b_hh), B, T, ndir, H, fpm(out), fpm(gates), (void*)stream),
          "irx_gru_forward");
    ctx->save_for_backward({x2, len32, w_ih, w_hh, out, gates});
    ctx->saved_data["dims"] = std::vector<int64_t>{B, T, I, ndir, H};
    ctx->saved_data["stream"] = stream;
    keep_alive(ctx, keep_a, keep_b);
    ctx->saved_data["slots"] = slot_ptrs;
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_out) {
    const auto sv = ctx->get_saved_variables();
    const Tensor &x2 = sv[0], &len32 = sv[1], &w_ih = sv[2], &w_hh = sv[3], &out = sv[4], &gates = sv[5];
    const auto dims = ctx->saved_data["dims"].toIntVector();
    const int B = (int)dims[0], T = (int)dims[1], I = (int)dims[2], ndir = (int)dims[3], H = (int)dims[4];
    void* stream = (void*)ctx->saved_data["stream"].toInt();
    const Sink sink(ctx->saved_data["slots"].toIntVector());
    const auto& slots = sink.slots;
    const Tensor dout = grad_out[0].contiguous().to(torch::kFloat32);
    const auto opt = x2.options();
    Tensor dgi = torch::empty({B, T, ndir, (int64_t)3 * H}, opt), dgh = torch::empty({B, T, ndir, (int64_t)3 * H}, opt);
    check(g_api.gru_bwd(fp(dout), fp(out), fp(gates), len32.data_ptr<int32_t>(), fp(w_hh), B, T, ndir, H, fpm(dgi), fpm(dgh),
                        stream),
          "irx_gru_backward");
    const int64_t BT = (int64_t)B * T, G = 3 * (int64_t)H;
    const Tensor dgi2 = dgi.view({BT, ndir * G});
    variable_list res(2 + 8 + 4);      // (stream, slot_ptrs, keep_a, keep_b: no gradients)
    // layout of the returned list = forward's arguments: x, len32, 8 parameters, stream, slot_ptrs, keep_a, keep_b
    if (ctx->needs_input_grad(0)) res[0] = dgi2.mm(w_ih).view({B, T, I});
    // the four weight gradients of both directions in one launch (irx_gru_wgrad reads h_{t-1} from `out` with the direction's
    // shift): straight into the optimizer's slots, or into fresh tensors handed to autograd
    const bool deliver = sink.deliver((size_t)4 * ndir);
    float* gp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // per direction: w_ih, w_hh, b_ih, b_hh
    for (int d = 0; d < ndir; ++d) {
      if (deliver) {
        for (int q = 0; q < 4; ++q) gp[4 * d + q] = (float*)slots[4 * d + q];
      } else {
        Tensor g_wi = torch::empty({G, I}, opt), g_wh = torch::empty({G, H}, opt), g_bi = torch::empty({G}, opt),
               g_bh = torch::empty({G}, opt);
        gp[4 * d] = g_wi.data_ptr<float>(); gp[4 * d + 1] = g_wh.data_ptr<float>();
        gp[4 * d + 2] = g_bi.data_ptr<float>(); gp[4 * d + 3] = g_bh.data_ptr<float>();
        res[2 + 4 * d] = g_wi; res[2 + 4 * d + 1] = g_wh; res[2 + 4 * d + 2] = g_bi; res[2 + 4 * d + 3] = g_bh;
      }
    }
    check(g_api.gru_wgrad(fp(dgi), fp(dgh), fp(x2), fp(out), B, T, I, ndir, H, gp[0], gp[4], gp[1], gp[5], gp[2], gp[6], gp[3], gp[7],
                          stream),
          "irx_gru_wgrad");
    if (deliver) sink.delivered(stream);
    return res;
  }
};

Tensor gru_layer(const Tensor& x, const Tensor& len32, std::vector<Tensor> params, int64_t stream, std::vector<int64_t> slot_ptrs,
                 std::vector<Tensor> keep) {
  TORCH_CHECK(g_api.gru_fwd && g_api.gru_bwd, "irx nodes: bind() has not been called");
  TORCH_CHECK(params.size() == 4 || params.size() == 8, "gru_layer: 4 parameters per direction, 1 or 2 directions");
  c10::optional<Tensor> q[4];
  if (params.size() == 8)
    for (int i = 0; i < 4; ++i) q[i] = params[4 + i];
  return GRULayerNode::apply(x, len32, params[0], params[1], params[2], params[3], q[0], q[1], q[2], q[3], stream,
                             std::move(slot_ptrs), opt_at(keep, 0), opt_at(keep, 1));
}

// nn.Sequential(Linear, ReLU, Dropout, Linear, ReLU) (reference models/lang_module.py:33-37, the word projection) on the same
// operator: norm 4 (none) + 8 (ReLU on the output); slot_ptrs: (w1, b1, w2, b2).
Tensor mlp_relu2(const Tensor& x, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, double drop_p, int64_t seed,
                 int64_t stream, std::vector<int64_t> slot_ptrs, std::vector<Tensor> keep) {
  TORCH_CHECK(g_api.mlp2_fwd && g_api.mlp2_bwd && g_api.mlp2_saved_floats, "irx nodes: bind() has not been called");
  return MLP2Node::apply(x, w1, b1, c10::optional<Tensor>(), c10::optional<Tensor>(), w2, b2, (int64_t)(4 | 8), 0.0, c10::optional<Tensor>(),
                         c10::optional<Tensor>(), 0.0, drop_p, seed, stream, std::move(slot_ptrs), opt_at(keep, 0), opt_at(keep, 1));
}

// The hash tables and 27-neighbour tables of every level of a coordinate pyramid in ONE call (reference: torchsparse builds a
// kernel map per (stride, kernel size) lazily inside spnn.Conv3d — models/basic_blocks.py:14,32-39): per level irx_hash_build +
// irx_kmap_build_s1 on tensors allocated here with ATen. From Python each level was two ctypes calls and three torch.empty —
// 0.4-0.6 ms of interpreter time per step at the head of the forward (5 levels x 2 encoders; round 5 host profile). Levels whose
// `have` flag is set, or that hold no rows, are skipped. -> per level (table keys int64 [cap], table values int32 [cap],
// neighbours int32 [27][max(n, 1)]) or three undefined tensors.
std::vector<Tensor> kmaps_build(const std::vector<Tensor>& keys, const std::vector<Tensor>& coords, const std::vector<int64_t>& strides,
                                const std::vector<int64_t>& have, int64_t stream) {
  TORCH_CHECK(g_api.kmaps_multi && g_api.hash_capacity, "irx nodes: bind() has not been called");
  TORCH_CHECK(keys.size() == coords.size() && keys.size() == strides.size() && keys.size() == have.size(), "kmaps_build: list sizes");
  const size_t nl = keys.size();
  std::vector<Tensor> out(3 * nl);
  // round 6: ONE library call (three launches) for every level that still needs its tables, three allocations shared by the levels
  std::vector<size_t> todo;
  int64_t cap_total = 0, nbr_total = 0;
  std::vector<size_t> caps;
  for (size_t l = 0; l < nl; ++l) {
    const int64_t n = coords[l].size(0);
    if (have[l] || n == 0) continue;
    todo.push_back(l);
    caps.push_back(g_api.hash_capacity((int)n));
    cap_total += (int64_t)caps.back();
    nbr_total += 27 * ((n + 63) / 64 * 64);                 // (every level's table starts on a 256-byte boundary)
  }
  if (todo.empty()) return out;
  TORCH_CHECK(todo.size() <= 8, "kmaps_build: more than 8 levels");
  const auto opt = keys[todo[0]].options();
  Tensor tk_all = torch::empty({cap_total}, opt.dtype(torch::kInt64));
  Tensor tv_all = torch::empty({cap_total}, opt.dtype(torch::kInt32));
  Tensor nbr_all = torch::empty({nbr_total}, opt.dtype(torch::kInt32));
  const uint64_t* kp[8]; const int32_t* cp[8]; int n_[8], st[8], ld[8]; uint64_t* tkp[8]; int32_t* tvp[8]; size_t cap[8]; int32_t* np_[8];
  int64_t co = 0, no = 0;
  for (size_t i = 0; i < todo.size(); ++i) {
    const size_t l = todo[i];
    const int64_t n = coords[l].size(0), ldl = (n + 63) / 64 * 64;
    Tensor tk = tk_all.narrow(0, co, (int64_t)caps[i]), tv = tv_all.narrow(0, co, (int64_t)caps[i]);
    Tensor nbr = nbr_all.narrow(0, no, 27 * ldl).view({27, ldl});
    co += (int64_t)caps[i];
    no += 27 * ldl;
    kp[i] = (const uint64_t*)keys[l].data_ptr<int64_t>();
    cp[i] = coords[l].data_ptr<int32_t>();
    n_[i] = (int)n; st[i] = (int)strides[l]; ld[i] = (int)ldl;
    tkp[i] = (uint64_t*)tk.data_ptr<int64_t>(); tvp[i] = tv.data_ptr<int32_t>(); cap[i] = caps[i]; np_[i] = nbr.data_ptr<int32_t>();
    out[3 * l] = tk; out[3 * l + 1] = tv; out[3 * l + 2] = nbr;
  }
  check(g_api.kmaps_multi((int)todo.size(), kp, cp, n_, st, tkp, tvp, cap, np_, ld, (void*)stream), "irx_kmaps_build_multi");
  return out;
}

// The same tables by OCTREE DESCENT (irx_kmaps_build_pyramid, round 6): `parents`, `koffs`, `children` are the down-sampling maps of
// level l -> l + 1 (irx_pyramid_build's arrays; one entry per level but the last). Only the coarsest level is searched (and gets a hash
// table, when it has more than 2048 rows); every finer level is two cached table reads per neighbour. -> per level (table keys, table
// values, neighbours) with undefined table tensors where none was built.
std::vector<Tensor> kmaps_build_pyramid(const std::vector<Tensor>& keys, const std::vector<Tensor>& coords, const std::vector<int64_t>& strides,
                                        const std::vector<Tensor>& parents, const std::vector<Tensor>& koffs, const std::vector<Tensor>& children,
                                        const std::vector<int64_t>& child_lds, int64_t stream) {
  TORCH_CHECK(g_api.kmaps_pyramid && g_api.hash_capacity, "irx nodes: bind() has not been called");
  const size_t nl = keys.size();
  TORCH_CHECK(nl >= 1 && nl <= 8 && coords.size() == nl && strides.size() == nl && parents.size() + 1 == nl && koffs.size() + 1 == nl &&
              children.size() + 1 == nl && child_lds.size() + 1 == nl, "kmaps_build_pyramid: list sizes");
  std::vector<Tensor> out(3 * nl);
  int64_t nbr_total = 0;
  int n_[8], st[8], ld[8], cld[8];
  for (size_t l = 0; l < nl; ++l) {
    const int64_t n = coords[l].size(0), ldl = (n + 63) / 64 * 64;
    n_[l] = (int)n; st[l] = (int)strides[l]; ld[l] = (int)(ldl > 0 ? ldl : 64);
    nbr_total += 27 * (int64_t)ld[l];
  }
  const auto opt = keys[0].options();
  Tensor nbr_all = torch::empty({nbr_total}, opt.dtype(torch::kInt32));
  int32_t* np_[8];
  const int32_t* pp[8]; const uint8_t* kp[8]; const int32_t* cp[8];
  int64_t no = 0;
  for (size_t l = 0; l < nl; ++l) {
    Tensor nbr = nbr_all.narrow(0, no, 27 * (int64_t)ld[l]).view({27, (int64_t)ld[l]});
    no += 27 * (int64_t)ld[l];
    np_[l] = nbr.data_ptr<int32_t>();
    out[3 * l + 2] = nbr;
    if (l + 1 < nl) {
      pp[l] = parents[l].data_ptr<int32_t>(); kp[l] = koffs[l].data_ptr<uint8_t>(); cp[l] = children[l].data_ptr<int32_t>();
      cld[l] = (int)child_lds[l];
    }
  }
  const size_t top = nl - 1;
  Tensor tk, tv;
  size_t cap = 0;
  if (n_[top] > 2048) {
    cap = g_api.hash_capacity(n_[top]);
    tk = torch::empty({(int64_t)cap}, opt.dtype(torch::kInt64));
    tv = torch::empty({(int64_t)cap}, opt.dtype(torch::kInt32));
    out[3 * top] = tk; out[3 * top + 1] = tv;
  }
  check(g_api.kmaps_pyramid((int)nl, n_, st, (const uint64_t*)keys[top].data_ptr<int64_t>(), coords[top].data_ptr<int32_t>(),
                            tk.defined() ? (uint64_t*)tk.data_ptr<int64_t>() : nullptr, tv.defined() ? tv.data_ptr<int32_t>() : nullptr, cap,
                            pp, kp, cp, cld, np_, ld, (void*)stream),
        "irx_kmaps_build_pyramid");
  return out;
}

// The tables only the encoders' BACKWARD passes read — compacted pair lists of the given neighbour / child tables (irx_pairs_build_multi:
// two launches for all of them) and the transposed child maps of the down-sampling layers (irx_kmap_down_transpose, one launch each) —
// in one call, so that the preparation stage builds them beside the kernel maps instead of the autograd engine's thread at the head
// of each encoder's backward (sparse/encoder_fn.py _backward_tables: ~0.1 ms of interpreter time per encoder, on the step's critical
// path). tbls[t] int32 [K[t]][ld[t]] with n_out[t] valid columns -> per table (in_list [K][ldp], out_list [K][ldp], counts [K]) as
// sparse/functional.py pairs_build_multi lays them out; parents / koffs [n] -> per map the (8, max(n, 1)) transposed table.
std::vector<Tensor> backward_tables(const std::vector<Tensor>& tbls, const std::vector<int64_t>& lds, const std::vector<int64_t>& n_outs,
                                    const std::vector<int64_t>& Ks, const std::vector<Tensor>& parents, const std::vector<Tensor>& koffs,
                                    int64_t stream) {
  TORCH_CHECK(g_api.pairs_multi && g_api.pairs_ws && g_api.down_transpose, "irx nodes: bind() has not been called");
  const size_t nt = tbls.size(), nm = parents.size();
  TORCH_CHECK(lds.size() == nt && n_outs.size() == nt && Ks.size() == nt && koffs.size() == nm && nt <= 16, "backward_tables: list sizes");
  std::vector<Tensor> out;
  out.reserve(3 * nt + nm);
  if (nt > 0) {
    const auto opt = tbls[0].options().dtype(torch::kInt32);
    int64_t total = 0, ktotal = 0;
    size_t wsb = 0;
    std::vector<int64_t> ldp(nt), sz(nt);
    for (size_t t = 0; t < nt; ++t) {
      ldp[t] = n_outs[t] > 0 ? n_outs[t] : 1;
      sz[t] = Ks[t] * ldp[t];
      total += 2 * sz[t];
      ktotal += Ks[t];
      wsb += g_api.pairs_ws((int)n_outs[t], (int)Ks[t]);
    }
    Tensor lists = torch::empty({total}, opt), counts = torch::empty({ktotal}, opt);
    Tensor ws = torch::empty({(int64_t)(wsb > 4 ? wsb : 4)}, opt.dtype(torch::kUInt8));
    const int32_t* tp[16]; int ld_[16], no_[16], k_[16], ldp_[16]; int32_t* il[16]; int32_t* ol[16]; int32_t* cn[16];
    int64_t lo = 0, co = 0;
    for (size_t t = 0; t < nt; ++t) {
      Tensor a = lists.narrow(0, lo, sz[t]).view({Ks[t], ldp[t]}), b = lists.narrow(0, lo + sz[t], sz[t]).view({Ks[t], ldp[t]});
      Tensor c = counts.narrow(0, co, Ks[t]);
      lo += 2 * sz[t];
      co += Ks[t];
      tp[t] = tbls[t].data_ptr<int32_t>(); ld_[t] = (int)lds[t]; no_[t] = (int)n_outs[t]; k_[t] = (int)Ks[t]; ldp_[t] = (int)ldp[t];
      il[t] = a.data_ptr<int32_t>(); ol[t] = b.data_ptr<int32_t>(); cn[t] = c.data_ptr<int32_t>();
      out.push_back(a); out.push_back(b); out.push_back(c);
    }
    check(g_api.pairs_multi((int)nt, tp, ld_, no_, k_, il, ol, ldp_, cn, ws.data_ptr(), wsb, (void*)stream), "irx_pairs_build_multi");
  }
  for (size_t m = 0; m < nm; ++m) {
    const int64_t n = parents[m].size(0), ld = n > 0 ? n : 1;
    Tensor tbl = torch::empty({8, ld}, parents[m].options().dtype(torch::kInt32));
    check(g_api.down_transpose(parents[m].data_ptr<int32_t>(), koffs[m].data_ptr<uint8_t>(), (int)n, tbl.data_ptr<int32_t>(), (int)ld,
                               (void*)stream),
          "irx_kmap_down_transpose");
    out.push_back(tbl);
  }
  return out;
}

// The autograd engine runs ready nodes in descending order of their sequence number, and that number comes from a THREAD-LOCAL counter
// (at::sequence_number). InstanceRefer builds part of its graph on a helper thread (language module, relation head): with one node per
// head the training thread creates ~10 nodes per step and the helper ~40, so after a few steps the helper's nodes outrank every node
// of the training thread and the engine issues the language / relation backward (~0.5 ms of host time, off the critical path) BEFORE
// the two encoders' backward passes (measured, round 6: fp32 8.49 -> 9.43 ms per step). The training thread therefore advances its own
// counter by `n` at the head of every forward: its nodes stay ahead of the helper's, in creation order among themselves.
int64_t bump_sequence(int64_t n) {
  uint64_t v = 0;
  for (int64_t i = 0; i < n; ++i) v = at::sequence_number::get_and_increment();
  return (int64_t)v;
}

// addresses of the C-ABI entry points, taken from the library instance _lib.py loaded
void bind(const std::unordered_map<std::string, uint64_t>& addr) {
  auto get = [&](const char* name) -> uint64_t {
    auto it = addr.find(name);
    TORCH_CHECK(it != addr.end() && it->second != 0, "irx nodes: missing entry point ", name);
    return it->second;
  };
  g_api.mlp2_saved_floats = (saved_floats_fn)get("irx_mlp2_saved_floats");
  g_api.mlp2_fwd = (mlp2_fwd_fn)get("irx_mlp2_fwd");
  g_api.mlp2_bwd = (mlp2_bwd_fn)get("irx_mlp2_bwd");
  g_api.gru_fwd = (gru_fwd_fn)get("irx_gru_forward");
  g_api.gru_bwd = (gru_bwd_fn)get("irx_gru_backward");
  g_api.gru_wgrad = (gru_wgrad_fn)get("irx_gru_wgrad");
  g_api.last_error = (last_error_fn)get("irx_last_error");
  irxn::g_last_error = g_api.last_error;
  g_api.hash_capacity = (hash_capacity_fn)get("irx_hash_capacity");
  g_api.hash_build = (hash_build_fn)get("irx_hash_build");
  g_api.kmap_s1 = (kmap_s1_fn)get("irx_kmap_build_s1");
  g_api.kmaps_multi = (kmaps_multi_fn)get("irx_kmaps_build_multi");
  g_api.kmaps_pyramid = (kmaps_pyramid_fn)get("irx_kmaps_build_pyramid");
  g_api.pairs_ws = (pairs_ws_fn)get("irx_pairs_workspace_bytes");
  g_api.pairs_multi = (pairs_multi_fn)get("irx_pairs_build_multi");
  g_api.down_transpose = (down_transpose_fn)get("irx_kmap_down_transpose");
}

}  // namespace

namespace irxn {
last_error_fn g_last_error = nullptr;
void register_heads(pybind11::module& m);      // heads_nodes.cpp
}  // namespace irxn

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd nodes over libirx's C-ABI (dense heads)";
  m.def("bind", &bind);
  m.def("mlp2", &mlp2);
  m.def("gru_layer", &gru_layer);
  m.def("mlp_relu2", &mlp_relu2);
  m.def("kmaps_build", &kmaps_build);
  m.def("bump_sequence", &bump_sequence);
  m.def("backward_tables", &backward_tables);
  m.def("kmaps_build_pyramid", &kmaps_build_pyramid);
  irxn::register_heads(m);
}
