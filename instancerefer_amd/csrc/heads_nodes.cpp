// heads_nodes.cpp — the matching heads and the loss as ONE autograd node each (round 6).
//
// Round 5's trace: between "loss issued" and the encoders' first backward convolution the GPU idles for ~1 ms while the autograd
// engine walks ~60 small nodes (cosine, MLP, pooling, BEV / Conv2d rows, BatchNorm rows, attention pooling, loss), a dozen of them
// Python autograd.Functions entered through the GIL at 40-130 us each, plus one AccumulateGrad evaluation per parameter. Here a head
// is one torch::autograd::Node whose forward and backward issue the SAME C-ABI calls of libirx in the SAME order as the operator-by-
// operator path (bit-identical results; tests/test_heads_gpu.py) from C++: no interpreter, no per-operator engine hop, parameter
// gradients straight into the optimizer's slots (optim.FlatAdam's sink) — or, without a sink, returned to autograd as usual.
//
//   SceneHeadNode : encoder output -> BEV rows -> BatchNorm2d/ReLU -> Conv2d 3x3 -> BatchNorm2d/ReLU -> Dropout -> Conv2d 3x3 ->
//                   language-guided attention pooling -> area classifier          (reference models/scene_module.py:61-96)
//   AttrHeadNode  : encoder output -> global max pooling -> attribute MLPs -> cosine score, and the scene score of every candidate
//                   against the scene vector                                      (models/attribute_module.py:105-126, scene_module.py:98-106)
//   LossNode      : irx_total_loss (lib/loss_helper.py:196-269) with its stored gradients
//
// Binding plumbing only: no arithmetic happens in this file.
#include "torch_nodes.h"

#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/functions/utils.h>

#include <c10/core/Stream.h>
#include <c10/core/StreamGuard.h>

#include <cmath>
#include <unordered_map>

namespace irxn {
namespace {

using at::Tensor;
using torch::autograd::Node;
using torch::autograd::edge_list;
using torch::autograd::variable_list;

typedef size_t (*saved_floats_fn)(int, int);
typedef int (*mlp2_fwd_fn)(const float*, int, int, int, int, const float*, const float*, int, const float*, const float*, float,
                           float*, float*, float, float, unsigned long long, const float*, const float*, float*, float*, void*);
typedef int (*mlp2_bwd_fn)(const float*, const float*, int, int, int, int, const float*, int, const float*, const float*,
                           const float*, float, float*, float*, float*, float*, float*, float*, float*, float*, void*);
typedef int (*segmax_fn)(const float*, const int32_t*, int, int, float*, int32_t*, void*);
typedef int (*segmax_bwd_fn)(const float*, const int32_t*, int, int, float*, void*);
typedef int (*cos_fwd_fn)(const float*, const float*, const int64_t*, int, int, float, float*, float*, void*);
typedef int (*cos_bwd_fn)(const float*, const float*, const int64_t*, const float*, const float*, const float*, int, int, int, float,
                          float*, float*, void*);
typedef size_t (*conv_ws_fn)(int, int, int, int, int);
typedef int (*conv_fwd_fn)(const float*, const float*, const int32_t*, int, int, int, int, int, int, int, float*, void*, size_t, void*);
typedef size_t (*wgrad_ws_fn)(int, int, int, int);
typedef int (*wgrad_fn)(const float*, const float*, const int32_t*, int, int, int, int, int, float*, void*, size_t, void*);
typedef size_t (*bn_ws_fn)(int, int);
typedef int (*bn_fwd_fn)(const float*, int, int, float, float, const float*, const float*, const float*, int, float*, float*, float*,
                         float*, float*, void*, size_t, void*);
typedef int (*bn_bwd_fn)(const float*, const float*, const float*, int, int, const float*, const float*, const float*, int, float*,
                         float*, float*, float*, void*, size_t, void*);
typedef int (*kdt_fn)(const int32_t*, const uint8_t*, int, int32_t*, int, void*);
typedef int (*attn_fwd_fn)(const float*, const float*, int, int, int, float, float*, float*, void*);
typedef int (*attn_bwd_fn)(const float*, const float*, const float*, const float*, const float*, int, int, int, float, float*, float*,
                           void*);
typedef int (*fork_fn)(void*, void*);
typedef int (*drop_fn)(const float*, size_t, float, unsigned long long, float*, void*);
typedef size_t (*ec_ws_fn)(int, int, int, int, int, int);
typedef int (*ec_fwd_fn)(const float*, const float*, const int64_t*, const int32_t*, int, int, int, int, int, int, const float* const*,
                         float*, int32_t*, void*);
typedef int (*ec_bwd_fn)(const float*, const float*, const int64_t*, const int32_t*, int, int, int, int, int, int, const float* const*,
                         const float*, const int32_t*, float* const*, float*, void*, size_t, void*);
typedef int (*lp_fwd_fn)(const float*, const float*, const int64_t*, int, int, int, int, const float* const*, const float* const*, float*,
                         float*, float*, float*, void*);
typedef int (*lp_bwd_fn)(const float*, const float*, const int64_t*, int, int, int, int, const float* const*, const float*, const float*,
                         const float*, const float*, const float*, float*, float*, float*, float*, float*, void*);
typedef int (*loss_fn)(const float*, const int64_t*, int, int, const float*, const int64_t*, int, const float*, const float*,
                       const float*, const float*, const int64_t*, const float*, int, float, float, float, int, float*, float*, float*,
                       float*, void*);

struct Api {
  saved_floats_fn mlp2_saved_floats = nullptr;
  mlp2_fwd_fn mlp2_fwd = nullptr;
  mlp2_bwd_fn mlp2_bwd = nullptr;
  segmax_fn segmax = nullptr;
  segmax_bwd_fn segmax_bwd = nullptr;
  cos_fwd_fn cos_fwd = nullptr;
  cos_bwd_fn cos_bwd = nullptr;
  conv_ws_fn conv_ws = nullptr;
  conv_fwd_fn conv_fwd = nullptr;
  wgrad_ws_fn wgrad_ws = nullptr;
  wgrad_fn wgrad = nullptr;
  bn_ws_fn bn_ws = nullptr;
  bn_fwd_fn bn_fwd = nullptr;
  bn_bwd_fn bn_bwd = nullptr;
  kdt_fn kdt = nullptr;
  attn_fwd_fn attn_fwd = nullptr;
  attn_bwd_fn attn_bwd = nullptr;
  drop_fn dropout = nullptr;
  fork_fn fork = nullptr;
  loss_fn total_loss = nullptr;
  ec_ws_fn ec_ws = nullptr;
  ec_fwd_fn ec_fwd = nullptr;
  ec_bwd_fn ec_bwd = nullptr;
  lp_fwd_fn lp_fwd = nullptr;
  lp_bwd_fn lp_bwd = nullptr;
} api;

inline Tensor f32c(const Tensor& t) { return t.contiguous().to(torch::kFloat32); }
inline Tensor bytes(size_t n, const Tensor& like) {
  return at::empty({(int64_t)(n > 16 ? n : 16)}, like.options().dtype(torch::kUInt8));
}

// ---- parameter-gradient destinations: the optimizer's slots (sink) or fresh tensors ------------------------------------------
struct PGrads {
  bool deliver = false;
  std::vector<float*> ptr;
  std::vector<Tensor> own;
  const std::vector<Tensor>* params = nullptr;
  void init(const Sink& sink, const std::vector<Tensor>& p) {
    params = &p;
    deliver = sink.deliver(p.size());
    ptr.resize(p.size());
    if (!deliver) own.resize(p.size());
    for (size_t i = 0; i < p.size(); ++i) {
      if (deliver) {
        ptr[i] = (float*)sink.slots[i];
      } else {
        own[i] = at::empty(p[i].sizes(), p[i].options().dtype(torch::kFloat32));
        ptr[i] = own[i].data_ptr<float>();
      }
    }
  }
  // a tensor view of destination i with the parameter's shape (for the gradients ATen writes: Conv2d weight / bias)
  Tensor view(size_t i) const {
    return deliver ? at::from_blob(ptr[i], (*params)[i].sizes(), (*params)[i].options().dtype(torch::kFloat32)) : own[i];
  }
};

// common state of a head node: data inputs first, then (only when no sink took them at forward time) the parameters
struct HeadNode : public Node {
  std::vector<Tensor> params;
  bool params_are_inputs = false;
  size_t n_x = 0;
  Sink sink;
  Tensor keep_a, keep_b;
  void* stream = nullptr;
  bool released = false;

  void guard() const { TORCH_CHECK(!released, name(), ": backward through a graph whose buffers were already freed (retain_graph)"); }

  // hand the parameter gradients over: sink record, autograd outputs, or (parameters hidden at forward time but the sink refuses now:
  // a second backward before zero_grad(), an optimizer replaced in between) accumulation into .grad by hand — what their
  // AccumulateGrad nodes would have done
  void finish(const PGrads& pg, variable_list& out) {
    if (pg.deliver) {
      sink.delivered(stream);
    } else if (params_are_inputs) {
      for (size_t i = 0; i < params.size(); ++i) out[n_x + i] = pg.own[i];
    } else {
      at::NoGradGuard ng;
      for (size_t i = 0; i < params.size(); ++i) {
        Tensor& g = params[i].mutable_grad();
        if (g.defined()) g.add_(pg.own[i]); else g = pg.own[i];
      }
    }
  }

  // edges: the data inputs, then the parameters unless a sink takes their gradients
  void wire(const std::vector<Tensor>& xs, const std::vector<int64_t>& slot_ptrs, const std::vector<Tensor>& keep) {
    n_x = xs.size();
    sink = Sink(slot_ptrs);
    params_are_inputs = sink.slots.size() != params.size();
    if (keep.size() > 0) keep_a = keep[0];
    if (keep.size() > 1) keep_b = keep[1];
    edge_list edges = torch::autograd::collect_next_edges(xs);
    if (params_are_inputs) {
      edge_list pe = torch::autograd::collect_next_edges(params);
      edges.insert(edges.end(), pe.begin(), pe.end());
    }
    set_next_edges(std::move(edges));
  }
};

// ---- one head MLP: nn.Sequential(Linear, BatchNorm1d | LayerNorm, ReLU, [Dropout], Linear) through irx_mlp2_fwd / _bwd ---------
struct Mlp {
  size_t p0 = 0;                 // index of w1 in the node's parameter list: (w1, b1, gamma, beta, w2, b2)
  Tensor rmean, rvar;            // BatchNorm1d running statistics (undefined: LayerNorm)
  int norm = 3;
  float eps = 1e-5f, momentum = 0.f;
  double drop_p = 0.0;           // (double: the backward's 1 / (1 - p) is evaluated as dense.py / MLP2Node evaluate it, bit for bit)
  unsigned long long seed = 0;
  Tensor x, saved;               // kept for the backward
  int rows = 0, din = 0, dh = 0, dout = 0;
  void reset() { x = Tensor(); saved = Tensor(); rmean = Tensor(); rvar = Tensor(); }
};

Tensor mlp_fwd(Mlp& m, const std::vector<Tensor>& P, const Tensor& x_in, void* stream) {
  m.x = f32c(x_in);
  const Tensor &w1 = P[m.p0], &b1 = P[m.p0 + 1], &gamma = P[m.p0 + 2], &beta = P[m.p0 + 3], &w2 = P[m.p0 + 4], &b2 = P[m.p0 + 5];
  m.rows = (int)m.x.size(0); m.din = (int)m.x.size(1); m.dh = (int)w1.size(0); m.dout = (int)w2.size(0);
  TORCH_CHECK(w1.size(1) == m.din && w2.size(1) == m.dh, "head MLP: shape mismatch");
  Tensor y = at::empty({m.rows, m.dout}, m.x.options());
  m.saved = at::empty({(int64_t)api.mlp2_saved_floats(m.rows, m.dh)}, m.x.options());
  check(api.mlp2_fwd(fp(m.x), m.rows, m.din, m.dh, m.dout, fp(w1), fp(b1), m.norm, fp(gamma), fp(beta), m.eps, fpm(m.rmean),
                     fpm(m.rvar), m.momentum, (float)m.drop_p, m.seed, fp(w2), fp(b2), fpm(m.saved), fpm(y), stream),
        "irx_mlp2_fwd");
  return y;
}

Tensor mlp_bwd(const Mlp& m, const std::vector<Tensor>& P, const Tensor& dy, bool want_dx, const PGrads& pg, void* stream) {
  const Tensor &w1 = P[m.p0], &gamma = P[m.p0 + 2], &w2 = P[m.p0 + 4];
  Tensor scratch = at::empty({(int64_t)m.rows * (m.dh + (want_dx ? m.din : 0))}, m.x.options());
  float* base = scratch.data_ptr<float>();
  float* dx_ptr = want_dx ? base + (size_t)m.rows * m.dh : nullptr;
  float* const* gp = pg.ptr.data() + m.p0;
  check(api.mlp2_bwd(fp(m.x), fp(dy), m.rows, m.din, m.dh, m.dout, fp(w1), m.norm, fp(gamma), fp(w2), fp(m.saved),
                     m.drop_p > 0 ? (float)(1.0 / (1.0 - m.drop_p)) : 1.f, base, dx_ptr, gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], stream),
        "irx_mlp2_bwd");
  return want_dx ? scratch.narrow(0, (int64_t)m.rows * m.dh, (int64_t)m.rows * m.din).view({m.rows, m.din}) : Tensor();
}

// ---- gather-GEMM over a row table (irx_spconv_fwd): y[q] = sum_k x[tbl[k][q]] @ w[k]  (trans: the data gradient) ----------------
Tensor conv_rows(const Tensor& x, const Tensor& w, const Tensor& tbl, int ld, int n_out, int flip, int trans, void* stream) {
  const int K = (int)w.size(0);
  const int cin = trans ? (int)w.size(2) : (int)w.size(1), cout = trans ? (int)w.size(1) : (int)w.size(2);
  TORCH_CHECK(x.size(1) == cin, "conv rows: channel mismatch");
  Tensor y = at::empty({n_out, cout}, x.options());
  const size_t wsb = api.conv_ws(n_out, K, cin, cout, trans);
  Tensor ws = wsb ? bytes(wsb, x) : Tensor();
  check(api.conv_fwd(fp(x), fp(w), tbl.data_ptr<int32_t>(), ld, n_out, K, cin, cout, flip, trans, fpm(y),
                     wsb ? ws.data_ptr() : nullptr, wsb, stream),
        "irx_spconv_fwd");
  return y;
}

void wgrad_rows(const Tensor& x, const Tensor& dy, const Tensor& tbl, int ld, int n_out, int K, float* dw, void* stream) {
  const int cin = (int)x.size(1), cout = (int)dy.size(1);
  const size_t wsb = api.wgrad_ws(n_out, K, cin, cout);
  Tensor ws = wsb ? bytes(wsb, x) : Tensor();
  check(api.wgrad(fp(x), fp(dy), tbl.data_ptr<int32_t>(), ld, n_out, K, cin, cout, dw, wsb ? ws.data_ptr() : nullptr, wsb, stream),
        "irx_spconv_wgrad");
}

// ---- train-mode BatchNorm (+ ReLU) over rows (irx_bn_forward / irx_bn_backward) -------------------------------------------------
struct BnRows {
  size_t p0 = 0;                 // index of (weight, bias) in the parameter list
  Tensor rmean, rvar, x, y, mean, invstd;
  float eps = 1e-5f, momentum = 0.1f;
  void reset() { rmean = rvar = x = y = mean = invstd = Tensor(); }
};

Tensor bn_fwd(BnRows& b, const std::vector<Tensor>& P, const Tensor& x, void* stream) {
  const int n = (int)x.size(0), c = (int)x.size(1);
  b.x = x;
  b.mean = at::empty({c}, x.options());
  b.invstd = at::empty({c}, x.options());
  b.y = at::empty_like(x);
  const size_t wsb = api.bn_ws(n, c);
  Tensor ws = bytes(wsb, x);
  check(api.bn_fwd(fp(x), n, c, b.eps, b.momentum, fp(P[b.p0]), fp(P[b.p0 + 1]), nullptr, 1, fpm(b.mean), fpm(b.invstd), fpm(b.rmean),
                   fpm(b.rvar), fpm(b.y), ws.data_ptr(), wsb, stream),
        "irx_bn_forward");
  return b.y;
}

Tensor bn_bwd(const BnRows& b, const std::vector<Tensor>& P, const Tensor& dy, const PGrads& pg, void* stream) {
  const int n = (int)b.x.size(0), c = (int)b.x.size(1);
  Tensor dx = at::empty_like(b.x);
  const size_t wsb = api.bn_ws(n, c);
  Tensor ws = bytes(wsb, b.x);
  check(api.bn_bwd(fp(b.x), fp(b.y), fp(dy), n, c, fp(b.mean), fp(b.invstd), fp(P[b.p0]), 1, fpm(dx), pg.ptr[b.p0], pg.ptr[b.p0 + 1],
                   nullptr, ws.data_ptr(), wsb, stream),
        "irx_bn_backward");
  return dx;
}

// ---- nn.Conv2d 3x3 (stride 1, no padding, bias) on channels-last rows through the constant grid tables --------------------------
struct Conv2dRows {
  size_t p0 = 0;                 // (weight [cout][cin][ks][ks], bias [cout])
  Tensor fwd_tbl, bwd_tbl, x, wk;
  int n_out = 0, n_in = 0, ks = 3;
  void reset() { fwd_tbl = bwd_tbl = x = wk = Tensor(); }
};

Tensor conv2d_fwd(Conv2dRows& c, const std::vector<Tensor>& P, const Tensor& x, void* stream) {
  const Tensor& w = P[c.p0];
  const int64_t cout = w.size(0), cin = w.size(1);
  c.x = x;
  c.wk = w.permute({2, 3, 1, 0}).reshape({(int64_t)c.ks * c.ks, cin, cout}).contiguous();
  Tensor y = conv_rows(x, c.wk, c.fwd_tbl, c.n_out, c.n_out, 0, 0, stream);
  y.add_(P[c.p0 + 1]);
  return y;
}

// data gradient only (the chain the encoder's backward waits for) ...
Tensor conv2d_dgrad(const Conv2dRows& c, const Tensor& dy, void* stream) {
  return conv_rows(dy, c.wk, c.bwd_tbl, c.n_in, c.n_in, 0, 1, stream);
}
// ... and the parameter gradients (bias: column sums; weight: pair-free weight gradient, permuted back to the Conv2d layout) — ATen
// operators here run on the thread's CURRENT stream: the caller sets it to `stream` when that is not the node's own
void conv2d_wgrad(const Conv2dRows& c, const std::vector<Tensor>& P, const Tensor& dy, const PGrads& pg, void* stream) {
  const Tensor& w = P[c.p0];
  const int64_t cout = w.size(0), cin = w.size(1);
  const int K = c.ks * c.ks;
  Tensor db = pg.view(c.p0 + 1);
  const std::vector<int64_t> dim0{0};
  at::sum_out(db, dy, dim0);
  Tensor dwk = at::empty({K, cin, cout}, dy.options());
  wgrad_rows(c.x, dy, c.fwd_tbl, c.n_out, c.n_out, K, fpm(dwk), stream);
  Tensor dw = pg.view(c.p0);
  dw.copy_(dwk.view({(int64_t)c.ks, (int64_t)c.ks, cin, cout}).permute({3, 2, 0, 1}));
}

// a second stream lent by the caller (torch.cuda.Stream: handle + the c10 triple), or none
struct AuxStream {
  void* ptr = nullptr;
  int64_t id = 0, device_index = 0, device_type = 0;
  bool on() const { return ptr != nullptr; }
  c10::Stream c10s() const { return c10::Stream::unpack3(id, (c10::DeviceIndex)device_index, (c10::DeviceType)device_type); }
};

// =================================================================================================================================
// SceneHeadNode
//   data inputs : feats [n][128] (the BEV encoder's output rows), lang: the language module's scene vector [B][256] — or, with
//                 pre_lang, that vector already through lang_emb_fc ([B][128]; the MLP then runs as a node of its own on the language
//                 stream, off this head's chain)
//   parameters  : to_bev.1.kernel, to_bev.2.{weight,bias}, vis_emb_fc.0.{weight,bias}, vis_emb_fc.1.{weight,bias},
//                 vis_emb_fc.4.{weight,bias}, [lang_emb_fc (6) unless pre_lang], cls (6)
//   outputs     : vis_atten [B][n_vis], seg_scores [B][9], scene vector [B][128]
// Backward: the data-gradient chain (classifier -> attention -> Conv2d -> Dropout -> BatchNorm -> Conv2d -> BatchNorm -> BEV rows) is
// what the scene encoder's backward — the step's long pole — waits for; the weight / bias gradients feed nothing downstream. With a
// lent second stream (aux) they are issued there, behind an event on the chain, and the chain is 21 launches instead of 35.
// =================================================================================================================================
struct SceneHeadNode : public HeadNode {
  Tensor feats, bev_tbl, bev_tbl_t;             // bev_tbl_t: the transposed table (8, max(n, 1)) of the data gradient
  int ncell = 0, B = 0, n_vis = 0;
  BnRows bn0, bn1;
  Conv2dRows cv0, cv1;
  Mlp lang, cls;
  bool pre_lang = false;
  AuxStream aux;
  float drop_p = 0.f;
  unsigned long long drop_seed = 0;
  Tensor rows4, lang_h, atten, scene_vec;

  std::string name() const override { return "irx::SceneHeadNode"; }
  void release_variables() override {
    released = true;
    feats = bev_tbl = bev_tbl_t = rows4 = lang_h = atten = scene_vec = Tensor();
    bn0.reset(); bn1.reset(); cv0.reset(); cv1.reset(); lang.reset(); cls.reset();
  }

  variable_list apply(variable_list&& grads) override {
    guard();
    at::NoGradGuard ng;
    variable_list out(num_outputs());
    PGrads pg;
    pg.init(sink, params);
    const Tensor d_seg = grads[1].defined() ? f32c(grads[1]) : at::zeros({B, cls.dout}, scene_vec.options());
    // area classifier, then the scene vector's two consumers summed in the engine's order (cosine scores first, classifier second)
    Tensor d_vec = mlp_bwd(cls, params, d_seg, true, pg, stream);
    if (grads[2].defined()) d_vec = f32c(grads[2]) + d_vec;
    const int d = (int)rows4.size(1);
    Tensor dfeats = at::empty_like(rows4), dlang_h = at::empty_like(lang_h);
    const Tensor datten = grads[0].defined() ? f32c(grads[0]) : Tensor();
    check(api.attn_bwd(fp(rows4), fp(lang_h), fp(atten), fp(d_vec), fp(datten), B, n_vis, d, 1.f / std::sqrt((float)d), fpm(dfeats),
                       fpm(dlang_h), stream),
          "irx_attn_pool_bwd");
    // ---- the chain ----
    Tensor g1 = conv2d_dgrad(cv1, dfeats, stream);
    if (drop_p > 0.f) check(api.dropout(fp(g1), (size_t)g1.numel(), drop_p, drop_seed, fpm(g1), stream), "irx_dropout_flat");
    Tensor g2 = bn_bwd(bn1, params, g1, pg, stream);
    Tensor g3 = conv2d_dgrad(cv0, g2, stream);
    Tensor g4 = bn_bwd(bn0, params, g3, pg, stream);
    const Tensor& kernel = params[0];
    const int K = (int)kernel.size(0), n = (int)feats.size(0), ld_b = n > 0 ? n : 1;
    if (should_compute_output(0)) {
      TORCH_CHECK(bev_tbl_t.size(0) == 8 && bev_tbl_t.size(1) == ld_b, "scene head: transposed BEV table shape");
      out[0] = conv_rows(g4, kernel, bev_tbl_t, ld_b, n, 0, 1, stream);
    }
    // the language MLP (unless it is a node of its own): its output gradient leaves this node, so it stays on the node's stream
    if (!pre_lang) {
      Tensor dlang = mlp_bwd(lang, params, dlang_h, should_compute_output(1), pg, stream);
      if (should_compute_output(1)) out[1] = dlang;
    } else if (should_compute_output(1)) {
      out[1] = dlang_h;
    }
    // ---- parameter gradients that feed nothing downstream ----
    void* ws = stream;
    if (aux.on() && aux.ptr != stream) {
      check(api.fork(stream, aux.ptr), "irx_stream_fork");
      ws = aux.ptr;
      const c10::Stream cs = aux.c10s();
      for (const Tensor* t : {&dfeats, &g2, &g4, &cv1.x, &cv0.x, &feats, &cv1.fwd_tbl, &cv0.fwd_tbl, &bev_tbl})
        if (t->defined()) t->record_stream(cs);
      c10::StreamGuard sg(cs);                       // (ATen operators of conv2d_wgrad and their allocations follow the current stream)
      conv2d_wgrad(cv1, params, dfeats, pg, ws);
      conv2d_wgrad(cv0, params, g2, pg, ws);
      wgrad_rows(feats, g4, bev_tbl, ncell, ncell, K, pg.ptr[0], ws);
    } else {
      conv2d_wgrad(cv1, params, dfeats, pg, ws);
      conv2d_wgrad(cv0, params, g2, pg, ws);
      wgrad_rows(feats, g4, bev_tbl, ncell, ncell, K, pg.ptr[0], ws);
    }
    if (pg.deliver) {
      // every slot is complete on `ws`: the chain's BatchNorm / MLP gradients were enqueued on the node's stream ahead of the fork
      sink.delivered(ws);
    } else {
      if (ws != stream) check(api.fork(ws, stream), "irx_stream_fork");      // autograd reads the gradients on the node's stream
      finish(pg, out);
    }
    return out;
  }
};

std::vector<Tensor> scene_head(const Tensor& feats_in, const Tensor& lang_in, bool pre_lang, const Tensor& bev_tbl, const Tensor& bev_tbl_t,
                               int64_t ncell, int64_t B, std::vector<Tensor> grid, std::vector<int64_t> grid_n,
                               std::vector<Tensor> params, std::vector<Tensor> stats, std::vector<double> f, std::vector<int64_t> seeds,
                               int64_t stream_i, std::vector<int64_t> aux_stream, std::vector<int64_t> slot_ptrs, std::vector<Tensor> keep) {
  TORCH_CHECK(api.conv_fwd && api.bn_fwd && api.mlp2_fwd && api.attn_fwd, "irx nodes: bind_heads() has not been called");
  TORCH_CHECK(params.size() == (pre_lang ? 15u : 21u) && stats.size() == 6 && f.size() == 9 && seeds.size() == 2 && grid.size() == 4 &&
              grid_n.size() == 4 && (aux_stream.empty() || aux_stream.size() == 4), "scene_head: argument lists");
  void* stream = (void*)stream_i;
  auto node = std::shared_ptr<SceneHeadNode>(new SceneHeadNode(), torch::autograd::deleteNode);
  SceneHeadNode& s = *node;
  s.params = std::move(params);
  s.stream = stream;
  s.pre_lang = pre_lang;
  if (aux_stream.size() == 4) {
    s.aux.ptr = (void*)aux_stream[0]; s.aux.id = aux_stream[1]; s.aux.device_index = aux_stream[2]; s.aux.device_type = aux_stream[3];
  }
  const bool rec = at::GradMode::is_enabled() && (feats_in.requires_grad() || lang_in.requires_grad() || s.params[0].requires_grad());
  std::vector<Tensor> outs;
  {
    at::NoGradGuard ng;
    const std::vector<Tensor>& P = s.params;
    s.feats = f32c(feats_in);
    s.bev_tbl = bev_tbl; s.bev_tbl_t = bev_tbl_t;
    s.ncell = (int)ncell; s.B = (int)B;
    s.bn0.p0 = 1; s.bn0.rmean = stats[0]; s.bn0.rvar = stats[1]; s.bn0.eps = (float)f[0]; s.bn0.momentum = (float)f[1];
    s.cv0.p0 = 3; s.cv0.fwd_tbl = grid[0]; s.cv0.bwd_tbl = grid[1]; s.cv0.n_out = (int)grid_n[0]; s.cv0.n_in = (int)grid_n[1];
    s.bn1.p0 = 5; s.bn1.rmean = stats[2]; s.bn1.rvar = stats[3]; s.bn1.eps = (float)f[2]; s.bn1.momentum = (float)f[3];
    s.cv1.p0 = 7; s.cv1.fwd_tbl = grid[2]; s.cv1.bwd_tbl = grid[3]; s.cv1.n_out = (int)grid_n[2]; s.cv1.n_in = (int)grid_n[3];
    s.cv0.ks = (int)P[3].size(2); s.cv1.ks = (int)P[7].size(2);
    s.drop_p = (float)f[4]; s.drop_seed = (unsigned long long)seeds[0];
    s.lang.p0 = 9; s.lang.norm = 3; s.lang.eps = (float)f[5]; s.lang.drop_p = f[6]; s.lang.seed = (unsigned long long)seeds[1];
    s.cls.p0 = pre_lang ? 9 : 15; s.cls.norm = 1; s.cls.rmean = stats[4]; s.cls.rvar = stats[5]; s.cls.eps = (float)f[7];
    s.cls.momentum = (float)f[8];
    Tensor rows = conv_rows(s.feats, P[0], bev_tbl, s.ncell, s.ncell, 0, 0, stream);          // (B * nx * ny, 128)
    rows = bn_fwd(s.bn0, P, rows, stream);
    rows = conv2d_fwd(s.cv0, P, rows, stream);
    rows = bn_fwd(s.bn1, P, rows, stream);
    if (s.drop_p > 0.f) {
      Tensor dropped = at::empty_like(rows);
      check(api.dropout(fp(rows), (size_t)rows.numel(), s.drop_p, s.drop_seed, fpm(dropped), stream), "irx_dropout_flat");
      rows = dropped;
    }
    s.rows4 = conv2d_fwd(s.cv1, P, rows, stream);                                             // (B * n_vis, D)
    TORCH_CHECK(s.rows4.size(0) % B == 0, "scene_head: rows do not divide by the batch size");
    s.n_vis = (int)(s.rows4.size(0) / B);
    const int d = (int)s.rows4.size(1);
    s.lang_h = pre_lang ? f32c(lang_in) : mlp_fwd(s.lang, P, lang_in, stream);
    TORCH_CHECK(s.lang_h.size(0) == B && s.lang_h.size(1) == d, "scene_head: language vector shape");
    s.atten = at::empty({B, s.n_vis}, s.rows4.options());
    s.scene_vec = at::empty({B, d}, s.rows4.options());
    check(api.attn_fwd(fp(s.rows4), fp(s.lang_h), s.B, s.n_vis, d, 1.f / std::sqrt((float)d), fpm(s.atten), fpm(s.scene_vec), stream),
          "irx_attn_pool_fwd");
    Tensor seg = mlp_fwd(s.cls, P, s.scene_vec, stream);
    outs = {s.atten.detach(), seg, s.scene_vec.detach()};
  }
  if (rec) {
    s.wire({feats_in, lang_in}, slot_ptrs, keep);
    torch::autograd::set_history(outs, node);
  }
  return outs;
}

// =================================================================================================================================
// AttrHeadNode
//   data inputs : feats [n][128] (the candidate encoder's output rows), lang: the attribute vector [B][256] (or, with pre_lang, already
//                 through attribute.lang_emb_fc)
//   parameters  : [attribute.lang_emb_fc (6) unless pre_lang], attribute.vis_emb_fc (6), scene.vis_emb_fc1 (6)
//   outputs     : obj_feats [Nc][128], attribute_scores [Nc], obj_h [Nc][128] = vis_emb_fc1(obj_feats) — the candidate side of the
//                 scene scores, which a CosineNode takes against the scene vector once the scene head has produced it (so that
//                 everything here runs beside the scene head, forward and backward)
// =================================================================================================================================
struct AttrHeadNode : public HeadNode {
  Tensor offsets, idx, arg, pooled, lang_h, vis_h, score_a, norms_a;
  int n_rows = 0, nc = 0, B = 0;
  Mlp lang, vis, fc1;
  bool pre_lang = false;
  float eps_a = 1e-12f;

  std::string name() const override { return "irx::AttrHeadNode"; }
  void release_variables() override {
    released = true;
    offsets = idx = arg = pooled = lang_h = vis_h = score_a = norms_a = Tensor();
    lang.reset(); vis.reset(); fc1.reset();
  }

  variable_list apply(variable_list&& grads) override {
    guard();
    at::NoGradGuard ng;
    variable_list out(num_outputs());
    PGrads pg;
    pg.init(sink, params);
    // candidate-side MLP of the scene scores (the later operator of the forward: first in the engine's order)
    const Tensor d_obj_h = grads[2].defined() ? f32c(grads[2]) : at::zeros({nc, fc1.dout}, pooled.options());
    Tensor d_pool = mlp_bwd(fc1, params, d_obj_h, true, pg, stream);
    // attribute scores -> visual MLP
    const Tensor d_as = grads[1].defined() ? f32c(grads[1]) : at::zeros_like(score_a);
    Tensor d_vis_h = at::empty_like(vis_h), d_lang_h = at::empty_like(lang_h);
    check(api.cos_bwd(fp(vis_h), fp(lang_h), idx.data_ptr<int64_t>(), fp(score_a), fp(norms_a), fp(d_as), nc, B, (int)vis_h.size(1), eps_a,
                      fpm(d_vis_h), fpm(d_lang_h), stream),
          "irx_cosine_rows_bwd");
    d_pool = d_pool + mlp_bwd(vis, params, d_vis_h, true, pg, stream);
    if (grads[0].defined()) d_pool = d_pool + f32c(grads[0]);
    if (should_compute_output(0)) {
      Tensor dx = at::zeros({n_rows, pooled.size(1)}, pooled.options());
      check(api.segmax_bwd(fp(d_pool), arg.data_ptr<int32_t>(), nc, (int)pooled.size(1), fpm(dx), stream), "irx_segment_max_backward");
      out[0] = dx;
    }
    if (!pre_lang) {
      Tensor dlang = mlp_bwd(lang, params, d_lang_h, should_compute_output(1), pg, stream);
      if (should_compute_output(1)) out[1] = dlang;
    } else if (should_compute_output(1)) {
      out[1] = d_lang_h;
    }
    finish(pg, out);
    return out;
  }
};

std::vector<Tensor> attr_head(const Tensor& feats_in, const Tensor& offsets, int64_t nseg, const Tensor& idx, const Tensor& lang_in,
                              bool pre_lang, std::vector<Tensor> params, std::vector<Tensor> stats, std::vector<double> f,
                              std::vector<int64_t> seeds, int64_t stream_i, std::vector<int64_t> slot_ptrs, std::vector<Tensor> keep) {
  TORCH_CHECK(api.segmax && api.cos_fwd && api.mlp2_fwd, "irx nodes: bind_heads() has not been called");
  TORCH_CHECK(params.size() == (pre_lang ? 12u : 18u) && stats.size() == 2 && f.size() == 6 && seeds.size() == 1, "attr_head: argument lists");
  void* stream = (void*)stream_i;
  auto node = std::shared_ptr<AttrHeadNode>(new AttrHeadNode(), torch::autograd::deleteNode);
  AttrHeadNode& s = *node;
  s.params = std::move(params);
  s.stream = stream;
  s.pre_lang = pre_lang;
  const bool rec = at::GradMode::is_enabled() && (feats_in.requires_grad() || lang_in.requires_grad() || s.params[0].requires_grad());
  std::vector<Tensor> outs;
  {
    at::NoGradGuard ng;
    const std::vector<Tensor>& P = s.params;
    const Tensor x = f32c(feats_in);
    const int c = (int)x.size(1);
    s.n_rows = (int)x.size(0); s.nc = (int)nseg; s.B = (int)lang_in.size(0);
    s.offsets = offsets; s.idx = idx;
    TORCH_CHECK(idx.size(0) == nseg && idx.scalar_type() == torch::kInt64 && offsets.scalar_type() == torch::kInt32, "attr_head: index tensors");
    const size_t o = pre_lang ? 0 : 6;
    s.lang.p0 = 0; s.lang.norm = 1; s.lang.rmean = stats[0]; s.lang.rvar = stats[1]; s.lang.eps = (float)f[0]; s.lang.momentum = (float)f[1];
    s.vis.p0 = o; s.vis.norm = 3; s.vis.eps = (float)f[2];
    s.fc1.p0 = o + 6; s.fc1.norm = 3; s.fc1.eps = (float)f[3]; s.fc1.drop_p = f[4]; s.fc1.seed = (unsigned long long)seeds[0];
    s.eps_a = (float)f[5];
    s.lang_h = pre_lang ? f32c(lang_in) : mlp_fwd(s.lang, P, lang_in, stream);
    s.pooled = at::empty({nseg, c}, x.options());
    s.arg = at::empty({nseg, c}, x.options().dtype(torch::kInt32));
    check(api.segmax(fp(x), offsets.data_ptr<int32_t>(), s.nc, c, fpm(s.pooled), s.arg.data_ptr<int32_t>(), stream), "irx_segment_max");
    s.vis_h = mlp_fwd(s.vis, P, s.pooled, stream);
    TORCH_CHECK(s.lang_h.size(1) == s.vis_h.size(1), "attr_head: language vector width");
    s.score_a = at::empty({nseg}, x.options());
    s.norms_a = at::empty({nseg > 0 ? nseg : 1, 2}, x.options());
    check(api.cos_fwd(fp(s.vis_h), fp(s.lang_h), idx.data_ptr<int64_t>(), s.nc, (int)s.vis_h.size(1), s.eps_a, fpm(s.score_a), fpm(s.norms_a),
                      stream),
          "irx_cosine_rows_fwd");
    Tensor obj_h = mlp_fwd(s.fc1, P, s.pooled, stream);
    outs = {s.pooled.detach(), s.score_a.detach(), obj_h};
  }
  if (rec) {
    s.wire({feats_in, lang_in}, slot_ptrs, keep);
    torch::autograd::set_history(outs, node);
  }
  return outs;
}

// =================================================================================================================================
// CosineNode: score[i] = cos(a_i, b_{idx[i]}) (irx_cosine_rows_fwd / _bwd; reference models/scene_module.py:104-106,
// relation_module.py:104-105, attribute_module.py:122-126). data inputs a [n][d], b [m][d]; no parameters.
// =================================================================================================================================
struct CosineNode : public HeadNode {
  Tensor a, b, idx, score, norms;
  float eps = 1e-8f;
  std::string name() const override { return "irx::CosineNode"; }
  void release_variables() override { released = true; a = b = idx = score = norms = Tensor(); }
  variable_list apply(variable_list&& grads) override {
    guard();
    at::NoGradGuard ng;
    variable_list out(num_outputs());
    if (!grads[0].defined()) return out;
    const Tensor ds = f32c(grads[0]);
    const bool wa = should_compute_output(0), wb = should_compute_output(1);
    Tensor da = wa ? at::empty_like(a) : Tensor(), db = wb ? at::empty_like(b) : Tensor();
    check(api.cos_bwd(fp(a), fp(b), idx.defined() ? idx.data_ptr<int64_t>() : nullptr, fp(score), fp(norms), fp(ds), (int)a.size(0),
                      (int)b.size(0), (int)a.size(1), eps, fpm(da), fpm(db), stream),
          "irx_cosine_rows_bwd");
    if (wa) out[0] = da;
    if (wb) out[1] = db;
    return out;
  }
};

Tensor cosine_rows(const Tensor& a_in, const Tensor& b_in, const c10::optional<Tensor>& idx, double eps, int64_t stream_i) {
  TORCH_CHECK(api.cos_fwd && api.cos_bwd, "irx nodes: bind_heads() has not been called");
  auto node = std::shared_ptr<CosineNode>(new CosineNode(), torch::autograd::deleteNode);
  CosineNode& s = *node;
  s.stream = (void*)stream_i;
  const bool rec = at::GradMode::is_enabled() && (a_in.requires_grad() || b_in.requires_grad());
  Tensor out;
  {
    at::NoGradGuard ng;
    s.a = f32c(a_in); s.b = f32c(b_in);
    if (idx.has_value() && idx->defined()) s.idx = idx->contiguous();
    s.eps = (float)eps;
    const int64_t n = s.a.size(0);
    TORCH_CHECK(s.a.size(1) == s.b.size(1) && (!s.idx.defined() || (s.idx.size(0) == n && s.idx.scalar_type() == torch::kInt64)),
                "cosine_rows: shapes");
    s.score = at::empty({n}, s.a.options());
    s.norms = at::empty({n > 0 ? n : 1, 2}, s.a.options());
    check(api.cos_fwd(fp(s.a), fp(s.b), s.idx.defined() ? s.idx.data_ptr<int64_t>() : nullptr, (int)n, (int)s.a.size(1), s.eps, fpm(s.score),
                      fpm(s.norms), s.stream),
          "irx_cosine_rows_fwd");
    out = s.score.detach();
  }
  if (rec) {
    s.wire({a_in, b_in}, {}, {});
    torch::autograd::set_history(out, node);
  }
  return out;
}

// =================================================================================================================================
// LossNode: irx_total_loss. data inputs: lang_scores [B][n_lang], seg_scores [B][n_seg], s1, s2, s3 [ns].
// outputs: loss (1,) — the only differentiable one —, ref_loss (1,), lang_loss (), seg_loss (), seg_acc ()
// =================================================================================================================================
struct LossNode : public HeadNode {
  Tensor dstore;
  int64_t o1 = 0, o2 = 0, ns = 0, B = 0, n_lang = 0, n_seg = 0;
  std::string name() const override { return "irx::LossNode"; }
  void release_variables() override { released = true; dstore = Tensor(); }
  variable_list apply(variable_list&& grads) override {
    guard();
    at::NoGradGuard ng;
    variable_list out(num_outputs());
    if (!grads[0].defined()) return out;
    Tensor scaled = dstore * grads[0].reshape({1}).to(dstore.scalar_type());
    if (should_compute_output(0)) out[0] = scaled.narrow(0, 0, o1).view({B, n_lang});
    if (should_compute_output(1)) out[1] = scaled.narrow(0, o1, o2 - o1).view({B, n_seg});
    Tensor ds = scaled.narrow(0, o2, ns);
    for (int i = 2; i < 5; ++i)
      if (should_compute_output(i)) out[i] = ds;
    return out;
  }
};

std::vector<Tensor> total_loss(const Tensor& lang_scores_in, const Tensor& seg_scores_in, const Tensor& s1_in, const Tensor& s2_in,
                               const Tensor& s3_in, const Tensor& lang_label, const Tensor& seg_label, const Tensor& lab,
                               const Tensor& seg_off, const Tensor& keep_t, double gamma, double margin, double ref_weight,
                               int64_t batch_size, int64_t stream_i) {
  TORCH_CHECK(api.total_loss, "irx nodes: bind_heads() has not been called");
  auto node = std::shared_ptr<LossNode>(new LossNode(), torch::autograd::deleteNode);
  LossNode& s = *node;
  const bool rec = at::GradMode::is_enabled() && (lang_scores_in.requires_grad() || seg_scores_in.requires_grad() || s1_in.requires_grad() ||
                                                  s2_in.requires_grad() || s3_in.requires_grad());
  std::vector<Tensor> outs;
  {
    at::NoGradGuard ng;
    const Tensor ls = f32c(lang_scores_in), ss = f32c(seg_scores_in), s1 = f32c(s1_in), s2 = f32c(s2_in), s3 = f32c(s3_in);
    s.B = ls.size(0); s.n_lang = ls.size(1); s.n_seg = ss.size(1); s.ns = s1.size(0);
    const int64_t nscored = keep_t.size(0);
    s.o1 = s.B * s.n_lang; s.o2 = s.o1 + s.B * s.n_seg;
    Tensor out5 = at::empty({5}, ls.options());
    s.dstore = nscored == 0 ? at::zeros({s.o2 + s.ns}, ls.options()) : at::empty({s.o2 + s.ns}, ls.options());
    float* g = s.dstore.data_ptr<float>();
    check(api.total_loss(fp(ls), lang_label.data_ptr<int64_t>(), (int)s.B, (int)s.n_lang, fp(ss), seg_label.data_ptr<int64_t>(), (int)s.n_seg,
                         fp(s1), fp(s2), fp(s3), fp(lab), seg_off.data_ptr<int64_t>(), fp(keep_t), (int)nscored, (float)gamma, (float)margin,
                         (float)ref_weight, (int)batch_size, fpm(out5), g, g + s.o1, g + s.o2, (void*)stream_i),
          "irx_total_loss");
    // (detach(): aliases of out5 that autograd does not track as views — the node attaches to a plain tensor)
    outs = {out5.narrow(0, 0, 1).detach(), out5.narrow(0, 1, 1).detach(), out5.select(0, 2).detach(), out5.select(0, 3).detach(),
            out5.select(0, 4).detach()};
  }
  if (rec) {
    s.stream = (void*)stream_i;
    s.wire({lang_scores_in, seg_scores_in, s1_in, s2_in, s3_in}, {}, {});
    torch::autograd::set_history(outs[0], node);
  }
  return outs;
}

// =================================================================================================================================
// RelationHeadNode (reference models/relation_module.py:84-107, basic_blocks.py:98-133)
//   data input  : lang [B][256] (the language module's relation vector). Node features / positions / kNN grid are data (prepared).
//   parameters  : lang_emb_fc (6), vis_emb_fc (6), gcn.weight.{0,2}.{weight,bias} (4), gcn.mlp.{0,2}.{weight,bias} (4)
//   output      : relation_scores [Nc]
// =================================================================================================================================
struct RelationHeadNode : public HeadNode {
  Tensor feats, pos, qidx, nbr, idx, arg, lang_h, vis_h, score, norms;
  int nq = 0, k = 0, fin = 0, nc = 0, hid = 0, fout = 0, B = 0;
  Mlp lang, vis;
  float eps_cos = 1e-8f;

  std::string name() const override { return "irx::RelationHeadNode"; }
  void release_variables() override {
    released = true;
    feats = pos = qidx = nbr = idx = arg = lang_h = vis_h = score = norms = Tensor();
    lang.reset(); vis.reset();
  }

  variable_list apply(variable_list&& grads) override {
    guard();
    at::NoGradGuard ng;
    variable_list out(num_outputs());
    PGrads pg;
    pg.init(sink, params);
    const Tensor ds = grads[0].defined() ? f32c(grads[0]) : at::zeros_like(score);
    Tensor d_vis_h = at::empty_like(vis_h), d_lang_h = at::empty_like(lang_h);
    check(api.cos_bwd(fp(vis_h), fp(lang_h), idx.data_ptr<int64_t>(), fp(score), fp(norms), fp(ds), nq, B, (int)vis_h.size(1), eps_cos,
                      fpm(d_vis_h), fpm(d_lang_h), stream),
          "irx_cosine_rows_bwd");
    Tensor d_gcn = mlp_bwd(vis, params, d_vis_h, true, pg, stream);
    const float* pp[8];
    float* gp[8];
    for (int i = 0; i < 8; ++i) { pp[i] = fp(params[12 + i]); gp[i] = pg.ptr[12 + i]; }
    const size_t wsb = api.ec_ws(nq, k, fin, nc, hid, fout);
    Tensor ws = bytes(wsb, feats);
    check(api.ec_bwd(fp(feats), fp(pos), qidx.data_ptr<int64_t>(), nbr.data_ptr<int32_t>(), nq, k, fin, nc, hid, fout, pp, fp(d_gcn),
                     arg.data_ptr<int32_t>(), gp, nullptr, ws.data_ptr(), wsb, stream),
          "irx_edgeconv_max_bwd");
    Tensor dlang = mlp_bwd(lang, params, d_lang_h, should_compute_output(0), pg, stream);
    if (should_compute_output(0)) out[0] = dlang;
    finish(pg, out);
    return out;
  }
};

std::vector<Tensor> relation_head(const Tensor& lang_in, const Tensor& feats, const Tensor& pos, const Tensor& qidx, const Tensor& nbr,
                                  const Tensor& idx, int64_t nc, std::vector<Tensor> params, std::vector<Tensor> stats,
                                  std::vector<double> f, std::vector<int64_t> seeds, int64_t stream_i, std::vector<int64_t> slot_ptrs,
                                  std::vector<Tensor> keep) {
  TORCH_CHECK(api.ec_fwd && api.cos_fwd && api.mlp2_fwd, "irx nodes: bind_heads() has not been called");
  TORCH_CHECK(params.size() == 20 && stats.size() == 2 && f.size() == 6 && seeds.size() == 2, "relation_head: argument lists");
  void* stream = (void*)stream_i;
  auto node = std::shared_ptr<RelationHeadNode>(new RelationHeadNode(), torch::autograd::deleteNode);
  RelationHeadNode& s = *node;
  s.params = std::move(params);
  s.stream = stream;
  const bool rec = at::GradMode::is_enabled() && (lang_in.requires_grad() || s.params[0].requires_grad());
  std::vector<Tensor> outs;
  {
    at::NoGradGuard ng;
    const std::vector<Tensor>& P = s.params;
    s.feats = f32c(feats); s.pos = f32c(pos); s.qidx = qidx.contiguous(); s.nbr = nbr.contiguous(); s.idx = idx;
    TORCH_CHECK(s.qidx.scalar_type() == torch::kInt64 && s.nbr.scalar_type() == torch::kInt32 && idx.scalar_type() == torch::kInt64,
                "relation_head: index tensors");
    s.nq = (int)s.nbr.size(0); s.k = (int)s.nbr.size(1); s.fin = (int)s.feats.size(1); s.nc = (int)nc;
    s.hid = (int)P[12].size(0); s.fout = (int)P[18].size(0); s.B = (int)lang_in.size(0);
    s.lang.p0 = 0; s.lang.norm = 1; s.lang.rmean = stats[0]; s.lang.rvar = stats[1]; s.lang.eps = (float)f[0]; s.lang.momentum = (float)f[1];
    s.lang.drop_p = f[2]; s.lang.seed = (unsigned long long)seeds[0];
    s.vis.p0 = 6; s.vis.norm = 3; s.vis.eps = (float)f[3]; s.vis.drop_p = f[4]; s.vis.seed = (unsigned long long)seeds[1];
    s.eps_cos = (float)f[5];
    s.lang_h = mlp_fwd(s.lang, P, lang_in, stream);
    Tensor gcn = at::empty({s.nq, s.fout}, s.feats.options());
    s.arg = at::empty({s.nq, s.fout}, s.feats.options().dtype(torch::kInt32));
    const float* pp[8];
    for (int i = 0; i < 8; ++i) pp[i] = fp(P[12 + i]);
    check(api.ec_fwd(fp(s.feats), fp(s.pos), s.qidx.data_ptr<int64_t>(), s.nbr.data_ptr<int32_t>(), s.nq, s.k, s.fin, s.nc, s.hid, s.fout, pp,
                     fpm(gcn), s.arg.data_ptr<int32_t>(), stream),
          "irx_edgeconv_max_fwd");
    s.vis_h = mlp_fwd(s.vis, P, gcn, stream);
    s.score = at::empty({s.nq}, s.feats.options());
    s.norms = at::empty({s.nq > 0 ? s.nq : 1, 2}, s.feats.options());
    check(api.cos_fwd(fp(s.vis_h), fp(s.lang_h), idx.data_ptr<int64_t>(), s.nq, (int)s.vis_h.size(1), s.eps_cos, fpm(s.score), fpm(s.norms),
                      stream),
          "irx_cosine_rows_fwd");
    outs = {s.score.detach()};
  }
  if (rec) {
    s.wire({lang_in}, slot_ptrs, keep);
    torch::autograd::set_history(outs, node);
  }
  return outs;
}

// =================================================================================================================================
// LangPoolNode: the language module's four attention heads (reference models/lang_module.py:61-83; irx_lang_pool_fwd / _bwd)
//   data inputs : feats [B][T][O] (GRU output), embed [B][T][E] (projected words)
//   parameters  : fc_a / fc_cls / fc_rel / fc_scene .{weight [1][O], bias [1]} (8)
//   outputs     : att [B][T][4], pooled vectors attr / cls / rel / scene, each [B][E] contiguous (no select nodes behind them)
// =================================================================================================================================
struct LangPoolNode : public HeadNode {
  Tensor feats, embed, length, att, prob, qsum;
  int B = 0, T = 0, O = 0, E = 0;
  std::string name() const override { return "irx::LangPoolNode"; }
  void release_variables() override { released = true; feats = embed = length = att = prob = qsum = Tensor(); }
  variable_list apply(variable_list&& grads) override {
    guard();
    at::NoGradGuard ng;
    variable_list out(num_outputs());
    PGrads pg;
    pg.init(sink, params);
    // d pooled [B][4][E] from the four vectors' gradients (a missing one is zero)
    Tensor dpooled = at::zeros({B, 4, E}, feats.options());
    for (int h = 0; h < 4; ++h)
      if (grads[1 + h].defined()) dpooled.select(1, h).copy_(grads[1 + h]);
    const Tensor datt = grads[0].defined() ? f32c(grads[0]) : Tensor();
    Tensor dfeats = at::empty_like(feats), dembed = at::empty_like(embed);
    Tensor dwb = at::empty({4 * (int64_t)O + 4}, feats.options()), part = at::empty({(int64_t)B * 4 * (O + 1)}, feats.options());
    const float* wp[4];
    for (int h = 0; h < 4; ++h) wp[h] = fp(params[2 * h]);
    check(api.lp_bwd(fp(feats), fp(embed), length.data_ptr<int64_t>(), B, T, O, E, wp, fp(att), fp(prob), fp(qsum), fp(dpooled), fp(datt),
                     fpm(dfeats), fpm(dembed), fpm(dwb), fpm(dwb) + 4 * (size_t)O, fpm(part), stream),
          "irx_lang_pool_bwd");
    for (int h = 0; h < 4; ++h) {
      pg.view(2 * h).copy_(dwb.narrow(0, (int64_t)h * O, O).view({1, O}));
      pg.view(2 * h + 1).copy_(dwb.narrow(0, 4 * (int64_t)O + h, 1));
    }
    if (should_compute_output(0)) out[0] = dfeats;
    if (should_compute_output(1)) out[1] = dembed;
    finish(pg, out);
    return out;
  }
};

std::vector<Tensor> lang_pool(const Tensor& feats_in, const Tensor& embed_in, const Tensor& length, std::vector<Tensor> params, int64_t stream_i,
                              std::vector<int64_t> slot_ptrs, std::vector<Tensor> keep) {
  TORCH_CHECK(api.lp_fwd && api.lp_bwd, "irx nodes: bind_heads() has not been called");
  TORCH_CHECK(params.size() == 8, "lang_pool: four (weight, bias) pairs");
  void* stream = (void*)stream_i;
  auto node = std::shared_ptr<LangPoolNode>(new LangPoolNode(), torch::autograd::deleteNode);
  LangPoolNode& s = *node;
  s.params = std::move(params);
  s.stream = stream;
  const bool rec = at::GradMode::is_enabled() && (feats_in.requires_grad() || embed_in.requires_grad() || s.params[0].requires_grad());
  std::vector<Tensor> outs;
  {
    at::NoGradGuard ng;
    s.feats = f32c(feats_in); s.embed = f32c(embed_in);
    s.length = length.contiguous().to(torch::kInt64);
    s.B = (int)s.feats.size(0); s.T = (int)s.feats.size(1); s.O = (int)s.feats.size(2); s.E = (int)s.embed.size(2);
    TORCH_CHECK(s.embed.size(0) == s.B && s.embed.size(1) == s.T, "lang_pool: feats / embed shapes");
    const auto opt = s.feats.options();
    s.att = at::empty({s.B, s.T, 4}, opt);
    s.prob = at::empty({s.B, s.T, 4}, opt);
    s.qsum = at::empty({s.B, 4}, opt);
    Tensor pooled = at::empty({s.B, 4, s.E}, opt);
    const float* wp[4];
    const float* bp[4];
    for (int h = 0; h < 4; ++h) { wp[h] = fp(s.params[2 * h]); bp[h] = fp(s.params[2 * h + 1]); }
    check(api.lp_fwd(fp(s.feats), fp(s.embed), s.length.data_ptr<int64_t>(), s.B, s.T, s.O, s.E, wp, bp, fpm(s.att), fpm(s.prob), fpm(s.qsum),
                     fpm(pooled), stream),
          "irx_lang_pool_fwd");
    outs = {s.att.detach()};
    for (int h = 0; h < 4; ++h) outs.push_back(pooled.select(1, h).contiguous());     // four small copies instead of four select nodes
  }
  if (rec) {
    s.wire({feats_in, embed_in}, slot_ptrs, keep);
    torch::autograd::set_history(outs, node);
  }
  return outs;
}

void bind_heads(const std::unordered_map<std::string, uint64_t>& addr) {
  auto get = [&](const char* name) -> uint64_t {
    auto it = addr.find(name);
    TORCH_CHECK(it != addr.end() && it->second != 0, "irx nodes: missing entry point ", name);
    return it->second;
  };
  api.mlp2_saved_floats = (saved_floats_fn)get("irx_mlp2_saved_floats");
  api.mlp2_fwd = (mlp2_fwd_fn)get("irx_mlp2_fwd");
  api.mlp2_bwd = (mlp2_bwd_fn)get("irx_mlp2_bwd");
  api.segmax = (segmax_fn)get("irx_segment_max");
  api.segmax_bwd = (segmax_bwd_fn)get("irx_segment_max_backward");
  api.cos_fwd = (cos_fwd_fn)get("irx_cosine_rows_fwd");
  api.cos_bwd = (cos_bwd_fn)get("irx_cosine_rows_bwd");
  api.conv_ws = (conv_ws_fn)get("irx_spconv_fwd_workspace_bytes");
  api.conv_fwd = (conv_fwd_fn)get("irx_spconv_fwd");
  api.wgrad_ws = (wgrad_ws_fn)get("irx_spconv_wgrad_workspace_bytes");
  api.wgrad = (wgrad_fn)get("irx_spconv_wgrad");
  api.bn_ws = (bn_ws_fn)get("irx_bn_workspace_bytes");
  api.bn_fwd = (bn_fwd_fn)get("irx_bn_forward");
  api.bn_bwd = (bn_bwd_fn)get("irx_bn_backward");
  api.kdt = (kdt_fn)get("irx_kmap_down_transpose");
  api.attn_fwd = (attn_fwd_fn)get("irx_attn_pool_fwd");
  api.attn_bwd = (attn_bwd_fn)get("irx_attn_pool_bwd");
  api.dropout = (drop_fn)get("irx_dropout_flat");
  api.fork = (fork_fn)get("irx_stream_fork");
  api.total_loss = (loss_fn)get("irx_total_loss");
  api.ec_ws = (ec_ws_fn)get("irx_edgeconv_workspace_bytes");
  api.ec_fwd = (ec_fwd_fn)get("irx_edgeconv_max_fwd");
  api.ec_bwd = (ec_bwd_fn)get("irx_edgeconv_max_bwd");
  api.lp_fwd = (lp_fwd_fn)get("irx_lang_pool_fwd");
  api.lp_bwd = (lp_bwd_fn)get("irx_lang_pool_bwd");
}

}  // namespace

void register_heads(pybind11::module& m) {
  m.def("bind_heads", &bind_heads);
  m.def("scene_head", &scene_head);
  m.def("attr_head", &attr_head);
  m.def("cosine_rows", &cosine_rows);
  m.def("total_loss", &total_loss);
  m.def("relation_head", &relation_head);
  m.def("lang_pool", &lang_pool);
}

}  // namespace irxn
