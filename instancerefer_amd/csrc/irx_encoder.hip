// irx_encoder.hip — whole-encoder executor: the 13 Conv3d -> BatchNorm (-> + residual) -> ReLU groups of
// SparseConvEncoder / BEVEncoder (reference models/basic_blocks.py:59-95,136-171) issued by ONE C-ABI call per direction.
//
// The Python executor (one ctypes call + ~2 tensor allocations per kernel) cost ~0.55 ms forward and ~0.9 ms backward
// per encoder of pure host time; with the step host-bound at ~14 ms that was ~20 % of it.  Here the host walks a
// descriptor table and launches the same kernels in the same order (results are bit-identical to the per-layer
// C-ABI calls); the caller owns every buffer (one activation / gradient arena) and the workspace.
#include "irx_common.h"
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include "../../include/irx.h"
#include <atomic>
#include <chrono>
#include <mutex>

static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

namespace {
struct Layer {
  int K, cin, cout, n_in, n_out, res;
  const int32_t* tbl; int ld;
  const int32_t* tbl_b; int ld_b, flip_b;
  const int32_t *pair_in, *pair_out, *pair_counts; int ld_pairs;
  const float *w, *gamma, *beta; float *running_mean, *running_var;
  float *x, *c, *y, *mean, *invstd;
  float *dw, *dgamma, *dbeta, *gy;
  float eps, momentum;
  int store;   // bf16 storage of the activations / gradients in flight (IRX_ENC_STORE)
  const int32_t* order;   // launch order of the output tiles or NULL (IRX_ENC_ORDER)
  void** prof; // 6 event handles or NULL (IRX_ENC_PROF)
};

Layer unpack(const int64_t* d, const double* f) {
  Layer L;
  L.K = (int)d[IRX_ENC_K]; L.cin = (int)d[IRX_ENC_CIN]; L.cout = (int)d[IRX_ENC_COUT];
  L.n_in = (int)d[IRX_ENC_N_IN]; L.n_out = (int)d[IRX_ENC_N_OUT]; L.res = (int)d[IRX_ENC_RES];
  L.tbl = (const int32_t*)d[IRX_ENC_TBL]; L.ld = (int)d[IRX_ENC_LD];
  L.tbl_b = (const int32_t*)d[IRX_ENC_TBL_B]; L.ld_b = (int)d[IRX_ENC_LD_B]; L.flip_b = (int)d[IRX_ENC_FLIP_B];
  L.pair_in = (const int32_t*)d[IRX_ENC_PAIR_IN]; L.pair_out = (const int32_t*)d[IRX_ENC_PAIR_OUT];
  L.pair_counts = (const int32_t*)d[IRX_ENC_PAIR_COUNTS]; L.ld_pairs = (int)d[IRX_ENC_LD_PAIRS];
  L.w = (const float*)d[IRX_ENC_W]; L.gamma = (const float*)d[IRX_ENC_GAMMA]; L.beta = (const float*)d[IRX_ENC_BETA];
  L.running_mean = (float*)d[IRX_ENC_RUNNING_MEAN]; L.running_var = (float*)d[IRX_ENC_RUNNING_VAR];
  L.x = (float*)d[IRX_ENC_X]; L.c = (float*)d[IRX_ENC_C]; L.y = (float*)d[IRX_ENC_Y];
  L.mean = (float*)d[IRX_ENC_MEAN]; L.invstd = (float*)d[IRX_ENC_INVSTD];
  L.dw = (float*)d[IRX_ENC_DW]; L.dgamma = (float*)d[IRX_ENC_DGAMMA]; L.dbeta = (float*)d[IRX_ENC_DBETA];
  L.gy = (float*)d[IRX_ENC_GY];
  L.store = (int)d[IRX_ENC_STORE];
  L.order = (const int32_t*)d[IRX_ENC_ORDER];
  L.prof = (void**)d[IRX_ENC_PROF];
  L.eps = (float)f[0]; L.momentum = (float)f[1];
  return L;
}

bool pairs_path(const Layer& L) {
  auto ok = [](int c) { return c == 32 || c == 64 || c == 128; };
  return L.pair_in != nullptr && ok(L.cin) && ok(L.cout);
}

// workspace regions: [conv | wgrad | bn]
struct Regions { size_t conv, wgrad, bn, wimg; };
Regions regions(const int64_t* desc, const double* fdesc, int n, bool backward) {
  Regions r = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const Layer L = unpack(desc + (size_t)i * IRX_ENC_NFIELDS, fdesc + (size_t)i * 2);
    size_t c, b = irx_bn_workspace_bytes(L.n_out, L.cout), wg = 0;
    if (!backward) {
      c = irx_spconv_fwd_workspace_bytes(L.n_out, L.K, L.cin, L.cout, 0);
    } else {
      c = irx_spconv_fwd_workspace_bytes(L.n_in, L.K, L.cout, L.cin, 1);
      wg = pairs_path(L) ? irx_spconv_wgrad_pairs_workspace_bytes(L.n_out, L.K, L.cin, L.cout)
                         : irx_spconv_wgrad_workspace_bytes(L.n_out, L.K, L.cin, L.cout);
    }
    r.wimg += align256((size_t)L.K * L.cin * L.cout * sizeof(float));
    if (c > r.conv) r.conv = c;
    if (wg > r.wgrad) r.wgrad = wg;
    if (b > r.bn) r.bn = b;
  }
  r.conv = align256(r.conv); r.wgrad = align256(r.wgrad); r.bn = align256(r.bn);
  return r;
}
}  // namespace

extern "C" size_t irx_encoder_workspace_bytes(const int64_t* desc, const double* fdesc, int n_layers, int backward) {
  if (!desc || !fdesc || n_layers <= 0) return 0;
  const IrxModeScope pin((int)desc[IRX_ENC_MODE]);
  const Regions r = regions(desc, fdesc, n_layers, backward != 0);
  return r.conv + r.wgrad + r.bn + r.wimg + 256;
}

namespace {
// Sync BatchNorm inside the one-call executor (include/irx.h irx_encoder_forward_sync): the fold over the ranks is the CALLER's
// (torch.distributed), reached through a callback between the statistics and the apply pass of every layer.
struct EncSync {
  irx_allreduce_fn cb = nullptr;
  void* user = nullptr;
  double* sums = nullptr;    // [n_layers][IRX_ENC_SYNC_STRIDE] float64: sum x | sum x^2 | row count   (kept for the backward)
  float* gsums = nullptr;    // [n_layers][IRX_ENC_SYNC_STRIDE] float32: sum g | sum g xhat (backward)
};
int encoder_forward_impl(const int64_t* desc, const double* fdesc, int n_layers, void* workspace, size_t workspace_bytes,
                         void* stream, const EncSync& sync);
int encoder_backward_impl(const int64_t* desc, const double* fdesc, int n_layers, float* dc_scratch, float* dx0, void* workspace,
                          size_t workspace_bytes, void* stream, const EncSync& sync);
}  // namespace

// ---- backward gate between two encoder passes of one training step (irx_encoder_gate_next, include/irx.h) ----
// Two encoders' backward passes run on two streams.  Each walks its pyramid from the smallest level up: first a chain of ~100
// short, latency-bound kernels, then a few long ones that fill every CU.  Left alone, the pass that reaches its long kernels
// first starves the other's short chain (a short kernel only gets CUs when a long kernel's workgroups retire, all at once, every
// 30-90 us): the candidate encoder's small levels took 2.6 ms beside the scene encoder's large ones against 0.5 ms alone
// (profiles/r05_timeline_bf16.txt).  The gate orders them: the RECORDER marks the point where its levels below `rows` are done,
// the WAITER does not start its levels of `rows` or more before that point.
// a polite spin (the x86 pause intrinsic is not portable host code)
static inline void irx_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#else
  std::this_thread::yield();
#endif
}
struct EncGate { int role = 0; int64_t rows = 0; uint64_t token = 0; };   // role 1 = recorder, 2 = waiter
static thread_local EncGate g_gate_next;
static std::atomic<uint64_t> g_gate_recorded{0};
static hipEvent_t g_gate_ev[2] = {nullptr, nullptr};
static std::mutex g_gate_mu;
static hipEvent_t gate_event(uint64_t token) {
  std::lock_guard<std::mutex> lk(g_gate_mu);
  hipEvent_t& e = g_gate_ev[token & 1];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
  return e;
}
extern "C" int irx_encoder_gate_next(int role, long long rows, unsigned long long token) {
  IRX_REQUIRE(role >= 0 && role <= 2, "irx_encoder_gate_next: role %d", role);
  g_gate_next.role = role;
  g_gate_next.rows = rows;
  g_gate_next.token = token;
  return IRX_OK;
}
// at the head of backward layer `n_out` rows (or behind the last layer: n_out < 0); true once the gate has fired
static bool gate_step(const EncGate& g, bool fired, int64_t n_out, void* stream) {
  if (!g.role || fired || (n_out >= 0 && n_out < g.rows)) return fired;
  hipEvent_t e = gate_event(g.token);
  if (!e) return true;
  if (g.role == 1) {
    if (hipEventRecord(e, (hipStream_t)stream) == hipSuccess) g_gate_recorded.store(g.token, std::memory_order_release);
  } else if (n_out >= 0) {
    // the recorder's mark must be ENQUEUED before this wait is (a wait on an event without a pending record is a no-op); it is
    // issued by another thread at about the same time: poll briefly, go on without the gate if it does not show up
    const auto t0 = std::chrono::steady_clock::now();
    while (g_gate_recorded.load(std::memory_order_acquire) < g.token &&
           std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(1500))
      irx_cpu_relax();
    if (g_gate_recorded.load(std::memory_order_acquire) == g.token) (void)hipStreamWaitEvent((hipStream_t)stream, e, 0);
  }
  return true;
}

extern "C" int irx_encoder_forward(const int64_t* desc, const double* fdesc, int n_layers, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return encoder_forward_impl(desc, fdesc, n_layers, workspace, workspace_bytes, stream, EncSync());
}

extern "C" int irx_encoder_forward_sync(const int64_t* desc, const double* fdesc, int n_layers, void* workspace,
                                        size_t workspace_bytes, void* stream, double* sums, irx_allreduce_fn allreduce,
                                        void* user) {
  IRX_REQUIRE(sums && allreduce, "irx_encoder_forward_sync: null sums buffer / callback");
  EncSync sy;
  sy.cb = allreduce; sy.user = user; sy.sums = sums;
  return encoder_forward_impl(desc, fdesc, n_layers, workspace, workspace_bytes, stream, sy);
}

extern "C" int irx_encoder_backward_sync(const int64_t* desc, const double* fdesc, int n_layers, float* dc_scratch, float* dx0,
                                         void* workspace, size_t workspace_bytes, void* stream, double* sums, float* gsums,
                                         irx_allreduce_fn allreduce, void* user) {
  IRX_REQUIRE(sums && gsums && allreduce, "irx_encoder_backward_sync: null buffers / callback");
  EncSync sy;
  sy.cb = allreduce; sy.user = user; sy.sums = sums; sy.gsums = gsums;
  return encoder_backward_impl(desc, fdesc, n_layers, dc_scratch, dx0, workspace, workspace_bytes, stream, sy);
}

namespace {
int encoder_forward_impl(const int64_t* desc, const double* fdesc, int n_layers, void* workspace, size_t workspace_bytes,
                         void* stream, const EncSync& sync) {
  IRX_REQUIRE(desc && fdesc && n_layers > 0, "irx_encoder_forward: empty descriptor table");
  const int mode = (int)desc[IRX_ENC_MODE];
  IRX_REQUIRE(mode >= 0 && mode <= 2, "irx_encoder_forward: IRX_ENC_MODE = %d", mode);
  const IrxModeScope pin(mode);                      // every launch of this call uses the table's mode
  const Regions r = regions(desc, fdesc, n_layers, false);
  IRX_REQUIRE(workspace && workspace_bytes >= r.conv + r.bn + r.wimg + 255, "irx_encoder_forward: workspace %zu < %zu",
              workspace_bytes, r.conv + r.bn + r.wimg + 255);
  char* ws_c = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  char* ws_b = ws_c + r.conv;
  // fragment-major weight images of every fast-path layer: one launch
  const float* wimg[64] = {nullptr};
  IRX_REQUIRE(n_layers <= 16, "irx_encoder_forward: more than 16 layers");
  const int st = (int)desc[IRX_ENC_STORE];
  {
    IrxPermuteJobs J, J3;                         // second- / third-generation images (the latter: bf16 inputs, irx_spconv3.hip)
    J.n = J3.n = 0;
    char* p = ws_b + r.bn;
    size_t run = 0;
    for (int i = 0; i < n_layers; ++i) {
      const Layer L = unpack(desc + (size_t)i * IRX_ENC_NFIELDS, fdesc + (size_t)i * 2);
      if (L.n_out > 0 && !irx_stem_supported(L.K, L.cin, L.cout) && irx_spconv_fast_path(L.x, L.w, L.c, L.cin, L.cout, 0)) {
        if (irx_spconv3_use(L.cin, L.cout, (st && i > 0) ? 1 : 0, L.n_in, 0)) {
          const int j = J3.n++;
          J3.w[j] = L.w; J3.dst[j] = (float*)p; J3.K[j] = L.K; J3.cin[j] = L.cin; J3.cout[j] = L.cout;
        } else {
          const int j = J.n++;
          J.w[j] = L.w; J.dst[j] = (float*)p; J.K[j] = L.K; J.cin[j] = L.cin; J.cout[j] = L.cout;
          run += (size_t)L.K * L.cin * L.cout / 4;
          J.end4[j] = run;
        }
        wimg[i] = (const float*)p;
      }
      p += align256((size_t)L.K * L.cin * L.cout * sizeof(float));
    }
    int rc = irx_permute_w_multi_launch(J, 0, (hipStream_t)stream);
    if (rc) return rc;
    rc = irx_permute_w3_multi_launch(J3, 0, (hipStream_t)stream);
    if (rc) return rc;
  }
  IRX_REQUIRE(!st || irx_conv_bf16(), "irx_encoder_forward: bf16 storage needs a bf16 compute mode in IRX_ENC_MODE");
  for (int i = 0; i < n_layers; ++i) {
    const Layer L = unpack(desc + (size_t)i * IRX_ENC_NFIELDS, fdesc + (size_t)i * 2);
    IRX_REQUIRE(L.res < i, "irx_encoder_forward: layer %d takes its residual from a later layer", i);
    IRX_REQUIRE(L.store == st && (!st || L.res < n_layers - 1), "irx_encoder_forward: inconsistent storage flags");
    IrxStore ty;
    ty.x = (st && i > 0) ? 1 : 0;                 // the encoder's input features are fp32
    ty.y = st;                                    // conv output c
    ty.order = L.order;
    ty.x_rows = L.n_in;
    const int y_bf = (st && i < n_layers - 1) ? 1 : 0;   // the encoder's output stays fp32
    // offset-split layers (the small levels): the slabs are folded by the BatchNorm statistics pass instead of a reduce launch
    float* slabs = nullptr;
    int nslabs = 0;
    if (!sync.cb && irx_knob(IRX_KNOB_FOLD_SLABS) != 0) {         // ("enc_fold_slabs" / IRX_ENC_FOLD_SLABS=0: dev A/B knob)
      ty.slabs_out = &slabs;
      ty.splits_out = &nslabs;
    }
    if (L.prof) irx_profile_next_kernel(L.prof[0], L.prof[1]);
    int rc = irx_spconv_fwd_impl(L.x, L.w, L.tbl, L.ld, L.n_out, L.K, L.cin, L.cout, 0, 0, L.c, 0, wimg[i], ws_c, r.conv,
                                 stream, ty);
    if (rc) return rc;
    if (sync.cb) {
      // this rank's float64 sums and row count -> the caller's all-reduce -> mean / invstd from the folded sums (the global count
      // stays on the device: sums[2 cout])
      IRX_REQUIRE(L.cout <= (IRX_ENC_SYNC_STRIDE - 2) / 2, "irx_encoder_forward_sync: %d channels exceed the sums stride", L.cout);
      double* sl = sync.sums + (size_t)i * IRX_ENC_SYNC_STRIDE;
      rc = irx_bn_sums_t(L.c, L.n_out, L.cout, sl, ws_b, r.bn, stream, st, sl + 2 * L.cout);
      if (rc) return rc;
      rc = sync.cb(sync.user, sl, 2 * L.cout + 1, 1, stream);
      IRX_REQUIRE(rc == 0, "irx_encoder_forward_sync: the all-reduce callback failed (%d) at layer %d", rc, i);
      rc = irx_bn_stats_from_sums(sl, 0.0, L.cout, L.eps, L.momentum, L.mean, L.invstd, L.running_mean, L.running_var, stream);
    }
    if (rc) return rc;
    const float* res = nullptr;
    if (L.res >= 0) res = (const float*)desc[(size_t)L.res * IRX_ENC_NFIELDS + IRX_ENC_Y];
    if (sync.cb)
      rc = irx_bn_apply_t(L.c, L.n_out, L.cout, L.mean, L.invstd, L.gamma, L.beta, res, 1, L.y, stream, st, st, y_bf);
    else if (slabs)
      rc = irx_bn_forward_slabs_t(slabs, nslabs, L.n_out, L.cout, L.eps, L.momentum, L.gamma, L.beta, res, 1, L.mean, L.invstd,
                                  L.running_mean, L.running_var, L.c, L.y, ws_b, r.bn, stream, st, st, y_bf);
    else      // statistics + apply: one launch for the levels that stay on-die (irx_norm.hip, k_bn_slice_fwd), two + one otherwise
      rc = irx_bn_forward_t(L.c, L.n_out, L.cout, L.eps, L.momentum, L.gamma, L.beta, res, 1, L.mean, L.invstd, L.running_mean,
                            L.running_var, L.y, ws_b, r.bn, stream, st, st, y_bf);
    if (rc) return rc;
  }
  return IRX_OK;
}
}  // namespace

// gy of the last layer holds d(loss)/d(output) on entry.  On return dw / dgamma / dbeta of every layer are written and,
// when dx0 != NULL, dx0 [n_in0][cin0] = d(loss)/d(input features).  gy of the other layers is scratch.
extern "C" int irx_encoder_backward(const int64_t* desc, const double* fdesc, int n_layers, float* dc_scratch,
                                    float* dx0, void* workspace, size_t workspace_bytes, void* stream) {
  return encoder_backward_impl(desc, fdesc, n_layers, dc_scratch, dx0, workspace, workspace_bytes, stream, EncSync());
}

namespace {
int encoder_backward_impl(const int64_t* desc, const double* fdesc, int n_layers, float* dc_scratch, float* dx0, void* workspace,
                          size_t workspace_bytes, void* stream, const EncSync& sync) {
  IRX_REQUIRE(desc && fdesc && n_layers > 0 && dc_scratch, "irx_encoder_backward: bad arguments");
  const int mode = (int)desc[IRX_ENC_MODE];
  IRX_REQUIRE(mode >= 0 && mode <= 2, "irx_encoder_backward: IRX_ENC_MODE = %d", mode);
  const IrxModeScope pin(mode);                      // the mode the forward pass ran under, whatever the setting is now
  const Regions r = regions(desc, fdesc, n_layers, true);
  IRX_REQUIRE(workspace && workspace_bytes >= r.conv + r.wgrad + r.bn + r.wimg + 255,
              "irx_encoder_backward: workspace %zu < %zu", workspace_bytes, r.conv + r.wgrad + r.bn + r.wimg + 255);
  char* ws_c = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  char* ws_w = ws_c + r.conv;
  char* ws_b = ws_w + r.wgrad;
  // data-gradient weight images (kernel-relative: reduction = conv Cout, outputs = conv Cin): one launch
  const float* wimg[64] = {nullptr};
  IRX_REQUIRE(n_layers <= 16, "irx_encoder_backward: more than 16 layers");
  {
    IrxPermuteJobs J, J3;
    J.n = J3.n = 0;
    const int st_ = (int)desc[IRX_ENC_STORE];
    char* p = ws_b + r.bn;
    size_t run = 0;
    for (int i = 0; i < n_layers; ++i) {
      const Layer L = unpack(desc + (size_t)i * IRX_ENC_NFIELDS, fdesc + (size_t)i * 2);
      const float* dxp = (i > 0) ? (const float*)desc[(size_t)(i - 1) * IRX_ENC_NFIELDS + IRX_ENC_GY] : dx0;
      if (dxp && L.n_in > 0 && irx_spconv_fast_path(dc_scratch, L.w, dxp, L.cout, L.cin, 1)) {
        if (irx_spconv3_use(L.cout, L.cin, st_, L.n_out, 0)) {      // x = d c [n_out][cout] in bf16
          const int j = J3.n++;
          J3.w[j] = L.w; J3.dst[j] = (float*)p; J3.K[j] = L.K; J3.cin[j] = L.cout; J3.cout[j] = L.cin;
        } else {
          const int j = J.n++;
          J.w[j] = L.w; J.dst[j] = (float*)p; J.K[j] = L.K; J.cin[j] = L.cout; J.cout[j] = L.cin;
          run += (size_t)L.K * L.cin * L.cout / 4;
          J.end4[j] = run;
        }
        wimg[i] = (const float*)p;
      }
      p += align256((size_t)L.K * L.cin * L.cout * sizeof(float));
    }
    int rc = irx_permute_w_multi_launch(J, 1, (hipStream_t)stream);
    if (rc) return rc;
    rc = irx_permute_w3_multi_launch(J3, 1, (hipStream_t)stream);
    if (rc) return rc;
  }
  // a layer whose OUTPUT feeds a later layer's shortcut receives that share (dresidual) first; the main-path
  // gradient of the layer after it is then accumulated on top
  bool is_res_source[64] = {false};
  IRX_REQUIRE(n_layers <= 64, "irx_encoder_backward: more than 64 layers");
  for (int i = 0; i < n_layers; ++i) {
    const int res = (int)desc[(size_t)i * IRX_ENC_NFIELDS + IRX_ENC_RES];
    if (res >= 0) is_res_source[res] = true;
  }
  const int st = (int)desc[IRX_ENC_STORE];
  IRX_REQUIRE(!st || irx_conv_bf16(), "irx_encoder_backward: bf16 storage needs a bf16 compute mode in IRX_ENC_MODE");
  IRX_REQUIRE(!st || !dx0, "irx_encoder_backward: the input gradient is not available with bf16 storage");
  // Weight gradients on a second stream (IRX_ENC_DC2, round 6). The chain of a backward pass is BatchNorm backward -> data gradient,
  // layer after layer; a layer's weight gradient reads the same d c and the saved input but feeds nothing downstream, and on the
  // large levels it is a third of the pass (profiles/r06_e_stepdump.txt: the scene encoder's last 0.6 ms run one kernel at a time, none
  // of them above 0.4 of its HBM bound). With two d c scratches alternating by layer the weight gradient of layer i runs on the
  // library's own stream while the chain goes on with layer i - 1; events order (a) wgrad(i) behind BN-backward(i), (b) BN-backward(i - 2)
  // — which overwrites the scratch wgrad(i) reads — behind wgrad(i), (c) the caller's stream behind every wgrad at the end.
  float* dc2 = (float*)desc[IRX_ENC_DC2];
  const bool two = dc2 != nullptr && !sync.cb && n_layers <= 16;
  hipStream_t sW = (hipStream_t)stream;
  hipEvent_t* ev = nullptr;                  // [0..16): BN-backward(i) done, [16..32): wgrad(i) done
  if (two) {
    static thread_local hipStream_t t_stream = nullptr;
    static thread_local hipEvent_t t_ev[32] = {nullptr};
    if (!t_ev[0])
      for (int e = 0; e < 32; ++e) IRX_CHECK_HIP(hipEventCreateWithFlags(&t_ev[e], hipEventDisableTiming), "irx_encoder_backward(event)");
    sW = (hipStream_t)desc[IRX_ENC_WSTREAM];
    if (!sW) {
      if (!t_stream) IRX_CHECK_HIP(hipStreamCreateWithFlags(&t_stream, hipStreamNonBlocking), "irx_encoder_backward(stream)");
      sW = t_stream;
    }
    ev = t_ev;
  }
  float* const dc_main = dc_scratch;
  const int64_t wrows = two ? desc[IRX_ENC_WROWS] : 0;
  IrxDySlabs pending;                        // gy of the layer about to be processed, still as its producer's offset-split slabs
  const EncGate gate = g_gate_next;          // (set for this call by irx_encoder_gate_next on this thread, or by the lane)
  g_gate_next = EncGate();
  bool gate_fired = false;
  for (int i = n_layers - 1; i >= 0; --i) {
    const Layer L = unpack(desc + (size_t)i * IRX_ENC_NFIELDS, fdesc + (size_t)i * 2);
    gate_fired = gate_step(gate, gate_fired, L.n_out, stream);
    if (two) {
      dc_scratch = (i & 1) ? dc2 : dc_main;
      // the weight gradient of layer i + 2 read this scratch: it must be done before BatchNorm backward overwrites it
      if (i + 2 < n_layers) IRX_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, ev[16 + i + 2], 0), "irx_encoder_backward(wait W)");
    }
    const bool lent = two && (wrows <= 0 || L.n_out < wrows);       // this layer's weight gradient on the second stream?
    void* const wstream = lent ? (void*)sW : stream;
    const IrxDySlabs dys = pending;
    pending = IrxDySlabs();
    float* dres = nullptr;
    if (L.res >= 0) dres = (float*)desc[(size_t)L.res * IRX_ENC_NFIELDS + IRX_ENC_GY];
    const int last = (i == n_layers - 1);
    // c: st | y, gy: fp32 for the last layer | dc scratch and the shortcut gradient (never the last layer's): st
    // a layer without a shortcut hands over beta: its ReLU mask is recomputed from c and y is not read (irx_norm.hip)
    static const bool remask = !(getenv("IRX_BN_REMASK") && atoi(getenv("IRX_BN_REMASK")) == 0);   // dev A/B knob
    int rc;
    const float* mk_beta = (L.res < 0 && remask) ? L.beta : nullptr;
    if (sync.cb) {
      // this rank's sums (= the parameter gradients d beta, d gamma, which the gradient all-reduce folds like every other) ->
      // folded copies through the caller's all-reduce -> the apply pass with the folded sums and the global row count
      float* gl = sync.gsums + (size_t)i * IRX_ENC_SYNC_STRIDE;
      const double* cnt = sync.sums + (size_t)i * IRX_ENC_SYNC_STRIDE + 2 * L.cout;
      rc = irx_bn_backward_t(L.c, L.y, L.gy, L.n_out, L.cout, L.mean, L.invstd, L.gamma, 1, nullptr, L.dgamma, L.dbeta, nullptr,
                             ws_b, r.bn, stream, st, (st && !last) ? 1 : 0, (st && !last) ? 1 : 0, st, st, 1);
      if (rc) return rc;
      rc = irx_bn_pack_sums(L.dbeta, L.dgamma, L.cout, gl, stream);
      if (rc) return rc;
      rc = sync.cb(sync.user, gl, 2 * L.cout, 0, stream);
      IRX_REQUIRE(rc == 0, "irx_encoder_backward_sync: the all-reduce callback failed (%d) at layer %d", rc, i);
      rc = irx_bn_backward_t(L.c, L.y, L.gy, L.n_out, L.cout, L.mean, L.invstd, L.gamma, 1, dc_scratch, nullptr, nullptr, dres,
                             nullptr, 0, stream, st, (st && !last) ? 1 : 0, (st && !last) ? 1 : 0, st, st, 2, gl, gl + L.cout,
                             0.0, cnt, mk_beta);
    } else {
      rc = irx_bn_backward_t(L.c, L.y, L.gy, L.n_out, L.cout, L.mean, L.invstd, L.gamma, 1, dc_scratch, L.dgamma,
                             L.dbeta, dres, ws_b, r.bn, stream, st, (st && !last) ? 1 : 0, (st && !last) ? 1 : 0, st, st,
                             3, nullptr, nullptr, 0.0, nullptr, mk_beta, dys);
    }
    if (rc) return rc;
    if (lent) {
      IRX_CHECK_HIP(hipEventRecord(ev[i], (hipStream_t)stream), "irx_encoder_backward(record E)");
      IRX_CHECK_HIP(hipStreamWaitEvent(sW, ev[i], 0), "irx_encoder_backward(wait E)");
    }
    const long abl = irx_knob(IRX_KNOB_ABL);         // dev, timing only: what the chain costs without a kernel family
    if (L.prof) irx_profile_next_kernel(L.prof[4], L.prof[5]);
    if (abl & 1) {
      rc = IRX_OK;
    } else if (pairs_path(L)) {
      IRX_REQUIRE(!st || i > 0, "irx_encoder_backward: bf16 storage expects a stem (Cin <= 8 or 129..136) as layer 0");
      rc = irx_spconv_wgrad_pairs_impl(L.x, dc_scratch, L.pair_in, L.pair_out, L.ld_pairs, L.pair_counts, L.n_out, L.K,
                                       L.cin, L.cout, L.dw, ws_w, r.wgrad, wstream, st, L.n_in);
    } else {
      IRX_REQUIRE(!st || i == 0, "irx_encoder_backward: layer %d (%d -> %d channels) has no bf16-storage weight-gradient path",
                  i, L.cin, L.cout);
      IrxPairLists pl;
      if (L.pair_in && irx_wide_stem(L.K, L.cin, L.cout)) {   // the multiview stem: its 128 leading channels through the pair lists
        pl.in_list = L.pair_in; pl.out_list = L.pair_out; pl.counts = L.pair_counts; pl.ldp = L.ld_pairs;
      }
      rc = irx_spconv_wgrad_impl(L.x, dc_scratch, L.tbl, L.ld, L.n_out, L.K, L.cin, L.cout, L.dw, ws_w, r.wgrad, wstream, st, pl);
    }
    if (rc) return rc;
    if (two) IRX_CHECK_HIP(hipEventRecord(ev[16 + i], (hipStream_t)wstream), "irx_encoder_backward(record W)");
    float* dx = (i > 0) ? (float*)desc[(size_t)(i - 1) * IRX_ENC_NFIELDS + IRX_ENC_GY] : dx0;
    if (dx && !(abl & 2)) {
      const int acc = (i > 0 && is_res_source[i - 1]) ? 1 : 0;
      IrxStore ty;
      ty.x = st;                                  // dc
      ty.y = st;                                  // gy of layer i - 1 (never the last layer)
      ty.order = (L.tbl_b == L.tbl) ? L.order : nullptr;   // stride-1: the forward table with flipped offsets, same tile costs
      ty.x_rows = L.n_out;
      if (L.prof) irx_profile_next_kernel(L.prof[2], L.prof[3]);
      // a 2^3 / stride-2 layer in fp32: tiled by parent rows over the FORWARD (child) table (irx_spconv2.hip, k_updgrad)
      // IRX_UPDGRAD=0 switches it off, IRX_UPDGRAD_MIN overrides the size threshold (dev A/B; the bit-identity test uses
      // irx_debug_set_knob)
      const bool updgrad = irx_knob(IRX_KNOB_UPDGRAD) != 0;
      const long updgrad_min = irx_knob(IRX_KNOB_UPDGRAD_MIN);
      // (from 40 k parents = 625 workgroups on: measured 120 -> 79 us at 259 k parents, 92 -> 58 at 489 k fine rows, 36 -> 29 at
      //  51 k, but 31 -> 38 at 29 k and 24 -> 39 at 9 k parents, where 64-parent tiles leave most CUs without a workgroup)
      if (updgrad && L.tbl_b != L.tbl && L.K == 8 && !st && !irx_conv_bf16() && !acc && wimg[i] && L.n_out >= updgrad_min &&
          irx_updgrad_supported(L.cout, L.cin))
        rc = irx_updgrad_launch(dc_scratch, wimg[i], L.tbl, L.ld, L.n_out, L.cout, L.cin, dx, (hipStream_t)stream);
      else {
        // offset-split layers: the slabs are folded by the NEXT iteration's BatchNorm statistics pass (gy of layer i - 1)
        float* slabs = nullptr;
        int nslabs = 0;
        if (!sync.cb && i > 0 && irx_knob(IRX_KNOB_FOLD_SLABS) != 0) {
          ty.slabs_out = &slabs;
          ty.splits_out = &nslabs;
        }
        rc = irx_spconv_fwd_impl(dc_scratch, L.w, L.tbl_b, L.ld_b, L.n_in, L.K, L.cout, L.cin, L.flip_b, 1, dx, acc, wimg[i],
                                 ws_c, r.conv, stream, ty);
        if (slabs) {
          pending.slabs = slabs;
          pending.S = nslabs;
          pending.acc = acc;
        }
      }
      if (rc) return rc;
    }
  }
  gate_step(gate, gate_fired, -1, stream);   // a recorder without a level of `rows` or more marks the end of its pass
  if (two)                                   // join: everything this pass wrote is complete on the caller's stream
    for (int i = 0; i < n_layers && i < 2; ++i)
      IRX_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, ev[16 + i], 0), "irx_encoder_backward(join)");
  return IRX_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Asynchronous issue (irx_encoder_submit / irx_encoder_wait): the ~50 (forward) / ~100 (backward) kernel launches of one
// encoder pass cost 0.17 / 0.21 ms of host time; a lane is a detached library thread that performs the same
// irx_encoder_forward / irx_encoder_backward call from a COPY of the descriptor table, so the caller's thread (Python,
// holding the GIL) goes on issuing the independent work of the other modules meanwhile. Jobs of a lane run in
// submission order. The caller keeps every device buffer alive and enqueues nothing that depends on the pass (nor
// records an event on its stream) before irx_encoder_wait(lane) returned.
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
struct EncJob {
  int backward, n, device;
  std::vector<int64_t> desc;
  std::vector<double> fdesc;
  float *dc, *dx0;
  void* ws;
  size_t wsb;
  void* stream;
  EncGate gate;
};
struct EncLane {
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<EncJob> q;
  std::atomic<int> queued{0};   // jobs in q (lets the worker poll without the lock)
  std::atomic<int> pending{0};  // queued + running
  int status = 0;         // first failure since the last wait
  std::string err;
  bool started = false;
};
constexpr int kLanes = 4;
EncLane* lane_of(int i) {
  static EncLane* lanes = new EncLane[kLanes];   // never destroyed: the detached workers may outlive static destructors
  return &lanes[i];
}
void lane_main(EncLane* L) {
  for (;;) {
    EncJob j;
    {
      // Poll briefly before sleeping (IRX_LANE_POLL_US, default 200): back-to-back submissions skip the futex wake-up.
      // Longer polling was measured equal (the caller only waits for the lane ~1 ms after submitting) and would burn
      // CPU quota in multi-rank runs.
      static const long poll_us = getenv("IRX_LANE_POLL_US") ? atol(getenv("IRX_LANE_POLL_US")) : 200;
      const auto t0 = std::chrono::steady_clock::now();
      while (L->queued.load(std::memory_order_acquire) == 0 &&
             std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(poll_us))
        irx_cpu_relax();
      std::unique_lock<std::mutex> lk(L->mu);
      L->cv_job.wait(lk, [&] { return !L->q.empty(); });
      j = std::move(L->q.front());
      L->q.pop_front();
      L->queued.fetch_sub(1, std::memory_order_release);
    }
    int rc = (int)hipSetDevice(j.device);
    g_gate_next = j.gate;
    if (rc == 0)
      rc = j.backward ? irx_encoder_backward(j.desc.data(), j.fdesc.data(), j.n, j.dc, j.dx0, j.ws, j.wsb, j.stream)
                      : irx_encoder_forward(j.desc.data(), j.fdesc.data(), j.n, j.ws, j.wsb, j.stream);
    {
      std::lock_guard<std::mutex> lk(L->mu);
      if (rc != 0 && L->status == 0) {
        L->status = rc;
        L->err = irx_last_error();
      }
      L->pending.fetch_sub(1, std::memory_order_release);
    }
    L->cv_done.notify_all();
  }
}
}  // namespace

extern "C" int irx_encoder_submit(int lane, int backward, const int64_t* desc, const double* fdesc, int n_layers,
                                  float* dc_scratch, float* dx0, void* workspace, size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(lane >= 0 && lane < kLanes, "irx_encoder_submit: lane %d outside [0, %d)", lane, kLanes);
  IRX_REQUIRE(desc && fdesc && n_layers > 0 && n_layers <= 16, "irx_encoder_submit: bad descriptor table");
  EncJob j;
  j.backward = backward != 0;
  j.n = n_layers;
  IRX_CHECK_HIP(hipGetDevice(&j.device), "irx_encoder_submit(hipGetDevice)");
  j.desc.assign(desc, desc + (size_t)n_layers * IRX_ENC_NFIELDS);
  j.fdesc.assign(fdesc, fdesc + (size_t)n_layers * 2);
  j.dc = dc_scratch;
  j.dx0 = dx0;
  j.ws = workspace;
  j.wsb = workspace_bytes;
  j.stream = stream;
  j.gate = backward ? g_gate_next : EncGate();
  if (backward) g_gate_next = EncGate();
  EncLane* L = lane_of(lane);
  {
    std::lock_guard<std::mutex> lk(L->mu);
    if (!L->started) {
      std::thread(lane_main, L).detach();
      L->started = true;
    }
    L->q.push_back(std::move(j));
    L->pending.fetch_add(1, std::memory_order_relaxed);
    L->queued.fetch_add(1, std::memory_order_release);
  }
  L->cv_job.notify_one();
  return IRX_OK;
}

// `to` continues behind everything enqueued on `from` so far (an event from a small per-thread ring is recorded on `from`
// and waited for on `to`): lets a caller that issues one operator's independent halves on two streams order them without owning
// HIP events (csrc/heads_nodes.cpp: the scene head's weight gradients beside its data-gradient chain).
extern "C" int irx_stream_fork(void* from, void* to) {
  IRX_REQUIRE(from != to, "irx_stream_fork: the two streams are the same");
  static thread_local hipEvent_t ring[16] = {nullptr};
  static thread_local unsigned next = 0;
  hipEvent_t& e = ring[next++ & 15];
  if (!e) IRX_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), "irx_stream_fork(event)");
  IRX_CHECK_HIP(hipEventRecord(e, (hipStream_t)from), "irx_stream_fork(record)");
  IRX_CHECK_HIP(hipStreamWaitEvent((hipStream_t)to, e, 0), "irx_stream_fork(wait)");
  return IRX_OK;
}

extern "C" int irx_encoder_wait(int lane) {
  IRX_REQUIRE(lane >= 0 && lane < kLanes, "irx_encoder_wait: lane %d outside [0, %d)", lane, kLanes);
  EncLane* L = lane_of(lane);
  {
    const auto t0 = std::chrono::steady_clock::now();   // the pass is usually done or nearly done: poll briefly first
    while (L->pending.load(std::memory_order_acquire) != 0 &&
           std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2))
      irx_cpu_relax();
  }
  std::unique_lock<std::mutex> lk(L->mu);
  L->cv_done.wait(lk, [&] { return L->pending.load(std::memory_order_acquire) == 0; });
  const int rc = L->status;
  if (rc != 0) {
    irx_set_error("%s", L->err.c_str());
    L->status = 0;
    L->err.clear();
  }
  return rc;
}
