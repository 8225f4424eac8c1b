// One residual block of the backbone as ONE host call each way (include/prn.h: prn_bottleneck_*; models/backbone.py:53-73, models/dcn.py:52-67).
//
// The reference calls Bottleneck.forward 33 times per image batch (PlaneRecNet_101); operator by operator that is 6-9 host calls forward and as many
// backward, each through the host framework's graph node, allocator and FFI.  Here the launch SEQUENCE of a block is issued from C: the caller hands in
// the block's input, its parameters (and the derived operand layouts it keeps: split-kernel images, input-gradient layouts, Winograd-domain weights),
// ONE buffer for everything the backward pass reads again (`save`), one for the gradients the weight-gradient launches read later (`gsave`) and a
// scratch workspace.  What used to be a protocol between host call sites -- a K-split GEMM leaving its partial sums to the BatchNorm kernel behind it,
// a Winograd convolution leaving its output transform to it, a BatchNorm kernel writing the input transform for the 3x3 convolution behind it
//  -- is internal state of this call: pointers into `ws` that never leave it.
//
// Nothing here computes: every launch is one of the library's own operators (prn_conv2d_fwd_counted, prn_bn_*, prn_gemm_batched, prn_winograd_*,
// prn_dcnv2_*), called with the arguments the operator-by-operator path passes, so the results are those of that path bit for bit
// (tests/test_blocks_gpu.py).  The weight gradients are NOT part of the block: nobody inside the backward pass reads them, the caller launches them
// (grouped, on its side stream) from the operands this call leaves in `save` / `gsave`.
#include "prn_common.h"
#include <string.h>

namespace {

constexpr int32_t PLAN_MAGIC = 0x424c4b36;                  // "BLK6"
constexpr int64_t WINOGRAD_MIN_TILES = 128;                 // the operator layer's rule (planerecnet_amd.ops.winograd_ok)

inline int64_t up256(int64_t v) { return (v + 255) & ~255LL; }

// Everything a call needs that depends on the descriptor only: sub-descriptors, which hand-overs apply, where things live in save / gsave / ws.
struct Plan {
  int32_t magic, pad_;
  prn_bottleneck_desc d;
  int32_t Ho, Wo, P4, conv2_path;
  int32_t small1, small2;        // one-launch BatchNorm kernels at H x W / at Ho x Wo
  int32_t lazy1, parts1;         // conv1's K-split sums are summed by bn1's kernel (parts1 partial sums, parts1_off bytes into conv1's workspace)
  int32_t v1, lazy2, keep_v;     // bn1 writes conv2's Winograd input transform; bn2's kernel applies conv2's output transform; V stays in `save`
  int32_t lazy3g, parts3g;       // backward: conv3's input gradient leaves its K-split sums to bn2's backward kernel
  int32_t v2g, lazy2g;           // bn2's backward writes the input transform of conv2's input-gradient convolution; bn1's backward applies its output transform
  int32_t pad2_;
  int64_t parts1_off, parts3g_off;
  prn_conv_desc c1, c2, c3, cd, c27, g1, g2, g3, gd, g27;
  prn_dcn_desc dcn;
  // save (forward -> backward)
  int64_t s_c1, s_a1, s_v, s_c2, s_a2, s_c3, s_cd, s_om, s_table, s_stats[4], save_bytes;
  // gsave (backward -> weight gradients); q_bn: floats of bn_grads
  int64_t q_d3, q_dd, q_d2, q_d1, q_dom, q_bn, gsave_bytes;
  // forward scratch
  int64_t f_c1, f_bn, f_v, f_yt, f_gemm, f_c2, f_c27, f_dcn, f_c3, f_cd, f_res, fwd_ws_bytes;
  // backward scratch
  int64_t b_bn, b_dres, b_g3, b_g3ws, b_v, b_yt, b_gemm, b_g2, b_g2ws, b_dcn, b_dx1, b_g27ws, b_g1ws, b_gdtmp, b_gdws, bwd_ws_bytes;
};

prn_conv_desc conv_desc(int B, int C, int H, int W, int M, int K, int stride, int pad, int Ho, int Wo, int mode, int dil, int ystride, int yH, int yW,
                        const prn_gemm_opts& o) {
  prn_conv_desc d;
  d.B = B; d.C = C; d.H = H; d.W = W; d.M = M; d.KH = d.KW = K; d.stride = stride; d.pad = pad; d.Ho = Ho; d.Wo = Wo;
  d.in_mode = mode; d.dil = dil; d.epilogue = PRN_EPI_NONE; d.ystride = ystride; d.yH = yH; d.yW = yW; d.reserved = 0;
  d.opts = o;
  return d;
}

bool winograd_ok(const prn_bottleneck_desc& d, int B, int C, int H, int W, int M) {      // 3x3 / stride 1 / pad 1 / zero padding / no epilogue
  return (d.flags & PRN_BLK_WINOGRAD) && (W % 4) == 0 && H >= 8 && C >= 64 && M >= 64 && (int64_t)B * ((H + 3) / 4) * (W / 4) >= WINOGRAD_MIN_TILES;
}

int make_plan(const prn_bottleneck_desc* dp, Plan& L) {
  PRN_REQUIRE(dp != nullptr, "prn_bottleneck_plan: null descriptor");
  const prn_bottleneck_desc& d = *dp;
  PRN_REQUIRE(d.B > 0 && d.C > 0 && d.H > 0 && d.W > 0 && d.planes > 0, "prn_bottleneck_plan: empty dimension");
  PRN_REQUIRE(d.stride == 1 || d.stride == 2, "prn_bottleneck_plan: stride %d (1 or 2)", d.stride);
  PRN_REQUIRE(d.downsample || (d.stride == 1 && d.C == 4 * d.planes), "prn_bottleneck_plan: without a downsample branch the block must keep its shape (C = 4 * planes, stride 1)");
  PRN_REQUIRE(!d.dcn || d.max_offset > 0.f, "prn_bottleneck_plan: the deformable variant needs max_offset (max(h, w) / 4)");
  memset(&L, 0, sizeof(L));
  L.magic = PLAN_MAGIC;
  L.d = d;
  const int B = d.B, C = d.C, H = d.H, W = d.W, P = d.planes, s = d.stride, Q = 4 * d.planes;
  const int Ho = (H + 2 - 3) / s + 1, Wo = (W + 2 - 3) / s + 1;
  L.Ho = Ho; L.Wo = Wo;
  const prn_gemm_opts& o = d.opts;
  const bool ho = (d.flags & PRN_BLK_HANDOVER) != 0;
  L.small1 = prn_bn_kernel_kind(B, H * W);
  L.small2 = prn_bn_kernel_kind(B, Ho * Wo);
  // ---- forward descriptors
  L.c1 = conv_desc(B, C, H, W, P, 1, 1, 0, H, W, PRN_IN_ZERO, 1, 0, 0, 0, o);
  L.c3 = conv_desc(B, P, Ho, Wo, Q, 1, 1, 0, Ho, Wo, PRN_IN_ZERO, 1, 0, 0, 0, o);
  if (d.downsample) L.cd = conv_desc(B, C, H, W, Q, 1, s, 0, Ho, Wo, PRN_IN_ZERO, 1, 0, 0, 0, o);
  const bool wino = !d.dcn && s == 1 && winograd_ok(d, B, P, H, W, P);
  L.conv2_path = d.dcn ? PRN_BLK_CONV2_DCN : (wino ? PRN_BLK_CONV2_WINOGRAD : PRN_BLK_CONV2_DIRECT);
  if (d.dcn) {
    L.c27 = conv_desc(B, P, H, W, 27, 3, s, 1, Ho, Wo, PRN_IN_ZERO, 1, 0, 0, 0, o);
    prn_dcn_desc& q = L.dcn;
    q.B = B; q.C = P; q.H = H; q.W = W; q.M = P; q.stride = s; q.pad = 1; q.Ho = Ho; q.Wo = Wo; q.raw = 1; q.max_offset = d.max_offset; q.epilogue = PRN_EPI_NONE;
    q.opts = o;
  } else if (!wino) {
    L.c2 = conv_desc(B, P, H, W, P, 3, s, 1, Ho, Wo, PRN_IN_ZERO, 1, 0, 0, 0, o);
  }
  if (wino) L.P4 = (int)prn_winograd_tiles(B, H, W);
  // ---- input-gradient descriptors: the same GEMM over dy with the flipped / transposed weights (planerecnet_amd.ops.conv_dgrad_raw)
  L.g3 = conv_desc(B, Q, Ho, Wo, P, 1, 1, 0, Ho, Wo, PRN_IN_ZERO, 1, 0, 0, 0, o);
  L.g1 = conv_desc(B, P, H, W, C, 1, 1, 0, H, W, PRN_IN_ZERO, 1, 0, 0, 0, o);
  if (!d.dcn && !wino)
    L.g2 = s == 1 ? conv_desc(B, P, Ho, Wo, P, 3, 1, 1, H, W, PRN_IN_ZERO, 1, 0, 0, 0, o) : conv_desc(B, P, Ho, Wo, P, 3, 1, 1, H, W, PRN_IN_DILATED, 2, 0, 0, 0, o);
  if (d.dcn)
    L.g27 = s == 1 ? conv_desc(B, 27, Ho, Wo, P, 3, 1, 1, H, W, PRN_IN_ZERO, 1, 0, 0, 0, o) : conv_desc(B, 27, Ho, Wo, P, 3, 1, 1, H, W, PRN_IN_DILATED, 2, 0, 0, 0, o);
  if (d.downsample)      // stride 2: the GEMM runs over dy's own pixel grid and scatters to the even positions of dx (four times fewer multiply-adds)
    L.gd = s == 1 ? conv_desc(B, Q, Ho, Wo, C, 1, 1, 0, H, W, PRN_IN_ZERO, 1, 0, 0, 0, o) : conv_desc(B, Q, Ho, Wo, C, 1, 1, 0, Ho, Wo, PRN_IN_ZERO, 1, 2, H, W, o);
  // ---- hand-overs (all of them need the one-launch BatchNorm kernels: small maps)
  int64_t off = 0;
  const int p1 = prn_conv2d_fwd_partials(&L.c1, &off);
  if (p1 < 0) return 2;
  L.lazy1 = ho && L.small1 && p1 > 1 && (((int64_t)B * P * H * W) & 3) == 0;
  L.parts1 = L.lazy1 ? p1 : 0; L.parts1_off = L.lazy1 ? off : 0;
  L.v1 = ho && wino && L.small1;
  L.lazy2 = ho && wino && L.small2 && L.P4 <= 768;
  L.keep_v = wino && (d.flags & PRN_BLK_KEEP_V) && 4LL * 36 * P * L.P4 <= (128LL << 20);
  const int p3 = prn_conv2d_fwd_partials(&L.g3, &off);
  if (p3 < 0) return 2;
  L.lazy3g = ho && L.small2 && p3 > 1 && (((int64_t)B * P * Ho * Wo) & 3) == 0;
  L.parts3g = L.lazy3g ? p3 : 0; L.parts3g_off = L.lazy3g ? off : 0;
  L.v2g = ho && wino && L.small2;
  L.lazy2g = ho && wino && L.small1 && L.P4 <= 768;
  // ---- sizes
  const int64_t n1 = (int64_t)B * P * H * W * 4, n2 = (int64_t)B * P * Ho * Wo * 4, n3 = (int64_t)B * Q * Ho * Wo * 4, nx = (int64_t)B * C * H * W * 4;
  const int64_t nv = wino ? 4LL * 36 * P * L.P4 : 0, nom = (int64_t)B * 27 * Ho * Wo * 4;
  const int Cmax = Q > C ? Q : C;
  const int64_t bnws = 2LL * Cmax * PRN_BN_SPLITS * 8;
#define PRN_WS(expr_, name_) const int64_t name_ = (expr_); if (name_ < 0) return 2
  PRN_WS(prn_conv2d_fwd_ws_bytes(&L.c1), w_c1);
  PRN_WS(prn_conv2d_fwd_ws_bytes(&L.c3), w_c3);
  PRN_WS(d.downsample ? prn_conv2d_fwd_ws_bytes(&L.cd) : 0, w_cd);
  PRN_WS(L.conv2_path == PRN_BLK_CONV2_DIRECT ? prn_conv2d_fwd_ws_bytes(&L.c2) : 0, w_c2);
  PRN_WS(d.dcn ? prn_conv2d_fwd_ws_bytes(&L.c27) : 0, w_c27);
  PRN_WS(d.dcn ? prn_dcnv2_fwd_ws_bytes(&L.dcn) : 0, w_dcnf);
  PRN_WS(d.dcn ? prn_dcnv2_table_bytes(&L.dcn) : 0, n_table);
  PRN_WS(d.dcn ? prn_dcnv2_bwd_ws_bytes(&L.dcn) : 0, w_dcnb);
  PRN_WS(wino ? prn_gemm_batched_ws_bytes(P, P, L.P4, 36, &o) : 0, w_gemm);
  PRN_WS(prn_conv2d_fwd_ws_bytes(&L.g3), w_g3);
  PRN_WS(prn_conv2d_fwd_ws_bytes(&L.g1), w_g1);
  PRN_WS(L.conv2_path == PRN_BLK_CONV2_DIRECT ? prn_conv2d_fwd_ws_bytes(&L.g2) : 0, w_g2);
  PRN_WS(d.dcn ? prn_conv2d_fwd_ws_bytes(&L.g27) : 0, w_g27);
  PRN_WS(d.downsample ? prn_conv2d_fwd_ws_bytes(&L.gd) : 0, w_gd);
#undef PRN_WS
  int64_t at = 0;
  auto put = [&at](int64_t& slot, int64_t bytes) { slot = at; at += up256(bytes); };
  // save
  put(L.s_c1, n1); put(L.s_a1, n1);
  put(L.s_v, L.keep_v ? nv : 0);
  put(L.s_c2, n2); put(L.s_a2, n2); put(L.s_c3, n3);
  put(L.s_cd, d.downsample ? n3 : 0);
  put(L.s_om, d.dcn ? nom : 0); put(L.s_table, n_table);
  put(L.s_stats[0], 8LL * P); put(L.s_stats[1], 8LL * P); put(L.s_stats[2], 8LL * Q); put(L.s_stats[3], d.downsample ? 8LL * Q : 0);
  L.save_bytes = at;
  // gsave
  at = 0;
  put(L.q_d3, n3); put(L.q_dd, d.downsample ? n3 : 0); put(L.q_d2, n2); put(L.q_d1, n1); put(L.q_dom, d.dcn ? nom : 0);
  L.q_bn = 2 * P + 2 * P + 2 * Q + (d.downsample ? 2 * Q : 0);      // floats of the caller's bn_grads: dgamma1 | dbeta1 | dgamma2 | dbeta2 | dgamma3 | dbeta3 [| dgamma_d | dbeta_d]
  L.gsave_bytes = at;
  // forward scratch (regions do not overlap: the stream orders the launches, nothing is read after the call returns)
  at = 0;
  put(L.f_c1, w_c1); put(L.f_bn, bnws);
  put(L.f_v, (wino && !L.keep_v) ? nv : 0); put(L.f_yt, nv); put(L.f_gemm, w_gemm);
  put(L.f_c2, w_c2); put(L.f_c27, w_c27); put(L.f_dcn, w_dcnf); put(L.f_c3, w_c3); put(L.f_cd, w_cd);
  put(L.f_res, d.downsample ? n3 : 0);
  L.fwd_ws_bytes = at;
  // backward scratch
  at = 0;
  put(L.b_bn, bnws); put(L.b_dres, n3);
  put(L.b_g3, n2); put(L.b_g3ws, w_g3);
  put(L.b_v, nv); put(L.b_yt, nv); put(L.b_gemm, w_gemm);
  put(L.b_g2, n1); put(L.b_g2ws, w_g2);
  put(L.b_dcn, w_dcnb); put(L.b_dx1, d.dcn ? n1 : 0); put(L.b_g27ws, w_g27);
  put(L.b_g1ws, w_g1);
  put(L.b_gdtmp, d.downsample ? nx : 0); put(L.b_gdws, w_gd);
  L.bwd_ws_bytes = at;
  return 0;
}

const Plan* plan_of(const void* plan, const char* who) {
  const Plan* L = (const Plan*)plan;
  if (L == nullptr || L->magic != PLAN_MAGIC) { prn_set_error("%s: not a plan filled by prn_bottleneck_plan", who); return nullptr; }
  return L;
}

inline float* f32(void* base, int64_t off) { return (float*)((char*)base + off); }
inline void* raw(void* base, int64_t off, int64_t used = 1) { return used ? (void*)((char*)base + off) : nullptr; }

#define PRN_TRY(call_) do { if (int e_ = (call_)) return e_; } while (0)

}  // namespace

extern "C" int64_t prn_bottleneck_plan_bytes(void) { return (int64_t)sizeof(Plan); }
extern "C" int64_t prn_bottleneck_params_bytes(void) { return (int64_t)sizeof(prn_bottleneck_params); }

extern "C" int prn_bottleneck_plan(const prn_bottleneck_desc* d, void* plan) {
  PRN_REQUIRE(plan != nullptr, "prn_bottleneck_plan: null plan buffer (prn_bottleneck_plan_bytes() bytes, caller-owned)");
  Plan L;
  if (int e = make_plan(d, L)) { memset(plan, 0, sizeof(int32_t)); return e; }
  memcpy(plan, &L, sizeof(L));
  return 0;
}

extern "C" int prn_bottleneck_plan_info(const void* plan, int64_t* out, int n) {
  const Plan* L = plan_of(plan, "prn_bottleneck_plan_info");
  if (!L) return 2;
  PRN_REQUIRE(out != nullptr && n >= PRN_BLK_INFO_COUNT, "prn_bottleneck_plan_info: out must hold PRN_BLK_INFO_COUNT (%d) values", PRN_BLK_INFO_COUNT);
  out[PRN_BLK_SAVE_BYTES] = L->save_bytes; out[PRN_BLK_GSAVE_BYTES] = L->gsave_bytes;
  out[PRN_BLK_FWD_WS_BYTES] = L->fwd_ws_bytes; out[PRN_BLK_BWD_WS_BYTES] = L->bwd_ws_bytes;
  out[PRN_BLK_HO] = L->Ho; out[PRN_BLK_WO] = L->Wo; out[PRN_BLK_CONV2_PATH] = L->conv2_path; out[PRN_BLK_KEEPS_V] = L->keep_v;
  out[PRN_BLK_OFF_A1] = L->s_a1; out[PRN_BLK_OFF_V] = L->s_v; out[PRN_BLK_OFF_A2] = L->s_a2; out[PRN_BLK_OFF_OM] = L->s_om; out[PRN_BLK_OFF_TABLE] = L->s_table;
  out[PRN_BLK_OFF_D1] = L->q_d1; out[PRN_BLK_OFF_D2] = L->q_d2; out[PRN_BLK_OFF_D3] = L->q_d3; out[PRN_BLK_OFF_DD] = L->q_dd; out[PRN_BLK_OFF_DOM] = L->q_dom;
  out[PRN_BLK_BN_GRAD_FLOATS] = L->q_bn;
  out[PRN_BLK_HANDOVERS] = L->lazy1 | (L->v1 << 1) | (L->lazy2 << 2) | (L->lazy3g << 3) | (L->v2g << 4) | (L->lazy2g << 5);
  out[PRN_BLK_NEEDS_W1_IMG] = prn_conv2d_kernel_kind(&L->c1) >= 2; out[PRN_BLK_NEEDS_W3_IMG] = prn_conv2d_kernel_kind(&L->c3) >= 2;
  out[PRN_BLK_NEEDS_WD_IMG] = (L->d.downsample && L->d.stride == 1 && prn_conv2d_kernel_kind(&L->cd) >= 2) ? 1 : 0;
  out[PRN_BLK_NEEDS_W1T_IMG] = prn_conv2d_kernel_kind(&L->g1) >= 2; out[PRN_BLK_NEEDS_W3T_IMG] = prn_conv2d_kernel_kind(&L->g3) >= 2;
  out[PRN_BLK_NEEDS_WDT_IMG] = (L->d.downsample && L->d.stride == 1 && prn_conv2d_kernel_kind(&L->gd) >= 2) ? 1 : 0;
  out[PRN_BLK_NEEDS_U_IMG] = (L->conv2_path == PRN_BLK_CONV2_WINOGRAD && prn_gemm_pipe(L->d.planes, L->d.planes, 1, L->P4, 36, &L->d.opts) >= 1) ? 1 : 0;
  out[PRN_BLK_NEEDS_COLT_IMG] = (L->d.dcn && prn_gemm_pipe(9 * L->d.planes, L->d.planes, L->d.B, L->Ho * L->Wo, 1, &L->d.opts) >= 1) ? 1 : 0;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------- forward
extern "C" int prn_bottleneck_train_fwd(const void* plan, const prn_bottleneck_params* p, const float* x, float* y, void* save, void* ws, void* stream) {
  const Plan* Lp = plan_of(plan, "prn_bottleneck_train_fwd");
  if (!Lp) return 2;
  const Plan& L = *Lp;
  const prn_bottleneck_desc& d = L.d;
  PRN_REQUIRE(p && x && y && save && (ws || L.fwd_ws_bytes == 0), "prn_bottleneck_train_fwd: null tensor / buffer");
  PRN_REQUIRE(p->w1 && p->w3 && (p->w2 || L.conv2_path == PRN_BLK_CONV2_WINOGRAD) && (p->u2 || L.conv2_path != PRN_BLK_CONV2_WINOGRAD) && (!d.downsample || p->wd),
              "prn_bottleneck_train_fwd: a weight operand the plan reads is null");
  PRN_REQUIRE(!d.dcn || (p->w27 && p->b27), "prn_bottleneck_train_fwd: the deformable variant needs the merged offset | modulator weights");
  for (int i = 0; i < (d.downsample ? 4 : 3); ++i)
    PRN_REQUIRE(p->gamma[i] && p->beta[i] && p->running_mean[i] && p->running_var[i], "prn_bottleneck_train_fwd: null BatchNorm parameter (layer %d)", i);
  PRN_REQUIRE(((reinterpret_cast<uintptr_t>(save) | reinterpret_cast<uintptr_t>(ws)) & 255) == 0, "prn_bottleneck_train_fwd: save / ws must be 256-byte aligned");
  const int B = d.B, H = d.H, W = d.W, P = d.planes, Q = 4 * P, Ho = L.Ho, Wo = L.Wo;
  float* c1 = f32(save, L.s_c1); float* a1 = f32(save, L.s_a1); float* c2 = f32(save, L.s_c2); float* a2 = f32(save, L.s_a2); float* c3 = f32(save, L.s_c3);
  float* st1 = f32(save, L.s_stats[0]); float* st2 = f32(save, L.s_stats[1]); float* st3 = f32(save, L.s_stats[2]); float* std_ = f32(save, L.s_stats[3]);
  double* bnws = (double*)raw(ws, L.f_bn);
  const int64_t n1 = (int64_t)B * P * H * W;
  const prn_gemm_opts* o = &d.opts;
  // conv1 (-> K-split partial sums when bn1's kernel will sum them)
  PRN_TRY(prn_conv2d_fwd_counted(&L.c1, x, p->w1, p->w1_img, nullptr, nullptr, c1, raw(ws, L.f_c1, L.f_bn - L.f_c1), nullptr, stream, L.lazy1 ? 1 : 0));
  // bn1 + ReLU (+ sum of the partials, + conv2's Winograd input transform)
  float* V = L.conv2_path == PRN_BLK_CONV2_WINOGRAD ? (L.keep_v ? f32(save, L.s_v) : f32(ws, L.f_v)) : nullptr;
  const float* in1 = L.lazy1 ? f32(ws, L.f_c1 + L.parts1_off) : c1;
  if (L.v1)
    PRN_TRY(prn_bn_train_fwd_winograd(in1, L.lazy1 ? L.parts1 : 1, L.lazy1 ? n1 : 0, L.lazy1 ? c1 : nullptr, st1, p->gamma[0], p->beta[0], nullptr, a1, p->running_mean[0],
                                      p->running_var[0], V, B, P, H, W, d.eps[0], d.momentum[0], 1, stream));
  else if (L.lazy1)
    PRN_TRY(prn_bn_train_fwd_partials(in1, L.parts1, n1, c1, st1, p->gamma[0], p->beta[0], nullptr, a1, p->running_mean[0], p->running_var[0], B, P, H * W, d.eps[0],
                                      d.momentum[0], 1, stream));
  else
    PRN_TRY(prn_bn_train_fwd(c1, st1, p->gamma[0], p->beta[0], nullptr, a1, p->running_mean[0], p->running_var[0], L.small1 ? nullptr : bnws, B, P, H * W, d.eps[0],
                             d.momentum[0], 1, stream));
  // conv2
  float* Yt = f32(ws, L.f_yt);
  if (L.conv2_path == PRN_BLK_CONV2_WINOGRAD) {
    if (!L.v1) PRN_TRY(prn_winograd_input(a1, V, B, P, H, W, PRN_IN_ZERO, stream));
    PRN_TRY(prn_gemm_batched(P, P, L.P4, 36, p->u2, p->u2_img, V, Yt, raw(ws, L.f_gemm, L.f_c2 - L.f_gemm), o, stream));
    if (!L.lazy2) PRN_TRY(prn_winograd_output(Yt, nullptr, nullptr, c2, B, P, H, W, PRN_EPI_NONE, stream));
  } else if (L.conv2_path == PRN_BLK_CONV2_DCN) {
    float* om = f32(save, L.s_om);
    void* table = raw(save, L.s_table);
    PRN_TRY(prn_conv2d_fwd_counted(&L.c27, a1, p->w27, nullptr, p->b27, nullptr, om, raw(ws, L.f_c27, L.f_dcn - L.f_c27), nullptr, stream, 0));
    PRN_TRY(prn_dcnv2_table(&L.dcn, om, nullptr, table, stream));
    PRN_TRY(prn_dcnv2_fwd(&L.dcn, a1, table, p->w2, p->b2, c2, raw(ws, L.f_dcn, L.f_c3 - L.f_dcn), stream));
  } else {
    PRN_TRY(prn_conv2d_fwd_counted(&L.c2, a1, p->w2, nullptr, nullptr, nullptr, c2, raw(ws, L.f_c2, L.f_c27 - L.f_c2), nullptr, stream, 0));
  }
  // bn2 + ReLU (+ conv2's Winograd output transform)
  if (L.conv2_path == PRN_BLK_CONV2_WINOGRAD && L.lazy2)
    PRN_TRY(prn_winograd_output_bn_fwd(Yt, c2, st2, p->gamma[1], p->beta[1], a2, p->running_mean[1], p->running_var[1], B, P, Ho, Wo, d.eps[1], d.momentum[1], 1, stream));
  else
    PRN_TRY(prn_bn_train_fwd(c2, st2, p->gamma[1], p->beta[1], nullptr, a2, p->running_mean[1], p->running_var[1], L.small2 ? nullptr : bnws, B, P, Ho * Wo, d.eps[1],
                             d.momentum[1], 1, stream));
  // conv3
  PRN_TRY(prn_conv2d_fwd_counted(&L.c3, a2, p->w3, p->w3_img, nullptr, nullptr, c3, raw(ws, L.f_c3, L.f_cd - L.f_c3), nullptr, stream, 0));
  // identity path
  const float* res = x;
  if (d.downsample) {
    float* cd = f32(save, L.s_cd);
    float* rd = f32(ws, L.f_res);
    PRN_TRY(prn_conv2d_fwd_counted(&L.cd, x, p->wd, d.stride == 1 ? p->wd_img : nullptr, nullptr, nullptr, cd, raw(ws, L.f_cd, L.f_res - L.f_cd), nullptr, stream, 0));
    PRN_TRY(prn_bn_train_fwd(cd, std_, p->gamma[3], p->beta[3], nullptr, rd, p->running_mean[3], p->running_var[3], L.small2 ? nullptr : bnws, B, Q, Ho * Wo, d.eps[3],
                             d.momentum[3], 0, stream));
    res = rd;
  }
  // bn3 + residual + ReLU
  return prn_bn_train_fwd(c3, st3, p->gamma[2], p->beta[2], res, y, p->running_mean[2], p->running_var[2], L.small2 ? nullptr : bnws, B, Q, Ho * Wo, d.eps[2], d.momentum[2],
                          1, stream);
}

// ---------------------------------------------------------------------------------------------------------------- backward
// dy: gradient of the block's output.  dx: gradient of its input, fully written -- except with dx_accumulate != 0 (stride-2 downsample blocks, PRN_BLK_SCATTER_ACC):
// dx then ALREADY holds another gradient of the block's input (what its other readers sent back) and is added to in place.
extern "C" int prn_bottleneck_train_bwd(const void* plan, const prn_bottleneck_params* p, const float* x, const float* y, const float* dy, float* dx,
                                        int dx_accumulate, const void* save_, void* gsave, float* bn_grads, void* ws, void* stream) {
  const Plan* Lp = plan_of(plan, "prn_bottleneck_train_bwd");
  if (!Lp) return 2;
  const Plan& L = *Lp;
  const prn_bottleneck_desc& d = L.d;
  PRN_REQUIRE(p && x && y && dy && dx && save_ && gsave && bn_grads && ws, "prn_bottleneck_train_bwd: null tensor / buffer");
  PRN_REQUIRE(p->w1_t && p->w3_t && (!d.downsample || p->wd_t), "prn_bottleneck_train_bwd: an input-gradient weight layout the plan reads is null");
  PRN_REQUIRE(L.conv2_path != PRN_BLK_CONV2_WINOGRAD || p->ut2, "prn_bottleneck_train_bwd: conv2's Winograd-domain input-gradient operand is null");
  PRN_REQUIRE(L.conv2_path != PRN_BLK_CONV2_DIRECT || p->w2_t, "prn_bottleneck_train_bwd: conv2's input-gradient layout is null");
  PRN_REQUIRE(!d.dcn || (p->w27_t && p->w2_cols_t), "prn_bottleneck_train_bwd: the deformable variant needs the input-gradient layouts of both convolutions");
  PRN_REQUIRE(!dx_accumulate || (d.downsample && d.stride == 2 && (d.flags & PRN_BLK_SCATTER_ACC)),
              "prn_bottleneck_train_bwd: dx_accumulate only for stride-2 downsample blocks planned with PRN_BLK_SCATTER_ACC");
  PRN_REQUIRE(((reinterpret_cast<uintptr_t>(save_) | reinterpret_cast<uintptr_t>(gsave) | reinterpret_cast<uintptr_t>(ws)) & 255) == 0,
              "prn_bottleneck_train_bwd: save / gsave / ws must be 256-byte aligned");
  void* save = const_cast<void*>(save_);
  const int B = d.B, C = d.C, H = d.H, W = d.W, P = d.planes, Q = 4 * P, Ho = L.Ho, Wo = L.Wo;
  const float* c1 = f32(save, L.s_c1); const float* a1 = f32(save, L.s_a1); const float* c2 = f32(save, L.s_c2); const float* c3 = f32(save, L.s_c3);
  const float* st1 = f32(save, L.s_stats[0]); const float* st2 = f32(save, L.s_stats[1]); const float* st3 = f32(save, L.s_stats[2]); const float* std_ = f32(save, L.s_stats[3]);
  float* d3 = f32(gsave, L.q_d3); float* d2 = f32(gsave, L.q_d2); float* d1 = f32(gsave, L.q_d1);
  float* dg1 = bn_grads; float* db1 = dg1 + P; float* dg2 = db1 + P; float* db2 = dg2 + P; float* dg3 = db2 + P; float* db3 = dg3 + Q; float* dgd = db3 + Q; float* dbd = dgd + Q;
  double* bnws = (double*)raw(ws, L.b_bn);
  float* dres = f32(ws, L.b_dres);
  const prn_gemm_opts* o = &d.opts;
  hipStream_t st = (hipStream_t)stream;
  // bn3 (+ residual + ReLU): d3 = gradient of conv3's result, dres = gradient of the identity path
  PRN_TRY(prn_bn_bwd(dy, c3, y, st3, p->gamma[2], p->beta[2], d3, dres, dg3, db3, L.small2 ? nullptr : bnws, B, Q, Ho * Wo, 1, 0, stream));
  // identity path with a downsample branch: BatchNorm, then the 1x1 convolution's input gradient -> `addx`, the addend of conv1's input-gradient epilogue
  const float* addx = dres;
  if (d.downsample) {
    float* dd = f32(gsave, L.q_dd);
    PRN_TRY(prn_bn_bwd(dres, f32(save, L.s_cd), nullptr, std_, p->gamma[3], p->beta[3], dd, nullptr, dgd, dbd, L.small2 ? nullptr : bnws, B, Q, Ho * Wo, 0, 0, stream));
    void* gws = raw(ws, L.b_gdws, L.bwd_ws_bytes - L.b_gdws);
    if (d.stride == 1) {
      float* tmp = f32(ws, L.b_gdtmp);
      PRN_TRY(prn_conv2d_fwd_counted(&L.gd, dd, p->wd_t, p->wd_t_img, nullptr, nullptr, tmp, gws, nullptr, stream, 0));
      addx = tmp;
    } else if (dx_accumulate) {                              // added INTO the gradient the block's other readers sent (the strided epilogue reads its addend at the output's index)
      PRN_TRY(prn_conv2d_fwd_counted(&L.gd, dd, p->wd_t, nullptr, nullptr, dx, dx, gws, nullptr, stream, 0));
      addx = dx;
    } else {
      float* tmp = f32(ws, L.b_gdtmp);
      const hipError_t e = hipMemsetAsync(tmp, 0, (size_t)B * C * H * W * 4, st);
      PRN_REQUIRE(e == hipSuccess, "prn_bottleneck_train_bwd: clearing the strided input gradient failed: %s", hipGetErrorString(e));
      PRN_TRY(prn_conv2d_fwd_counted(&L.gd, dd, p->wd_t, nullptr, nullptr, nullptr, tmp, gws, nullptr, stream, 0));
      addx = tmp;
    }
  }
  // conv3's input gradient (-> K-split partial sums when bn2's backward kernel will sum them)
  float* g3 = f32(ws, L.b_g3);
  PRN_TRY(prn_conv2d_fwd_counted(&L.g3, d3, p->w3_t, p->w3_t_img, nullptr, nullptr, g3, raw(ws, L.b_g3ws, L.b_v - L.b_g3ws), nullptr, stream, L.lazy3g ? 1 : 0));
  // bn2 backward (ReLU sign recomputed from conv2's result): d2 = gradient of conv2's result (+ its Winograd input transform for conv2's input-gradient convolution)
  const float* in2 = L.lazy3g ? f32(ws, L.b_g3ws + L.parts3g_off) : g3;
  const int64_t n2 = (int64_t)B * P * Ho * Wo;
  float* Vg = f32(ws, L.b_v);
  if (L.v2g)
    PRN_TRY(prn_bn_bwd_winograd(in2, L.lazy3g ? L.parts3g : 1, L.lazy3g ? n2 : 0, c2, nullptr, st2, p->gamma[1], p->beta[1], d2, nullptr, dg2, db2, Vg, B, P, Ho, Wo, 1, 0, stream));
  else if (L.lazy3g)
    PRN_TRY(prn_bn_bwd_partials(in2, L.parts3g, n2, c2, nullptr, st2, p->gamma[1], p->beta[1], d2, nullptr, dg2, db2, B, P, Ho * Wo, 1, 0, stream));
  else
    PRN_TRY(prn_bn_bwd(g3, c2, nullptr, st2, p->gamma[1], p->beta[1], d2, nullptr, dg2, db2, L.small2 ? nullptr : bnws, B, P, Ho * Wo, 1, 0, stream));
  // conv2's input gradient -> g2 (or, in the Winograd domain, left to bn1's backward kernel)
  float* g2 = f32(ws, L.b_g2);
  float* Ytg = f32(ws, L.b_yt);
  if (L.conv2_path == PRN_BLK_CONV2_WINOGRAD) {
    if (!L.v2g) PRN_TRY(prn_winograd_input(d2, Vg, B, P, H, W, PRN_IN_ZERO, stream));
    PRN_TRY(prn_gemm_batched(P, P, L.P4, 36, p->ut2, p->ut2_img, Vg, Ytg, raw(ws, L.b_gemm, L.b_g2 - L.b_gemm), o, stream));
    if (!L.lazy2g) PRN_TRY(prn_winograd_output(Ytg, nullptr, nullptr, g2, B, P, H, W, PRN_EPI_NONE, stream));
  } else if (L.conv2_path == PRN_BLK_CONV2_DCN) {
    const float* om = f32(save, L.s_om);
    float* dom = f32(gsave, L.q_dom);
    float* dx1 = f32(ws, L.b_dx1);
    void* cws = raw(ws, L.b_dcn);
    PRN_TRY(prn_dcnv2_bwd_input(&L.dcn, d2, p->w2_cols_t, p->w2_cols_t_img, om, nullptr, dx1, cws, stream));      // column gradient into cws, sampler's input gradient
    PRN_TRY(prn_dcnv2_bwd_offset_mask(&L.dcn, a1, om, nullptr, dom, nullptr, cws, stream));                        // gradient of the raw offset | modulator map
    PRN_TRY(prn_conv2d_fwd_counted(&L.g27, dom, p->w27_t, nullptr, nullptr, dx1, g2, raw(ws, L.b_g27ws, L.b_g1ws - L.b_g27ws), nullptr, stream, 0));
  } else {
    PRN_TRY(prn_conv2d_fwd_counted(&L.g2, d2, p->w2_t, nullptr, nullptr, nullptr, g2, raw(ws, L.b_g2ws, L.b_dcn - L.b_g2ws), nullptr, stream, 0));
  }
  // bn1 backward: d1 = gradient of conv1's result
  if (L.conv2_path == PRN_BLK_CONV2_WINOGRAD && L.lazy2g)
    PRN_TRY(prn_winograd_output_bn_bwd(Ytg, c1, st1, p->gamma[0], p->beta[0], d1, dg1, db1, B, P, H, W, 1, stream));
  else
    PRN_TRY(prn_bn_bwd(g2, c1, nullptr, st1, p->gamma[0], p->beta[0], d1, nullptr, dg1, db1, L.small1 ? nullptr : bnws, B, P, H * W, 1, 0, stream));
  // conv1's input gradient; the identity path's gradient joins in its epilogue
  return prn_conv2d_fwd_counted(&L.g1, d1, p->w1_t, p->w1_t_img, nullptr, addx, dx, raw(ws, L.b_g1ws, L.b_gdtmp - L.b_g1ws), nullptr, stream, 0);
}
