// Token-skip gate of the reference's residual-MoE block (models/resMoE.py:32-85 `Gate`, used at 126-145), as HIP:
//   smoe_gate_ln_router  : [LayerNorm +] gate (+ router) in ONE pass over the activations -- the glue around both halves
//                          of forward_residule_moe: norm1 + dense_gate in front of attention, norm2 + moe_gate + the
//                          MoE router in front of the experts
//   smoe_zero_row_output : what every skipped (all-zero) token receives from the MoE, computed once per layer
// The kernel is router16_kernel (router16_kernel.h) with GATE = 1 (gate + router) or 2 (gate only).
#include "router16_kernel.h"

namespace {
using namespace r16;

struct GLnArgs {
  const float* g; const float* b; float eps; void* xn16; int xn16_dtype; float* xn32;
  int32_t* hist = nullptr;   // chunk histogram for the dispatch plan (gate + router only), or NULL
};

template <int NJ, bool LN, typename NT, int GATE>
int launch_gate(const float* x, const GLnArgs& ln, const SkipGateArgs& ga, const float* wg, const float* bg, int64_t T, int d,
                int E, int k, int32_t* rc, int32_t* rl, int64_t* idx, float* score, hipStream_t s, bool ws_zero) {
  constexpr int EB = 8;
  constexpr size_t smem = router16_smem<NJ, LN, EB, GATE>();
  const int64_t tok_per_block = (R16_THREADS / 64) * 4;
  int64_t need = (T + tok_per_block - 1) / tok_per_block;
  constexpr int64_t max_wg = 768;
  const int64_t iters = (need + max_wg - 1) / max_wg;
  const int grid = (int)(need < 1 ? 1 : (need + iters - 1) / (iters < 1 ? 1 : iters));
  HistArgs ha{GATE == 1 ? ln.hist : nullptr, R16_HIST_TOK};
  const int grid0 = ha.hist ? (int)((T + R16_HIST_TOK - 1) / R16_HIST_TOK) : grid;
  if (!ws_zero) {   // (a kept workspace is zero already: the redo pass clears its counter words on the way out)
    hipError_t me = smoe_zero_words(rc, 4, s);
    if (me != hipSuccess) {
      smoe_set_error("smoe_gate_ln_router: counter clear failed: %s", hipGetErrorString(me));
      return (int)me;
    }
  }
  if (smem > 64 * 1024) {
    SMOE_ENSURE_SMEM(router16_kernel<float, NJ, 0, LN, NT, EB, GATE>);
    SMOE_ENSURE_SMEM(router16_kernel<float, NJ, 1, LN, NT, EB, GATE>);
  }
#define G_LAUNCH(MODE, GRID)                                                                                          \
  hipLaunchKernelGGL((router16_kernel<float, NJ, MODE, LN, NT, EB, GATE>), dim3(GRID), dim3(R16_THREADS), smem, s, x,   \
                     ln.g, ln.b, ln.eps, (NT*)ln.xn16, ln.xn32, wg, bg, (const float*)nullptr, T, d, E, k,            \
                     (int)SMOE_GATE_NAIVE, rc, rl, idx, score, (float*)nullptr, (float*)nullptr, ga, ha)
  G_LAUNCH(0, grid0);
  SMOE_CHECK_LAUNCH("smoe_gate_ln_router/f32");
  G_LAUNCH(1, (grid < 16 ? grid : 16));
  SMOE_CHECK_LAUNCH("smoe_gate_ln_router/redo");
#undef G_LAUNCH
  return 0;
}

template <bool LN, typename NT, int GATE>
int gate_by_d(const float* x, const GLnArgs& ln, const SkipGateArgs& ga, const float* wg, const float* bg, int64_t T, int d,
              int E, int k, int32_t* rc, int32_t* rl, int64_t* idx, float* score, hipStream_t s, bool ws_zero) {
  switch (d) {
    case 192: return launch_gate<3, LN, NT, GATE>(x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
    case 384: return launch_gate<6, LN, NT, GATE>(x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
    case 768: return launch_gate<12, LN, NT, GATE>(x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
    case 1024: return launch_gate<16, LN, NT, GATE>(x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
  }
  smoe_set_error("smoe_gate_ln_router: unsupported d=%d", d);
  return 1;
}

template <int GATE>
int gate_by_ln(bool with_ln, const float* x, const GLnArgs& ln, const SkipGateArgs& ga, const float* wg, const float* bg,
               int64_t T, int d, int E, int k, int32_t* rc, int32_t* rl, int64_t* idx, float* score, hipStream_t s,
               bool ws_zero) {
  const bool bf = ln.xn16_dtype == SMOE_BF16;
  if (with_ln) {
    if (bf) return gate_by_d<true, bf16_bits, GATE>(x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
    return gate_by_d<true, f16, GATE>(x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
  }
  if (bf) return gate_by_d<false, bf16_bits, GATE>(x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
  return gate_by_d<false, f16, GATE>(x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
}

// out[c] = sum_j score_j (sum_h W2[e_j][c,h] gelu(b1[e_j][h]) + b2[e_j][c]);  (e_j, score_j) = the NaiveGate routing of an
// all-zero row: top-k of the gate bias (ties -> lowest id, descending value), softmax over the kept k.  One wave per column.
__global__ __launch_bounds__(256) void zero_row_output_kernel(const float* __restrict__ bg, int E, int k,
                                                              const float* __restrict__ w2, const float* __restrict__ b1,
                                                              const float* __restrict__ b2, int d, int h, int e_base,
                                                              int E_local, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= d) return;
  int chosen[R16_MAX_K];
  float cval[R16_MAX_K];
  for (int r = 0; r < k; ++r) {
    int bi = -1;
    float bv = 0.f;
    for (int e = 0; e < E; ++e) {
      bool taken = false;
      for (int r2 = 0; r2 < r; ++r2) taken |= (chosen[r2] == e);
      if (taken) continue;
      const float v = bg ? bg[e] : 0.f;
      if (bi < 0 || v > bv) { bi = e; bv = v; }
    }
    chosen[r] = bi;
    cval[r] = bv;
  }
  float ssum = 0.f, ex[R16_MAX_K];
  for (int r = 0; r < k; ++r) { ex[r] = expf(cval[r] - cval[0]); ssum += ex[r]; }
  float total = 0.f;
  for (int r = 0; r < k; ++r) {
    const int e = chosen[r] - e_base;              // expert parallel: only this rank's experts contribute to its partial sum
    if (e < 0 || e >= E_local) continue;
    const float* wrow = w2 + ((int64_t)e * d + c) * h;
    float acc = 0.f;
    for (int j = lane; j < h; j += 64) {
      const float v = b1 ? b1[(int64_t)e * h + j] : 0.f;
      const float gl = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
      acc = fmaf(wrow[j], gl, acc);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    total += (ex[r] / ssum) * (acc + (b2 ? b2[(int64_t)e * d + c] : 0.f));
  }
  if (lane == 0) out[c] = total;
}

// Backward of one gated half of the residual-MoE block in TRAINING (models/resMoE.py:68-77 hard masks with the straight-through
// estimator, 131-143): with xn the normed activations, keep / skip the hard decisions and p = sigmoid(<xn, w> + b),
//     tk = xn * m1,  skip_tk = xn * m0,  out = f(tk) + tk + skip_tk,   m1 = keep + p.detach() - p,  m0 = skip + (1 - p).detach() - (1 - p)
// and the gradients g_f = dL/d(tk as the operator's input), g_out = dL/d(out):
//     dL/dp = -<g_f, xn>            (the two <g_out, xn> terms of m1 and m0 cancel: d m1 / dp = -1, d m0 / dp = +1)
//     dz    = dL/dp * p (1 - p)     (d loss / d gate logit; dw = dz^T xn and db = sum dz follow outside: smoe_gate_wgrad)
//     dxn   = g_f * keep + g_out + dz * w
// One pass: 16 lanes per token (the router's layout), p recomputed from xn (no saved activations beyond xn and the decisions).
// gate_on = 0 (a disabled gate passes everything and has no gradient): dxn = g_f + g_out, dz = 0.
template <typename GT, int NJ>
__global__ __launch_bounds__(R16_THREADS, 4) void skip_gate_bwd_kernel(const float* __restrict__ xn, const GT* __restrict__ g_f,
                                                                       const float* __restrict__ g_out, const float* __restrict__ w,
                                                                       const float* __restrict__ b, const float* __restrict__ mask,
                                                                       int gate_on, int64_t T, float* __restrict__ dxn,
                                                                       float* __restrict__ dz_out) {
  constexpr int d = 64 * NJ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, u = lane & 15;
  const int64_t slot_stride = (int64_t)gridDim.x * (R16_THREADS / 64) * 4;
  const float bias = (b && gate_on) ? *b : 0.f;
  for (int64_t it0 = ((int64_t)blockIdx.x * (R16_THREADS / 64) + wave) * 4; it0 < T; it0 += slot_stride) {
    const int64_t t = it0 + q;
    const bool live = t < T;
    const int64_t tc = live ? t : T - 1;
    const int64_t rowoff = tc * (int64_t)d + u * 4;
    f32x4 xv[NJ], gv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xv[j] = *reinterpret_cast<const f32x4*>(xn + rowoff + 64 * j);
      float tmp[4];
      load4(g_f + rowoff + 64 * j, tmp);
      gv[j] = f32x4{tmp[0], tmp[1], tmp[2], tmp[3]};
    }
    float az = 0.f, ad = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + u * 4 + 64 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        az = fmaf(xv[j][i], wv[i], az);
        ad = fmaf(gv[j][i], xv[j][i], ad);
      }
    }
    const float z = row16_sum(az) + bias, dot = row16_sum(ad);
    const float p = 1.0f / (1.0f + expf(-z));
    const float dz = gate_on ? -dot * p * (1.0f - p) : 0.f;
    const float keep = gate_on ? mask[tc * 2 + 1] : 1.f;
    if (live) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + u * 4 + 64 * j);
        f32x4 o = gv[j] * f32x4{keep, keep, keep, keep} + wv * f32x4{dz, dz, dz, dz};
        if (g_out) o += *reinterpret_cast<const f32x4*>(g_out + rowoff + 64 * j);
        *reinterpret_cast<f32x4*>(dxn + rowoff + 64 * j) = o;
      }
      if (u == 0 && dz_out) dz_out[t] = dz;
    }
  }
}

template <typename GT>
int launch_skip_gate_bwd(const float* xn, const void* g_f, const float* g_out, const float* w, const float* b, const float* mask,
                         int gate_on, int64_t T, int d, float* dxn, float* dz, hipStream_t s) {
  int64_t need = (T + 15) / 16;
  const int grid = (int)(need < 1 ? 1 : (need > 4096 ? 4096 : need));
#define SGB(NJ) hipLaunchKernelGGL((skip_gate_bwd_kernel<GT, NJ>), dim3(grid), dim3(R16_THREADS), 0, s, xn, (const GT*)g_f, g_out, w, b, mask, gate_on, T, dxn, dz)
  switch (d) {
    case 192: SGB(3); break;
    case 384: SGB(6); break;
    case 768: SGB(12); break;
    case 1024: SGB(16); break;
    default: smoe_set_error("smoe_skip_gate_bwd: unsupported d=%d", d); return 1;
  }
#undef SGB
  SMOE_CHECK_LAUNCH("smoe_skip_gate_bwd");
  return 0;
}

}  // namespace

extern "C" int smoe_skip_gate_bwd(const float* xn, const void* g_f, int g_f_dtype, const float* g_out, const float* gate_w,
                                  const float* gate_b, const float* mask, int gate_on, int64_t T, int d, float* dxn, float* dz,
                                  void* stream) {
  if (T == 0) return 0;
  SMOE_REQUIRE(xn && g_f && gate_w && dxn, "smoe_skip_gate_bwd: null pointer");
  SMOE_REQUIRE(!gate_on || mask, "smoe_skip_gate_bwd: an enabled gate needs the forward's decisions (mask)");
  SMOE_REQUIRE(T > 0 && T < (1ll << 31), "smoe_skip_gate_bwd: bad T");
  hipStream_t s = (hipStream_t)stream;
  switch (g_f_dtype) {
    case SMOE_F32: return launch_skip_gate_bwd<float>(xn, g_f, g_out, gate_w, gate_b, mask, gate_on, T, d, dxn, dz, s);
    case SMOE_F16: return launch_skip_gate_bwd<f16>(xn, g_f, g_out, gate_w, gate_b, mask, gate_on, T, d, dxn, dz, s);
    case SMOE_BF16: return launch_skip_gate_bwd<bf16_bits>(xn, g_f, g_out, gate_w, gate_b, mask, gate_on, T, d, dxn, dz, s);
  }
  smoe_set_error("smoe_skip_gate_bwd: bad g_f dtype %d", g_f_dtype);
  return 1;
}

extern "C" int smoe_gate_ln_router_supported(int d, int E, int k) {
  return ((d == 192 || d == 384 || d == 768 || d == 1024) && E >= 0 && E <= 8 && (E == 0 || (k >= 1 && k <= E && k <= R16_MAX_K))) ? 1 : 0;
}

extern "C" int smoe_gate_ln_router(const void* x, int x_dtype, int with_ln, const float* ln_gamma, const float* ln_beta,
                                   float ln_eps, const float* gate_w, const float* gate_b, const float* threshold,
                                   void* xn16, int xn16_dtype, float* xn32, const float* zero_out, const float* wg,
                                   const float* bg, int64_t T, int d, int E, int k, int64_t* idx, int64_t* idx_plan,
                                   float* score, float* mask, int32_t* skip_count, int32_t* chunk_hist, float* tk32,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (T == 0) return 0;
  SMOE_REQUIRE(x && gate_w, "smoe_gate_ln_router: null pointer");
  SMOE_REQUIRE(x_dtype == SMOE_F32, "smoe_gate_ln_router: x must be f32 (the residual stream / the normed activations)");
  SMOE_REQUIRE(smoe_gate_ln_router_supported(d, E, k), "smoe_gate_ln_router: unsupported shape d=%d E=%d k=%d", d, E, k);
  SMOE_REQUIRE(T > 0 && T < (1ll << 31), "smoe_gate_ln_router: bad T");
  SMOE_REQUIRE(E == 0 || (wg && idx && score), "smoe_gate_ln_router: router outputs missing");
  SMOE_REQUIRE(xn16_dtype == SMOE_F16 || xn16_dtype == SMOE_BF16, "smoe_gate_ln_router: xn16 must be f16 or bf16");
  SMOE_REQUIRE(workspace && workspace_bytes >= 16 + (((size_t)T * 4 + 15) & ~(size_t)15), "smoe_gate_ln_router: workspace too small");
  int32_t* rc = reinterpret_cast<int32_t*>(workspace);
  int32_t* rl = reinterpret_cast<int32_t*>((char*)workspace + 16);
  GLnArgs ln{ln_gamma, ln_beta, ln_eps, xn16, xn16_dtype, xn32};
  ln.hist = (chunk_hist && E > 0 && 1024 % (R16_HIST_TOK * k) == 0) ? chunk_hist : nullptr;
  SkipGateArgs ga{gate_w, gate_b, threshold, skip_count, mask, zero_out, E > 0 ? idx_plan : nullptr, tk32};
  hipStream_t s = (hipStream_t)stream;
  const bool ws_zero = (with_ln & 2) != 0;   // bit 1: the caller keeps the workspace's counter words zero between calls
  if (E == 0) return gate_by_ln<2>((with_ln & 1) != 0, (const float*)x, ln, ga, nullptr, nullptr, T, d, 0, 1, rc, rl, nullptr, nullptr, s, ws_zero);
  return gate_by_ln<1>((with_ln & 1) != 0, (const float*)x, ln, ga, wg, bg, T, d, E, k, rc, rl, idx, score, s, ws_zero);
}

extern "C" int smoe_zero_row_output(const float* bg, int E, int k, const float* w2, const float* b1, const float* b2, int d,
                                    int h, int e_base, int E_local, float* out, void* stream) {
  SMOE_REQUIRE(w2 && out, "smoe_zero_row_output: null pointer");
  SMOE_REQUIRE(E >= 1 && k >= 1 && k <= E && k <= R16_MAX_K && d > 0 && h > 0, "smoe_zero_row_output: bad sizes E=%d k=%d d=%d h=%d", E, k, d, h);
  SMOE_REQUIRE(e_base >= 0 && E_local >= 1 && e_base + E_local <= E, "smoe_zero_row_output: local experts [%d, %d) outside [0, %d)",
               e_base, e_base + E_local, E);
  hipLaunchKernelGGL(zero_row_output_kernel, dim3((d + 3) / 4), dim3(256), 0, (hipStream_t)stream, bg, E, k, w2, b1, b2, d, h, e_base,
                     E_local, out);
  SMOE_CHECK_LAUNCH("smoe_zero_row_output");
  return 0;
}
