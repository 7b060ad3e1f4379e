// Router: gate logits (f64 accumulate) + top-k + softmax.  HBM-bound: reads the token matrix once.
// Replaces fmoe NaiveGate / SwitchGate forward.
//
// Layout: one wave per token row (coalesced 16-B loads of the whole row: lane l owns elements
// [256c + 4l, 256c + 4l + 4) of chunk c), eight experts at a time.  The router weights sit in LDS as
// f64 (or are read through L2 as f32 when E*d*8 exceeds the LDS budget).  The 8 per-lane partial sums
// of a group of 8 experts are reduced with a transposed butterfly (10 shuffles instead of 48): after
// it, lanes with equal (lane>>3) hold the full logit of expert 4*b5 + 2*b4 + b3.
#include "smoe_common.h"

namespace {

constexpr int ROUTER_THREADS = 512;
constexpr int ROUTER_WAVES = ROUTER_THREADS / 64;
constexpr int ROUTER_MAX_K = 8;

__device__ __forceinline__ double shfl_xor_f64(double v, int m) { return __shfl_xor(v, m, 64); }

template <typename XT, int NCH, bool W_LDS>
__global__ __launch_bounds__(ROUTER_THREADS) void router_kernel(
    const XT* __restrict__ x, const float* __restrict__ wg, const float* __restrict__ bg,
    const float* __restrict__ noise, int64_t T, int d, int E, int k, int gate_kind,
    int64_t* __restrict__ idx_out, float* __restrict__ score_out, float* __restrict__ logits_out,
    float* __restrict__ probs_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Epad = (E + 7) & ~7;
  // carve: [ROUTER_WAVES][Epad] f32 logits, then (W_LDS) [Epad][d] f64 weights
  float* lds_logit = reinterpret_cast<float*>(smem);
  const size_t logit_bytes = ((size_t)ROUTER_WAVES * Epad * 4 + 15) & ~(size_t)15;
  double* lds_w = reinterpret_cast<double*>(smem + logit_bytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  if (W_LDS) {
    for (int i = tid; i < Epad * d; i += ROUTER_THREADS) {
      const int e = i / d;
      lds_w[i] = (e < E) ? (double)wg[i] : 0.0;
    }
    __syncthreads();
  }
  float* my_logit = lds_logit + wave * Epad;

  const int64_t wave_gid = (int64_t)blockIdx.x * ROUTER_WAVES + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * ROUTER_WAVES;

  bool cvalid[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) cvalid[c] = (c * 256 + lane * 4) < d;

  float xn[NCH][4];
  auto load_row = [&](int64_t t) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (cvalid[c]) {
        load4(x + t * (int64_t)d + c * 256 + lane * 4, xn[c]);
      } else {
        xn[c][0] = xn[c][1] = xn[c][2] = xn[c][3] = 0.f;
      }
    }
  };

  int64_t t = wave_gid;
  if (t < T) load_row(t);
  for (; t < T; t += wave_stride) {
    double xd[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) xd[c][j] = (double)xn[c][j];
    const int64_t tn = t + wave_stride;
    if (tn < T) load_row(tn);  // prefetch the next row under this row's arithmetic

    for (int eb = 0; eb < Epad; eb += 8) {
      double acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (cvalid[c]) {
          const int col = c * 256 + lane * 4;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            double w0, w1, w2, w3;
            if (W_LDS) {
              const double* wp = lds_w + (size_t)(eb + i) * d + col;
              w0 = wp[0]; w1 = wp[1]; w2 = wp[2]; w3 = wp[3];
            } else {
              if (eb + i < E) {
                float wf[4];
                load4(wg + (size_t)(eb + i) * d + col, wf);
                w0 = wf[0]; w1 = wf[1]; w2 = wf[2]; w3 = wf[3];
              } else {
                w0 = w1 = w2 = w3 = 0.0;
              }
            }
            acc[i] = fma(xd[c][0], w0, acc[i]);
            acc[i] = fma(xd[c][1], w1, acc[i]);
            acc[i] = fma(xd[c][2], w2, acc[i]);
            acc[i] = fma(xd[c][3], w3, acc[i]);
          }
        }
      }
      // transposed butterfly: 8 values/lane -> 1 value/lane
      double a4[4], a2[2], a1;
      {
        const bool hi = lane & 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double send = hi ? acc[i] : acc[i + 4];
          const double keep = hi ? acc[i + 4] : acc[i];
          a4[i] = keep + shfl_xor_f64(send, 32);
        }
      }
      {
        const bool hi = lane & 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const double send = hi ? a4[i] : a4[i + 2];
          const double keep = hi ? a4[i + 2] : a4[i];
          a2[i] = keep + shfl_xor_f64(send, 16);
        }
      }
      {
        const bool hi = lane & 8;
        const double send = hi ? a2[0] : a2[1];
        const double keep = hi ? a2[1] : a2[0];
        a1 = keep + shfl_xor_f64(send, 8);
      }
      a1 += shfl_xor_f64(a1, 4);
      a1 += shfl_xor_f64(a1, 2);
      a1 += shfl_xor_f64(a1, 1);
      const int el = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
      const int e = eb + el;
      if ((lane & 7) == 0 && e < E) {
        const double b = bg ? (double)bg[e] : 0.0;
        float lg = (float)(a1 + b);
        if (logits_out) logits_out[t * (int64_t)E + e] = lg;
        if (gate_kind == SMOE_GATE_SWITCH && noise) lg += noise[t * (int64_t)E + e];
        my_logit[e] = lg;
      }
    }
    // LDS writes by this wave are visible to this wave after the wait (same wave, in-order DS pipe)
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();

    // ---- top-k over my_logit[0..E) : ties -> lowest expert id ----
    int chosen[ROUTER_MAX_K];
    float chosen_val[ROUTER_MAX_K];
#pragma unroll
    for (int r = 0; r < ROUTER_MAX_K; ++r) { chosen[r] = -1; chosen_val[r] = 0.f; }
#pragma unroll
    for (int r = 0; r < ROUTER_MAX_K; ++r) {
      if (r < k) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int e = lane; e < E; e += 64) {
          bool taken = false;
#pragma unroll
          for (int q = 0; q < ROUTER_MAX_K; ++q) taken |= (q < r) && (chosen[q] == e);
          const float v = my_logit[e];
          if (!taken && (v > bv || (v == bv && e < bi) || bi == 0x7fffffff)) { bv = v; bi = e; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
          const float ov = __shfl_xor(bv, m, 64);
          const int oi = __shfl_xor(bi, m, 64);
          const bool take = (oi != 0x7fffffff) && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi));
          if (take) { bv = ov; bi = oi; }
        }
        chosen[r] = bi;
        chosen_val[r] = bv;
      }
    }

    if (gate_kind == SMOE_GATE_NAIVE) {
      // softmax over the k kept logits only; chosen_val[0] is the max
      float ex[ROUTER_MAX_K];
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < ROUTER_MAX_K; ++r) {
        ex[r] = (r < k) ? expf(chosen_val[r] - chosen_val[0]) : 0.f;
        s += ex[r];
      }
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < ROUTER_MAX_K; ++r) {
          if (r < k) {
            idx_out[t * (int64_t)k + r] = chosen[r];
            score_out[t * (int64_t)k + r] = ex[r] / s;
          }
        }
      }
    } else {
      // switch gate: full softmax over E, k == 1
      const float mx = chosen_val[0];
      float s = 0.f;
      for (int e = lane; e < E; e += 64) s += expf(my_logit[e] - mx);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
      if (probs_out) {
        for (int e = lane; e < E; e += 64) probs_out[t * (int64_t)E + e] = expf(my_logit[e] - mx) / s;
      }
      if (lane == 0) {
        idx_out[t] = chosen[0];
        score_out[t] = 1.0f / s;  // exp(0)/s
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename XT, int NCH>
int launch_router(const void* x, const float* wg, const float* bg, const float* noise, int64_t T, int d,
                  int E, int k, int gate_kind, int64_t* idx, float* score, float* logits_out, float* probs,
                  hipStream_t stream) {
  const int Epad = (E + 7) & ~7;
  const size_t logit_bytes = ((size_t)ROUTER_WAVES * Epad * 4 + 15) & ~(size_t)15;
  const size_t w_bytes = (size_t)Epad * d * 8;
  const bool w_lds = (logit_bytes + w_bytes) <= 64 * 1024;
  const size_t smem = logit_bytes + (w_lds ? w_bytes : 0);
  int64_t need = (T + ROUTER_WAVES - 1) / ROUTER_WAVES;
  int grid = (int)(need < 512 ? (need < 1 ? 1 : need) : 512);
  if (w_lds) {
    hipLaunchKernelGGL((router_kernel<XT, NCH, true>), dim3(grid), dim3(ROUTER_THREADS), smem, stream,
                       (const XT*)x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs);
  } else {
    hipLaunchKernelGGL((router_kernel<XT, NCH, false>), dim3(grid), dim3(ROUTER_THREADS), smem, stream,
                       (const XT*)x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs);
  }
  SMOE_CHECK_LAUNCH("smoe_router_topk");
  return 0;
}

template <typename XT>
int dispatch_nch(const void* x, const float* wg, const float* bg, const float* noise, int64_t T, int d, int E,
                 int k, int gate_kind, int64_t* idx, float* score, float* logits_out, float* probs,
                 hipStream_t s) {
  const int nch = (d + 255) / 256;
  switch (nch) {
    case 1: return launch_router<XT, 1>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case 2: return launch_router<XT, 2>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case 3: return launch_router<XT, 3>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case 4: return launch_router<XT, 4>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case 5: return launch_router<XT, 5>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case 6: return launch_router<XT, 6>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case 7: return launch_router<XT, 7>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case 8: return launch_router<XT, 8>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
  }
  smoe_set_error("smoe_router_topk: d=%d unsupported (d <= 2048)", d);
  return 1;
}

}  // namespace

extern "C" int smoe_router_topk(const void* x, int x_dtype, const float* wg, const float* bg, const float* noise,
                                int64_t T, int d, int E, int k, int gate_kind, int64_t* idx, float* score,
                                float* logits_out, float* probs, void* stream) {
  SMOE_REQUIRE(x && wg && idx && score, "smoe_router_topk: null pointer");
  SMOE_REQUIRE(T >= 0 && d > 0 && E > 0, "smoe_router_topk: bad sizes T=%lld d=%d E=%d", (long long)T, d, E);
  SMOE_REQUIRE(d % 8 == 0 && d <= 2048, "smoe_router_topk: d=%d must be a multiple of 8 and <= 2048", d);
  SMOE_REQUIRE(k >= 1 && k <= E && k <= ROUTER_MAX_K, "smoe_router_topk: k=%d out of range (E=%d, max %d)", k, E,
               ROUTER_MAX_K);
  SMOE_REQUIRE(E <= 4096, "smoe_router_topk: E=%d too large", E);
  SMOE_REQUIRE(gate_kind == SMOE_GATE_NAIVE || gate_kind == SMOE_GATE_SWITCH, "smoe_router_topk: bad gate_kind %d",
               gate_kind);
  SMOE_REQUIRE(gate_kind != SMOE_GATE_SWITCH || k == 1, "smoe_router_topk: switch gate needs k == 1");
  if (T == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  switch (x_dtype) {
    case SMOE_F32: return dispatch_nch<float>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case SMOE_F16: return dispatch_nch<f16>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
    case SMOE_BF16: return dispatch_nch<bf16_bits>(x, wg, bg, noise, T, d, E, k, gate_kind, idx, score, logits_out, probs, s);
  }
  smoe_set_error("smoe_router_topk: bad x_dtype %d", x_dtype);
  return 1;
}
