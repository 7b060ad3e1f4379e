// Router, 16-lanes-per-token layout (E <= 8 at d in {192, 384, 768, 1024}; E <= 32 at d in {768, 1024}): the fast
// path of smoe_router_topk.  EB = experts held per lane (8, 16 or 32 accumulator pairs).
//
// Four tokens per wave: lane = 16 q + u handles the float4 chunks u, u+16, u+32, ... of token slot q (every load
// instruction covers 4 x 256 contiguous bytes).  Per-token reductions are 4 DPP-modified adds inside a 16-lane
// DPP row (quad_perm xor-1, xor-2, row_ror 4, row_ror 8) -- plain VALU, no LDS round trips, no ds_bpermute --
// after which every lane of the row holds all E logits in registers and the top-(k+1) selection is a short
// unrolled compare chain.  Weights sit in LDS as f32; the four token slots read the same addresses (broadcast).
// Same contract as router.hip: f32 logits with a rigorous error bound, tokens whose deciding gaps fall inside the
// bound go to the redo list and are recomputed with f64 accumulation (MODE 1, same layout).
#include "smoe_common.h"
#include <type_traits>

namespace {

constexpr int R16_THREADS = 256;
constexpr int R16_MAX_K = 4;

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
  return v + __builtin_bit_cast(float, moved);
}
// sum over the 16 lanes of a DPP row; every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x128>(v);  // row_ror:8
  return v;
}
__device__ __forceinline__ double row16_sum(double v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}

__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// LN = fused LayerNorm in front of the router (models/vision_transformer.py:321 `mlp(norm2(x))`): the row is
// normalised in registers (two-pass mean / variance over the 16-lane row, f32), written once as the 16-bit
// operand image the expert GEMM gathers from (xn16) and optionally as f32 (xn32), and routed on its f32 value.
// The f64 redo pass recomputes the same normalisation with the same lane layout, hence bit-identical inputs.
template <typename XT, int NJ, int MODE, bool LN, typename NT, int EB>
__global__ __launch_bounds__(R16_THREADS, (MODE == 0 ? (EB <= 8 ? 4 : (EB <= 16 ? 2 : 1)) : (EB <= 16 ? 2 : 1))) void router16_kernel(
    const XT* __restrict__ x, const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    NT* __restrict__ xn16, float* __restrict__ xn32, const float* __restrict__ wg, const float* __restrict__ bg,
    const float* __restrict__ noise, int64_t T, int d, int E, int k, int gate_kind,
    int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_list, int64_t* __restrict__ idx_out,
    float* __restrict__ score_out, float* __restrict__ logits_out, float* __restrict__ probs_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_w = reinterpret_cast<float*>(smem);            // [EB][d], rows >= E zero
  float* lds_wn2 = lds_w + EB * d;                        // [EB]
  float* lds_bias = lds_wn2 + EB;                         // [EB] gate bias, zero where absent (branch-free add)
  float* lds_g = lds_bias + EB;                           // [d] LayerNorm weight, then [d] bias (LN only)
  float* lds_be = lds_g + d;
  d = 64 * NJ;  // the launcher only dispatches exact multiples: makes every chunk bound below compile-time
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, u = lane & 15;
  if (MODE == 1 && redo_list && *redo_count == 0) return;  // nothing to redo (the common case): exit before any setup

  int64_t n_items = T;
  if (MODE == 1 && redo_list) {  // a trip count read from device memory is never trusted: at most T tokens can be listed
    n_items = *redo_count;
    n_items = n_items < 0 ? 0 : (n_items > T ? T : n_items);
  }
  const int64_t slot_gid = ((int64_t)blockIdx.x * (R16_THREADS / 64) + wave) * 4 + q;
  const int64_t slot_stride = (int64_t)gridDim.x * (R16_THREADS / 64) * 4;
  // The row of the NEXT item is fetched as soon as the current one's registers are free -- the first one before
  // the weight staging below, so the HBM latency of the first rows runs under the prologue.  Dead slots of the
  // last group re-read the last item (no predication on the loads; only stores are guarded).
  int64_t it0 = slot_gid - q;  // all four slots of a wave iterate together (DPP needs the whole row active)
  f32x4 xv[NJ];
  int64_t t_next = 0;
  bool live_next = false;
  auto fetch = [&](int64_t i0) {
    const int64_t it = i0 + q;
    live_next = it < n_items;
    const int64_t itc = live_next ? it : n_items - 1;
    t_next = (MODE == 1 && redo_list) ? (int64_t)redo_list[itc] : itc;
    if (MODE == 1) t_next = t_next < 0 ? 0 : (t_next >= T ? T - 1 : t_next);  // list entries index x: keep them inside it
    const XT* src = x + t_next * (int64_t)(64 * NJ) + u * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float tmp[4];
      load4(src + 64 * j, tmp);
      xv[j] = f32x4{tmp[0], tmp[1], tmp[2], tmp[3]};
    }
  };
  if (it0 < n_items) fetch(it0);

  for (int i = tid * 4; i < EB * d; i += R16_THREADS * 4) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i / d < E) v = *reinterpret_cast<const f32x4*>(wg + i);
    *reinterpret_cast<f32x4*>(lds_w + i) = v;
  }
  if (tid < EB) lds_bias[tid] = (bg && tid < E) ? bg[tid] : 0.f;
  if (LN) {
    for (int i = tid; i < d; i += R16_THREADS) {
      lds_g[i] = ln_g ? ln_g[i] : 1.f;
      lds_be[i] = ln_b ? ln_b[i] : 0.f;
    }
  }
  __syncthreads();
  for (int e = wave; MODE == 0 && e < EB; e += R16_THREADS / 64) {  // squared row norms, one wave per expert row
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s = fmaf(lds_w[e * d + c], lds_w[e * d + c], s);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) lds_wn2[e] = s;
  }
  __syncthreads();
  float wmax2 = 0.f;
#pragma unroll
  for (int e = 0; e < EB; ++e) wmax2 = fmaxf(wmax2, lds_wn2[e]);

  while (it0 < n_items) {
    const int64_t t = t_next;
    const bool live = live_next;
    const int64_t rowoff = t * (int64_t)d + u * 4;  // this lane's first chunk; chunk j sits 64 j elements further
    // LDS offset of this lane's first chunk through a per-iteration opaque zero: otherwise the loop-invariant
    // LDS reads (gamma, beta, all weights) are hoisted out of the token loop and spilled
    int lz = 0;
    asm volatile("" : "+v"(lz));
    const int ub = u * 4 + lz;
    if constexpr (LN) {
      constexpr float inv_d = 1.0f / (float)(64 * NJ);
      f32x2 s1 = f32x2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) s1 += lo2(xv[j]) + hi2(xv[j]);
      const float mean = row16_sum(s1[0] + s1[1]) * inv_d;
      const f32x2 mean2 = f32x2{mean, mean};
      f32x2 s2 = f32x2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x2 a = lo2(xv[j]) - mean2, b = hi2(xv[j]) - mean2;
        s2 = __builtin_elementwise_fma(a, a, s2);
        s2 = __builtin_elementwise_fma(b, b, s2);
      }
      const float rstd = rsqrtf(row16_sum(s2[0] + s2[1]) * inv_d + ln_eps);
      const f32x4 mean4 = f32x4{mean, mean, mean, mean}, rstd4 = f32x4{rstd, rstd, rstd, rstd};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(lds_g + ub + 64 * j);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(lds_be + ub + 64 * j);
        xv[j] = __builtin_elementwise_fma((xv[j] - mean4) * rstd4, gg, bb);
      }
      if (MODE == 0 && live) {
        if (xn32) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(xn32 + rowoff + 64 * j) = xv[j];
        }
        if (xn16) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if constexpr (std::is_same<NT, f16>::value) {
              f16x4 o; o[0] = (f16)xv[j][0]; o[1] = (f16)xv[j][1]; o[2] = (f16)xv[j][2]; o[3] = (f16)xv[j][3];
              *reinterpret_cast<f16x4*>(xn16 + rowoff + 64 * j) = o;
            } else {
              s16x4 o; o[0] = (short)f32_to_bf16(xv[j][0]); o[1] = (short)f32_to_bf16(xv[j][1]);
              o[2] = (short)f32_to_bf16(xv[j][2]); o[3] = (short)f32_to_bf16(xv[j][3]);
              *reinterpret_cast<s16x4*>(xn16 + rowoff + 64 * j) = o;
            }
          }
        }
      }
    }
    float lg[EB];
    if constexpr (MODE == 0) {
      // two partial sums per expert (even / odd element pairs): packed f32 FMAs, half the issue slots
      f32x2 acc[EB];
#pragma unroll
      for (int e = 0; e < EB; ++e) acc[e] = f32x2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(lds_w + ub + (e * d + 64 * j));
          acc[e] = __builtin_elementwise_fma(lo2(xv[j]), lo2(w), acc[e]);
          acc[e] = __builtin_elementwise_fma(hi2(xv[j]), hi2(w), acc[e]);
        }
        __builtin_amdgcn_sched_barrier(0);  // one chunk's weight reads next to their FMAs (else: e-major reorder + spills)
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) lg[e] = row16_sum(acc[e][0] + acc[e][1]) + lds_bias[e];
    } else {
      double acc[EB];
#pragma unroll
      for (int e = 0; e < EB; ++e) acc[e] = 0.0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(lds_w + ub + (e * d + 64 * j));
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[e] = fma((double)xv[j][i], (double)w[i], acc[e]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int e = 0; e < EB; ++e) lg[e] = (float)(row16_sum(acc[e]) + (double)lds_bias[e]);
    }
    float xs = 0.f;  // |x|^2 for the error bound (MODE 0); last use of the row registers
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) xs = fmaf(xv[j][i], xv[j][i], xs);
      xs = row16_sum(xs);
    }
    it0 += slot_stride;
    if (it0 < n_items) fetch(it0);  // next row on its way while this one is ranked and stored
    if (logits_out && live && u == 0) {
#pragma unroll
      for (int e = 0; e < EB; ++e)
        if (e < E) logits_out[t * (int64_t)E + e] = lg[e];
    }
    if (gate_kind == SMOE_GATE_SWITCH && noise && live) {
#pragma unroll
      for (int e = 0; e < EB; ++e)
        if (e < E) lg[e] += noise[t * (int64_t)E + e];
    }
    // top-kc in registers: ties -> lowest id, descending value.  Working copy with absent / already chosen
    // experts at -inf; strict > keeps the lowest id among equals.
    const int kc = (MODE == 0 && k < E) ? k + 1 : k;
    int chosen[R16_MAX_K + 1];
    float cval[R16_MAX_K + 1];
    float lw[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) lw[e] = (e < E) ? lg[e] : -INFINITY;
#pragma unroll
    for (int r = 0; r <= R16_MAX_K; ++r) {
      chosen[r] = 0;
      cval[r] = 0.f;
      if (r < kc) {
        float bv = lw[0];
        int bi = 0;
#pragma unroll
        for (int e = 1; e < EB; ++e) {
          const bool gt = lw[e] > bv;
          bv = gt ? lw[e] : bv;
          bi = gt ? e : bi;
        }
        chosen[r] = bi;
        cval[r] = bv;
        if (r + 1 < kc) {
#pragma unroll
          for (int e = 0; e < EB; ++e) lw[e] = (e == bi) ? -INFINITY : lw[e];
        }
      }
    }
    if constexpr (MODE == 0) {
      float amax = 0.f;
#pragma unroll
      for (int r = 0; r <= R16_MAX_K; ++r)
        if (r < kc) amax = fmaxf(amax, fabsf(cval[r]));
      // per-lane FMA chain 4*NJ, 4 reduction levels, bias add; factor 2 (two logits) x 2 (safety)
      const float bound = 4.0f * (float)(NJ * 4 + 6) * 5.9604645e-8f * sqrtf(xs * wmax2) + 9.6e-7f * (amax + 1.0f);
      bool ambiguous = false;
#pragma unroll
      for (int r = 0; r < R16_MAX_K; ++r)
        if (r + 1 < kc) ambiguous |= !((cval[r] - cval[r + 1]) > bound);
      if (ambiguous && live && u == 0) redo_list[atomicAdd(redo_count, 1)] = (int32_t)t;
    }
    if (live && u == 0) {
      if (gate_kind == SMOE_GATE_NAIVE) {
        float ex[R16_MAX_K];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < R16_MAX_K; ++r) {
          ex[r] = (r < k) ? expf(cval[r] - cval[0]) : 0.f;
          s += ex[r];
        }
#pragma unroll
        for (int r = 0; r < R16_MAX_K; ++r)
          if (r < k) {
            idx_out[t * (int64_t)k + r] = chosen[r];
            score_out[t * (int64_t)k + r] = ex[r] / s;
          }
      } else {
        const float mx = cval[0];
        float pe[EB];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          pe[e] = (e < E) ? expf(lg[e] - mx) : 0.f;
          s += pe[e];
        }
        if (probs_out) {
#pragma unroll
          for (int e = 0; e < EB; ++e)
            if (e < E) probs_out[t * (int64_t)E + e] = pe[e] / s;
        }
        idx_out[t] = chosen[0];
        score_out[t] = 1.0f / s;
      }
    }
  }
}

// LayerNorm alone, same 16-lanes-per-token layout and arithmetic as the fused router (block glue for the
// attention half of the block: `attn(norm1(x))`, models/vision_transformer.py:320): one pass, 16-bit or f32 output.
template <typename XT, int NJ, typename OT>
__global__ __launch_bounds__(R16_THREADS, 4) void layernorm16_kernel(const XT* __restrict__ x, const float* __restrict__ g,
                                                                     const float* __restrict__ b, float eps, int64_t T,
                                                                     int d, OT* __restrict__ out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, u = lane & 15;
  const int nchunk = d >> 2;
  const int64_t slot_gid = ((int64_t)blockIdx.x * (R16_THREADS / 64) + wave) * 4 + q;
  const int64_t slot_stride = (int64_t)gridDim.x * (R16_THREADS / 64) * 4;
  for (int64_t it0 = slot_gid - q; it0 < T; it0 += slot_stride) {
    const int64_t t = it0 + q;
    const bool live = t < T;
    float xv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = u + 16 * j;
      if (live && c < nchunk) load4(x + t * (int64_t)d + c * 4, xv[j]);
      else xv[j][0] = xv[j][1] = xv[j][2] = xv[j][3] = 0.f;
    }
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s1 += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
    const float mean = row16_sum(s1) / (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = u + 16 * j;
      if (c < nchunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dv = xv[j][i] - mean; s2 = fmaf(dv, dv, s2); }
      }
    }
    const float rstd = rsqrtf(row16_sum(s2) / (float)d + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = u + 16 * j;
      if (live && c < nchunk) {
        // gamma / beta come from L1/L2 at use (6 KB, shared by every wave): holding them in registers would cost
        // 96 VGPRs and spill
        const f32x4 gg = g ? *reinterpret_cast<const f32x4*>(g + c * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 bb = b ? *reinterpret_cast<const f32x4*>(b + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaf((xv[j][i] - mean) * rstd, gg[i], bb[i]);
        OT* dst = out + t * (int64_t)d + c * 4;
        if constexpr (std::is_same<OT, float>::value) {
          *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        } else if constexpr (std::is_same<OT, f16>::value) {
          f16x4 v; v[0] = (f16)o[0]; v[1] = (f16)o[1]; v[2] = (f16)o[2]; v[3] = (f16)o[3];
          *reinterpret_cast<f16x4*>(dst) = v;
        } else {
          s16x4 v; v[0] = (short)f32_to_bf16(o[0]); v[1] = (short)f32_to_bf16(o[1]); v[2] = (short)f32_to_bf16(o[2]); v[3] = (short)f32_to_bf16(o[3]);
          *reinterpret_cast<s16x4*>(dst) = v;
        }
      }
    }
  }
}

template <typename XT, typename OT>
int ln_dispatch_nj(const void* x, const float* g, const float* b, float eps, int64_t T, int d, void* out, hipStream_t s) {
  // one 16-token group per workgroup (no loop): the dispatcher back-fills CUs as groups retire, so there is no
  // 1-vs-2-iteration imbalance between resident workgroups
  int64_t need = (T + 15) / 16;
  const int grid = (int)(need < 1 ? 1 : (need > (1 << 20) ? (1 << 20) : need));
#define LN_LAUNCH(NJ) hipLaunchKernelGGL((layernorm16_kernel<XT, NJ, OT>), dim3(grid), dim3(R16_THREADS), 0, s, (const XT*)x, g, b, eps, T, d, (OT*)out)
  switch (d) {
    case 192: LN_LAUNCH(3); break;
    case 384: LN_LAUNCH(6); break;
    case 768: LN_LAUNCH(12); break;
    case 1024: LN_LAUNCH(16); break;
    default: smoe_set_error("smoe_layernorm: unsupported d=%d", d); return 1;
  }
#undef LN_LAUNCH
  SMOE_CHECK_LAUNCH("smoe_layernorm");
  return 0;
}

struct LnArgs {
  const float* g; const float* b; float eps; void* xn16; int xn16_dtype; float* xn32; bool on;
};

template <typename XT, int NJ, bool LN, typename NT, int EB>
int launch16(const void* x, const LnArgs& ln, const float* wg, const float* bg, const float* noise, int64_t T, int d,
             int E, int k, int gate_kind, int force_f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score,
             float* logits_out, float* probs, hipStream_t s) {
  const size_t smem = ((size_t)EB * d + 2 * EB + (LN ? 2 * (size_t)d : 0)) * 4;
  const int64_t tok_per_block = (R16_THREADS / 64) * 4;
  int64_t need = (T + tok_per_block - 1) / tok_per_block;
  // <= 768 workgroups (3 resident per CU; 512 = 2 per CU for the 16-expert image: the LDS weight image is loaded once
  // per workgroup), every workgroup the same number of 16-token groups
  constexpr int64_t max_wg = EB <= 8 ? 768 : (EB <= 16 ? 512 : 256);
  const int64_t iters = (need + max_wg - 1) / max_wg;
  const int grid = (int)(need < 1 ? 1 : (need + iters - 1) / (iters < 1 ? 1 : iters));
  if (smem > 64 * 1024) {  // 16 experts x d 1024 (+ LayerNorm vectors): above the default dynamic-LDS limit
    SMOE_ENSURE_SMEM(router16_kernel<XT, NJ, 0, LN, NT, EB>);
    SMOE_ENSURE_SMEM(router16_kernel<XT, NJ, 1, LN, NT, EB>);
  }
#define R16_LAUNCH(MODE, GRID, RC, RL)                                                                               \
  hipLaunchKernelGGL((router16_kernel<XT, NJ, MODE, LN, NT, EB>), dim3(GRID), dim3(R16_THREADS), smem, s, (const XT*)x,  \
                     ln.g, ln.b, ln.eps, (NT*)ln.xn16, ln.xn32, wg, bg, noise, T, d, E, k, gate_kind, RC, RL, idx,   \
                     score, logits_out, probs)
  if (force_f64 && !LN) {
    R16_LAUNCH(1, grid, nullptr, nullptr);
    SMOE_CHECK_LAUNCH("smoe_router_topk/f64");
    return 0;
  }
  hipError_t me = hipMemsetAsync(rc, 0, 16, s);
  if (me != hipSuccess) {
    smoe_set_error("smoe_router_topk: memset failed: %s", hipGetErrorString(me));
    return (int)me;
  }
  if (force_f64 && LN) {  // f64 mode still needs the normalised rows written: run the f32 pass for its stores first
    R16_LAUNCH(0, grid, rc, rl);
    SMOE_CHECK_LAUNCH("smoe_router_topk/f32");
    R16_LAUNCH(1, grid, nullptr, nullptr);
    SMOE_CHECK_LAUNCH("smoe_router_topk/f64");
    return 0;
  }
  R16_LAUNCH(0, grid, rc, rl);
  SMOE_CHECK_LAUNCH("smoe_router_topk/f32");
  R16_LAUNCH(1, (grid < 16 ? grid : 16), rc, rl);
  SMOE_CHECK_LAUNCH("smoe_router_topk/redo");
#undef R16_LAUNCH
  return 0;
}

template <typename XT, bool LN, typename NT>
int dispatch16(const void* x, const LnArgs& ln, const float* wg, const float* bg, const float* noise, int64_t T, int d,
               int E, int k, int gate_kind, int f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score, float* lo,
               float* pr, hipStream_t s) {
  if (E > 16) {  // 32 experts per lane (one workgroup per CU: the f32 weight image is 96-128 KB); ViT-B / ViT-L widths
    switch (d) {
      case 768: return launch16<XT, 12, LN, NT, 32>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
      case 1024: return launch16<XT, 16, LN, NT, 32>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
    }
    return -1;
  }
  if (E > 8) {  // 16 experts per lane: instantiated for the ViT-B / ViT-L widths only (compile time)
    switch (d) {
      case 768: return launch16<XT, 12, LN, NT, 16>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
      case 1024: return launch16<XT, 16, LN, NT, 16>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
    }
    return -1;
  }
  switch (d) {
    case 192: return launch16<XT, 3, LN, NT, 8>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
    case 384: return launch16<XT, 6, LN, NT, 8>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
    case 768: return launch16<XT, 12, LN, NT, 8>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
    case 1024: return launch16<XT, 16, LN, NT, 8>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
  }
  return -1;
}

template <typename XT>
int dispatch16_ln(const void* x, const LnArgs& ln, const float* wg, const float* bg, const float* noise, int64_t T,
                  int d, int E, int k, int gate_kind, int f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score,
                  float* lo, float* pr, hipStream_t s) {
  if (!ln.on) return dispatch16<XT, false, f16>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
  if (ln.xn16_dtype == SMOE_BF16)
    return dispatch16<XT, true, bf16_bits>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
  return dispatch16<XT, true, f16>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
}

}  // namespace

static bool shape_ok16(int d, int E, int k) {
  if (k > R16_MAX_K) return false;
  if (E <= 8) return d == 192 || d == 384 || d == 768 || d == 1024;
  return E <= 32 && (d == 768 || d == 1024);
}

// returns -1 when the shape is not covered by this fast path (caller falls back to router.hip)
int smoe_router16_try(const void* x, int x_dtype, const float* wg, const float* bg, const float* noise, int64_t T,
                      int d, int E, int k, int gate_kind, int force_f64, int32_t* rc, int32_t* rl, int64_t* idx,
                      float* score, float* logits_out, float* probs, hipStream_t s) {
  if (!shape_ok16(d, E, k)) return -1;
  LnArgs ln{nullptr, nullptr, 0.f, nullptr, SMOE_F16, nullptr, false};
  switch (x_dtype) {
    case SMOE_F32: return dispatch16_ln<float>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
    case SMOE_F16: return dispatch16_ln<f16>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
    case SMOE_BF16: return dispatch16_ln<bf16_bits>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
  }
  return -1;
}

extern "C" int smoe_ln_router_supported(int d, int E, int k) { return shape_ok16(d, E, k) ? 1 : 0; }

// LayerNorm + router in one pass over x (block glue fusion): xn = LN(x) * gamma + beta is
// written as the 16-bit operand image (xn16, f16 or bf16; may be NULL) and / or as f32 (xn32; may be NULL) and
// routed exactly like smoe_router_topk routes xn.  Shapes: smoe_ln_router_supported(d, E, k).
extern "C" int smoe_ln_router_topk(const void* x, int x_dtype, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                   void* xn16, int xn16_dtype, float* xn32, const float* wg, const float* bg,
                                   const float* noise, int64_t T, int d, int E, int k, int gate_kind, int64_t* idx,
                                   float* score, float* logits_out, float* probs, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  const int force_f64 = (gate_kind & 0x100) ? 1 : 0;
  gate_kind &= 0xff;
  if (T == 0) return 0;
  SMOE_REQUIRE(x && wg && idx && score, "smoe_ln_router_topk: null pointer");
  SMOE_REQUIRE(shape_ok16(d, E, k), "smoe_ln_router_topk: unsupported shape d=%d E=%d k=%d", d, E, k);
  SMOE_REQUIRE(T >= 0 && T < (1ll << 31), "smoe_ln_router_topk: bad T");
  SMOE_REQUIRE(k >= 1 && k <= E, "smoe_ln_router_topk: bad k");
  SMOE_REQUIRE(gate_kind == SMOE_GATE_NAIVE || (gate_kind == SMOE_GATE_SWITCH && k == 1), "smoe_ln_router_topk: bad gate");
  SMOE_REQUIRE(xn16_dtype == SMOE_F16 || xn16_dtype == SMOE_BF16, "smoe_ln_router_topk: xn16 must be f16 or bf16");
  SMOE_REQUIRE(workspace && workspace_bytes >= 16 + (((size_t)T * 4 + 15) & ~(size_t)15), "smoe_ln_router_topk: workspace too small");
  if (T == 0) return 0;
  int32_t* rc = reinterpret_cast<int32_t*>(workspace);
  int32_t* rl = reinterpret_cast<int32_t*>((char*)workspace + 16);
  LnArgs ln{ln_gamma, ln_beta, ln_eps, xn16, xn16_dtype, xn32, true};
  hipStream_t s = (hipStream_t)stream;
  switch (x_dtype) {
    case SMOE_F32: return dispatch16_ln<float>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
    case SMOE_F16: return dispatch16_ln<f16>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
    case SMOE_BF16: return dispatch16_ln<bf16_bits>(x, ln, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
  }
  smoe_set_error("smoe_ln_router_topk: bad x_dtype %d", x_dtype);
  return 1;
}

// LayerNorm over the last dimension, d in {192, 384, 768, 1024}; out dtype f32 / f16 / bf16.
extern "C" int smoe_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, float eps, int64_t T,
                              int d, void* out, int out_dtype, void* stream) {
  SMOE_REQUIRE(T >= 0 && (d == 192 || d == 384 || d == 768 || d == 1024), "smoe_layernorm: unsupported shape T=%lld d=%d", (long long)T, d);
  if (T == 0) return 0;
  SMOE_REQUIRE(x && out, "smoe_layernorm: null pointer");
  hipStream_t s = (hipStream_t)stream;
#define LN_OUT(XT)                                                                                   \
  switch (out_dtype) {                                                                               \
    case SMOE_F32: return ln_dispatch_nj<XT, float>(x, gamma, beta, eps, T, d, out, s);             \
    case SMOE_F16: return ln_dispatch_nj<XT, f16>(x, gamma, beta, eps, T, d, out, s);               \
    case SMOE_BF16: return ln_dispatch_nj<XT, bf16_bits>(x, gamma, beta, eps, T, d, out, s);        \
  }
  switch (x_dtype) {
    case SMOE_F32: LN_OUT(float) break;
    case SMOE_F16: LN_OUT(f16) break;
    case SMOE_BF16: LN_OUT(bf16_bits) break;
  }
#undef LN_OUT
  smoe_set_error("smoe_layernorm: bad dtype");
  return 1;
}
