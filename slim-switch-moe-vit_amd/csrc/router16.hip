// Router, 16-lanes-per-token layout (E <= 8, d in {192, 384, 768, 1024}): the fast path of smoe_router_topk.
//
// Four tokens per wave: lane = 16 q + u handles the float4 chunks u, u+16, u+32, ... of token slot q (every load
// instruction covers 4 x 256 contiguous bytes).  Per-token reductions are 4 DPP-modified adds inside a 16-lane
// DPP row (quad_perm xor-1, xor-2, row_ror 4, row_ror 8) -- plain VALU, no LDS round trips, no ds_bpermute --
// after which every lane of the row holds all E logits in registers and the top-(k+1) selection is a short
// unrolled compare chain.  Weights sit in LDS as f32; the four token slots read the same addresses (broadcast).
// Same contract as router.hip: f32 logits with a rigorous error bound, tokens whose deciding gaps fall inside the
// bound go to the redo list and are recomputed with f64 accumulation (MODE 1, same layout).
#include "smoe_common.h"

namespace {

constexpr int R16_THREADS = 256;
constexpr int R16_MAX_K = 4;
constexpr int R16_E = 8;

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

template <typename XT, int NJ, int MODE>
__global__ __launch_bounds__(R16_THREADS, (MODE == 0 ? 3 : 2)) void router16_kernel(
    const XT* __restrict__ x, const float* __restrict__ wg, const float* __restrict__ bg,
    const float* __restrict__ noise, int64_t T, int d, int E, int k, int gate_kind,
    int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_list, int64_t* __restrict__ idx_out,
    float* __restrict__ score_out, float* __restrict__ logits_out, float* __restrict__ probs_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_w = reinterpret_cast<float*>(smem);            // [R16_E][d], rows >= E zero
  float* lds_wn2 = lds_w + R16_E * d;                        // [R16_E]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, u = lane & 15;

  for (int i = tid * 4; i < R16_E * d; i += R16_THREADS * 4) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i / d < E) v = *reinterpret_cast<const f32x4*>(wg + i);
    *reinterpret_cast<f32x4*>(lds_w + i) = v;
  }
  __syncthreads();
  if (tid < R16_E) {
    float s = 0.f;
    for (int c = 0; c < d; ++c) s = fmaf(lds_w[tid * d + c], lds_w[tid * d + c], s);
    lds_wn2[tid] = s;
  }
  __syncthreads();
  float wmax2 = 0.f;
#pragma unroll
  for (int e = 0; e < R16_E; ++e) wmax2 = fmaxf(wmax2, lds_wn2[e]);

  const int nchunk = d >> 2;  // float4 chunks per row
  int64_t n_items = T;
  if (MODE == 1 && redo_list) n_items = *redo_count;
  const int64_t slot_gid = ((int64_t)blockIdx.x * (R16_THREADS / 64) + wave) * 4 + q;
  const int64_t slot_stride = (int64_t)gridDim.x * (R16_THREADS / 64) * 4;
  // all four slots of a wave iterate together (DPP needs the whole row active): loop on the wave's first slot
  for (int64_t it0 = slot_gid - q; it0 < n_items; it0 += slot_stride) {
    const int64_t it = it0 + q;
    const bool live = it < n_items;
    const int64_t t = live ? ((MODE == 1 && redo_list) ? (int64_t)redo_list[it] : it) : 0;
    float xv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = u + 16 * j;
      if (live && c < nchunk) load4(x + t * (int64_t)d + c * 4, xv[j]);
      else xv[j][0] = xv[j][1] = xv[j][2] = xv[j][3] = 0.f;
    }
    float lg[R16_E];
    if constexpr (MODE == 0) {
      float acc[R16_E];
#pragma unroll
      for (int e = 0; e < R16_E; ++e) acc[e] = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = u + 16 * j;
        if (c < nchunk) {
#pragma unroll
          for (int e = 0; e < R16_E; ++e) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(lds_w + e * d + c * 4);
            acc[e] = fmaf(xv[j][0], w[0], acc[e]);
            acc[e] = fmaf(xv[j][1], w[1], acc[e]);
            acc[e] = fmaf(xv[j][2], w[2], acc[e]);
            acc[e] = fmaf(xv[j][3], w[3], acc[e]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < R16_E; ++e) lg[e] = row16_sum(acc[e]) + ((bg && e < E) ? bg[e] : 0.f);
    } else {
      double acc[R16_E];
#pragma unroll
      for (int e = 0; e < R16_E; ++e) acc[e] = 0.0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = u + 16 * j;
        if (c < nchunk) {
#pragma unroll
          for (int e = 0; e < R16_E; ++e) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(lds_w + e * d + c * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[e] = fma((double)xv[j][i], (double)w[i], acc[e]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < R16_E; ++e) lg[e] = (float)(row16_sum(acc[e]) + ((bg && e < E) ? (double)bg[e] : 0.0));
    }
    if (logits_out && live && u == 0) {
#pragma unroll
      for (int e = 0; e < R16_E; ++e)
        if (e < E) logits_out[t * (int64_t)E + e] = lg[e];
    }
    if (gate_kind == SMOE_GATE_SWITCH && noise && live) {
#pragma unroll
      for (int e = 0; e < R16_E; ++e)
        if (e < E) lg[e] += noise[t * (int64_t)E + e];
    }
    // top-kc in registers: ties -> lowest id, descending value
    const int kc = (MODE == 0 && k < E) ? k + 1 : k;
    int chosen[R16_MAX_K + 1];
    float cval[R16_MAX_K + 1];
    unsigned taken = 0;
#pragma unroll
    for (int r = 0; r <= R16_MAX_K; ++r) {
      chosen[r] = 0;
      cval[r] = 0.f;
      if (r < kc) {
        float bv = -INFINITY;
        int bi = -1;
#pragma unroll
        for (int e = 0; e < R16_E; ++e) {
          const bool ok = (e < E) && !((taken >> e) & 1u);
          if (ok && (bi < 0 || lg[e] > bv)) { bv = lg[e]; bi = e; }
        }
        chosen[r] = bi;
        cval[r] = bv;
        taken |= 1u << bi;
      }
    }
    if constexpr (MODE == 0) {
      float xs = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) xs = fmaf(xv[j][i], xv[j][i], xs);
      xs = row16_sum(xs);
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
        float pe[R16_E];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < R16_E; ++e) {
          pe[e] = (e < E) ? expf(lg[e] - mx) : 0.f;
          s += pe[e];
        }
        if (probs_out) {
#pragma unroll
          for (int e = 0; e < R16_E; ++e)
            if (e < E) probs_out[t * (int64_t)E + e] = pe[e] / s;
        }
        idx_out[t] = chosen[0];
        score_out[t] = 1.0f / s;
      }
    }
  }
}

template <typename XT, int NJ>
int launch16(const void* x, const float* wg, const float* bg, const float* noise, int64_t T, int d, int E, int k,
             int gate_kind, int force_f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score, float* logits_out,
             float* probs, hipStream_t s) {
  const size_t smem = ((size_t)R16_E * d + R16_E) * 4;
  const int64_t tok_per_block = (R16_THREADS / 64) * 4;
  int64_t need = (T + tok_per_block - 1) / tok_per_block;
  const int grid = (int)(need < 2048 ? (need < 1 ? 1 : need) : 2048);
#define R16_LAUNCH(MODE, GRID, RC, RL)                                                                              \
  hipLaunchKernelGGL((router16_kernel<XT, NJ, MODE>), dim3(GRID), dim3(R16_THREADS), smem, s, (const XT*)x, wg, bg, \
                     noise, T, d, E, k, gate_kind, RC, RL, idx, score, logits_out, probs)
  if (force_f64) {
    R16_LAUNCH(1, grid, nullptr, nullptr);
    SMOE_CHECK_LAUNCH("smoe_router_topk/f64");
    return 0;
  }
  hipError_t me = hipMemsetAsync(rc, 0, 16, s);
  if (me != hipSuccess) {
    smoe_set_error("smoe_router_topk: memset failed: %s", hipGetErrorString(me));
    return (int)me;
  }
  R16_LAUNCH(0, grid, rc, rl);
  SMOE_CHECK_LAUNCH("smoe_router_topk/f32");
  R16_LAUNCH(1, (grid < 16 ? grid : 16), rc, rl);
  SMOE_CHECK_LAUNCH("smoe_router_topk/redo");
#undef R16_LAUNCH
  return 0;
}

template <typename XT>
int dispatch16(const void* x, const float* wg, const float* bg, const float* noise, int64_t T, int d, int E, int k,
               int gate_kind, int f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score, float* lo, float* pr,
               hipStream_t s) {
  switch (d) {
    case 192: return launch16<XT, 3>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
    case 384: return launch16<XT, 6>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
    case 768: return launch16<XT, 12>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
    case 1024: return launch16<XT, 16>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
  }
  return -1;
}

}  // namespace

// returns -1 when the shape is not covered by this fast path (caller falls back to router.hip)
int smoe_router16_try(const void* x, int x_dtype, const float* wg, const float* bg, const float* noise, int64_t T,
                      int d, int E, int k, int gate_kind, int force_f64, int32_t* rc, int32_t* rl, int64_t* idx,
                      float* score, float* logits_out, float* probs, hipStream_t s) {
  if (E > R16_E || k > R16_MAX_K || !(d == 192 || d == 384 || d == 768 || d == 1024)) return -1;
  switch (x_dtype) {
    case SMOE_F32: return dispatch16<float>(x, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
    case SMOE_F16: return dispatch16<f16>(x, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
    case SMOE_BF16: return dispatch16<bf16_bits>(x, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
  }
  return -1;
}
