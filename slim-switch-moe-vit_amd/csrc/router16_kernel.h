// Router / LayerNorm / token-skip-gate kernel in the 16-lanes-per-token layout (shared by router16.hip and gate.hip).
//
// Four tokens per wave: lane = 16 q + u handles the float4 chunks u, u+16, u+32, ... of token slot q (every load
// instruction covers 4 x 256 contiguous bytes).  Per-token reductions are 4 DPP-modified adds inside a 16-lane
// DPP row (quad_perm xor-1, xor-2, row_ror 4, row_ror 8) -- plain VALU, no LDS round trips, no ds_bpermute --
// after which every lane of the row holds all E logits in registers and the top-(k+1) selection is a short
// unrolled compare chain.  Weights sit in LDS as f32; the four token slots read the same addresses (broadcast).
// Contract (as router.hip): decisions are DEFINED on f64-accumulated values.  The f32 pass (MODE 0) carries a
// rigorous error bound; a token whose deciding gap falls inside the bound goes to the redo list and is recomputed
// with f64 accumulation (MODE 1, same layout, hence bit-identical inputs).
//
// GATE (token-skip gate of the reference's residual-MoE block, models/resMoE.py:32-85 `Gate` + 126-145):
//   z = <row, gate_w> + gate_b;  the token is SKIPPED iff sigmoid(z) > threshold  <=>  z > logit(threshold).
//   A skipped token enters the following operator as an all-zero row (resMoE.py:133-136 `x * mask`), so
//     - its 16-bit operand image row is written as zeros,
//     - its router logits are the gate biases (a zero row's logits), i.e. it routes like every other zero row, and
//       idx_plan gets -1: the row is not dispatched, its expert output is the per-layer constant `zero_out`
//       (= sum_j score_j (W2[e_j] gelu(b1[e_j]) + b2[e_j]), smoe_zero_row_output) added to its f32 image here,
//     - the f32 image row (the residual the reference takes from the NORMED activations, `+ tk + skip_tk`) is
//       written for every token.
//   GATE = 1: gate + router;  GATE = 2: gate only (the attention half: no experts, no routing outputs).
#pragma once
#include "smoe_common.h"
#include <type_traits>

namespace r16 {

constexpr int R16_THREADS = 256;
constexpr int R16_MAX_K = 4;

struct SkipGateArgs {
  const float* w;          // [d] f32 gate weight (head.1.weight)
  const float* b;          // [1] f32 gate bias, or NULL
  const float* thr;        // [1] f32 threshold in DEVICE memory (the module's buffer); NULL = gate disabled (all pass)
  int32_t* skip_count;     // [1] += number of skipped tokens, or NULL
  float* mask;             // [T,2] f32 (skip, keep), or NULL
  const float* zero_out;   // [d] f32 added to the f32 image of skipped rows, or NULL
  int64_t* idx_plan;       // [T,k] idx, -1 for skipped tokens; NULL when there is no router
  float* tk32;             // [T,d] f32: the row, zeros for a skipped token (the training path's masked operand image), or NULL
};

// Chunk histogram for the dispatch plan (smoe_dispatch_plan_hist): hist[c][e] = dispatched entries of expert e among the tokens
// [c * tok, (c + 1) * tok).  The f32 pass then walks CONTIGUOUS chunks (workgroup b = chunk b) and counts in LDS what it decides
// itself; the redo pass adds the tokens it decides (a handful, global atomics).  The plan's own counting launch and its pass
// over idx go away.
constexpr int R16_HIST_TOK = 64;
struct HistArgs {
  int32_t* hist;           // [ceil(T / tok)][E] i32, or NULL
  int tok;                 // tokens per chunk (R16_HIST_TOK)
};

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

template <typename NT> __device__ __forceinline__ void store4_16(NT* dst, const f32x4& v) {
  if constexpr (std::is_same<NT, f16>::value) {
    f16x4 o; o[0] = (f16)v[0]; o[1] = (f16)v[1]; o[2] = (f16)v[2]; o[3] = (f16)v[3];
    *reinterpret_cast<f16x4*>(dst) = o;
  } else {
    s16x4 o; o[0] = (short)f32_to_bf16(v[0]); o[1] = (short)f32_to_bf16(v[1]);
    o[2] = (short)f32_to_bf16(v[2]); o[3] = (short)f32_to_bf16(v[3]);
    *reinterpret_cast<s16x4*>(dst) = o;
  }
}

// NTH = threads per workgroup: 256, or 512 for the 16- / 32-expert images (their f32 weight image fills most of the LDS, so
// only one or two workgroups fit a CU: twice the waves behind the same image doubles what is in flight per CU).
// LN = fused LayerNorm in front (models/vision_transformer.py:321 `mlp(norm2(x))`, resMoE.py:126 / 137): the row is
// normalised in registers (two-pass mean / variance over the 16-lane row, f32), written once as the 16-bit
// operand image the GEMMs read (xn16) and optionally as f32 (xn32), and routed / gated on its f32 value.
// The f64 redo pass recomputes the same normalisation with the same lane layout, hence bit-identical inputs.
template <typename XT, int NJ, int MODE, bool LN, typename NT, int EB, int GATE = 0, int NTH = R16_THREADS>
__global__ __launch_bounds__(NTH, (NTH / R16_THREADS) * (MODE == 0 ? (EB <= 8 ? (GATE != 0 ? 3 : 4) : (EB <= 16 ? 2 : 1)) : (EB <= 16 ? 2 : 1))) void router16_kernel(
    const XT* __restrict__ x, const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    NT* __restrict__ xn16, float* __restrict__ xn32, const float* __restrict__ wg, const float* __restrict__ bg,
    const float* __restrict__ noise, int64_t T, int d, int E, int k, int gate_kind,
    int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_list, int64_t* __restrict__ idx_out,
    float* __restrict__ score_out, float* __restrict__ logits_out, float* __restrict__ probs_out, SkipGateArgs ga,
    HistArgs ha) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool ROUTE = GATE != 2;
  constexpr int EW = ROUTE ? EB : 0;                        // expert rows held in LDS
  d = 64 * NJ;  // the launcher only dispatches exact multiples: makes every chunk bound below compile-time
  float* lds_w = reinterpret_cast<float*>(smem);            // [EW][d], rows >= E zero
  float* lds_wn2 = lds_w + EW * d;                          // [EB]
  float* lds_bias = lds_wn2 + EB;                           // [EB] gate bias, zero where absent (branch-free add)
  float* lds_g = lds_bias + EB;                             // [d] LayerNorm weight, then [d] bias (LN only)
  float* lds_be = lds_g + (LN ? d : 0);
  float* lds_gw = lds_be + (LN ? d : 0);                    // [d] skip-gate weight (GATE only)
  double* lds_gz = reinterpret_cast<double*>(lds_gw + (GATE ? d : 0));  // [0] logit(threshold)  [1] |gate_w|^2  [2] gate bias
  int* lds_hist = reinterpret_cast<int*>(lds_gz + (GATE ? 4 : 0));       // [EB] this chunk's histogram (ha.hist only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, u = lane & 15;
  if (MODE == 1 && redo_list && *redo_count == 0) return;  // nothing to redo (the common case): exit before any setup

  int64_t n_items = T;
  if (MODE == 1 && redo_list) {  // a trip count read from device memory is never trusted: at most T tokens can be listed
    n_items = *redo_count;
    n_items = n_items < 0 ? 0 : (n_items > T ? T : n_items);
  }
  // tokens of this workgroup: 16-token groups strided over the grid -- or, with a chunk histogram (f32 pass), ONE contiguous
  // chunk of ha.tok tokens
  const bool chunked = MODE == 0 && ROUTE && ha.hist != nullptr;
  const int64_t slot_gid = chunked ? (int64_t)blockIdx.x * ha.tok + wave * 4 + q : ((int64_t)blockIdx.x * (NTH / 64) + wave) * 4 + q;
  const int64_t slot_stride = chunked ? (int64_t)(NTH / 64) * 4 : (int64_t)gridDim.x * (NTH / 64) * 4;
  if (chunked) {
    const int64_t chunk_end = ((int64_t)blockIdx.x + 1) * ha.tok;
    if (chunk_end < n_items) n_items = chunk_end;
  }
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

  if constexpr (ROUTE) {
    // weight image -> LDS, eight 16-byte loads in flight per thread (one load per trip made the staging a chain of
    // dependent L2 round trips: 16 of them for the 32-expert image of a 512-thread workgroup)
    constexpr int WU = 8;
    for (int base = tid * 4; base < EB * d; base += NTH * 4 * WU) {
      f32x4 v[WU];
#pragma unroll
      for (int q8 = 0; q8 < WU; ++q8) {
        const int i = base + q8 * NTH * 4;
        v[q8] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < EB * d && i / d < E) v[q8] = *reinterpret_cast<const f32x4*>(wg + i);
      }
#pragma unroll
      for (int q8 = 0; q8 < WU; ++q8) {
        const int i = base + q8 * NTH * 4;
        if (i < EB * d) *reinterpret_cast<f32x4*>(lds_w + i) = v[q8];
      }
    }
    if (tid < EB) lds_bias[tid] = (bg && tid < E) ? bg[tid] : 0.f;
  }
  if (LN) {
    for (int i = tid; i < d; i += NTH) {
      lds_g[i] = ln_g ? ln_g[i] : 1.f;
      lds_be[i] = ln_b ? ln_b[i] : 0.f;
    }
  }
  if constexpr (GATE != 0) {
    for (int i = tid; i < d; i += NTH) lds_gw[i] = ga.w[i];
  }
  if constexpr (ROUTE) {
    if (tid < EB) lds_hist[tid] = 0;
  }
  __syncthreads();
  if constexpr (ROUTE) {
    for (int e = wave; MODE == 0 && e < EB; e += NTH / 64) {  // squared row norms, one wave per expert row
      float s = 0.f;
      for (int c = lane; c < d; c += 64) s = fmaf(lds_w[e * d + c], lds_w[e * d + c], s);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
      if (lane == 0) lds_wn2[e] = s;
    }
  }
  if constexpr (GATE != 0) {
    if (wave == NTH / 64 - 1) {  // the last wave: |gate_w|^2, logit(threshold), bias
      float s = 0.f;
      for (int c = lane; c < d; c += 64) s = fmaf(lds_gw[c], lds_gw[c], s);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
      if (lane == 0) {
        double zt = __builtin_huge_val();  // disabled gate: nothing exceeds +inf
        if (ga.thr) {
          const double th = (double)*ga.thr;
          zt = th >= 1.0 ? __builtin_huge_val() : (th <= 0.0 ? -__builtin_huge_val() : log(th / (1.0 - th)));
        }
        lds_gz[0] = zt;
        lds_gz[1] = (double)s;
        lds_gz[2] = ga.b ? (double)*ga.b : 0.0;
      }
    }
  }
  __syncthreads();
  float wmax2 = 0.f;
  if constexpr (ROUTE) {
#pragma unroll
    for (int e = 0; e < EB; ++e) wmax2 = fmaxf(wmax2, lds_wn2[e]);
  }
  double g_zthr = 0.0;
  float g_wn2 = 0.f, g_bias = 0.f;
  if constexpr (GATE != 0) {
    g_zthr = lds_gz[0];
    g_wn2 = (float)lds_gz[1];
    g_bias = (float)lds_gz[2];
  }

  int my_skips = 0;  // skipped tokens this wave decided (GATE): ONE global atomic per workgroup at the end -- thousands of
                     // waves adding to the single counter word one by one serialise at the memory side (~100 us per launch)
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
    }
    float xs = 0.f;  // |row|^2 for the error bounds (MODE 0); without a gate it is computed after the logits (the
                     // last use of the row registers: one live register fewer through the FMA chains)
    if constexpr (MODE == 0 && GATE != 0) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) xs = fmaf(xv[j][i], xv[j][i], xs);
      xs = row16_sum(xs);
    }
    // ---- token-skip gate: z = <row, gate_w> + b; skipped iff z > logit(threshold) ---------------------------------
    bool skip = false, gate_ambiguous = false;
    if constexpr (GATE != 0) {
      if constexpr (MODE == 0) {
        f32x2 ag = f32x2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(lds_gw + ub + 64 * j);
          ag = __builtin_elementwise_fma(lo2(xv[j]), lo2(w), ag);
          ag = __builtin_elementwise_fma(hi2(xv[j]), hi2(w), ag);
        }
        const float z = row16_sum(ag[0] + ag[1]) + g_bias;
        const float bound = 4.0f * (float)(NJ * 4 + 6) * 5.9604645e-8f * sqrtf(xs * g_wn2) + 9.6e-7f * (fabsf(z) + 1.0f);
        const double gap = (double)z - g_zthr;
        skip = gap > 0.0;
        gate_ambiguous = !(fabs(gap) > (double)bound);  // +-inf thresholds are never ambiguous
      } else {
        double ag = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(lds_gw + ub + 64 * j);
#pragma unroll
          for (int i = 0; i < 4; ++i) ag = fma((double)xv[j][i], (double)w[i], ag);
        }
        skip = (row16_sum(ag) + lds_gz[2]) > g_zthr;
      }
    }
    // ---- operand images: the 16-bit row (zeros for a skipped token) and the f32 row (+ zero-row expert output) ------
    if constexpr (LN || GATE != 0) {
      const bool writer = live && (MODE == 0 || GATE != 0);  // the redo pass re-writes only what a gate decision changes
      if (writer) {
        if (xn32) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            f32x4 v = xv[j];
            if constexpr (GATE != 0) {
              if (ga.zero_out) {  // wave-uniform; the per-token part is a select (four tokens of a wave may differ)
                const f32x4 z = *reinterpret_cast<const f32x4*>(ga.zero_out + u * 4 + 64 * j);
                v += skip ? z : f32x4{0.f, 0.f, 0.f, 0.f};
              }
            }
            *reinterpret_cast<f32x4*>(xn32 + rowoff + 64 * j) = v;
          }
        }
        if (xn16) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            store4_16<NT>(xn16 + rowoff + 64 * j, (GATE != 0 && skip) ? f32x4{0.f, 0.f, 0.f, 0.f} : xv[j]);
        }
        if constexpr (GATE != 0) {
          if (ga.tk32) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              *reinterpret_cast<f32x4*>(ga.tk32 + rowoff + 64 * j) = skip ? f32x4{0.f, 0.f, 0.f, 0.f} : xv[j];
          }
        }
      }
    }
    if constexpr (!ROUTE) {
      // gate only: bookkeeping, then the next row
      it0 += slot_stride;
      const bool redo = (MODE == 0) && gate_ambiguous;
      if (redo && live && u == 0) list_push(redo_count, redo_list, T, t);
      if (live && u == 0 && ga.mask) {
        ga.mask[t * 2] = skip ? 1.f : 0.f;
        ga.mask[t * 2 + 1] = skip ? 0.f : 1.f;
      }
      my_skips += __popcll(__ballot(live && u == 0 && skip && !redo));
      if (it0 < n_items) fetch(it0);
      continue;
    } else {
    float lg[EB];
    if constexpr (MODE == 0) {
      // two partial sums per expert (even / odd element pairs): packed f32 FMAs, half the issue slots
      f32x2 acc[EB];
#pragma unroll
      for (int e = 0; e < EB; ++e) acc[e] = f32x2{0.f, 0.f};
      if constexpr (EB <= 8) {
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
      } else {
        // 16 / 32 experts: only one or two waves per SIMD fit (the weight image fills the LDS), so a chunk's reads followed by
        // its FMAs left both pipes idle most of the time (E = 32: LDS 25 % busy, VALU 28 %, waves waiting 66 % of their cycles).
        // Software pipeline over groups of 8 experts: the next group's eight 16-byte reads are in flight under the current
        // group's sixteen packed FMAs (two register sets of 8 x 4).
        constexpr int G = 8, GPJ = EB / G, NSTEP = NJ * GPJ;
        f32x4 wq[2][G];
        auto ldw = [&](int step, f32x4 (&dst)[G]) {
          const int j = step / GPJ, e0 = (step % GPJ) * G;
#pragma unroll
          for (int g8 = 0; g8 < G; ++g8) dst[g8] = *reinterpret_cast<const f32x4*>(lds_w + ub + ((e0 + g8) * d + 64 * j));
        };
        ldw(0, wq[0]);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
          const int j = step / GPJ, e0 = (step % GPJ) * G;
          if (step + 1 < NSTEP) ldw(step + 1, wq[(step + 1) & 1]);
#pragma unroll
          for (int g8 = 0; g8 < G; ++g8) {
            acc[e0 + g8] = __builtin_elementwise_fma(lo2(xv[j]), lo2(wq[step & 1][g8]), acc[e0 + g8]);
            acc[e0 + g8] = __builtin_elementwise_fma(hi2(xv[j]), hi2(wq[step & 1][g8]), acc[e0 + g8]);
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the steps in order: reads of step + 1, then the FMAs of step
        }
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
    if constexpr (MODE == 0 && GATE == 0) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) xs = fmaf(xv[j][i], xv[j][i], xs);
      xs = row16_sum(xs);
    }
    if constexpr (GATE != 0) {
      if (skip) {  // a skipped token is an all-zero row for the router: its logits are the biases, exactly
#pragma unroll
        for (int e = 0; e < EB; ++e) lg[e] = lds_bias[e];
        xs = 0.f;
      }
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
    bool redo = false;
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
      // an all-zero row (a skipped token, or a genuinely zero input) has logits that ARE the biases -- no arithmetic, no
      // rounding: a tie between biases is resolved by the lowest id here exactly as the f64 pass would resolve it
      redo = (ambiguous && xs > 0.f) || gate_ambiguous;
      if (redo && live && u == 0) list_push(redo_count, redo_list, T, t);
    }
    if (ha.hist && live && u == 0 && !redo && !(GATE != 0 && skip)) {   // (a redo token is counted by the pass that decides it)
#pragma unroll
      for (int r = 0; r < R16_MAX_K; ++r)
        if (r < k) {
          if constexpr (MODE == 0) atomicAdd(&lds_hist[chosen[r]], 1);
          else atomicAdd(&ha.hist[(t / ha.tok) * (int64_t)E + chosen[r]], 1);
        }
    }
    if constexpr (GATE != 0) {
      if (live && u == 0 && ga.mask) {
        ga.mask[t * 2] = skip ? 1.f : 0.f;
        ga.mask[t * 2 + 1] = skip ? 0.f : 1.f;
      }
      // tokens handed to the redo pass are counted there, by their final decision
      my_skips += __popcll(__ballot(live && u == 0 && skip && !redo));
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
            if (GATE != 0 && ga.idx_plan) ga.idx_plan[t * (int64_t)k + r] = skip ? (int64_t)-1 : (int64_t)chosen[r];
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
        if (GATE != 0 && ga.idx_plan) ga.idx_plan[t] = skip ? (int64_t)-1 : (int64_t)chosen[0];
      }
    }
    }  // ROUTE
  }
  if constexpr (ROUTE && MODE == 0) {
    if (ha.hist) {         // kernel-uniform: this chunk's row of the table
      __syncthreads();
      if (tid < E) ha.hist[(int64_t)blockIdx.x * E + tid] = lds_hist[tid];
    }
  }
  if constexpr (GATE != 0) {
    if (ga.skip_count) {   // kernel-uniform
      __syncthreads();     // the weight image in LDS is dead from here on: reuse its first word
      int* wg_cnt = reinterpret_cast<int*>(smem);
      if (tid == 0) *wg_cnt = 0;
      __syncthreads();
      if (lane == 0 && my_skips) atomicAdd(wg_cnt, my_skips);
      __syncthreads();
      if (tid == 0 && *wg_cnt) atomicAdd(ga.skip_count, *wg_cnt);
    }
  }
  // The redo pass leaves the counter words as it needs to find them next time: the workgroup that finishes last (every
  // workgroup has read the count long before it finishes) clears the count and the arrival word, so a caller that keeps the
  // workspace (ops: one per device and stream) never launches a clearing kernel (gate_kind | 0x200 says so).
  if (MODE == 1 && redo_list) {
    __syncthreads();
    if (tid == 0) {
      const int old = atomicAdd(&redo_count[1], 1);
      if (old == (int)gridDim.x - 1) {
        redo_count[1] = 0;
        redo_count[0] = 0;
      }
    }
  }
}

// dynamic LDS of one workgroup of router16_kernel
template <int NJ, bool LN, int EB, int GATE> constexpr size_t router16_smem() {
  constexpr int d = 64 * NJ;
  return ((size_t)(GATE != 2 ? EB : 0) * d + 2 * EB + (LN ? 2 * (size_t)d : 0) + (GATE != 0 ? (size_t)d : 0)) * 4 + (GATE != 0 ? 32 : 0)
         + (GATE != 2 ? (size_t)EB * 4 : 0);
}

}  // namespace r16
