// LayerNorm + router for 16 / 32 experts with the logits on the f32 matrix cores (router16.hip dispatches here for E > 8).
//
// Why a second layout: router16_kernel keeps a token on 16 lanes and reads every weight from LDS once per token --
// E x d x 4 bytes of LDS traffic per token with nothing to reuse it on.  Up to 8 experts that hides under the HBM time;
// at E = 32, d = 1024 it is 128 KB per token, the weight image leaves room for two waves per SIMD, and the pass ran at
// 96-140 us for 113 MB (LDS 25 % busy, VALU 28 %, waves waiting 66 % of their cycles).
//
// Here a workgroup of four waves takes a TILE of 16 tokens and wave w owns a quarter of the columns of all 16 rows: lane
// (ti = lane & 15, kk = lane >> 4) holds 8-float pieces at columns w d/4 + 32 m + 8 kk (+ 0..7) of token ti.  That is the
// operand layout of v_mfma_f32_16x16x4_f32 as it stands -- operand lane (row = lane & 15, k-slot = lane >> 4); a
// contraction does not care WHICH k a slot carries as long as both operands agree, so MFMA step (m, c) takes k = w d/4 +
// 32 m + 8 kk + c from both the row registers and the LDS weight image, 128 MFMAs per wave and tile at E = 32, d = 1024.
// One 16-byte LDS read feeds four MFMAs of 16 tokens x 16 experts: LDS traffic per token falls 10x.  Row statistics
// (LayerNorm mean / variance, |row|^2) and the partial logits of the four column quarters meet in LDS in a fixed order.
// A (expert) x B (token) puts a token on the lane and four consecutive experts in the accumulator registers: one 16-byte
// store per 16-expert block into the exchange area.  The tail (top-(k+1), error bound, softmax, stores) is router16's,
// with the wave's four tokens on four 16-lane rows.
//
// Contract as router16 / router.hip: decisions are DEFINED on f64-accumulated logits of the f32 row the kernel
// normalised.  MODE 0 = f32 MFMA pass with a rigorous bound (four interleaved accumulators: chains of 2 MP MFMA steps;
// the bound keeps router16's per-lane chain length 8 MP + 6, > 2.5x what the arithmetic needs); MODE 1 = f64 re-do over
// the redo list with the SAME load / LayerNorm stage, hence bit-identical rows.
#pragma once
#include "router16_kernel.h"

namespace rmt {

using r16::R16_MAX_K;
using r16::store4_16;

constexpr int MT_THREADS = 256;

// HV = 1: one four-wave workgroup, its exchange area double-buffered by the tile's parity.  HV = 2 (round 5; the E = 32 image leaves
// room for ONE workgroup per CU): two four-wave HALVES share the weight image, each walks tiles of its own through an exchange area
// of its own -- single-buffered, one more barrier per tile -- so that every SIMD holds TWO waves: one computes or waits for LDS while
// the other's loads are out (the medicine that took the attention backward from 199 to 163 us in round 4).
template <int MP, bool LN, int EB, int HV = 1> constexpr size_t router_mt_smem() {
  constexpr size_t d = 128 * MP;
  return ((size_t)EB * (d + 16) + 2 * EB + (LN ? 2 * d : 0) + HV * (2 * 64 + (HV == 1 ? 2 : 1) * 4 * 16 * (EB + 4))) * 4;
}

// sum over the four k-slot lanes of a token (lanes ti, ti + 16, ti + 32, ti + 48); every one of them gets the total
__device__ __forceinline__ float kk_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ double kk_sum(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// max / min over the 16 lanes of a DPP row; every lane of the row gets the result
template <int CTRL> __device__ __forceinline__ float dpp_maxf(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
  return fmaxf(v, __builtin_bit_cast(float, moved));
}
template <int CTRL> __device__ __forceinline__ int dpp_mini(int v) {
  const int moved = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
  return moved < v ? moved : v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = dpp_maxf<0xB1>(v); v = dpp_maxf<0x4E>(v); v = dpp_maxf<0x124>(v); v = dpp_maxf<0x128>(v);
  return v;
}
__device__ __forceinline__ int row16_min(int v) {
  v = dpp_mini<0xB1>(v); v = dpp_mini<0x4E>(v); v = dpp_mini<0x124>(v); v = dpp_mini<0x128>(v);
  return v;
}

template <typename XT, int MP, int MODE, bool LN, typename NT, int EB, int HV = 1>
__global__ __launch_bounds__(MT_THREADS * HV, 1) void router_mt_kernel(
    const XT* __restrict__ x, const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps,
    NT* __restrict__ xn16, float* __restrict__ xn32, const float* __restrict__ wg, const float* __restrict__ bg,
    const float* __restrict__ noise, int64_t T, int E, int k, int gate_kind, int32_t* __restrict__ redo_count,
    int32_t* __restrict__ redo_list, int64_t* __restrict__ idx_out, float* __restrict__ score_out,
    float* __restrict__ logits_out, float* __restrict__ probs_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int d = 128 * MP, dq = 32 * MP, WP = d + 16, NTL = EB / 16, PS = EB + 4, NACC = 4;
  constexpr int NPAR = HV == 1 ? 2 : 1;            // exchange buffers per half
  constexpr int NTHREADS = MT_THREADS * HV;
  static_assert(EB == 16 || EB == 32, "16 or 32 expert rows");
  static_assert(HV == 1 || MODE == 0, "the f64 re-do pass runs on one half");
  float* lds_w = reinterpret_cast<float*>(smem);   // [EB][WP = d + 16] (the 64-byte row pad and the XOR below make the fragment reads
                                                   // conflict-free), rows >= E repeat row E - 1 (masked by every consumer); 16-byte slot s of a 128-byte group of row e sits at s ^ (e & 7)
  float* lds_wn2 = lds_w + EB * WP;                // [EB]
  float* lds_bias = lds_wn2 + EB;                  // [EB]
  float* lds_g = lds_bias + EB;                    // [d] LayerNorm weight, [d] bias (LN only)
  float* lds_be = lds_g + (LN ? d : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, half = HV == 1 ? 0 : (tid >> 8), gwave = tid >> 6;
  constexpr int AREA = 2 * 64 + NPAR * 4 * 16 * PS;   // floats of one half's exchange area
  float* st1 = lds_be + (LN ? d : 0) + half * AREA; // [4 waves][16 tokens] row sums
  float* st2 = st1 + 64;                           // [4][16] centred sums of squares
  float* part = st2 + 64;                          // [NPAR parities][4 waves][16 tokens][PS]: partial logits, slot EB = |row|^2 part
  double* dpart = reinterpret_cast<double*>(part); // MODE 1: [4][16][EB] f64 partial logits (same bytes, one buffer)
  static_assert(MODE == 0 || 4 * 16 * EB * 8 <= 2 * 4 * 16 * PS * 4, "f64 partials fit the exchange area");
  const int ti = lane & 15, kk = lane >> 4;
  if (MODE == 1 && redo_list && *redo_count == 0) return;  // nothing to redo (the common case)

  int64_t n_items = T;
  if (MODE == 1 && redo_list) {  // a trip count read from device memory is never trusted
    n_items = *redo_count;
    n_items = n_items < 0 ? 0 : (n_items > T ? T : n_items);
  }
  const int64_t n_tiles = (n_items + 15) >> 4;
  auto token_of = [&](int64_t item, bool& live) -> int64_t {
    live = item < n_items;
    const int64_t itc = live ? item : n_items - 1;   // dead slots of the last tile re-read the last item (stores are guarded)
    int64_t t = (MODE == 1 && redo_list) ? (int64_t)redo_list[itc] : itc;
    if (MODE == 1) t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    return t;
  };
  // Two register sets: the rows of tile n + 1 are requested at the TOP of tile n and consumed a whole tile later -- with the
  // 128-KB weight image there is one wave per SIMD, nobody else to run while a load is out (a fetch issued after the MFMA
  // stage left ~2.5 us of HBM latency exposed per tile).  One wave per SIMD owns 512 registers, so the second set is free.
  f32x4 xv[MP][2], xnx[MP][2];
  auto fetch = [&](int64_t tile, f32x4 (&dst)[MP][2]) {
    bool lv;
    const int64_t t = token_of(tile * 16 + ti, lv);
    const XT* src = x + t * (int64_t)d + wave * dq + 8 * kk;
#pragma unroll
    for (int m = 0; m < MP; ++m) {
      float a[4], b[4];
      load4(src + 32 * m, a);
      load4(src + 32 * m + 4, b);
      dst[m][0] = f32x4{a[0], a[1], a[2], a[3]};
      dst[m][1] = f32x4{b[0], b[1], b[2], b[3]};
    }
  };
  // tiles: half h of workgroup b takes b HV + h, then every (gridDim.x HV)-th; BOTH halves run the same number of iterations (the
  // workgroup's barriers order all eight waves), a tile past the end is dead: clamped reads, no stores
  const int64_t tile0 = (int64_t)blockIdx.x * HV + half, tstep = (int64_t)gridDim.x * HV;
  const int64_t n_iter = (n_tiles - (int64_t)blockIdx.x * HV + tstep - 1) / tstep;   // of half 0; >= half 1's
  if (tile0 < n_tiles) fetch(tile0, xv);   // the first rows travel under the weight staging
  else fetch(n_tiles > 0 ? n_tiles - 1 : 0, xv);

  {  // weight image -> LDS by LDS-DMA (no registers, every piece in flight at once: staging through VGPRs was four dependent
     // rounds of loads, ~8 us per launch).  A wave-instruction fills 1 KiB of LDS linearly, so the SOURCE address carries the
     // layout: 16-byte unit L of the image = row e = L / (WP / 4), slot s; slot s holds source slot s ^ (e & 7) of its 8-slot
     // group.  Pad slots read any valid address (never read back); rows >= E repeat row E - 1 (every consumer masks e >= E).
    constexpr int UNITS_PER_ROW = WP / 4, N_KIB = EB * WP * 4 / 1024;
    static_assert(EB * WP * 4 % 1024 == 0, "whole 1-KiB pieces");
    const int wave_u = __builtin_amdgcn_readfirstlane(gwave);   // the DMA's LDS base is wave-uniform (M0)
    for (int pc = wave_u; pc < N_KIB; pc += NTHREADS / 64) {
      const int L = pc * 64 + lane;
      const int e = L / UNITS_PER_ROW, sl = L - e * UNITS_PER_ROW;
      const int ec = e < E ? e : E - 1;
      const int src_slot = sl < d / 4 ? ((sl & ~7) | ((sl & 7) ^ (e & 7))) : 0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (int64_t)ec * d + 4 * src_slot),
                                       (__attribute__((address_space(3))) void*)(smem + pc * 1024), 16, 0, 0);
    }
    if (tid < EB) lds_bias[tid] = (bg && tid < E) ? bg[tid] : 0.f;
    if (LN) {
      for (int i = tid; i < d; i += NTHREADS) {
        lds_g[i] = ln_g ? ln_g[i] : 1.f;
        lds_be[i] = ln_b ? ln_b[i] : 0.f;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces (and the first rows) have landed
  __syncthreads();
  for (int e = gwave; MODE == 0 && e < EB; e += NTHREADS / 64) {  // squared row norms (any column order): 16-byte reads, all in flight
    f32x4 q4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < d; c += 256) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(lds_w + e * WP + c + 4 * lane);
      q4 = __builtin_elementwise_fma(w, w, q4);
    }
    float s = (q4[0] + q4[1]) + (q4[2] + q4[3]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) lds_wn2[e] = s;
  }
  __syncthreads();
  float wmax2 = 0.f;
  if constexpr (MODE == 0) {
#pragma unroll
    for (int e = 0; e < EB; ++e) wmax2 = fmaxf(wmax2, lds_wn2[e]);
  }
  const int colw = wave * dq + 8 * kk;   // this lane's first column; piece m sits 32 m further

  int par = 0;
  int64_t tile = tile0;
  for (int64_t it = 0; it < n_iter; ++it, tile += tstep, par ^= (NPAR - 1)) {
    const bool tile_live = tile < n_tiles;
    bool live_l;
    const int64_t t_l = token_of((tile_live ? tile : n_tiles - 1) * 16 + ti, live_l);
    live_l = live_l && tile_live;
    if (tile + tstep < n_tiles) fetch(tile + tstep, xnx);
    int lz = 0;
    asm volatile("" : "+v"(lz));  // per-tile opaque zero: keeps the loop-invariant LDS reads inside the loop
    // ---- LayerNorm: two-pass statistics, the four column quarters meet in LDS (fixed order) -----------------------
    if constexpr (LN) {
      constexpr float inv_d = 1.0f / (float)d;
      f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MP; ++m) s4 += xv[m][0] + xv[m][1];
      const float s1 = kk_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
      if (kk == 0) st1[wave * 16 + ti] = s1;
      __syncthreads();
      const float mean = ((st1[ti + lz] + st1[16 + ti + lz]) + (st1[32 + ti + lz] + st1[48 + ti + lz])) * inv_d;
      const f32x4 mean4 = f32x4{mean, mean, mean, mean};
      f32x4 q4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MP; ++m) {
        const f32x4 a = xv[m][0] - mean4, b = xv[m][1] - mean4;
        q4 = __builtin_elementwise_fma(a, a, q4);
        q4 = __builtin_elementwise_fma(b, b, q4);
      }
      const float s2 = kk_sum((q4[0] + q4[1]) + (q4[2] + q4[3]));
      if (kk == 0) st2[wave * 16 + ti] = s2;
      __syncthreads();
      const float var = ((st2[ti + lz] + st2[16 + ti + lz]) + (st2[32 + ti + lz] + st2[48 + ti + lz])) * inv_d;
      const float rstd = rsqrtf(var + ln_eps);
      const f32x4 rstd4 = f32x4{rstd, rstd, rstd, rstd};
#pragma unroll
      for (int m = 0; m < MP; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 gg = *reinterpret_cast<const f32x4*>(lds_g + lz + colw + 32 * m + 4 * h);
          const f32x4 bb = *reinterpret_cast<const f32x4*>(lds_be + lz + colw + 32 * m + 4 * h);
          xv[m][h] = __builtin_elementwise_fma((xv[m][h] - mean4) * rstd4, gg, bb);
        }
      if (MODE == 0 && live_l) {   // operand images: 64 contiguous bytes (16-bit) / 128 (f32) per token and instruction
        const int64_t rowoff = t_l * (int64_t)d + colw;
        if (xn32) {
#pragma unroll
          for (int m = 0; m < MP; ++m) {
            *reinterpret_cast<f32x4*>(xn32 + rowoff + 32 * m) = xv[m][0];
            *reinterpret_cast<f32x4*>(xn32 + rowoff + 32 * m + 4) = xv[m][1];
          }
        }
        if (xn16) {
#pragma unroll
          for (int m = 0; m < MP; ++m) {
            const float v8[8] = {xv[m][0][0], xv[m][0][1], xv[m][0][2], xv[m][0][3], xv[m][1][0], xv[m][1][1], xv[m][1][2], xv[m][1][3]};
            store8(xn16 + rowoff + 32 * m, v8);   // one 16-byte store: 64 contiguous bytes per token and instruction
          }
        }
      }
    }
    // ---- partial logits of this wave's column quarter ---------------------------------------------------------------
    if constexpr (MODE == 0) {
      f32x4 q4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MP; ++m) {
        q4 = __builtin_elementwise_fma(xv[m][0], xv[m][0], q4);
        q4 = __builtin_elementwise_fma(xv[m][1], xv[m][1], q4);
      }
      const float xsp = kk_sum((q4[0] + q4[1]) + (q4[2] + q4[3]));
      f32x4 acc[NTL][NACC];
#pragma unroll
      for (int n = 0; n < NTL; ++n)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[n][a] = f32x4{0.f, 0.f, 0.f, 0.f};
      // weight row (16 n + ti): its swizzle is ti & 7 for every n and touches the 16-byte slot inside a 32-column group only,
      // so one address per half-piece h and immediates for m and n
      const float* wrow = lds_w + lz + ti * WP + wave * dq;
      const int wlo[2] = {(8 * kk) ^ ((ti & 7) << 2), (8 * kk + 4) ^ ((ti & 7) << 2)};
#pragma unroll
      for (int m = 0; m < MP; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 wf[NTL];
#pragma unroll
          for (int n = 0; n < NTL; ++n) wf[n] = *reinterpret_cast<const f32x4*>(wrow + wlo[h] + (n * 16 * WP + 32 * m));
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int n = 0; n < NTL; ++n)
              acc[n][(2 * m + h) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][c], xv[m][h][c], acc[n][(2 * m + h) % NACC], 0, 0, 0);
        }
      float* pw = part + ((par * 4 + wave) * 16 + ti) * PS;
#pragma unroll
      for (int n = 0; n < NTL; ++n)   // lane: token ti, experts 16 n + 4 kk + (0..3)
        *reinterpret_cast<f32x4*>(pw + 16 * n + 4 * kk) = (acc[n][0] + acc[n][1]) + (acc[n][2] + acc[n][3]);
      if (kk == 0) pw[EB] = xsp;
    } else {
      // the same contraction on the f64 matrix cores (v_mfma_f64_16x16x4_f64: operands as the f32 form, one f64 per lane; the
      // products of f32-origin values are exact in f64, the accumulation is f64).  C/D: col = lane & 15 (token), row = kk + 4 r.
      typedef double f64x4 __attribute__((ext_vector_type(4)));
      f64x4 acc[NTL][2];
#pragma unroll
      for (int n = 0; n < NTL; ++n) acc[n][0] = acc[n][1] = f64x4{0.0, 0.0, 0.0, 0.0};
      const float* wrow = lds_w + lz + ti * WP + wave * dq;
      const int wlo[2] = {(8 * kk) ^ ((ti & 7) << 2), (8 * kk + 4) ^ ((ti & 7) << 2)};
#pragma unroll
      for (int m = 0; m < MP; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 wf[NTL];
#pragma unroll
          for (int n = 0; n < NTL; ++n) wf[n] = *reinterpret_cast<const f32x4*>(wrow + wlo[h] + (n * 16 * WP + 32 * m));
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int n = 0; n < NTL; ++n)
              acc[n][h] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)wf[n][c], (double)xv[m][h][c], acc[n][h], 0, 0, 0);
        }
      double* pw = dpart + (wave * 16 + ti) * EB;
#pragma unroll
      for (int n = 0; n < NTL; ++n) {
        const f64x4 sacc = acc[n][0] + acc[n][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[16 * n + kk + 4 * r] = sacc[r];
      }
    }
    __syncthreads();

    // ---- tail: wave w ranks tokens 4 w .. 4 w + 3 of the tile, one per 16-lane DPP row; lane u of the row owns experts
    //      u and u + 16.  (Every lane scanning all E logits, as router16's tail does, cost 7.4 k cycles per tile here -- more
    //      than the 128 MFMAs.)  Arg-max = row maximum, then the lowest expert id among the lanes that hold it: ties -> lowest id.
    {
      const int q = lane >> 4, u = lane & 15, tl = 4 * wave + q;
      bool live;
      const int64_t t = token_of((tile_live ? tile : n_tiles - 1) * 16 + tl, live);
      live = live && tile_live;
      float lgv[NTL];
      float xs = 0.f;
      if constexpr (MODE == 0) {
        const float* p0 = part + lz + ((par * 4 + 0) * 16 + tl) * PS;
        constexpr int WS = 16 * PS;   // wave stride
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          const int e = u + 16 * j;
          lgv[j] = ((p0[e] + p0[WS + e]) + (p0[2 * WS + e] + p0[3 * WS + e])) + lds_bias[e];
        }
        xs = (p0[EB] + p0[WS + EB]) + (p0[2 * WS + EB] + p0[3 * WS + EB]);
      } else {
        const double* p0 = dpart + lz + tl * EB;
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          const int e = u + 16 * j;
          lgv[j] = (float)(((p0[e] + p0[16 * EB + e]) + (p0[32 * EB + e] + p0[48 * EB + e])) + (double)lds_bias[e]);
        }
      }
      if (logits_out && live) {
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          if (u + 16 * j < E) logits_out[t * (int64_t)E + u + 16 * j] = lgv[j];
      }
      if (gate_kind == SMOE_GATE_SWITCH && noise && live) {
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          if (u + 16 * j < E) lgv[j] += noise[t * (int64_t)E + u + 16 * j];
      }
      const int kc = (MODE == 0 && k < E) ? k + 1 : k;
      int chosen[R16_MAX_K + 1];
      float cval[R16_MAX_K + 1];
      float lw[NTL];
#pragma unroll
      for (int j = 0; j < NTL; ++j) lw[j] = (u + 16 * j < E) ? lgv[j] : -INFINITY;
#pragma unroll
      for (int r = 0; r <= R16_MAX_K; ++r) {
        chosen[r] = 0;
        cval[r] = 0.f;
        if (r < kc) {   // wave-uniform
          float bv = lw[0];
          int bi = u;
#pragma unroll
          for (int j = 1; j < NTL; ++j) {
            const bool gt = lw[j] > bv;   // strict: the lower id wins a tie inside the lane
            bv = gt ? lw[j] : bv;
            bi = gt ? u + 16 * j : bi;
          }
          const float mx = row16_max(bv);
          const int best = row16_min(bv == mx ? bi : 0x7fffffff);
          chosen[r] = best;
          cval[r] = mx;
          if (r + 1 < kc) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) lw[j] = (u + 16 * j == best) ? -INFINITY : lw[j];
          }
        }
      }
      if constexpr (MODE == 0) {
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r <= R16_MAX_K; ++r)
          if (r < kc) amax = fmaxf(amax, fabsf(cval[r]));
        // router16's bound (per-lane chain 8 MP, reduction levels, bias; x 2 logits x 2 safety); the MFMA chains here are 2 MP
        // steps of 4 products + 2 + 2 combine levels, well inside it (tests/test_gpu_parity.py measures <= 2 % of it)
        const float bound = 4.0f * (float)(MP * 8 + 6) * 5.9604645e-8f * sqrtf(xs * wmax2) + 9.6e-7f * (amax + 1.0f);
        bool ambiguous = false;
#pragma unroll
        for (int r = 0; r < R16_MAX_K; ++r)
          if (r + 1 < kc) ambiguous |= !((cval[r] - cval[r + 1]) > bound);
        if (ambiguous && xs > 0.f && live && u == 0) list_push(redo_count, redo_list, T, t);
      }
      if (gate_kind == SMOE_GATE_NAIVE) {
        if (live && u == 0) {
          float ex[R16_MAX_K];
          float sden = 0.f;
#pragma unroll
          for (int r = 0; r < R16_MAX_K; ++r) {
            ex[r] = (r < k) ? expf(cval[r] - cval[0]) : 0.f;
            sden += ex[r];
          }
#pragma unroll
          for (int r = 0; r < R16_MAX_K; ++r)
            if (r < k) {
              idx_out[t * (int64_t)k + r] = chosen[r];
              score_out[t * (int64_t)k + r] = ex[r] / sden;
            }
        }
      } else {   // switch gate: softmax over all E, the row's lanes each own their experts' terms
        const float mx = cval[0];
        float pe[NTL];
        float sl = 0.f;
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
          pe[j] = (u + 16 * j < E) ? expf(lgv[j] - mx) : 0.f;
          sl += pe[j];
        }
        const float sden = r16::row16_sum(sl);
        if (probs_out && live) {
#pragma unroll
          for (int j = 0; j < NTL; ++j)
            if (u + 16 * j < E) probs_out[t * (int64_t)E + u + 16 * j] = pe[j] / sden;
        }
        if (live && u == 0) {
          idx_out[t] = chosen[0];
          score_out[t] = 1.0f / sden;
        }
      }
    }
    if constexpr (MODE == 1 || NPAR == 1) __syncthreads();   // a single exchange buffer is rewritten by the next tile
#pragma unroll
    for (int m = 0; m < MP; ++m) {
      xv[m][0] = xnx[m][0];
      xv[m][1] = xnx[m][1];
    }
  }
  // the redo pass leaves the counter words zero for the next call (router16_kernel.h: the last workgroup to finish clears them)
  if (MODE == 1 && redo_list) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int old = atomicAdd(&redo_count[1], 1);
      if (old == (int)gridDim.x - 1) {
        redo_count[1] = 0;
        redo_count[0] = 0;
      }
    }
  }
}

}  // namespace rmt
