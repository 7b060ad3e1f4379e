// Router: gate logits + top-k + softmax.  HBM-bound: reads the token matrix once.
// Replaces fmoe NaiveGate / SwitchGate forward.
//
// Contract (oracle/moe_oracle.py): routing is decided on logits accumulated in f64 and rounded once to
// f32, so it cannot depend on summation order.  f64 FMAs for every token would make this kernel
// compute-bound (measured 1.05 TB/s), so each token first gets f32 logits with a rigorous error bound
//     |logit_f32 - exact| <= (n1 + 7) u ||x|| ||w||max   (n1 = per-lane FMA chain, 6 butterfly levels + bias)
// and only when one of the gaps that decide the result (between consecutive entries of the top-(k+1))
// is inside that bound is the token re-done in f64 -- a wave-uniform, rare branch (~1e-4 of tokens).
// Either way idx equals the oracle's; score/logits carry f32 rounding (<= ~1e-6).
//
// Layout: one wave per token row (coalesced 16-B loads: lane l owns elements [256c + 4l, +4) of chunk c),
// eight experts at a time; router weights in LDS as f32.  The 8 per-lane partial sums of a group of 8
// experts are reduced with a transposed butterfly (10 shuffles instead of 48): afterwards lanes with
// equal (lane>>3) hold the full logit of expert 4*b5 + 2*b4 + b3 of the group.
#include "smoe_common.h"

namespace {

constexpr int ROUTER_THREADS = 256;
constexpr int ROUTER_WAVES = ROUTER_THREADS / 64;
constexpr int ROUTER_MAX_K = 4;
constexpr int NONE_IDX = 0x7fffffff;

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int m) { return __shfl_xor(v, m, 64); }

// 8 values per lane (value i = partial sum of expert i) -> every lane gets the wave total of expert
// 4*b5 + 2*b4 + b3 (b = bits of its lane id)
template <typename T> __device__ __forceinline__ T butterfly8(const T (&acc)[8], int lane) {
  T a4[4], a2[2], a1;
  {
    const bool hi = lane & 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const T send = hi ? acc[i] : acc[i + 4];
      const T keep = hi ? acc[i + 4] : acc[i];
      a4[i] = keep + shfl_xor_t(send, 32);
    }
  }
  {
    const bool hi = lane & 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const T send = hi ? a4[i] : a4[i + 2];
      const T keep = hi ? a4[i + 2] : a4[i];
      a2[i] = keep + shfl_xor_t(send, 16);
    }
  }
  {
    const bool hi = lane & 8;
    const T send = hi ? a2[0] : a2[1];
    const T keep = hi ? a2[1] : a2[0];
    a1 = keep + shfl_xor_t(send, 8);
  }
  a1 += shfl_xor_t(a1, 4);
  a1 += shfl_xor_t(a1, 2);
  a1 += shfl_xor_t(a1, 1);
  return a1;
}

// MODE 0: f32 pass over all tokens; a token whose deciding gaps are inside the error bound is appended to
//         redo_list (its provisional outputs are overwritten later).  No f64 code -> low VGPR, high occupancy.
// MODE 1: f64 pass over redo_list[0 .. *redo_count)  (or over all tokens when redo_list == nullptr).
template <typename XT, int NCH, bool W_LDS, int MODE>
__global__ __launch_bounds__(ROUTER_THREADS, (MODE == 0 ? 4 : 2)) void router_kernel(
    const XT* __restrict__ x, const float* __restrict__ wg, const float* __restrict__ bg,
    const float* __restrict__ noise, int64_t T, int d, int E, int k, int gate_kind,
    int32_t* __restrict__ redo_count, int32_t* __restrict__ redo_list,
    int64_t* __restrict__ idx_out, float* __restrict__ score_out, float* __restrict__ logits_out,
    float* __restrict__ probs_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Epad = (E + 7) & ~7;
  // carve: [ROUTER_WAVES][Epad] f32 logits | [Epad] f32 squared weight norms (+pad) | (W_LDS) [Epad][d] f32 weights
  float* lds_logit = reinterpret_cast<float*>(smem);
  const size_t logit_bytes = ((size_t)(ROUTER_WAVES + 1) * Epad * 4 + 15) & ~(size_t)15;
  float* lds_wn2 = lds_logit + ROUTER_WAVES * Epad;
  float* lds_w = reinterpret_cast<float*>(smem + logit_bytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  if (W_LDS) {
    for (int i = tid * 4; i < Epad * d; i += ROUTER_THREADS * 4) {
      const int e = i / d;  // d % 4 == 0: a float4 never straddles two experts
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < E) v = *reinterpret_cast<const f32x4*>(wg + i);
      *reinterpret_cast<f32x4*>(lds_w + i) = v;
    }
  }
  // squared row norms of the router weights (for the error bound)
  for (int e = wave; e < Epad; e += ROUTER_WAVES) {
    float s = 0.f;
    if (e < E)
      for (int c = lane; c < d; c += 64) { const float w = wg[(size_t)e * d + c]; s = fmaf(w, w, s); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) lds_wn2[e] = s;
  }
  __syncthreads();
  float wmax2 = 0.f;
  for (int e = 0; e < E; ++e) wmax2 = fmaxf(wmax2, lds_wn2[e]);

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
  auto load_w = [&](int e, int col, float (&wf)[4]) {
    if (W_LDS) {
      f32x4 v = *reinterpret_cast<const f32x4*>(lds_w + (size_t)e * d + col);
      wf[0] = v[0]; wf[1] = v[1]; wf[2] = v[2]; wf[3] = v[3];
    } else if (e < E) {
      load4(wg + (size_t)e * d + col, wf);
    } else {
      wf[0] = wf[1] = wf[2] = wf[3] = 0.f;
    }
  };

  // work items: MODE 0 -> tokens 0..T-1; MODE 1 with a list -> list entries 0..count-1
  int64_t n_items = T;
  if (MODE == 1 && redo_list) {  // a trip count read from device memory is never trusted: at most T tokens can be listed
    n_items = *redo_count;
    n_items = n_items < 0 ? 0 : (n_items > T ? T : n_items);
  }
  auto item_token = [&](int64_t it) -> int64_t {
    if (!(MODE == 1 && redo_list)) return it;
    const int64_t t = (int64_t)redo_list[it];
    return t < 0 ? 0 : (t >= T ? T - 1 : t);  // list entries index x: keep them inside it
  };
  for (int64_t it = wave_gid; it < n_items; it += wave_stride) {
    const int64_t t = item_token(it);
    load_row(t);  // latency is hidden by occupancy (4 blocks of 4 waves per CU), not by software prefetch
    float (&xc)[NCH][4] = xn;

    auto publish = [&](int eb, float lg_wave, bool f64_path, double lg64) {
      const int el = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
      const int e = eb + el;
      if ((lane & 7) == 0 && e < E) {
        float lg;
        if (f64_path) lg = (float)(lg64 + (bg ? (double)bg[e] : 0.0));
        else lg = lg_wave + (bg ? bg[e] : 0.f);
        if (logits_out) logits_out[t * (int64_t)E + e] = lg;
        if (gate_kind == SMOE_GATE_SWITCH && noise) lg += noise[t * (int64_t)E + e];
        my_logit[e] = lg;
      }
    };
    auto logits_f32 = [&]() {
      for (int eb = 0; eb < Epad; eb += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (cvalid[c]) {
            const int col = c * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              float wf[4];
              load_w(eb + i, col, wf);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i] = fmaf(xc[c][j], wf[j], acc[i]);
              if (i & 1) __builtin_amdgcn_sched_barrier(0);  // keep at most two weight loads in flight (VGPR budget)
            }
          }
        }
        publish(eb, butterfly8(acc, lane), false, 0.0);
      }
    };
    auto logits_f64 = [&]() {
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
              float wf[4];
              load_w(eb + i, col, wf);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i] = fma((double)xc[c][j], (double)wf[j], acc[i]);
            }
          }
        }
        publish(eb, 0.f, true, butterfly8(acc, lane));
      }
    };
    // top-kc over my_logit[0..E): ties -> lowest expert id, order = descending value
    int chosen[ROUTER_MAX_K + 1];
    float chosen_val[ROUTER_MAX_K + 1];
    auto select = [&](int kc) {
#pragma unroll
      for (int r = 0; r <= ROUTER_MAX_K; ++r) { chosen[r] = -1; chosen_val[r] = 0.f; }
#pragma unroll
      for (int r = 0; r <= ROUTER_MAX_K; ++r) {
        if (r < kc) {
          float bv = -INFINITY;
          int bi = NONE_IDX;
          for (int e = lane; e < E; e += 64) {
            bool taken = false;
#pragma unroll
            for (int q = 0; q <= ROUTER_MAX_K; ++q) taken |= (q < r) && (chosen[q] == e);
            const float v = my_logit[e];
            if (!taken && (bi == NONE_IDX || v > bv || (v == bv && e < bi))) { bv = v; bi = e; }
          }
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(bv, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            const bool take = (oi != NONE_IDX) && (bi == NONE_IDX || ov > bv || (ov == bv && oi < bi));
            if (take) { bv = ov; bi = oi; }
          }
          chosen[r] = bi;
          chosen_val[r] = bv;
        }
      }
    };

    const int kc = (k < E) ? k + 1 : k;  // the (k+1)-th value guards the boundary of the kept set
    if constexpr (MODE == 0) {
      float xs = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) xs = fmaf(xc[c][j], xc[c][j], xs);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) xs += __shfl_xor(xs, m, 64);
      logits_f32();
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes precede its reads
      __builtin_amdgcn_wave_barrier();
      select(kc);
      // 2 x (bound on each logit) with a 2x safety factor; + rounding of the stored f32 values (+ noise add)
      float amax = 0.f;
#pragma unroll
      for (int r = 0; r <= ROUTER_MAX_K; ++r)
        if (r < kc) amax = fmaxf(amax, fabsf(chosen_val[r]));
      const float bound = 4.0f * (float)(NCH * 4 + 8) * 5.9604645e-8f * sqrtf(xs * wmax2) + 9.6e-7f * (amax + 1.0f);
      bool ambiguous = false;
#pragma unroll
      for (int r = 0; r < ROUTER_MAX_K; ++r)
        if (r + 1 < kc) ambiguous |= !((chosen_val[r] - chosen_val[r + 1]) > bound);
      if (ambiguous && lane == 0) list_push(redo_count, redo_list, T, t);  // wave-uniform condition
    } else {
      logits_f64();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      select(k);
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

template <typename XT, int NCH>
int launch_router(const void* x, const float* wg, const float* bg, const float* noise, int64_t T, int d, int E,
                  int k, int gate_kind, int force_f64, int32_t* redo_count, int32_t* redo_list, int64_t* idx,
                  float* score, float* logits_out, float* probs, hipStream_t stream) {
  const int Epad = (E + 7) & ~7;
  const size_t logit_bytes = ((size_t)(ROUTER_WAVES + 1) * Epad * 4 + 15) & ~(size_t)15;
  const size_t w_bytes = (size_t)Epad * d * 4;
  const bool w_lds = (logit_bytes + w_bytes) <= 64 * 1024;
  const size_t smem = logit_bytes + (w_lds ? w_bytes : 0);
  int64_t need = (T + ROUTER_WAVES - 1) / ROUTER_WAVES;
  const int grid = (int)(need < 2048 ? (need < 1 ? 1 : need) : 2048);
#define ROUTER_LAUNCH(WL, MODE, GRID, RC, RL)                                                                      \
  hipLaunchKernelGGL((router_kernel<XT, NCH, WL, MODE>), dim3(GRID), dim3(ROUTER_THREADS), smem, stream,          \
                     (const XT*)x, wg, bg, noise, T, d, E, k, gate_kind, RC, RL, idx, score, logits_out, probs)
  const bool ws_zero = (force_f64 & 2) != 0;
  force_f64 &= 1;
  if (force_f64) {
    if (w_lds) ROUTER_LAUNCH(true, 1, grid, nullptr, nullptr);
    else ROUTER_LAUNCH(false, 1, grid, nullptr, nullptr);
    SMOE_CHECK_LAUNCH("smoe_router_topk/f64");
    return 0;
  }
  if (!ws_zero) {
    hipError_t me = smoe_zero_words(redo_count, 4, stream);
    if (me != hipSuccess) {
      smoe_set_error("smoe_router_topk: counter clear failed: %s", hipGetErrorString(me));
      return (int)me;
    }
  }
  if (w_lds) ROUTER_LAUNCH(true, 0, grid, redo_count, redo_list);
  else ROUTER_LAUNCH(false, 0, grid, redo_count, redo_list);
  SMOE_CHECK_LAUNCH("smoe_router_topk/f32");
  const int redo_grid = grid < 32 ? grid : 32;  // the redo list is short (~1e-4 T); surplus waves exit at once
  if (w_lds) ROUTER_LAUNCH(true, 1, redo_grid, redo_count, redo_list);
  else ROUTER_LAUNCH(false, 1, redo_grid, redo_count, redo_list);
  SMOE_CHECK_LAUNCH("smoe_router_topk/redo");
#undef ROUTER_LAUNCH
  return 0;
}

template <typename XT>
int dispatch_nch(const void* x, const float* wg, const float* bg, const float* noise, int64_t T, int d, int E,
                 int k, int gate_kind, int f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score,
                 float* logits_out, float* probs, hipStream_t s) {
  const int nch = (d + 255) / 256;
  switch (nch) {
    case 1: return launch_router<XT, 1>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, logits_out, probs, s);
    case 2: return launch_router<XT, 2>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, logits_out, probs, s);
    case 3: return launch_router<XT, 3>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, logits_out, probs, s);
    case 4: return launch_router<XT, 4>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, logits_out, probs, s);
    case 5: return launch_router<XT, 5>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, logits_out, probs, s);
    case 6: return launch_router<XT, 6>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, logits_out, probs, s);
    case 7: return launch_router<XT, 7>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, logits_out, probs, s);
    case 8: return launch_router<XT, 8>(x, wg, bg, noise, T, d, E, k, gate_kind, f64, rc, rl, idx, score, logits_out, probs, s);
  }
  smoe_set_error("smoe_router_topk: d=%d unsupported (d <= 2048)", d);
  return 1;
}

}  // namespace

int smoe_router16_try(const void* x, int x_dtype, const float* wg, const float* bg, const float* noise, int64_t T,
                      int d, int E, int k, int gate_kind, int force_f64, int32_t* rc, int32_t* rl, int64_t* idx,
                      float* score, float* logits_out, float* probs, hipStream_t s);  // router16.hip

// workspace: [16 B: redo counter][T x i32 redo list]
extern "C" size_t smoe_router_workspace_bytes(int64_t T) { return T < 0 ? 0 : 16 + (((size_t)T * 4 + 15) & ~(size_t)15); }

// gate_kind bit 8 (0x100) forces the all-f64 path (test hook: every token through the f64 kernel)
extern "C" int smoe_router_topk(const void* x, int x_dtype, const float* wg, const float* bg, const float* noise,
                                int64_t T, int d, int E, int k, int gate_kind, int64_t* idx, float* score,
                                float* logits_out, float* probs, void* workspace, size_t workspace_bytes,
                                void* stream) {
  const int force_f64 = ((gate_kind & 0x100) ? 1 : 0) | ((gate_kind & 0x200) ? 2 : 0);   // bit 1: counter words kept zero by the caller
  gate_kind &= 0xff;
  if (T == 0) return 0;  // empty batch: nothing to route (zero-sized tensors have null data pointers)
  SMOE_REQUIRE(x && wg && idx && score, "smoe_router_topk: null pointer");
  SMOE_REQUIRE(T >= 0 && T < (1ll << 31) && d > 0 && E > 0, "smoe_router_topk: bad sizes T=%lld d=%d E=%d", (long long)T, d, E);
  SMOE_REQUIRE(d % 8 == 0 && d <= 2048, "smoe_router_topk: d=%d must be a multiple of 8 and <= 2048", d);
  SMOE_REQUIRE(k >= 1 && k <= E && k <= ROUTER_MAX_K, "smoe_router_topk: k=%d out of range (E=%d, max %d)", k, E,
               ROUTER_MAX_K);
  SMOE_REQUIRE(E <= 4096, "smoe_router_topk: E=%d too large", E);
  SMOE_REQUIRE(gate_kind == SMOE_GATE_NAIVE || gate_kind == SMOE_GATE_SWITCH, "smoe_router_topk: bad gate_kind %d",
               gate_kind);
  SMOE_REQUIRE(gate_kind != SMOE_GATE_SWITCH || k == 1, "smoe_router_topk: switch gate needs k == 1");
  SMOE_REQUIRE(workspace && workspace_bytes >= smoe_router_workspace_bytes(T), "smoe_router_topk: workspace too small");
  if (T == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  int32_t* rc = reinterpret_cast<int32_t*>(workspace);
  int32_t* rl = reinterpret_cast<int32_t*>((char*)workspace + 16);
  {
    const int r16 = smoe_router16_try(x, x_dtype, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score,
                                      logits_out, probs, s);
    if (r16 != -1) return r16;
  }
  switch (x_dtype) {
    case SMOE_F32: return dispatch_nch<float>(x, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
    case SMOE_F16: return dispatch_nch<f16>(x, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
    case SMOE_BF16: return dispatch_nch<bf16_bits>(x, wg, bg, noise, T, d, E, k, gate_kind, force_f64, rc, rl, idx, score, logits_out, probs, s);
  }
  smoe_set_error("smoe_router_topk: bad x_dtype %d", x_dtype);
  return 1;
}
