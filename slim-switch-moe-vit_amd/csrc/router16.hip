// Router, 16-lanes-per-token layout (E <= 8 at d in {192, 384, 768, 1024}; E <= 32 at d in {768, 1024}): the fast
// path of smoe_router_topk, the fused LayerNorm + router, and the LayerNorm kernel.  The kernel template lives in
// router16_kernel.h (shared with gate.hip, which instantiates the token-skip-gate variants).
#include "router16_kernel.h"
#include "router_mt_kernel.h"
#include <cstdlib>

namespace {
using namespace r16;

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

// LayerNorm alone, ONE WAVE PER ROW (lane l holds the elements [8 l + 512 i, 8 l + 512 i + 8) of the row: whole 32-byte pieces,
// two-pass statistics over the wave).  Same arithmetic order inside a lane's pieces as the 16-lanes-per-token kernel? No -- the
// reduction tree differs, so the two kernels agree to f32 rounding, not bit for bit; a shape uses ONE of them, always (below).
// At d = 1024 the 16-lane layout holds 64 row elements per lane and reaches 3.1 TB/s (cfg 4's model: 72.6 us for 227 MB,
// profiles/r05_cfg4_kernel_stats.csv); a wave per row holds 16.
template <typename XT, int NI, typename OT>
__global__ __launch_bounds__(256) void layernorm_wave_kernel(const XT* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ b, float eps, int64_t T, int d,
                                                             OT* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave_gid; t < T; t += nwaves) {
    float v[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 8 + 512 * i;
      if (c < d) load8(x + t * (int64_t)d + c, v[i]);
      else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[i][q] = 0.f;
      }
    }
    float mean, rstd;
    wave_row_stats<NI>(v, d, lane, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 8 + 512 * i;
      if (c < d) {
        float gg[8], bb[8], o[8];
        if (g) load8(g + c, gg);
        if (b) load8(b + c, bb);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = wave_row_affine(v[i][q], mean, rstd, g ? gg[q] : 1.f, b ? bb[q] : 0.f);
        store8(out + t * (int64_t)d + c, o);
      }
    }
  }
}

template <typename XT, typename OT>
int ln_dispatch_nj(const void* x, const float* g, const float* b, float eps, int64_t T, int d, void* out, hipStream_t s) {
  if (smoe_ln_wave_layout(d) && d % 8 == 0 && d <= 1024) {
    int64_t blocks = (T + 3) / 4;
    const int wg = (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
    if (d <= 512) hipLaunchKernelGGL((layernorm_wave_kernel<XT, 1, OT>), dim3(wg), dim3(256), 0, s, (const XT*)x, g, b, eps, T, d, (OT*)out);
    else hipLaunchKernelGGL((layernorm_wave_kernel<XT, 2, OT>), dim3(wg), dim3(256), 0, s, (const XT*)x, g, b, eps, T, d, (OT*)out);
    SMOE_CHECK_LAUNCH("smoe_layernorm/wave");
    return 0;
  }
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
  int32_t* hist = nullptr;   // chunk histogram for the dispatch plan (E <= 8 images only), or NULL
};

template <typename XT, int NJ, bool LN, typename NT, int EB>
int launch16(const void* x, const LnArgs& ln, const float* wg, const float* bg, const float* noise, int64_t T, int d,
             int E, int k, int gate_kind, int force_f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score,
             float* logits_out, float* probs, hipStream_t s) {
  // 32 experts: the f32 weight image (96-128 KB) leaves room for ONE workgroup per CU; 512 threads put 8 waves behind it
  // instead of 4 (the f32 pass only: the f64 redo pass would spill at that size and handles a handful of tokens)
  constexpr int NTH = EB > 16 ? 512 : R16_THREADS;
  const size_t smem = router16_smem<NJ, LN, EB, 0>();
  const int64_t tok_per_block = (NTH / 64) * 4;
  int64_t need = (T + tok_per_block - 1) / tok_per_block;
  // <= 768 workgroups (3 resident per CU; 512 = 2 per CU for the 16-expert image: the LDS weight image is loaded once
  // per workgroup), every workgroup the same number of 16-token groups
  constexpr int64_t max_wg = EB <= 8 ? 768 : (EB <= 16 ? 512 : 256);
  const int64_t iters = (need + max_wg - 1) / max_wg;
  int grid = (int)(need < 1 ? 1 : (need + iters - 1) / (iters < 1 ? 1 : iters));
  // with a chunk histogram every workgroup of the f32 pass walks ONE contiguous chunk of R16_HIST_TOK tokens
  HistArgs ha{(EB == 8 && !(force_f64 & 1)) ? ln.hist : nullptr, R16_HIST_TOK};
  const int grid0 = ha.hist ? (int)((T + R16_HIST_TOK - 1) / R16_HIST_TOK) : grid;
  if (smem > 64 * 1024) {  // 16 experts x d 1024 (+ LayerNorm vectors): above the default dynamic-LDS limit
    SMOE_ENSURE_SMEM(router16_kernel<XT, NJ, 0, LN, NT, EB, 0, NTH>);
    SMOE_ENSURE_SMEM(router16_kernel<XT, NJ, 1, LN, NT, EB, 0, R16_THREADS>);
  }
#define R16_LAUNCH(MODE, GRID, RC, RL)                                                                               \
  hipLaunchKernelGGL((router16_kernel<XT, NJ, MODE, LN, NT, EB, 0, (MODE == 0 ? NTH : R16_THREADS)>), dim3(GRID),       \
                     dim3(MODE == 0 ? NTH : R16_THREADS), smem, s, (const XT*)x,                                      \
                     ln.g, ln.b, ln.eps, (NT*)ln.xn16, ln.xn32, wg, bg, noise, T, d, E, k, gate_kind, RC, RL, idx,   \
                     score, logits_out, probs, SkipGateArgs{}, ha)
  const bool ws_zero = (force_f64 & 2) != 0;
  force_f64 &= 1;
  if (force_f64 && !LN) {
    R16_LAUNCH(1, grid, nullptr, nullptr);
    SMOE_CHECK_LAUNCH("smoe_router_topk/f64");
    return 0;
  }
  if (!ws_zero) {   // (a kept workspace is zero already: the redo pass clears its counter on the way out)
    hipError_t me = smoe_zero_words(rc, 4, s);
    if (me != hipSuccess) {
      smoe_set_error("smoe_router_topk: counter clear failed: %s", hipGetErrorString(me));
      return (int)me;
    }
  }
  if (force_f64 && LN) {  // f64 mode still needs the normalised rows written: run the f32 pass for its stores first
    R16_LAUNCH(0, grid, rc, rl);
    SMOE_CHECK_LAUNCH("smoe_router_topk/f32");
    R16_LAUNCH(1, grid, nullptr, nullptr);
    SMOE_CHECK_LAUNCH("smoe_router_topk/f64");
    return 0;
  }
  R16_LAUNCH(0, grid0, rc, rl);
  SMOE_CHECK_LAUNCH("smoe_router_topk/f32");
  R16_LAUNCH(1, (grid < 16 ? grid : 16), rc, rl);
  SMOE_CHECK_LAUNCH("smoe_router_topk/redo");
#undef R16_LAUNCH
  return 0;
}

// 16 / 32 experts on the f32 matrix cores (router_mt_kernel.h); SMOE_ROUTER_MT=0 keeps the 16-lanes-per-token kernel (A/B)
static bool use_router_mt() {
  static const bool on = [] { const char* v = getenv("SMOE_ROUTER_MT"); return !(v && v[0] == '0'); }();
  return on;
}

template <typename XT, int MP, bool LN, typename NT, int EB>
int launch_mt(const void* x, const LnArgs& ln, const float* wg, const float* bg, const float* noise, int64_t T, int E, int k,
              int gate_kind, int force_f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score, float* logits_out,
              float* probs, hipStream_t s) {
  using namespace rmt;
  constexpr size_t smem1 = router_mt_smem<MP, LN, EB, 1>();
  static_assert(smem1 <= 160 * 1024, "router_mt: LDS image too large");
  // An image that leaves room for ONE workgroup per CU (E = 32) runs the f32 pass on TWO four-wave halves per workgroup (HV = 2:
  // two waves per SIMD on the same image; router_mt_kernel.h) when both halves' exchange areas fit; SMOE_ROUTER_MT_HALVES=1 = the
  // one-half form (A/B).  The f64 re-do pass always runs one half.
  constexpr bool can2 = 2 * smem1 > 160 * 1024 && router_mt_smem<MP, LN, EB, 2>() <= 160 * 1024;
  static const bool want2 = [] { const char* v = getenv("SMOE_ROUTER_MT_HALVES"); return !(v && v[0] == '1'); }();
  const bool two = can2 && want2;
  const size_t smem = smem1;
  const int64_t n_tiles = (T + 15) / 16;
  const int per_cu = (int)((160 * 1024) / smem) < 1 ? 1 : (int)((160 * 1024) / smem);
  const int64_t max_wg = (int64_t)smoe_num_cus() * (per_cu > 2 ? 2 : per_cu);
  // every workgroup the same number of tiles (the weight image is staged once per workgroup)
  const int64_t iters = (n_tiles + max_wg - 1) / max_wg;
  const int grid = (int)(n_tiles < 1 ? 1 : (n_tiles + iters - 1) / (iters < 1 ? 1 : iters));
  SMOE_ENSURE_SMEM(router_mt_kernel<XT, MP, 0, LN, NT, EB>);
  SMOE_ENSURE_SMEM(router_mt_kernel<XT, MP, 1, LN, NT, EB>);
#define MT_LAUNCH(MODE, GRID, RC, RL)                                                                                  \
  hipLaunchKernelGGL((router_mt_kernel<XT, MP, MODE, LN, NT, EB>), dim3(GRID), dim3(MT_THREADS), smem, s, (const XT*)x, \
                     ln.g, ln.b, ln.eps, (NT*)ln.xn16, ln.xn32, wg, bg, noise, T, E, k, gate_kind, RC, RL, idx, score,  \
                     logits_out, probs)
  // the f32 pass on two halves: half as many workgroups, each walking the tiles of two
  auto launch_f32 = [&](int32_t* RC, int32_t* RL) {
    if constexpr (can2) {
      if (two) {
        constexpr size_t smem2 = router_mt_smem<MP, LN, EB, 2>();
        const int64_t pairs = (n_tiles + 1) / 2, cap = smoe_num_cus();
        const int64_t it2 = (pairs + cap - 1) / cap;
        const int grid2 = (int)(pairs < 1 ? 1 : (pairs + it2 - 1) / (it2 < 1 ? 1 : it2));
        hipLaunchKernelGGL((router_mt_kernel<XT, MP, 0, LN, NT, EB, 2>), dim3(grid2), dim3(2 * MT_THREADS), smem2, s, (const XT*)x,
                           ln.g, ln.b, ln.eps, (NT*)ln.xn16, ln.xn32, wg, bg, noise, T, E, k, gate_kind, RC, RL, idx, score,
                           logits_out, probs);
        return;
      }
    }
    MT_LAUNCH(0, grid, RC, RL);
  };
  if constexpr (can2) SMOE_ENSURE_SMEM(router_mt_kernel<XT, MP, 0, LN, NT, EB, 2>);
  const bool ws_zero = (force_f64 & 2) != 0;
  force_f64 &= 1;
  if (force_f64 && !LN) {
    MT_LAUNCH(1, grid, nullptr, nullptr);
    SMOE_CHECK_LAUNCH("smoe_router_topk/mt f64");
    return 0;
  }
  if (!ws_zero) {
    hipError_t me = smoe_zero_words(rc, 4, s);
    if (me != hipSuccess) {
      smoe_set_error("smoe_router_topk: counter clear failed: %s", hipGetErrorString(me));
      return (int)me;
    }
  }
  launch_f32(rc, rl);
  SMOE_CHECK_LAUNCH("smoe_router_topk/mt f32");
  if (force_f64) {   // f64 mode with LayerNorm: the f32 pass above wrote the normalised rows
    MT_LAUNCH(1, grid, nullptr, nullptr);
    SMOE_CHECK_LAUNCH("smoe_router_topk/mt f64");
    return 0;
  }
  MT_LAUNCH(1, (grid < 16 ? grid : 16), rc, rl);
  SMOE_CHECK_LAUNCH("smoe_router_topk/mt redo");
#undef MT_LAUNCH
  return 0;
}

template <typename XT, bool LN, typename NT>
int dispatch16(const void* x, const LnArgs& ln, const float* wg, const float* bg, const float* noise, int64_t T, int d,
               int E, int k, int gate_kind, int f64, int32_t* rc, int32_t* rl, int64_t* idx, float* score, float* lo,
               float* pr, hipStream_t s) {
  if (E > 8 && use_router_mt()) {
    if (E > 16) {
      switch (d) {
        case 768: return launch_mt<XT, 6, LN, NT, 32>(x, ln, wg, bg, noise, T, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
        case 1024: return launch_mt<XT, 8, LN, NT, 32>(x, ln, wg, bg, noise, T, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
      }
    } else {
      switch (d) {
        case 768: return launch_mt<XT, 6, LN, NT, 16>(x, ln, wg, bg, noise, T, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
        case 1024: return launch_mt<XT, 8, LN, NT, 16>(x, ln, wg, bg, noise, T, E, k, gate_kind, f64, rc, rl, idx, score, lo, pr, s);
      }
    }
    return -1;
  }
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

// Tokens per row of the chunk histogram that smoe_ln_router_topk / smoe_gate_ln_router write for this shape when the caller hands
// them a table (smoe_dispatch_plan_hist then needs no counting pass); 0 = this shape's router writes none.
extern "C" int smoe_router_chunk_hist_tokens(int d, int E, int k) {
  return (shape_ok16(d, E, k) && E <= 8 && k >= 1 && 1024 % (R16_HIST_TOK * k) == 0) ? R16_HIST_TOK : 0;
}

// LayerNorm + router in one pass over x (block glue fusion): xn = LN(x) * gamma + beta is
// written as the 16-bit operand image (xn16, f16 or bf16; may be NULL) and / or as f32 (xn32; may be NULL) and
// routed exactly like smoe_router_topk routes xn.  Shapes: smoe_ln_router_supported(d, E, k).
extern "C" int smoe_ln_router_topk(const void* x, int x_dtype, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                   void* xn16, int xn16_dtype, float* xn32, const float* wg, const float* bg,
                                   const float* noise, int64_t T, int d, int E, int k, int gate_kind, int64_t* idx,
                                   float* score, float* logits_out, float* probs, int32_t* chunk_hist, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  const int force_f64 = ((gate_kind & 0x100) ? 1 : 0) | ((gate_kind & 0x200) ? 2 : 0);   // bit 1: the workspace's counter words are kept zero by the caller
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
  ln.hist = (chunk_hist && smoe_router_chunk_hist_tokens(d, E, k)) ? chunk_hist : nullptr;
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
