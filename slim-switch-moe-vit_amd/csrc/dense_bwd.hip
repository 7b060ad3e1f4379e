// Backward of the dense half of the ViT block for the training step (engine.py:52-74 runs the whole model forward +
// backward under autocast; models/vision_transformer.py:283-322 is the block): LayerNorm backward here; the linears'
// backward reuses the grouped GEMM (dgrad, one row group), smoe_grouped_wgrad_rows and smoe_group_colsum.
//
// LayerNorm backward (norm1 / norm2 / final norm: nn.LayerNorm(d, eps = 1e-6), models/vision_transformer.py:303-311):
//   xhat = (x - mean) rstd,  g = dy gamma,  dx = rstd (g - mean(g) - xhat mean(g xhat)) [+ dres],
//   dgamma = sum_rows dy xhat,  dbeta = sum_rows dy.
// One wave per row (a row of d <= 1024 floats = up to 16 per lane), statistics recomputed from x in registers (two-pass
// variance, as the forward kernel) -- nothing but x itself is saved by the forward.  Bound: HBM (x + dy read, dx written).
// The column sums are deterministic: every wave keeps its own partial sums in registers over the rows it walks, the
// four waves of a workgroup meet in LDS in wave order, every workgroup writes one partial row, and a second launch adds
// the partial rows in workgroup order (no atomics).
#include "smoe_common.h"
#include <type_traits>

namespace {

constexpr int LNB_THREADS = 256;
constexpr int LNB_WAVES = LNB_THREADS / 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <typename DYT, int NJ>
__global__ __launch_bounds__(LNB_THREADS) void layernorm_bwd_kernel(const float* __restrict__ x, const DYT* __restrict__ dy,
                                                                    const float* __restrict__ gamma, const float* __restrict__ dres,
                                                                    float eps, int64_t T, int d, float* __restrict__ dx,
                                                                    float* __restrict__ partial) {
  __shared__ float red[LNB_WAVES][2][NJ * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = d >> 2;
  float ag[NJ][4], ab[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) ag[j][i] = ab[j][i] = 0.f;
  f32x4 gm[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    gm[j] = (gamma && c < nchunk) ? *reinterpret_cast<const f32x4*>(gamma + c * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
  }
  const float inv_d = 1.0f / (float)d;
  const int64_t row0 = (int64_t)blockIdx.x * LNB_WAVES + wave, stride = (int64_t)gridDim.x * LNB_WAVES;
  for (int64_t t = row0; t < T; t += stride) {
    float xv[NJ][4], gv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < nchunk) {
        load4(x + t * (int64_t)d + c * 4, xv[j]);
        load4(dy + t * (int64_t)d + c * 4, gv[j]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[j][i] = gv[j][i] = 0.f;
      }
    }
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s1 += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
    const float mean = wave_sum(s1) * inv_d;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (lane + 64 * j < nchunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dv = xv[j][i] - mean; s2 = fmaf(dv, dv, s2); }
      }
    }
    const float rstd = rsqrtf(wave_sum(s2) * inv_d + eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (lane + 64 * j < nchunk) ? (xv[j][i] - mean) * rstd : 0.f;
        const float dyv = gv[j][i];
        ag[j][i] = fmaf(dyv, xh, ag[j][i]);     // dgamma
        ab[j][i] += dyv;                        // dbeta
        const float g = dyv * gm[j][i];
        c1 += g;
        c2 = fmaf(g, xh, c2);
        xv[j][i] = xh;
        gv[j][i] = g;
      }
    }
    c1 = wave_sum(c1) * inv_d;
    c2 = wave_sum(c2) * inv_d;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < nchunk) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = rstd * (gv[j][i] - c1 - xv[j][i] * c2);
        if (dres) o += *reinterpret_cast<const f32x4*>(dres + t * (int64_t)d + c * 4);
        *reinterpret_cast<f32x4*>(dx + t * (int64_t)d + c * 4) = o;
      }
    }
  }
  // the workgroup's column sums, waves added in wave order
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[wave][0][(lane + 64 * j) * 4 + i] = ag[j][i];
      red[wave][1][(lane + 64 * j) * 4 + i] = ab[j][i];
    }
  __syncthreads();
  for (int c = tid; c < 2 * d; c += LNB_THREADS) {
    const int which = c >= d, col = which ? c - d : c;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_WAVES; ++w) s += red[w][which][col];
    partial[(int64_t)blockIdx.x * 2 * d + c] = s;
  }
}

// One gated half of the residual-MoE block, backward, in ONE pass (models/resMoE.py:126-131 / 137-140 with Gate.forward's hard
// branch, 68-77): the gate's straight-through gradients (gate.hip skip_gate_bwd_kernel's arithmetic), the LayerNorm backward above
// and the gate's parameter gradients -- as three passes they wrote and re-read the [T, d] gradient of the normed activations and
// read the normed activations twice more.  Everything is recomputed from x in registers:
//     xhat = (x - mean) rstd,  xn = xhat gamma + beta,  p = sigmoid(<xn, w> + b),
//     dz = -<g_f, xn> p (1 - p),  dxn = g_f keep + g_out + dz w,              (gate_on = 0: dz = 0, dxn = g_f + g_out)
//     dx = rstd (dxn gamma - mean(dxn gamma) - xhat mean(dxn gamma xhat)),
//     dgamma = sum dxn xhat,  dbeta = sum dxn,  dw = sum dz xn,  db = sum dz.
// g_f = dL/d(the operator's masked input), g_out = dL/d(the half's output, i.e. of the residual xn), keep = mask[t, 1].
// Partial rows per workgroup: [dgamma d | dbeta d | dw d | db, 0, 0, 0]; the same two-stage reduction as the LayerNorm's.
template <typename GT, int NJ>
__global__ __launch_bounds__(LNB_THREADS) void gate_ln_bwd_kernel(const float* __restrict__ x, const GT* __restrict__ g_f,
                                                                  const float* __restrict__ g_out, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps,
                                                                  const float* __restrict__ gate_w, const float* __restrict__ gate_b,
                                                                  const float* __restrict__ mask, int gate_on, int64_t T, int d,
                                                                  float* __restrict__ dx, float* __restrict__ dz_out,
                                                                  float* __restrict__ partial) {
  __shared__ float red[LNB_WAVES][3][NJ * 256];
  __shared__ float red_b[LNB_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = d >> 2;
  const int L = 3 * d + 4;
  float ag[NJ][4], ab[NJ][4], aw[NJ][4];
  float agb = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) ag[j][i] = ab[j][i] = aw[j][i] = 0.f;
  f32x4 gm[NJ], bt[NJ], wv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    const bool in = c < nchunk;
    gm[j] = (gamma && in) ? *reinterpret_cast<const f32x4*>(gamma + c * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
    bt[j] = (beta && in) ? *reinterpret_cast<const f32x4*>(beta + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    wv[j] = in ? *reinterpret_cast<const f32x4*>(gate_w + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float bias = (gate_b && gate_on) ? *gate_b : 0.f;
  const float inv_d = 1.0f / (float)d;
  const int64_t row0 = (int64_t)blockIdx.x * LNB_WAVES + wave, stride = (int64_t)gridDim.x * LNB_WAVES;
  for (int64_t t = row0; t < T; t += stride) {
    float xv[NJ][4], gv[NJ][4], go[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < nchunk) {
        load4(x + t * (int64_t)d + c * 4, xv[j]);
        load4(g_f + t * (int64_t)d + c * 4, gv[j]);
        if (g_out) load4(g_out + t * (int64_t)d + c * 4, go[j]);
        else go[j][0] = go[j][1] = go[j][2] = go[j][3] = 0.f;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[j][i] = gv[j][i] = go[j][i] = 0.f;
      }
    }
    const float keep = gate_on ? mask[t * 2 + 1] : 1.f;
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s1 += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
    const float mean = wave_sum(s1) * inv_d;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (lane + 64 * j < nchunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dv = xv[j][i] - mean; s2 = fmaf(dv, dv, s2); }
      }
    }
    const float rstd = rsqrtf(wave_sum(s2) * inv_d + eps);
    // the gate: logit and <g_f, xn> on the recomputed normed row (xv becomes xhat, xn kept beside it)
    float xn[NJ][4];
    float az = 0.f, ad = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool in = lane + 64 * j < nchunk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = in ? (xv[j][i] - mean) * rstd : 0.f;
        xv[j][i] = xh;
        xn[j][i] = in ? fmaf(xh, gm[j][i], bt[j][i]) : 0.f;
        az = fmaf(xn[j][i], wv[j][i], az);
        ad = fmaf(gv[j][i], xn[j][i], ad);
      }
    }
    const float z = wave_sum(az) + bias, dot = wave_sum(ad);
    const float p = 1.0f / (1.0f + expf(-z));
    const float dz = gate_on ? -dot * p * (1.0f - p) : 0.f;
    agb += dz;
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dyv = fmaf(gv[j][i], keep, fmaf(dz, wv[j][i], go[j][i]));     // dxn
        const float xh = xv[j][i];
        ag[j][i] = fmaf(dyv, xh, ag[j][i]);
        ab[j][i] += dyv;
        aw[j][i] = fmaf(dz, xn[j][i], aw[j][i]);
        const float g = dyv * gm[j][i];
        c1 += g;
        c2 = fmaf(g, xh, c2);
        gv[j][i] = g;
      }
    }
    c1 = wave_sum(c1) * inv_d;
    c2 = wave_sum(c2) * inv_d;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < nchunk) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = rstd * (gv[j][i] - c1 - xv[j][i] * c2);
        *reinterpret_cast<f32x4*>(dx + t * (int64_t)d + c * 4) = o;
      }
    }
    if (lane == 0 && dz_out) dz_out[t] = dz;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[wave][0][(lane + 64 * j) * 4 + i] = ag[j][i];
      red[wave][1][(lane + 64 * j) * 4 + i] = ab[j][i];
      red[wave][2][(lane + 64 * j) * 4 + i] = aw[j][i];
    }
  if (lane == 0) red_b[wave] = agb;
  __syncthreads();
  for (int c = tid; c < 3 * d; c += LNB_THREADS) {
    const int which = c / d, col = c - which * d;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_WAVES; ++w) s += red[w][which][col];
    partial[(int64_t)blockIdx.x * L + c] = s;
  }
  if (tid < 4) {
    float s = 0.f;
    if (tid == 0) {
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) s += red_b[w];
    }
    partial[(int64_t)blockIdx.x * L + 3 * d + tid] = s;
  }
}

// dgamma_dbeta[c] (c < 2 d: dgamma then dbeta) = sum over the workgroups' partial rows.  A workgroup takes 32 columns (one
// 128-byte segment of every partial row); thread (column c, slice r of 8) adds rows r, r + 8, ... in order, the eight slices meet
// in LDS in slice order: deterministic, and the 6-MB table is read at streaming rate (the first version gave every column to one
// thread walking all 1,024 rows: 226 us per call, 10 % of the training step).
// Two stages (LNB_STAGE1 = 16 row groups, then one): with the whole table behind 48 workgroups the second version still took 38 us
// per call -- 128 dependent-latency trips per thread on a fifth of the chip (profiles/r03_dispatch_kernels.md).
constexpr int LNB_STAGE1 = 16;
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ partial, int n_rows, int two_d,
                                                                   int rows_per_block, float* __restrict__ out) {
  __shared__ float red[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
  const int row0 = blockIdx.y * rows_per_block;
  int row1 = row0 + rows_per_block;
  if (row1 > n_rows) row1 = n_rows;
  float s = 0.f;
  if (c < two_d)
    for (int row = row0 + r; row < row1; row += 8) s += partial[(int64_t)row * two_d + c];
  red[r][threadIdx.x & 31] = s;
  __syncthreads();
  if (r == 0 && c < two_d) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][threadIdx.x & 31];
    out[(int64_t)blockIdx.y * two_d + c] = t;
  }
}

inline int lnb_grid(int64_t T) {
  int64_t need = (T + LNB_WAVES - 1) / LNB_WAVES;
  const int64_t cap = (int64_t)smoe_num_cus() * 4;      // 16 waves per CU: every lane keeps 2 x NJ 16-byte loads in flight
  if (need > cap) need = cap;
  return (int)(need < 1 ? 1 : need);
}

template <typename DYT>
int lnb_launch(const float* x, const void* dy, const float* gamma, const float* dres, float eps, int64_t T, int d, float* dx,
               float* partial, float* dgamma_dbeta, hipStream_t s) {
  const int grid = lnb_grid(T);
#define LNB(NJ) hipLaunchKernelGGL((layernorm_bwd_kernel<DYT, NJ>), dim3(grid), dim3(LNB_THREADS), 0, s, x, (const DYT*)dy, gamma, dres, eps, T, d, dx, partial)
  const int nj = (d / 4 + 63) / 64;
  switch (nj) {
    case 1: LNB(1); break;
    case 2: LNB(2); break;
    case 3: LNB(3); break;
    case 4: LNB(4); break;
    default: smoe_set_error("smoe_layernorm_bwd: d=%d out of range (d <= 1024)", d); return 1;
  }
#undef LNB
  SMOE_CHECK_LAUNCH("smoe_layernorm_bwd");
  float* stage1 = partial + (size_t)grid * 2 * d;        // [LNB_STAGE1][2 d] behind the per-workgroup rows
  const int rpb = (grid + LNB_STAGE1 - 1) / LNB_STAGE1;
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * d + 31) / 32, LNB_STAGE1), dim3(256), 0, s, partial, grid, 2 * d, rpb,
                     stage1);
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * d + 31) / 32, 1), dim3(256), 0, s, stage1, LNB_STAGE1, 2 * d, LNB_STAGE1,
                     dgamma_dbeta);
  SMOE_CHECK_LAUNCH("smoe_layernorm_bwd/reduce");
  return 0;
}

template <typename GT>
int glnb_launch(const float* x, const void* g_f, const float* g_out, const float* gamma, const float* beta, float eps,
                const float* gate_w, const float* gate_b, const float* mask, int gate_on, int64_t T, int d, float* dx, float* dz,
                float* partial, float* out, hipStream_t s) {
  const int grid = lnb_grid(T);
  const int L = 3 * d + 4;
#define GLNB(NJ) hipLaunchKernelGGL((gate_ln_bwd_kernel<GT, NJ>), dim3(grid), dim3(LNB_THREADS), 0, s, x, (const GT*)g_f, g_out, gamma, beta, eps, gate_w, gate_b, mask, gate_on, T, d, dx, dz, partial)
  const int nj = (d / 4 + 63) / 64;
  switch (nj) {
    case 1: GLNB(1); break;
    case 2: GLNB(2); break;
    case 3: GLNB(3); break;
    case 4: GLNB(4); break;
    default: smoe_set_error("smoe_gate_ln_bwd: d=%d out of range (d <= 1024)", d); return 1;
  }
#undef GLNB
  SMOE_CHECK_LAUNCH("smoe_gate_ln_bwd");
  float* stage1 = partial + (size_t)grid * L;
  const int rpb = (grid + LNB_STAGE1 - 1) / LNB_STAGE1;
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((L + 31) / 32, LNB_STAGE1), dim3(256), 0, s, partial, grid, L, rpb, stage1);
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((L + 31) / 32, 1), dim3(256), 0, s, stage1, LNB_STAGE1, L, LNB_STAGE1, out);
  SMOE_CHECK_LAUNCH("smoe_gate_ln_bwd/reduce");
  return 0;
}

// ---- router input gradient: dx[t, :] = dl[t, :] W  (dl [T, E] f32 = d loss / d logits, W [E, d] f32: gate.gate.weight) ----------
// The backward of the gate's nn.Linear w.r.t. its input (models/resmoe_flop_hook.py:7 names that layer): K = E <= 64 is far too
// thin for a matrix-core GEMM -- the kernel is the [T, d] store.  One thread = 4 consecutive columns of one row; the E weights of
// those columns come from L1 (W is E x d floats, shared by every row), the row's E gradients are wave-broadcast loads.
template <typename OT> __device__ __forceinline__ void gd_store4(OT* dst, const f32x4& acc) {
  if constexpr (std::is_same<OT, float>::value) {
    *reinterpret_cast<f32x4*>(dst) = acc;
  } else if constexpr (std::is_same<OT, f16>::value) {
    f16x4 v; v[0] = (f16)acc[0]; v[1] = (f16)acc[1]; v[2] = (f16)acc[2]; v[3] = (f16)acc[3];
    *reinterpret_cast<f16x4*>(dst) = v;
  } else {
    s16x4 v; v[0] = (short)f32_to_bf16(acc[0]); v[1] = (short)f32_to_bf16(acc[1]); v[2] = (short)f32_to_bf16(acc[2]); v[3] = (short)f32_to_bf16(acc[3]);
    *reinterpret_cast<s16x4*>(dst) = v;
  }
}
// A thread owns 4 columns and keeps their EB weights in registers; the workgroup walks rows, so a row's gradients are
// workgroup-uniform (scalar loads) and the loop body is EB x 4 FMAs and one 16-byte store: the kernel is its [T, d] store.
// (The first version re-read the weights from L1 for every row -- 8 vector loads per store, 69 us for the 155-MB store at
// ViT-B -- and paid a 64-bit division per element; profiles/r03_dispatch_kernels.md.)
template <typename OT, int EB>
__global__ __launch_bounds__(256) void gate_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ w, int64_t T, int E,
                                                          int d, OT* __restrict__ out) {
  const int nchunk = d >> 2;
  for (int cb = 0; cb < nchunk; cb += blockDim.x) {        // (one trip for d <= 1024)
    const int c = (cb + (int)threadIdx.x) * 4;
    const bool live = c < d;
    f32x4 wv[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e)
      wv[e] = (live && e < E) ? *reinterpret_cast<const f32x4*>(w + (int64_t)e * d + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
      const float* dr = dl + t * E;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const float g = e < E ? dr[e] : 0.f;
        acc[0] = fmaf(g, wv[e][0], acc[0]); acc[1] = fmaf(g, wv[e][1], acc[1]);
        acc[2] = fmaf(g, wv[e][2], acc[2]); acc[3] = fmaf(g, wv[e][3], acc[3]);
      }
      if (live) gd_store4<OT>(out + t * (int64_t)d + c, acc);
    }
  }
}
// any E: weights from L1 per row (rows per workgroup as above, no per-element division)
template <typename OT>
__global__ __launch_bounds__(256) void gate_dgrad_any_kernel(const float* __restrict__ dl, const float* __restrict__ w, int64_t T,
                                                              int E, int d, OT* __restrict__ out) {
  const int nchunk = d >> 2;
  for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
    const float* dr = dl + t * E;
    for (int ci = threadIdx.x; ci < nchunk; ci += blockDim.x) {
      const int c = ci * 4;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int e = 0; e < E; ++e) {
        const float g = dr[e];
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (int64_t)e * d + c);
        acc[0] = fmaf(g, wv[0], acc[0]); acc[1] = fmaf(g, wv[1], acc[1]); acc[2] = fmaf(g, wv[2], acc[2]); acc[3] = fmaf(g, wv[3], acc[3]);
      }
      gd_store4<OT>(out + t * (int64_t)d + c, acc);
    }
  }
}

template <typename OT>
void gate_dgrad_launch(const float* dl, const float* w, int64_t T, int E, int d, OT* out, hipStream_t s) {
  const int nchunk = d / 4;
  int threads = ((nchunk < 256 ? nchunk : 256) + 63) / 64 * 64;
  int64_t blocks = (int64_t)smoe_num_cus() * 16;          // rows are walked with this stride: plenty of stores in flight per CU
  if (blocks > T) blocks = T;
  if (E <= 8) hipLaunchKernelGGL((gate_dgrad_kernel<OT, 8>), dim3((int)blocks), dim3(threads), 0, s, dl, w, T, E, d, out);
  else if (E <= 16) hipLaunchKernelGGL((gate_dgrad_kernel<OT, 16>), dim3((int)blocks), dim3(threads), 0, s, dl, w, T, E, d, out);
  else hipLaunchKernelGGL((gate_dgrad_any_kernel<OT>), dim3((int)blocks), dim3(threads), 0, s, dl, w, T, E, d, out);
}

}  // namespace

extern "C" int smoe_gate_dgrad(const float* dl, const float* w, int64_t T, int E, int d, void* out, int out_dtype, void* stream) {
  SMOE_REQUIRE(T >= 0 && E >= 1 && d > 0 && d % 4 == 0, "smoe_gate_dgrad: bad sizes T=%lld E=%d d=%d", (long long)T, E, d);
  if (T == 0) return 0;
  SMOE_REQUIRE(dl && w && out && smoe_dtype_ok(out_dtype), "smoe_gate_dgrad: null pointer / bad dtype");
  hipStream_t s = (hipStream_t)stream;
  switch (out_dtype) {
    case SMOE_F32: gate_dgrad_launch<float>(dl, w, T, E, d, (float*)out, s); break;
    case SMOE_F16: gate_dgrad_launch<f16>(dl, w, T, E, d, (f16*)out, s); break;
    default: gate_dgrad_launch<bf16_bits>(dl, w, T, E, d, (bf16_bits*)out, s); break;
  }
  SMOE_CHECK_LAUNCH("smoe_gate_dgrad");
  return 0;
}

extern "C" size_t smoe_gate_ln_bwd_workspace_bytes(int64_t T, int d) {
  if (T < 0 || d <= 0) return 0;
  return ((size_t)lnb_grid(T) + LNB_STAGE1) * (3 * (size_t)d + 4) * sizeof(float);
}

// see gate_ln_bwd_kernel.  x f32 [T, d] (the half's input), g_f [T, d] (f32 / f16 / bf16), g_out f32 [T, d] or NULL, gamma / beta f32 [d]
// (NULL = 1 / 0), gate_w f32 [d], gate_b f32 [1] or NULL, mask f32 [T, 2] (the forward's decisions; needed when gate_on);
// dx f32 [T, d]; out f32 [3 d + 4] = dgamma | dbeta | dgate_w | dgate_b, 0, 0, 0; dz f32 [T] or NULL (d loss / d gate logit).
extern "C" int smoe_gate_ln_bwd(const float* x, const void* g_f, int g_f_dtype, const float* g_out, const float* gamma,
                                const float* beta, float eps, const float* gate_w, const float* gate_b, const float* mask,
                                int gate_on, int64_t T, int d, float* dx, float* dz, float* out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  SMOE_REQUIRE(T >= 0 && d > 0 && d % 4 == 0 && d <= 1024, "smoe_gate_ln_bwd: bad sizes T=%lld d=%d (d %% 4 == 0, d <= 1024)",
               (long long)T, d);
  SMOE_REQUIRE(out && workspace && gate_w, "smoe_gate_ln_bwd: null pointer");
  SMOE_REQUIRE(workspace_bytes >= smoe_gate_ln_bwd_workspace_bytes(T, d), "smoe_gate_ln_bwd: workspace too small");
  SMOE_REQUIRE(T == 0 || (x && g_f && dx), "smoe_gate_ln_bwd: null pointer");
  SMOE_REQUIRE(!gate_on || mask, "smoe_gate_ln_bwd: an enabled gate needs the forward's decisions (mask)");
  SMOE_REQUIRE(smoe_dtype_ok(g_f_dtype), "smoe_gate_ln_bwd: bad g_f dtype");
  hipStream_t s = (hipStream_t)stream;
  float* partial = reinterpret_cast<float*>(workspace);
  switch (g_f_dtype) {
    case SMOE_F32: return glnb_launch<float>(x, g_f, g_out, gamma, beta, eps, gate_w, gate_b, mask, gate_on, T, d, dx, dz, partial, out, s);
    case SMOE_F16: return glnb_launch<f16>(x, g_f, g_out, gamma, beta, eps, gate_w, gate_b, mask, gate_on, T, d, dx, dz, partial, out, s);
    default: return glnb_launch<bf16_bits>(x, g_f, g_out, gamma, beta, eps, gate_w, gate_b, mask, gate_on, T, d, dx, dz, partial, out, s);
  }
}

extern "C" size_t smoe_layernorm_bwd_workspace_bytes(int64_t T, int d) {
  if (T < 0 || d <= 0) return 0;
  return ((size_t)lnb_grid(T) + LNB_STAGE1) * 2 * (size_t)d * sizeof(float);
}

extern "C" int smoe_layernorm_bwd(const float* x, const void* dy, int dy_dtype, const float* gamma, const float* dres, float eps,
                                  int64_t T, int d, float* dx, float* dgamma_dbeta, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  SMOE_REQUIRE(T >= 0 && d > 0 && d % 4 == 0 && d <= 1024, "smoe_layernorm_bwd: bad sizes T=%lld d=%d (d %% 4 == 0, d <= 1024)",
               (long long)T, d);
  SMOE_REQUIRE(dgamma_dbeta && workspace, "smoe_layernorm_bwd: null pointer");
  SMOE_REQUIRE(workspace_bytes >= smoe_layernorm_bwd_workspace_bytes(T, d), "smoe_layernorm_bwd: workspace too small");
  SMOE_REQUIRE(T == 0 || (x && dy && dx), "smoe_layernorm_bwd: null pointer");
  SMOE_REQUIRE(smoe_dtype_ok(dy_dtype), "smoe_layernorm_bwd: bad dy dtype");
  hipStream_t s = (hipStream_t)stream;
  float* partial = reinterpret_cast<float*>(workspace);
  switch (dy_dtype) {
    case SMOE_F32: return lnb_launch<float>(x, dy, gamma, dres, eps, T, d, dx, partial, dgamma_dbeta, s);
    case SMOE_F16: return lnb_launch<f16>(x, dy, gamma, dres, eps, T, d, dx, partial, dgamma_dbeta, s);
    default: return lnb_launch<bf16_bits>(x, dy, gamma, dres, eps, T, d, dx, partial, dgamma_dbeta, s);
  }
}
