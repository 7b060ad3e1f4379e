// Backward-side helpers of the MoE operator (BASELINE cfg 5; SURVEY.md N5, Appendix B 'backward'):
//   smoe_gelu            A = gelu(H)                      (training forward keeps the pre-activation)
//   smoe_rowdot          dscore[i] = <dout[i/k], y[inv_pos[i]]>
//   smoe_pad_offsets / smoe_transpose_pad
//                        expert-sorted rows [n,C] -> K-major image [C, Lp] whose per-expert column ranges
//                        start on multiples of 64 (zero padded) = the operand layout of the wgrad GEMM
//   smoe_group_colsum    bias gradients: out[e,c] = sum of rows of expert e
#include "smoe_common.h"
#include <type_traits>

namespace {

__device__ __forceinline__ float gelu_fwd(float v) {
  const float a = fabsf(v);
  const float z = a * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float ex = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  return 0.5f * fmaf(a, fmaf(-p, ex, 1.0f), v);
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i + 8 <= n; i += stride) {
    float v[8];
    load8(src + i, v);
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = gelu_fwd(v[q]);
    store8(dst + i, v);
  }
}

template <typename DT, typename YT>
__global__ __launch_bounds__(256) void rowdot_kernel(const DT* __restrict__ dout, const YT* __restrict__ y,
                                                     const int64_t* __restrict__ inv_pos, int64_t n, int k, int d,
                                                     float* __restrict__ dscore) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave_gid; i < n; i += nwaves) {
    const int64_t slot = inv_pos[i];
    float acc = 0.f;
    if (slot >= 0) {
      for (int c = lane * 8; c < d; c += 512) {
        float a[8], b[8];
        load8(dout + (i / k) * (int64_t)d + c, a);
        load8(y + slot * (int64_t)d + c, b);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(a[q], b[q], acc);
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) dscore[i] = acc;
  }
}

// offsets_pad[e] = sum_{e'<e} round_up(offsets[e'+1]-offsets[e'], 64)
__global__ void pad_offsets_kernel(const int32_t* __restrict__ offsets, int E, int32_t* __restrict__ offsets_pad) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int run = 0;
    for (int e = 0; e < E; ++e) {
      offsets_pad[e] = run;
      run += ((offsets[e + 1] - offsets[e]) + 63) & ~63;
    }
    offsets_pad[E] = run;
  }
}

// 64 x 64 tiles: block (cb, rb) transposes padded columns [64 rb, +64) x source columns [64 cb, +64).
// The grid covers the whole padded length Lp, so pad columns are written as zeros.
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ src, const int32_t* __restrict__ offsets,
                                                            const int32_t* __restrict__ offsets_pad, int E, int C, int Lp,
                                                            T* __restrict__ dst) {
  __shared__ T tile[64][66];
  const int p0 = blockIdx.y * 64;  // padded-column tile
  const int c0 = blockIdx.x * 64;
  // padded ranges start on multiples of 64, so a 64-wide padded tile lies inside one expert's range
  int e = -1, src_row0 = 0, valid = 0;
  for (int q = 0; q < E; ++q) {
    const int lo = offsets_pad[q], hi = offsets_pad[q + 1];
    if (p0 >= lo && p0 < hi) {
      e = q;
      src_row0 = offsets[q] + (p0 - lo);
      const int left = offsets[q + 1] - src_row0;
      valid = left < 0 ? 0 : (left > 64 ? 64 : left);
      break;
    }
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row groups
  for (int r = ty; r < 64; r += 4) {
    T v = (T)0;
    if (e >= 0 && r < valid && c0 + tx < C) v = src[(int64_t)(src_row0 + r) * C + c0 + tx];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    if (c0 + c < C && p0 + tx < Lp) dst[(int64_t)(c0 + c) * Lp + p0 + tx] = tile[tx][c];
  }
}

// Batched matrix transpose with a cast: dst[b][c][r] = (DT) src[b][r][c], R % 64 == 0, C % 64 == 0.  One 64 x 64 tile per
// workgroup: 16-byte / 8-byte loads along c, f32 tile in LDS (row pitch 65 words: the transposed reads walk 65-word steps,
// conflict-free), stores of four consecutive r.  The backward pass wants every expert weight [E, out, in] a second time as
// [E, in, out] (the dgrad GEMMs contract over `out`): straight from the f32 master instead of a strided torch copy of the
// 16-bit shadow (89 us for ViT-B's 37.7 MB; this pass: one read of the master, one 16-bit write).
template <typename ST, typename DT>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const ST* __restrict__ src, DT* __restrict__ dst, int R, int C) {
  __shared__ float tile[64][65];
  const int64_t mat = (int64_t)blockIdx.z * R * C;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x, tc = (tid & 15) * 4, tr = tid >> 4;   // 16 threads x 4 elements across, 16 rows per pass
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = tr + 16 * i;
    float v[4];
    load4(src + mat + (int64_t)(r0 + r) * C + c0 + tc, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[r][tc + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tr + 16 * i;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = tile[tc + j][c];
    DT* o = dst + mat + (int64_t)(c0 + c) * R + r0 + tc;
    if constexpr (std::is_same<DT, float>::value) {
      *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    } else if constexpr (std::is_same<DT, f16>::value) {
      f16x4 h; h[0] = (f16)v[0]; h[1] = (f16)v[1]; h[2] = (f16)v[2]; h[3] = (f16)v[3];
      *reinterpret_cast<f16x4*>(o) = h;
    } else {
      s16x4 h; h[0] = (short)f32_to_bf16(v[0]); h[1] = (short)f32_to_bf16(v[1]); h[2] = (short)f32_to_bf16(v[2]); h[3] = (short)f32_to_bf16(v[3]);
      *reinterpret_cast<s16x4*>(o) = h;
    }
  }
}

// SwitchGate backward in one pass over [T, E] (fmoe.gates.SwitchGate: score = softmax(logits)[idx], aux = E sum_e frac_e prob_e):
//   g[t, e] = coef[e] + (e == idx[t] ? dscore[t] : 0)          coef[e] = d aux / d p[t, e] = daux * E * frac_e / kept (any t)
//   dlogits[t, e] = p[t, e] * (g[t, e] - sum_j p[t, j] g[t, j])                                     (softmax backward)
// One thread per token; the row (E <= 64 floats) is read twice from L1.
// coef_scale (device scalar, optional): the loss's gradient w.r.t. aux, multiplied onto coef here (not by a host-launched [E] multiply)
__global__ __launch_bounds__(256) void switch_gate_bwd_kernel(const float* __restrict__ probs, const int64_t* __restrict__ idx,
                                                              const float* __restrict__ dscore, const float* __restrict__ coef,
                                                              const float* __restrict__ coef_scale,
                                                              int64_t T, int E, float* __restrict__ dlogits) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float* p = probs + t * E;
  const int64_t sel = idx[t];
  const float ds = dscore ? dscore[t] : 0.f;
  const float cs = coef_scale ? *coef_scale : 1.0f;
  float dot = 0.f;
  for (int e = 0; e < E; ++e) dot = fmaf(p[e], (coef ? coef[e] * cs : 0.f) + (e == sel ? ds : 0.f), dot);
  float* o = dlogits + t * E;
  for (int e = 0; e < E; ++e) o[e] = p[e] * ((coef ? coef[e] * cs : 0.f) + (e == sel ? ds : 0.f) - dot);
}

// SwitchGate load-balance loss, forward (fmoe.gates.SwitchGate): aux = E sum_e frac_e prob_e with
// frac_e = counts[e] / kept, prob_e = sum_t p[t, e] / kept, kept = max(sum_e counts[e], 1); and coef[e] = E frac_e / kept =
// d aux / d p[t, e] (any t) for the backward.  Two deterministic launches instead of ~14 tiny host-launched ones: column sums of
// p [T, E] per SA_ROWS-row chunk (thread = (row of the pass, column), coalesced), then ONE workgroup adds the chunks in order.
constexpr int SA_ROWS = 1024;
constexpr int SA_MAX_E = 256;

__global__ __launch_bounds__(256) void switch_aux_partial_kernel(const float* __restrict__ probs, int64_t T, int E,
                                                                 float* __restrict__ partial) {
  __shared__ float red[256];
  const int rpp = 256 / E;                       // rows per pass (E <= 256)
  const int rr = threadIdx.x / E, e = threadIdx.x - rr * E;
  const int64_t r0 = (int64_t)blockIdx.x * SA_ROWS;
  const int64_t r1 = r0 + SA_ROWS < T ? r0 + SA_ROWS : T;
  float a0 = 0.f, a1 = 0.f;
  if (rr < rpp) {
    int64_t r = r0 + rr;
    for (; r + rpp < r1; r += 2 * rpp) {
      a0 += probs[r * E + e];
      a1 += probs[(r + rpp) * E + e];
    }
    if (r < r1) a0 += probs[r * E + e];
  }
  red[threadIdx.x] = a0 + a1;
  __syncthreads();
  if (threadIdx.x < E) {
    float acc = 0.f;
    for (int q = 0; q < rpp; ++q) acc += red[q * E + threadIdx.x];
    partial[(int64_t)blockIdx.x * E + threadIdx.x] = acc;
  }
}

__global__ __launch_bounds__(256) void switch_aux_final_kernel(const float* __restrict__ partial, int n_chunks,
                                                               const int32_t* __restrict__ counts, int E,
                                                               float* __restrict__ aux, float* __restrict__ coef) {
  __shared__ float term[SA_MAX_E];
  __shared__ float kept_s;
  const int e = threadIdx.x;
  if (e == 0) {
    int64_t k = 0;
    for (int q = 0; q < E; ++q) k += counts[q];
    kept_s = (float)(k < 1 ? 1 : k);
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < E) {
    int c = 0;
    for (; c + 3 < n_chunks; c += 4) {
      a0 += partial[(int64_t)c * E + e];
      a1 += partial[(int64_t)(c + 1) * E + e];
      a2 += partial[(int64_t)(c + 2) * E + e];
      a3 += partial[(int64_t)(c + 3) * E + e];
    }
    for (; c < n_chunks; ++c) a0 += partial[(int64_t)c * E + e];
  }
  __syncthreads();
  const float kept = kept_s;
  if (e < E) {
    const float frac = (float)counts[e] / kept;
    term[e] = frac * (((a0 + a1) + (a2 + a3)) / kept);
    coef[e] = (float)E * frac / kept;
  }
  __syncthreads();
  if (e == 0) {
    float acc = 0.f;
    for (int q = 0; q < E; ++q) acc += term[q];
    *aux = (float)E * acc;
  }
}

// Bias gradients, two deterministic passes.  Pass 1: one workgroup of 16 waves per (CS_ROWS-row chunk of one group, slab of 64 x VEC
// columns: VEC = 8 elements = one 16-byte load per lane for 16-bit rows, 4 for f32); wave w sums the chunk's rows w, w+16, ...
// with four rows in flight per lane, the sixteen waves meet in LDS (fixed order) and the workgroup stores one f32 partial row.
// The grid is an upper bound (the row counts live on the device); chunks are numbered group by group, so pass 2 adds each
// group's partial rows in chunk order (four waves x two chains per 64-column slab).  [25216, 768] f16 = 99 chunks x 2 slabs of
// 1024 threads: every CU busy, which the 256-thread / 512-row form (150 workgroups) was not.
constexpr int CS_ROWS = 256;
constexpr int CS_COLS = 256;    // slab of the router weight gradient below
constexpr int CS_WAVES = 16;

// (ends: optional i32 [E] -- group e's rows are [offsets[e], ends[e]) instead of [offsets[e], offsets[e + 1]): the slots of a static
//  expert exchange, whose padding rows are never read)
__device__ __forceinline__ bool colsum_find_chunk(const int32_t* __restrict__ offsets, const int32_t* __restrict__ ends, int E,
                                                  int chunk, int& e_out, int& r0, int& r1) {
  int base = 0;
  for (int e = 0; e < E; ++e) {
    const int lo = offsets[e], hi = ends ? ends[e] : offsets[e + 1];
    const int nc = (hi - lo + CS_ROWS - 1) / CS_ROWS;
    if (chunk < base + nc) {
      e_out = e;
      r0 = lo + (chunk - base) * CS_ROWS;
      r1 = r0 + CS_ROWS < hi ? r0 + CS_ROWS : hi;
      return true;
    }
    base += nc;
  }
  return false;
}

template <typename T, int VEC>
__global__ __launch_bounds__(64 * CS_WAVES) void group_colsum_partial_kernel(const T* __restrict__ src,
                                                                             const int32_t* __restrict__ offsets,
                                                                             const int32_t* __restrict__ ends, int E, int C,
                                                                             float* __restrict__ partial) {
  static_assert(VEC == 4 || (VEC == 8 && !std::is_same<T, float>::value), "16-byte loads at most");
  constexpr int SLAB = 64 * VEC;
  __shared__ float red[CS_WAVES][SLAB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * SLAB + lane * VEC;
  int e, r0, r1;
  const bool have = colsum_find_chunk(offsets, ends, E, (int)blockIdx.y, e, r0, r1);
  float a[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) a[q] = 0.f;
  if (have && c < C) {  // C % VEC == 0 (checked on the host)
    auto ld = [&](int row, float (&v)[VEC]) {
      const T* p = src + (int64_t)row * C + c;
      if constexpr (std::is_same<T, float>::value) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
      } else if constexpr (VEC == 8) {
        load8(p, v);
      } else {
        load4(p, v);
      }
    };
    int r = r0 + wave;
    for (; r + 3 * CS_WAVES < r1; r += 4 * CS_WAVES) {  // four rows in flight per lane
      float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
      ld(r, v0); ld(r + CS_WAVES, v1); ld(r + 2 * CS_WAVES, v2); ld(r + 3 * CS_WAVES, v3);
#pragma unroll
      for (int q = 0; q < VEC; ++q) a[q] += (v0[q] + v1[q]) + (v2[q] + v3[q]);
    }
    for (; r < r1; r += CS_WAVES) {
      float v0[VEC];
      ld(r, v0);
#pragma unroll
      for (int q = 0; q < VEC; ++q) a[q] += v0[q];
    }
  }
#pragma unroll
  for (int q = 0; q < VEC; ++q) red[wave][lane * VEC + q] = a[q];
  __syncthreads();
  if ((int)threadIdx.x < SLAB) {
    const int cc = blockIdx.x * SLAB + threadIdx.x;
    if (cc < C) {
      float s4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        s4[g] = (red[4 * g][threadIdx.x] + red[4 * g + 1][threadIdx.x]) + (red[4 * g + 2][threadIdx.x] + red[4 * g + 3][threadIdx.x]);
      partial[(int64_t)blockIdx.y * C + cc] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
  }
}

// one workgroup per (64-column slab, group): wave w adds the group's chunks w, w+4, ... (lane = column; two chains), the four meet in LDS
__global__ __launch_bounds__(256) void group_colsum_final_kernel(const float* __restrict__ partial,
                                                                 const int32_t* __restrict__ offsets,
                                                                 const int32_t* __restrict__ ends, int C,
                                                                 float* __restrict__ out) {
  __shared__ float red[4][64];
  const int e = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  int base = 0;
  for (int q = 0; q < e; ++q) base += ((ends ? ends[q] : offsets[q + 1]) - offsets[q] + CS_ROWS - 1) / CS_ROWS;
  const int nc = ((ends ? ends[e] : offsets[e + 1]) - offsets[e] + CS_ROWS - 1) / CS_ROWS;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    int k = wave;
    for (; k + 4 < nc; k += 8) {
      a0 += partial[(int64_t)(base + k) * C + c];
      a1 += partial[(int64_t)(base + k + 4) * C + c];
    }
    if (k < nc) a0 += partial[(int64_t)(base + k) * C + c];
  }
  red[wave][lane] = a0 + a1;
  __syncthreads();
  if (wave == 0 && c < C) out[(int64_t)e * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// Router weight gradient dWg[e, c] = sum_t dl[t, e] * x[t, c]: a [E x T] x [T x d] product whose output is tiny (E <= 16
// rows) and whose K dimension is the token count -- an HBM-bound weighted column sum, not a GEMM (a library GEMM picks a
// 32x16 tile and needs 180 us for 155 MB).  Same two deterministic passes as the bias gradients: 512-row chunk x
// 256-column slab partials, then a sum over chunks.
constexpr int GW_MAX_E = 16;
constexpr int GW_ROWS = 256;  // rows per chunk: ~600 workgroups at T = 50k, all resident

template <typename T, int EB>
__global__ __launch_bounds__(256) void gate_wgrad_partial_kernel(const float* __restrict__ dl, const T* __restrict__ x,
                                                                 int64_t n_rows, int E, int C, float* __restrict__ partial,
                                                                 float* __restrict__ bias_partial) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);  // [4 waves][EB][CS_COLS]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * CS_COLS + lane * 4;
  const int64_t r0 = (int64_t)blockIdx.y * GW_ROWS;
  const int64_t r1 = r0 + GW_ROWS < n_rows ? r0 + GW_ROWS : n_rows;
  float acc[EB][4];
  float bacc[EB];   // the bias gradient dbg[e] = sum_t dl[t, e] (x's "ones column"): the first slab's workgroups carry it
  const bool with_bias = bias_partial != nullptr && blockIdx.x == 0;
#pragma unroll
  for (int e = 0; e < EB; ++e) acc[e][0] = acc[e][1] = acc[e][2] = acc[e][3] = bacc[e] = 0.f;
  if (c < C) {
    auto row = [&](int64_t r, const float (&v)[4]) {
      const float* g = dl + r * E;
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const float w = e < E ? g[e] : 0.f;
        if (with_bias) bacc[e] += w;
        acc[e][0] = fmaf(w, v[0], acc[e][0]);
        acc[e][1] = fmaf(w, v[1], acc[e][1]);
        acc[e][2] = fmaf(w, v[2], acc[e][2]);
        acc[e][3] = fmaf(w, v[3], acc[e][3]);
      }
    };
    int64_t r = r0 + wave;
    for (; r + 12 < r1; r += 16) {  // four rows of x in flight per lane
      float v0[4], v1[4], v2[4], v3[4];
      load4(x + r * C + c, v0);
      load4(x + (r + 4) * C + c, v1);
      load4(x + (r + 8) * C + c, v2);
      load4(x + (r + 12) * C + c, v3);
      row(r, v0); row(r + 4, v1); row(r + 8, v2); row(r + 12, v3);
    }
    for (; r < r1; r += 4) {
      float v0[4];
      load4(x + r * C + c, v0);
      row(r, v0);
    }
  }
#pragma unroll
  for (int e = 0; e < EB; ++e)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * EB + e) * CS_COLS + lane * 4 + i] = acc[e][i];
  __syncthreads();
  const int cc = blockIdx.x * CS_COLS + threadIdx.x;
  if (cc < C) {
    for (int e = 0; e < E; ++e) {
      const float s01 = red[(0 * EB + e) * CS_COLS + threadIdx.x] + red[(1 * EB + e) * CS_COLS + threadIdx.x];
      const float s23 = red[(2 * EB + e) * CS_COLS + threadIdx.x] + red[(3 * EB + e) * CS_COLS + threadIdx.x];
      partial[((int64_t)blockIdx.y * E + e) * C + cc] = s01 + s23;
    }
  }
  if (with_bias) {   // block-uniform; lane 0 of a wave saw every row its wave handled (the slab's first columns: c < C holds)
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < EB; ++e) red[wave * EB + e] = bacc[e];
    }
    __syncthreads();
    if ((int)threadIdx.x < E)
      bias_partial[(int64_t)blockIdx.y * E + threadIdx.x] =
          (red[0 * EB + threadIdx.x] + red[1 * EB + threadIdx.x]) + (red[2 * EB + threadIdx.x] + red[3 * EB + threadIdx.x]);
  }
}

// one workgroup per (64-column slab, expert): wave w adds chunks w, w+4, ... (lane = column), the four meet in LDS
__global__ __launch_bounds__(256) void gate_wgrad_final_kernel(const float* __restrict__ partial, int n_chunks, int E, int C,
                                                               float* __restrict__ out, const float* __restrict__ bias_partial,
                                                               float* __restrict__ bias_out) {
  __shared__ float red[4][64];
  const int e = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  if (bias_out && blockIdx.x == gridDim.x - 1 && wave == 3) {   // the bias column: lane l adds chunks l, l + 64, ..., then a fixed tree
    float b = 0.f;
    for (int k = lane; k < n_chunks; k += 64) b += bias_partial[(int64_t)k * E + e];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) b += __shfl_xor(b, m, 64);
    if (lane == 0) bias_out[e] = b;
  }
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    int k = wave;
    for (; k + 4 < n_chunks; k += 8) {
      a0 += partial[((int64_t)k * E + e) * C + c];
      a1 += partial[((int64_t)(k + 4) * E + e) * C + c];
    }
    if (k < n_chunks) a0 += partial[((int64_t)k * E + e) * C + c];
  }
  red[wave][lane] = a0 + a1;
  __syncthreads();
  if (wave == 0 && c < C) out[(int64_t)e * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// Zero-row groups of the MoE operator's training path (tokens the skip gate masked: all-zero input rows, every one routed by the
// gate bias to expert gmap[E + j]; autograd._GroupFFN.backward): their weight gradients need no GEMM.  The rows of A = gelu(b1[e])
// are identical, so group E + j adds the rank-1 term colsum(dY_g) (x) A_row to dW2[gmap[E + j]]; the bias gradients of an expert
// are the column sums of its own group plus those of the zero groups that use it.  ONE launch instead of ~12 host-launched
// [E]-sized ones per layer; groups are folded in index order (deterministic).
template <typename AT>
__global__ __launch_bounds__(256) void zero_group_fold_kernel(const float* __restrict__ cs2, const float* __restrict__ cs1,
                                                              const AT* __restrict__ A, const int32_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ gmap, int E, int Z, int d, int h,
                                                              int64_t n_rows, float* __restrict__ dW2, float* __restrict__ db2,
                                                              float* __restrict__ db1) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_w = (int64_t)d * (h / 4);       // f32x4 pieces of one [d, h] matrix
  if (gid < per_w) {
    const int c = (int)(gid / (h / 4)), r = (int)(gid % (h / 4)) * 4;
    for (int j = 0; j < Z; ++j) {
      const int e = gmap[E + j];
      if (e < 0 || e >= E) continue;                 // (caller data)
      int64_t first = offsets[E + j];
      if (first >= offsets[E + j + 1]) continue;     // an empty group contributes nothing
      if (first > n_rows - 1) first = n_rows - 1;
      float a[4];
      load4(A + first * h + r, a);
      const float w = cs2[(int64_t)(E + j) * d + c];
      f32x4* dst = reinterpret_cast<f32x4*>(dW2 + ((int64_t)e * d + c) * h + r);
      f32x4 v = *dst;
      v[0] = fmaf(w, a[0], v[0]); v[1] = fmaf(w, a[1], v[1]); v[2] = fmaf(w, a[2], v[2]); v[3] = fmaf(w, a[3], v[3]);
      *dst = v;
    }
  }
  const int64_t nb2 = db2 ? (int64_t)E * d : 0, nb1 = db1 ? (int64_t)E * h : 0;
  if (gid < nb2 + nb1) {
    const bool two = gid < nb2;
    const int64_t i = two ? gid : gid - nb2;
    const int C = two ? d : h;
    const float* cs = two ? cs2 : cs1;
    const int e = (int)(i / C), c = (int)(i % C);
    float acc = cs[(int64_t)e * C + c];
    for (int j = 0; j < Z; ++j)
      if (gmap[E + j] == e) acc += cs[(int64_t)(E + j) * C + c];
    (two ? db2 : db1)[i] = acc;
  }
}

template <typename F> int by_dtype(int code, F&& f) {
  switch (code) {
    case SMOE_F32: return f((float*)nullptr);
    case SMOE_F16: return f((f16*)nullptr);
    case SMOE_BF16: return f((bf16_bits*)nullptr);
  }
  smoe_set_error("bad dtype %d", code);
  return 1;
}

}  // namespace

extern "C" int smoe_gelu(const void* src, void* dst, int dtype, int64_t n, void* stream) {
  SMOE_REQUIRE(n >= 0 && n % 8 == 0, "smoe_gelu: n=%lld must be a multiple of 8", (long long)n);
  if (n == 0) return 0;
  SMOE_REQUIRE(src && dst, "smoe_gelu: null pointer");
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  return by_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL((gelu_kernel<T>), dim3((int)blocks), dim3(256), 0, s, (const T*)src, (T*)dst, n);
    SMOE_CHECK_LAUNCH("smoe_gelu");
    return 0;
  });
}

extern "C" int smoe_rowdot(const void* dout, int dout_dtype, const void* y, int y_dtype, const int64_t* inv_pos,
                           int64_t n, int k, int d, float* dscore, void* stream) {
  SMOE_REQUIRE(n >= 0 && k >= 1 && d > 0 && d % 8 == 0, "smoe_rowdot: bad sizes");
  if (n == 0) return 0;
  SMOE_REQUIRE(dout && y && inv_pos && dscore, "smoe_rowdot: null pointer");
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = (n + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  return by_dtype(dout_dtype, [&](auto* t1) {
    using DT = std::remove_pointer_t<decltype(t1)>;
    return by_dtype(y_dtype, [&](auto* t2) {
      using YT = std::remove_pointer_t<decltype(t2)>;
      hipLaunchKernelGGL((rowdot_kernel<DT, YT>), dim3((int)blocks), dim3(256), 0, s, (const DT*)dout, (const YT*)y,
                         inv_pos, n, k, d, dscore);
      SMOE_CHECK_LAUNCH("smoe_rowdot");
      return 0;
    });
  });
}

// out[g * S + s] = offsets[g] + min(s * step_g, count_g), step_g = count_g cut into S pieces of whole 64-row chunks; out[G * S] =
// offsets[G]: every row group cut into S pseudo-groups (a weight gradient over few, long groups fills few CUs: smoe_grouped_wgrad_rows'
// grid is groups x output tiles -- 24 workgroups for DeiT-Tiny's dW1 [8, 768, 192]).
namespace {
__global__ void split_offsets_kernel(const int32_t* __restrict__ offsets, int G, int S, int32_t* __restrict__ out) {
  for (int i = threadIdx.x; i < G * S; i += blockDim.x) {
    const int g = i / S, s = i - g * S;
    const int32_t lo = offsets[g], cnt = offsets[g + 1] - lo;
    const int32_t step = ((cnt + S * 64 - 1) / (S * 64)) * 64;
    const int32_t at = s * step;
    out[i] = lo + (at < cnt ? at : cnt);
  }
  if (threadIdx.x == 0) out[G * S] = offsets[G];
}
}  // namespace

extern "C" int smoe_split_offsets(const int32_t* offsets, int G, int S, int32_t* out, void* stream) {
  SMOE_REQUIRE(offsets && out && G >= 1 && S >= 1 && (int64_t)G * S <= 65536, "smoe_split_offsets: bad arguments G=%d S=%d", G, S);
  hipLaunchKernelGGL(split_offsets_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, offsets, G, S, out);
  SMOE_CHECK_LAUNCH("smoe_split_offsets");
  return 0;
}

extern "C" int smoe_pad_offsets(const int32_t* offsets, int E, int32_t* offsets_pad, void* stream) {
  SMOE_REQUIRE(offsets && offsets_pad && E >= 1, "smoe_pad_offsets: bad arguments");
  hipLaunchKernelGGL(pad_offsets_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, offsets, E, offsets_pad);
  SMOE_CHECK_LAUNCH("smoe_pad_offsets");
  return 0;
}

extern "C" int smoe_transpose_pad(const void* src, int dtype, const int32_t* offsets, const int32_t* offsets_pad, int E,
                                  int64_t n_rows, int C, int Lp, void* dst, void* stream) {
  SMOE_REQUIRE(src && dst && offsets && offsets_pad, "smoe_transpose_pad: null pointer");
  SMOE_REQUIRE(E >= 1 && C > 0 && Lp > 0 && Lp % 64 == 0 && (int64_t)Lp >= ((n_rows + 63) / 64) * 64,
               "smoe_transpose_pad: bad sizes (Lp=%d must be a multiple of 64 and >= padded rows)", Lp);
  SMOE_REQUIRE(dtype == SMOE_F16 || dtype == SMOE_BF16 || dtype == SMOE_F32, "smoe_transpose_pad: bad dtype");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((C + 63) / 64, Lp / 64);
  if (dtype == SMOE_F32) {
    hipLaunchKernelGGL((transpose_pad_kernel<float>), grid, dim3(256), 0, s, (const float*)src, offsets, offsets_pad, E, C, Lp, (float*)dst);
  } else {
    hipLaunchKernelGGL((transpose_pad_kernel<unsigned short>), grid, dim3(256), 0, s, (const unsigned short*)src, offsets,
                       offsets_pad, E, C, Lp, (unsigned short*)dst);
  }
  SMOE_CHECK_LAUNCH("smoe_transpose_pad");
  return 0;
}

static inline int64_t colsum_chunks(int64_t n_rows_max, int E) { return (n_rows_max + CS_ROWS - 1) / CS_ROWS + E; }

extern "C" size_t smoe_group_colsum_workspace_bytes(int64_t n_rows_max, int E, int C) {
  if (n_rows_max < 0 || E < 1 || C < 1) return 0;
  return (size_t)colsum_chunks(n_rows_max, E) * (size_t)C * 4;
}

extern "C" int smoe_group_colsum(const void* src, int dtype, const int32_t* offsets, const int32_t* group_end, int E,
                                 int64_t n_rows_max, int C, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  SMOE_REQUIRE(offsets && out && E >= 1 && C > 0 && C % 4 == 0 && n_rows_max >= 0, "smoe_group_colsum: bad arguments");
  SMOE_REQUIRE(n_rows_max == 0 || src, "smoe_group_colsum: null pointer");
  SMOE_REQUIRE(workspace && workspace_bytes >= smoe_group_colsum_workspace_bytes(n_rows_max, E, C),
               "smoe_group_colsum: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t chunks = colsum_chunks(n_rows_max, E);
  SMOE_REQUIRE(chunks <= 65535, "smoe_group_colsum: too many rows (%lld)", (long long)n_rows_max);
  float* partial = reinterpret_cast<float*>(workspace);
  const bool wide = dtype != SMOE_F32 && C % 8 == 0;     // 16-byte loads: 8 elements of a 16-bit row
  const int slab = 64 * (wide ? 8 : 4);
  dim3 grid1((C + slab - 1) / slab, (unsigned)chunks), grid2((C + 63) / 64, E);
  return by_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    if constexpr (std::is_same<T, float>::value) {
      hipLaunchKernelGGL((group_colsum_partial_kernel<T, 4>), grid1, dim3(64 * CS_WAVES), 0, s, (const T*)src, offsets, group_end, E, C, partial);
    } else {
      if (wide) hipLaunchKernelGGL((group_colsum_partial_kernel<T, 8>), grid1, dim3(64 * CS_WAVES), 0, s, (const T*)src, offsets, group_end, E, C, partial);
      else hipLaunchKernelGGL((group_colsum_partial_kernel<T, 4>), grid1, dim3(64 * CS_WAVES), 0, s, (const T*)src, offsets, group_end, E, C, partial);
    }
    SMOE_CHECK_LAUNCH("smoe_group_colsum/partial");
    hipLaunchKernelGGL(group_colsum_final_kernel, grid2, dim3(256), 0, s, partial, offsets, group_end, C, out);
    SMOE_CHECK_LAUNCH("smoe_group_colsum/final");
    return 0;
  });
}

extern "C" size_t smoe_gate_wgrad_workspace_bytes(int64_t n_rows, int E, int C) {
  if (n_rows < 0 || E < 1 || C < 1) return 0;
  return (size_t)((n_rows + GW_ROWS - 1) / GW_ROWS) * (size_t)E * ((size_t)C + 1) * 4;   // + the bias column's partials
}

// dWg [E, C] (f32) = dl^T x for dl [n_rows, E] f32 and x [n_rows, C] (f32 / f16 / bf16); E <= 16, C % 4 == 0.
// db (optional, f32 [E]) = column sums of dl (the gate bias' gradient) from the same pass.
extern "C" int smoe_gate_wgrad(const float* dl, const void* x, int x_dtype, int64_t n_rows, int E, int C, float* out, float* db,
                               void* workspace, size_t workspace_bytes, void* stream) {
  SMOE_REQUIRE(out && E >= 1 && E <= GW_MAX_E && C > 0 && C % 4 == 0 && n_rows >= 0, "smoe_gate_wgrad: bad arguments (E <= %d, C %% 4 == 0)", GW_MAX_E);
  hipStream_t s = (hipStream_t)stream;
  if (n_rows == 0) {
    hipError_t me = smoe_zero_words(out, (int64_t)E * C, s);
    if (me == hipSuccess && db) me = smoe_zero_words(db, E, s);
    SMOE_REQUIRE(me == hipSuccess, "smoe_gate_wgrad: clear failed");
    return 0;
  }
  SMOE_REQUIRE(dl && x, "smoe_gate_wgrad: null pointer");
  SMOE_REQUIRE(workspace && workspace_bytes >= smoe_gate_wgrad_workspace_bytes(n_rows, E, C), "smoe_gate_wgrad: workspace too small");
  const int64_t chunks = (n_rows + GW_ROWS - 1) / GW_ROWS;
  SMOE_REQUIRE(chunks <= 65535, "smoe_gate_wgrad: too many rows (%lld)", (long long)n_rows);
  float* partial = reinterpret_cast<float*>(workspace);
  float* bpart = db ? partial + (size_t)chunks * E * C : nullptr;
  dim3 grid1((C + CS_COLS - 1) / CS_COLS, (unsigned)chunks), grid2((C + 63) / 64, E);
  return by_dtype(x_dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    if (E <= 8) {
      hipLaunchKernelGGL((gate_wgrad_partial_kernel<T, 8>), grid1, dim3(256), (size_t)4 * 8 * CS_COLS * 4, s, dl, (const T*)x, n_rows, E, C, partial, bpart);
    } else {
      auto kern = gate_wgrad_partial_kernel<T, 16>;
      SMOE_ENSURE_SMEM(gate_wgrad_partial_kernel<T, 16>);
      hipLaunchKernelGGL(kern, grid1, dim3(256), (size_t)4 * 16 * CS_COLS * 4, s, dl, (const T*)x, n_rows, E, C, partial, bpart);
    }
    SMOE_CHECK_LAUNCH("smoe_gate_wgrad/partial");
    hipLaunchKernelGGL(gate_wgrad_final_kernel, grid2, dim3(256), 0, s, partial, (int)chunks, E, C, out, bpart, db);
    SMOE_CHECK_LAUNCH("smoe_gate_wgrad/final");
    return 0;
  });
}

// see zero_group_fold_kernel.  cs2 f32 [E + Z, d], cs1 f32 [E + Z, h] (the column sums of dY / dH over ALL groups), A [n_rows, h]
// (f16 / bf16 / f32: the kept activations), offsets i32 [E + Z + 1], gmap i32 [E + Z]; dW2 f32 [E, d, h] is updated IN PLACE;
// db2 f32 [E, d] / db1 f32 [E, h] (either may be NULL; db1 needs cs1) are written.  h % 4 == 0.
extern "C" int smoe_zero_group_fold(const float* cs2, const float* cs1, const void* A, int a_dtype, const int32_t* offsets,
                                    const int32_t* gmap, int E, int Z, int d, int h, int64_t n_rows, float* dW2, float* db2,
                                    float* db1, void* stream) {
  SMOE_REQUIRE(E >= 1 && Z >= 0 && d > 0 && h > 0 && h % 4 == 0 && n_rows >= 0, "smoe_zero_group_fold: bad sizes");
  SMOE_REQUIRE(cs2 && offsets && gmap && dW2 && (!db1 || cs1) && (n_rows == 0 || A), "smoe_zero_group_fold: null pointer");
  if (n_rows == 0) Z = 0;
  const int64_t work = (int64_t)d * (h / 4) > (int64_t)E * (d + h) ? (int64_t)d * (h / 4) : (int64_t)E * (d + h);
  const int64_t blocks = (work + 255) / 256;
  SMOE_REQUIRE(blocks <= 0x7fffffff, "smoe_zero_group_fold: too large");
  hipStream_t s = (hipStream_t)stream;
  return by_dtype(a_dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL((zero_group_fold_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, cs2, cs1, (const T*)A, offsets, gmap, E, Z,
                       d, h, n_rows, dW2, db2, db1);
    SMOE_CHECK_LAUNCH("smoe_zero_group_fold");
    return 0;
  });
}

extern "C" size_t smoe_switch_aux_workspace_bytes(int64_t T, int E) {
  if (T < 0 || E < 1) return 0;
  return (size_t)((T + SA_ROWS - 1) / SA_ROWS + 1) * (size_t)E * 4;
}

// SwitchGate load-balance loss: aux [1] = E sum_e (counts[e] / kept) (sum_t probs[t, e] / kept), coef [E] = E counts[e] / kept^2
// (kept = max(sum counts, 1)); probs f32 [T, E], counts i32 [E] (the dispatch plan's kept counts); E <= 256.
extern "C" int smoe_switch_aux(const float* probs, const int32_t* counts, int64_t T, int E, float* aux, float* coef,
                               void* workspace, size_t workspace_bytes, void* stream) {
  SMOE_REQUIRE(T >= 0 && E >= 1 && E <= SA_MAX_E, "smoe_switch_aux: bad sizes T=%lld E=%d (E <= %d)", (long long)T, E, SA_MAX_E);
  SMOE_REQUIRE(counts && aux && coef && (T == 0 || probs), "smoe_switch_aux: null pointer");
  SMOE_REQUIRE(workspace && workspace_bytes >= smoe_switch_aux_workspace_bytes(T, E), "smoe_switch_aux: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t chunks = (T + SA_ROWS - 1) / SA_ROWS;
  SMOE_REQUIRE(chunks <= (1 << 24), "smoe_switch_aux: too many rows");
  float* partial = reinterpret_cast<float*>(workspace);
  if (chunks > 0) {
    hipLaunchKernelGGL(switch_aux_partial_kernel, dim3((unsigned)chunks), dim3(256), 0, s, probs, T, E, partial);
    SMOE_CHECK_LAUNCH("smoe_switch_aux/partial");
  }
  hipLaunchKernelGGL(switch_aux_final_kernel, dim3(1), dim3(256), 0, s, partial, (int)chunks, counts, E, aux, coef);
  SMOE_CHECK_LAUNCH("smoe_switch_aux/final");
  return 0;
}

template <typename ST>
static int transpose_cast_launch(const void* src, void* dst, int dst_dtype, int B, int R, int C, hipStream_t s) {
  const dim3 grid(C / 64, R / 64, B);
  switch (dst_dtype) {
    case SMOE_F32: hipLaunchKernelGGL((transpose_cast_kernel<ST, float>), grid, dim3(256), 0, s, (const ST*)src, (float*)dst, R, C); break;
    case SMOE_F16: hipLaunchKernelGGL((transpose_cast_kernel<ST, f16>), grid, dim3(256), 0, s, (const ST*)src, (f16*)dst, R, C); break;
    default: hipLaunchKernelGGL((transpose_cast_kernel<ST, bf16_bits>), grid, dim3(256), 0, s, (const ST*)src, (bf16_bits*)dst, R, C); break;
  }
  SMOE_CHECK_LAUNCH("smoe_transpose_cast");
  return 0;
}

extern "C" int smoe_transpose_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int R, int C, void* stream) {
  SMOE_REQUIRE(src && dst, "smoe_transpose_cast: null pointer");
  SMOE_REQUIRE(smoe_dtype_ok(src_dtype) && smoe_dtype_ok(dst_dtype), "smoe_transpose_cast: bad dtype");
  SMOE_REQUIRE(B >= 1 && B <= 65535 && R > 0 && C > 0 && R % 64 == 0 && C % 64 == 0 && R / 64 <= 65535,
               "smoe_transpose_cast: B=%d R=%d C=%d (R and C must be multiples of 64)", B, R, C);
  hipStream_t s = (hipStream_t)stream;
  switch (src_dtype) {
    case SMOE_F32: return transpose_cast_launch<float>(src, dst, dst_dtype, B, R, C, s);
    case SMOE_F16: return transpose_cast_launch<f16>(src, dst, dst_dtype, B, R, C, s);
    default: return transpose_cast_launch<bf16_bits>(src, dst, dst_dtype, B, R, C, s);
  }
}

extern "C" int smoe_switch_gate_bwd(const float* probs, const int64_t* idx, const float* dscore, const float* coef,
                                    const float* coef_scale, int64_t T, int E, float* dlogits, void* stream) {
  SMOE_REQUIRE(T >= 0 && E >= 1 && E <= 4096, "smoe_switch_gate_bwd: bad sizes T=%lld E=%d", (long long)T, E);
  if (T == 0) return 0;
  SMOE_REQUIRE(probs && idx && dlogits, "smoe_switch_gate_bwd: null pointer");
  hipLaunchKernelGGL(switch_gate_bwd_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, probs, idx, dscore,
                     coef, coef_scale, T, E, dlogits);
  SMOE_CHECK_LAUNCH("smoe_switch_gate_bwd");
  return 0;
}
