// Grouped (variable-batch) expert GEMM on MFMA.  Replaces fmoe_cuda.linear_forward (MOELinear /
// FMoELinear):  out[r,:] = epi(A[r,:] @ W[e]^T + bias[e])  for r in expert e's slice.
//
// One launch covers every expert.  The grid is an upper bound (ceil(M/BM) + E m-tiles); each workgroup
// finds its (expert, m-tile) from `offsets` on device, surplus workgroups exit.  No host sync.
//
// variant 0 ("t128"): 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 (f16/bf16) or
// 16x16x4 (f32-exact).  Operands are staged global -> VGPR -> LDS (XOR-swizzled 128-B rows, 16-B chunks:
// chunk' = chunk ^ ((row>>1)&7), conflict-free for the ds_read_b128 fragment reads), double-buffered.
// MFMA runs "swapped" (W fragment as the A operand) so every lane ends up with 4 consecutive output
// columns of one token row -> 8-byte LDS writes in the epilogue; the output tile then leaves through
// LDS as whole 16-B-per-lane row segments (optionally scattered through row_map = fused combine).
#include "smoe_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK_BYTES = 128;  // K-step = 128 bytes of a row (64 halfs / 32 floats)
constexpr int GEMM_THREADS = 256;
constexpr int STAGE_BYTES = (BM + BN) * BK_BYTES;  // 32 KiB
constexpr int C_PAD = 16;                          // bytes

// exact-erf GELU, erf by Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7): 0.5*(v + |v|*erf(|v|/sqrt2)).
// ~14 VALU ops incl. one v_rcp_f32 and one v_exp_f32; absolute GELU error < 1e-6 for |v| < 10.
__device__ __forceinline__ float gelu_erf(float v) {
  const float a = fabsf(v);
  const float z = a * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float ex = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  const float erf_abs = fmaf(-p, ex, 1.0f);
  return 0.5f * fmaf(a, erf_abs, v);
}

// GELU for the 16-bit-operand kernels: v * sigmoid(2 u(v)),  u(v) = v (c0 + c1 v^2 + c2 v^4 + c3 v^6 + c4 v^8) fitted to
// the exact-erf GELU (least squares on [-6, 6]); max |error| 3.6e-6 in f32 arithmetic -- far inside the 2^-11
// rounding of the f16 / bf16 store that follows -- at 11 issue slots (4 FMA, 3 mul, add, v_exp, v_rcp) instead of
// 17 for the erf form.  The coefficients carry the factor -2 log2(e).  No clamp is needed: the polynomial stays
// below -2.3 for every v^2 >= 0 (checked over the whole f32 range; tests/test_gpu_parity.py sweeps it), so beyond the fitted interval
// v * p only runs further towards -inf / +inf, where the sigmoid has saturated to 1 / 0 anyway.
__device__ __forceinline__ float gelu_fast(float v) {
  const float v2 = v * v;
  float p = fmaf(v2, -3.28856595e-06f, 8.92457392e-05f);
  p = fmaf(p, v2, 3.55226046e-04f);
  p = fmaf(p, v2, -1.05218634e-01f);
  p = fmaf(p, v2, -2.30204797e+00f);
  const float e = __builtin_amdgcn_exp2f(v * p);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// the same function on a pair: the eight regular operations become v_pk_mul / v_pk_fma / v_pk_add (two elements per
// issue slot; bit-identical per element), only v_exp / v_rcp stay scalar: 64 instead of 104 issue cycles per pair in
// the epilogue, where no MFMA competes for the slots
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 v) {
  const f32x2 v2 = v * v;
  f32x2 p = __builtin_elementwise_fma(v2, f32x2{-3.28856595e-06f, -3.28856595e-06f}, f32x2{8.92457392e-05f, 8.92457392e-05f});
  p = __builtin_elementwise_fma(p, v2, f32x2{3.55226046e-04f, 3.55226046e-04f});
  p = __builtin_elementwise_fma(p, v2, f32x2{-1.05218634e-01f, -1.05218634e-01f});
  p = __builtin_elementwise_fma(p, v2, f32x2{-2.30204797e+00f, -2.30204797e+00f});
  const f32x2 z = v * p;
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + f32x2{1.0f, 1.0f};
  return v * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x4 gelu_fast4(f32x4 v) {
  const f32x2 lo = gelu_fast2(__builtin_shufflevector(v, v, 0, 1)), hi = gelu_fast2(__builtin_shufflevector(v, v, 2, 3));
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename OT> struct OutPack;
template <> struct OutPack<float> {
  static constexpr int bytes = 4;
  __device__ static void write4(char* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct OutPack<f16> {
  static constexpr int bytes = 2;
  __device__ static void write4(char* p, const f32x4& v) {
    f16x4 t; t[0] = (f16)v[0]; t[1] = (f16)v[1]; t[2] = (f16)v[2]; t[3] = (f16)v[3];
    *reinterpret_cast<f16x4*>(p) = t;
  }
};
template <> struct OutPack<bf16_bits> {
  static constexpr int bytes = 2;
  __device__ static void write4(char* p, const f32x4& v) {
    s16x4 t; t[0] = (short)f32_to_bf16(v[0]); t[1] = (short)f32_to_bf16(v[1]); t[2] = (short)f32_to_bf16(v[2]); t[3] = (short)f32_to_bf16(v[3]);
    *reinterpret_cast<s16x4*>(p) = t;
  }
};

// four f32 -> four 16-bit outputs in two dwords (element 0 in the low half of the first), rounded as OutPack::write4 does
template <typename OT> __device__ __forceinline__ void pack4(const f32x4& v, uint32_t& lo, uint32_t& hi);
template <> __device__ __forceinline__ void pack4<f16>(const f32x4& v, uint32_t& lo, uint32_t& hi) {
  f16x4 t; t[0] = (f16)v[0]; t[1] = (f16)v[1]; t[2] = (f16)v[2]; t[3] = (f16)v[3];
  const u32x2 r = __builtin_bit_cast(u32x2, t);
  lo = r[0]; hi = r[1];
}
template <> __device__ __forceinline__ void pack4<bf16_bits>(const f32x4& v, uint32_t& lo, uint32_t& hi) {
  lo = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  hi = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
}
template <> __device__ __forceinline__ void pack4<float>(const f32x4&, uint32_t&, uint32_t&) {}   // never used (DIRECT is 16-bit only)

// scale 16 bytes of output elements in place (fused combine)
template <typename OT> __device__ __forceinline__ u32x4 scale16(u32x4 raw, float s);
template <> __device__ __forceinline__ u32x4 scale16<float>(u32x4 raw, float s) {
  f32x4 v = __builtin_bit_cast(f32x4, raw);
  v *= s;
  return __builtin_bit_cast(u32x4, v);
}
template <> __device__ __forceinline__ u32x4 scale16<f16>(u32x4 raw, float s) {
  f16x8 v = __builtin_bit_cast(f16x8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (f16)((float)v[i] * s);
  return __builtin_bit_cast(u32x4, v);
}
template <> __device__ __forceinline__ u32x4 scale16<bf16_bits>(u32x4 raw, float s) {
  s16x8 v = __builtin_bit_cast(s16x8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (short)f32_to_bf16(bf16_to_f32((unsigned short)v[i]) * s);
  return __builtin_bit_cast(u32x4, v);
}

// elementwise add of 16 bytes of output elements (fused residual)
template <typename OT> __device__ __forceinline__ u32x4 add16(u32x4 a, u32x4 b);
template <> __device__ __forceinline__ u32x4 add16<float>(u32x4 a, u32x4 b) {
  return __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, a) + __builtin_bit_cast(f32x4, b));
}
template <> __device__ __forceinline__ u32x4 add16<f16>(u32x4 a, u32x4 b) {
  f16x8 x = __builtin_bit_cast(f16x8, a), y = __builtin_bit_cast(f16x8, b);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (f16)((float)x[i] + (float)y[i]);
  return __builtin_bit_cast(u32x4, x);
}
template <> __device__ __forceinline__ u32x4 add16<bf16_bits>(u32x4 a, u32x4 b) {
  s16x8 x = __builtin_bit_cast(s16x8, a), y = __builtin_bit_cast(s16x8, b);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    x[i] = (short)f32_to_bf16(bf16_to_f32((unsigned short)x[i]) + bf16_to_f32((unsigned short)y[i]));
  return __builtin_bit_cast(u32x4, x);
}

// d/dv gelu(v) = Phi(v) + v phi(v), same erf approximation as gelu_erf
__device__ __forceinline__ float gelu_grad(float v) {
  const float a = fabsf(v);
  const float z = a * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float ex = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);  // exp(-v^2/2)
  const float erf_abs = fmaf(-p, ex, 1.0f);
  const float cdf = 0.5f + 0.5f * copysignf(erf_abs, v);
  return fmaf(v * 0.3989422804014327f, ex, cdf);
}
// 16 bytes of output elements times gelu'(16 bytes of saved pre-activations)  (SMOE_EPI_GELU_GRAD)
template <typename OT> __device__ __forceinline__ u32x4 mulgrad16(u32x4 aux, u32x4 val);
template <> __device__ __forceinline__ u32x4 mulgrad16<float>(u32x4 aux, u32x4 val) {
  f32x4 h = __builtin_bit_cast(f32x4, aux), v = __builtin_bit_cast(f32x4, val);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] *= gelu_grad(h[i]);
  return __builtin_bit_cast(u32x4, v);
}
template <> __device__ __forceinline__ u32x4 mulgrad16<f16>(u32x4 aux, u32x4 val) {
  f16x8 h = __builtin_bit_cast(f16x8, aux), v = __builtin_bit_cast(f16x8, val);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (f16)((float)v[i] * gelu_grad((float)h[i]));
  return __builtin_bit_cast(u32x4, v);
}
template <> __device__ __forceinline__ u32x4 mulgrad16<bf16_bits>(u32x4 aux, u32x4 val) {
  s16x8 h = __builtin_bit_cast(s16x8, aux), v = __builtin_bit_cast(s16x8, val);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    v[i] = (short)f32_to_bf16(bf16_to_f32((unsigned short)v[i]) * gelu_grad(bf16_to_f32((unsigned short)h[i])));
  return __builtin_bit_cast(u32x4, v);
}
template <typename OT> __device__ __forceinline__ u32x4 fuse_aux(int epilogue, u32x4 aux, u32x4 val) {
  return epilogue == SMOE_EPI_GELU_GRAD ? mulgrad16<OT>(aux, val) : add16<OT>(aux, val);
}

// locate the (expert, row range) of global m-tile `mt`; returns false if there is no such tile
__device__ __forceinline__ bool find_tile(const int32_t* __restrict__ offsets, int E, int mt, int& e_out, int& m0,
                                          int& m_end, int bm = BM) {
  int tile_base = 0;
  for (int e = 0; e < E; ++e) {
    const int lo = offsets[e], hi = offsets[e + 1];
    const int nt = (hi - lo + bm - 1) / bm;
    if (mt < tile_base + nt) {
      e_out = e;
      m0 = lo + (mt - tile_base) * bm;
      m_end = hi;
      return true;
    }
    tile_base += nt;
  }
  return false;
}

// The launch grid is an upper bound (ceil(rows / bm) + E m-tiles: the per-expert row counts live on the device).
// Each workgroup counts the m-tiles that really exist and the XCD-contiguous remap splits THOSE evenly over the
// eight XCDs (hardware sends workgroup b to XCD b % 8): without this the surplus slots all fall into the last
// XCD's range and the other seven carry up to 1/7 more tiles -- a whole extra round when the real tiles fit one.
// Returns false for a surplus workgroup; otherwise bid becomes the remapped tile slot.
__device__ __forceinline__ bool remap_balanced(const int32_t* __restrict__ offsets, int E, int bm, int group_m,
                                               int n_tiles_n, int& bid) {
  int total_mt = 0;
  for (int e = 0; e < E; ++e) total_mt += (offsets[e + 1] - offsets[e] + bm - 1) / bm;
  const int real = ((total_mt + group_m - 1) / group_m) * group_m * n_tiles_n;
  const int q = real / 8, r = real % 8, xcd = bid % 8, loc = bid / 8;
  if (loc >= q + (xcd < r ? 1 : 0)) return false;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  return true;
}

template <typename AB, typename OT>
__global__ __launch_bounds__(GEMM_THREADS, 2) void grouped_gemm_t128(
    const AB* __restrict__ A, const AB* __restrict__ W, const float* __restrict__ bias,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ group_expert, int E, int K, int N, int epilogue,
    const int64_t* __restrict__ row_map, const float* __restrict__ row_scale, const OT* residual, OT* out, int n_tiles_n,
    int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(AB);
  constexpr int BKE = BK_BYTES / ES;  // elements per K-step

  // ---- tile id: XCD-contiguous remap, then grouped (m, n) order -------------------------------------
  int bid = blockIdx.x;
  if (!remap_balanced(offsets, E, BM, group_m, n_tiles_n, bid)) return;
  const int per_group = group_m * n_tiles_n;
  const int g = bid / per_group, rem = bid % per_group;
  const int mt = g * group_m + rem % group_m;
  const int nt = rem / group_m;

  int e, m0, m_end;
  if (!find_tile(offsets, E, mt, e, m0, m_end)) return;
  if (group_expert) e = group_expert[e];  // group -> weight index
  const int n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- global -> register staging addresses --------------------------------------------------------
  const int ld_chunk = tid & 7, ld_row = tid >> 3;  // 32 rows per pass, 4 passes for A and for B
  const AB* a_ptr[4];
  const AB* w_ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ar = m0 + ld_row + 32 * i;
    if (ar >= m_end) ar = m_end - 1;
    a_ptr[i] = A + (int64_t)ar * K + ld_chunk * (16 / ES);
    int wr = n0 + ld_row + 32 * i;
    if (wr >= N) wr = N - 1;
    w_ptr[i] = W + ((int64_t)e * N + wr) * K + ld_chunk * (16 / ES);
  }
  u32x4 ra[4], rw[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + k0);
      rw[i] = *reinterpret_cast<const u32x4*>(w_ptr[i] + k0);
    }
  };
  auto lstore = [&](int buf) {
    char* sa = smem + buf * STAGE_BYTES;
    char* sw = sa + BM * BK_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(sa + swz(ld_row + 32 * i, ld_chunk)) = ra[i];
      *reinterpret_cast<u32x4*>(sw + swz(ld_row + 32 * i, ld_chunk)) = rw[i];
    }
  };

  f32x4 acc[4][4];  // [mi][ni]; lane holds n = 4*(lane>>4)+r (r=0..3) of token m = lane&15
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BKE;
  gload(0);
  lstore(0);
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BKE);
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sw = sa + BM * BK_BYTES;
    if constexpr (ES == 2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x4 af[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = *reinterpret_cast<const u32x4*>(sa + swz(wm * 64 + i * 16 + fr, kk * 4 + fq));
          wf[i] = *reinterpret_cast<const u32x4*>(sw + swz(wn * 64 + i * 16 + fr, kk * 4 + fq));
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            if constexpr (std::is_same<AB, f16>::value)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[ni]),
                                                                    __builtin_bit_cast(f16x8, af[mi]), acc[mi][ni], 0, 0, 0);
            else
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[ni]),
                                                                     __builtin_bit_cast(bf16x8_t, af[mi]), acc[mi][ni], 0, 0, 0);
          }
      }
    } else {
      // f32-exact: 16x16x4, K-step of 32 floats = 8 MFMA k-steps; lane reads element k = 4*ks + fq of row fr
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        float af[4], wf[4];
        const int kel = ks * 4 + fq;  // float index within the 32-float K-step
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = *reinterpret_cast<const float*>(sa + swz(wm * 64 + i * 16 + fr, kel >> 2) + (kel & 3) * 4);
          wf[i] = *reinterpret_cast<const float*>(sw + swz(wn * 64 + i * 16 + fr, kel >> 2) + (kel & 3) * 4);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias (+GELU), convert, stage the tile through LDS, store whole row segments -------
  constexpr int OB = OutPack<OT>::bytes;
  constexpr int C_STRIDE = BN * OB + C_PAD;
  {
    const float* bias_e = bias ? bias + (int64_t)e * N : nullptr;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int nl = wn * 64 + ni * 16 + fq * 4;  // tile-local column of r=0
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bias_e && n0 + nl < N) bv = *reinterpret_cast<const f32x4*>(bias_e + n0 + nl);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        f32x4 v = acc[mi][ni] + bv;
        if (epilogue == SMOE_EPI_GELU) {
          v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
        }
        const int ml = wm * 64 + mi * 16 + fr;
        OutPack<OT>::write4(smem + ml * C_STRIDE + nl * OB, v);
      }
    }
  }
  __syncthreads();
  {
    constexpr int CHUNKS = BN * OB / 16;        // 16-B chunks per tile row: 16 (2-byte out) or 32 (f32)
    constexpr int ROWS_PER_PASS = GEMM_THREADS / CHUNKS;
    const int ch = tid % CHUNKS, r0 = tid / CHUNKS;
    const int ncol = n0 + ch * (16 / OB);
    if (ncol < N) {
      for (int r = r0; r < BM; r += ROWS_PER_PASS) {
        const int m = m0 + r;
        if (m >= m_end) break;
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + r * C_STRIDE + ch * 16);
        int64_t orow = m;
        if (row_map) {
          orow = row_map[m];
          if (row_scale) v = scale16<OT>(v, row_scale[orow]);
        }
        const int64_t ooff = (orow * (int64_t)N + ncol) * OB;
        if (residual) v = fuse_aux<OT>(epilogue, *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(residual) + ooff), v);
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + ooff) = v;
      }
    }
  }
}

template <typename AB, typename OT>
int launch_t128(const void* A, const void* W, const float* bias, const int32_t* offsets, const int32_t* group_expert,
                int E, int64_t m_rows_max, int K, int N, int epilogue, const int64_t* row_map, const float* row_scale,
                const void* residual, void* out, hipStream_t s) {
  const int n_tiles_n = (N + BN - 1) / BN;
  const int max_m_tiles = (int)((m_rows_max + BM - 1) / BM) + E;
  const int group_m = 8;
  const int m_groups = (max_m_tiles + group_m - 1) / group_m;
  const int grid = m_groups * group_m * n_tiles_n;
  constexpr int OB = OutPack<OT>::bytes;
  size_t smem = 2 * STAGE_BYTES;
  const size_t ctile = (size_t)BM * (BN * OB + C_PAD);
  if (ctile > smem) smem = ctile;
  auto kern = grouped_gemm_t128<AB, OT>;
  SMOE_ENSURE_SMEM(grouped_gemm_t128<AB, OT>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(GEMM_THREADS), smem, s, (const AB*)A, (const AB*)W, bias, offsets, group_expert,
                     E, K, N, epilogue, row_map, row_scale, (const OT*)residual, (OT*)out, n_tiles_n, group_m);
  SMOE_CHECK_LAUNCH("smoe_grouped_gemm");
  return 0;
}


// =====================================================================================================
// glds variants: global -> LDS by DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write), source
// address pre-swizzled so the lane-linear LDS image is the same XOR-swizzled image the fragment reads expect
// (cdna_hip_programming.md rule 21).  Tile TBM x TBN x 64, WM x WN waves, two LDS stages, one barrier per
// K-tile ("2-phase" structure).  16-bit operands only.
template <typename AB, typename OT, int TBM, int TBN, int WM, int WN, int MINW>
__global__ __launch_bounds__(64 * WM * WN, MINW) void grouped_gemm_glds(
    const AB* __restrict__ A, const AB* __restrict__ W, const float* __restrict__ bias,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ group_expert, int E, int K, int N, int epilogue,
    const int64_t* __restrict__ row_map, const float* __restrict__ row_scale, const OT* residual, OT* out, int n_tiles_n,
    int group_m) {
  static_assert(sizeof(AB) == 2, "glds variants take 16-bit operands");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 64 * WM * WN, NW = WM * WN;
  constexpr int TM = TBM / WM, TN = TBN / WN, MI = TM / 16, NI = TN / 16;
  constexpr int STAGE = (TBM + TBN) * BK_BYTES;
  constexpr int A_SLOTS = TBM / 8 / NW, W_SLOTS = TBN / 8 / NW;  // 1-KiB (8-row) DMA pieces per wave
  static_assert(TBM % (8 * NW) == 0 && TBN % (8 * NW) == 0, "tile rows must split into 8-row pieces per wave");

  int bid = blockIdx.x;
  if (!remap_balanced(offsets, E, TBM, group_m, n_tiles_n, bid)) return;
  const int per_group = group_m * n_tiles_n;
  const int g = bid / per_group, rem = bid % per_group;
  const int mt = g * group_m + rem % group_m;
  const int nt = rem / group_m;
  int e, m0, m_end;
  if (!find_tile(offsets, E, mt, e, m0, m_end, TBM)) return;
  if (group_expert) e = group_expert[e];
  const int n0 = nt * TBN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // DMA source pointers: piece s of this wave covers tile rows [8*(s*NW+wave), +8); lane -> (row, LDS chunk pos)
  const int l_row = lane >> 3, l_pos = lane & 7;
  const AB* a_src[A_SLOTS];
  const AB* w_src[W_SLOTS];
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int r = 8 * (s * NW + wave) + l_row;
    int gr = m0 + r;
    if (gr >= m_end) gr = m_end - 1;
    a_src[s] = A + (int64_t)gr * K + ((l_pos ^ ((r >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int s = 0; s < W_SLOTS; ++s) {
    const int r = 8 * (s * NW + wave) + l_row;
    int gr = n0 + r;
    if (gr >= N) gr = N - 1;
    w_src[s] = W + ((int64_t)e * N + gr) * K + ((l_pos ^ ((r >> 1) & 7)) << 3);
  }
  auto stage = [&](int kt, int buf) {
    char* sa = smem + buf * STAGE;
    char* sw = sa + TBM * BK_BYTES;
    const int k0 = kt * 64;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[s] + k0),
                                       (__attribute__((address_space(3))) void*)(sa + (s * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int s = 0; s < W_SLOTS; ++s)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[s] + k0),
                                       (__attribute__((address_space(3))) void*)(sw + (s * NW + wave) * 1024), 16, 0, 0);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / 64;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
    const char* sa = smem + cur * STAGE + (wm * TM) * BK_BYTES;
    const char* sw = smem + cur * STAGE + (TBM + wn * TN) * BK_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      u32x4 af[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *reinterpret_cast<const u32x4*>(sw + swz(i * 16 + fr, kk * 4 + fq));
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const u32x4*>(sa + swz(i * 16 + fr, kk * 4 + fq));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if constexpr (std::is_same<AB, f16>::value)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[ni]),
                                                                  __builtin_bit_cast(f16x8, af[mi]), acc[mi][ni], 0, 0, 0);
          else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[ni]),
                                                                   __builtin_bit_cast(bf16x8_t, af[mi]), acc[mi][ni], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue in row passes through LDS ----------------------------------------------------------------
  constexpr int OB = OutPack<OT>::bytes;
  constexpr int C_STRIDE = TBN * OB + C_PAD;
  constexpr int LDS_BYTES = 2 * STAGE;
  constexpr int RP = (TBM * C_STRIDE <= LDS_BYTES) ? TBM : ((TBM / 2) * C_STRIDE <= LDS_BYTES ? TBM / 2 : TBM / 4);
  static_assert(RP * C_STRIDE <= LDS_BYTES, "epilogue pass does not fit in LDS");
  static_assert(RP % 16 == 0, "pass rows must be whole fragments");
  constexpr int NPASS = TBM / RP;
  constexpr int CHUNKS = TBN * OB / 16;
  constexpr int ROWS_PER_IT = NT / CHUNKS;
  static_assert(NT % CHUNKS == 0, "threads must tile the row chunks");
  const float* bias_e = bias ? bias + (int64_t)e * N : nullptr;
  f32x4 bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int nl = wn * TN + ni * 16 + fq * 4;
    bv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias_e && n0 + nl < N) bv[ni] = *reinterpret_cast<const f32x4*>(bias_e + n0 + nl);
  }
  const int ch = tid % CHUNKS, r0 = tid / CHUNKS;
  const int ncol = n0 + ch * (16 / OB);
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row = wm * TM + mi * 16;  // wave-uniform
      if (row / RP == p) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          f32x4 v = acc[mi][ni] + bv[ni];
          if (epilogue == SMOE_EPI_GELU) {
            v = gelu_fast4(v);
          }
          const int nl = wn * TN + ni * 16 + fq * 4;
          OutPack<OT>::write4(smem + (row - p * RP + fr) * C_STRIDE + nl * OB, v);
        }
      }
    }
    __syncthreads();
    if (ncol < N) {
      for (int r = r0; r < RP; r += ROWS_PER_IT) {
        const int m = m0 + p * RP + r;
        if (m >= m_end) break;
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + r * C_STRIDE + ch * 16);
        int64_t orow = m;
        if (row_map) {
          orow = row_map[m];
          if (row_scale) v = scale16<OT>(v, row_scale[orow]);
        }
        const int64_t ooff = (orow * (int64_t)N + ncol) * OB;
        if (residual) v = fuse_aux<OT>(epilogue, *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(residual) + ooff), v);
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + ooff) = v;
      }
    }
    if (p + 1 < NPASS) __syncthreads();
  }
}

template <typename AB, typename OT, int TBM, int TBN, int WM, int WN, int MINW>
int launch_glds(const void* A, const void* W, const float* bias, const int32_t* offsets, const int32_t* group_expert,
                int E, int64_t m_rows_max, int K, int N, int epilogue, const int64_t* row_map, const float* row_scale,
                const void* residual, void* out, int group_m, hipStream_t s) {
  const int n_tiles_n = (N + TBN - 1) / TBN;
  const int max_m_tiles = (int)((m_rows_max + TBM - 1) / TBM) + E;
  const int m_groups = (max_m_tiles + group_m - 1) / group_m;
  const int grid = m_groups * group_m * n_tiles_n;
  const size_t smem = 2 * (size_t)(TBM + TBN) * BK_BYTES;
  auto kern = grouped_gemm_glds<AB, OT, TBM, TBN, WM, WN, MINW>;
  SMOE_ENSURE_SMEM(grouped_gemm_glds<AB, OT, TBM, TBN, WM, WN, MINW>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), smem, s, (const AB*)A, (const AB*)W, bias, offsets, group_expert,
                     E, K, N, epilogue, row_map, row_scale, (const OT*)residual, (OT*)out, n_tiles_n, group_m);
  SMOE_CHECK_LAUNCH("smoe_grouped_gemm");
  return 0;
}


// =====================================================================================================
// variant 4 ("pp256"): 256x256x64 tile, 8 waves as 2(M) x 4(N), wave tile 128x64, glds staging into two
// 64-KiB LDS buffers.  Waves w and w+4 share a SIMD; group 1 (waves 4-7) runs ONE barrier interval behind
// group 0, so in every interval one wave of each SIMD issues its 16-MFMA cluster while its partner issues
// the ds_read_b128 fragment reads (and the DMA prefetch) for its next cluster: LDS reads and MFMA overlap
// instead of alternating.  Per K-tile a wave walks 4 quadrants of its 128x64 output (A-half x B-half),
// re-reading only the operand half that changes: (A0,B0) (A0,B1) (A1,B1) (A1,B0).
//
// Interval clock (group-0 time, 8 intervals per K-tile t):  R1 M1 R2 M2 R3 M3 R4 M4; group 1 is +1.
//   LDS regions of a buffer are re-staged one by one as soon as their last reader has retired its reads
//   (A rows 0-127 belong to group 0, rows 128-255 to group 1, W rows are shared); see the schedule comment
//   in the K loop.  Every wave drains its own DMAs with a COUNTED vmcnt before the barrier that closes
//   global interval 8t+7 (program interval 7 for group 0, 6 for group 1).
#define PP_BARRIER()                         \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

// ABL: timing-only ablation bits (diagnostic variants 40-47; results are wrong by construction):
//   1 = no DMA inside the K loop, 2 = no fragment ds_reads inside the K loop, 4 = no MFMA
// MODE 1 = weight-gradient GEMM: C[e] = P^T[:, k-range e] (Q^T[:, k-range e])^T with both operands stored K-major
// ([rows, Lp], per-expert column ranges padded to multiples of 64: smoe_transpose_pad); `offsets` are the
// padded ranges, K = Lp (row stride), N = rows of Q^T, m_rows = rows of P^T, one [m_rows, N] output per expert.
// MODE 2 = the same weight gradient straight from the TOKEN-major operands P [rows, m_rows], Q [rows, N] (no transposed
// copies): `offsets` are the experts' plain row ranges, K is unused.  A K-tile is 64 token rows; in LDS every 64-column
// slab of the tile is stored [token][64 columns] in exactly the bytes the K-major layout gives 64 tile rows (so the DMA
// pieces, regions and the re-staging schedule are unchanged), and the MFMA fragments -- 8 k-values of one output
// column per lane -- come from ds_read_b64_tr_b16 (two transposed reads per fragment, as attention.hip reads V).  Both
// operands use the same k-slot permutation (slot j of lane group g <-> token 32 kk + 16 (j >> 2) + 4 g + (j & 3)), which
// a contraction does not see.  Rows past an expert's range read a 16-byte zero page (`residual` carries its address).
// AFR = A row fragments per wave per A-half: tile height TBM = 64 * AFR (256 or 320 rows).  The taller tile cuts the
// tile count (fc2 / proj at cfg 2: 600 -> 480 tiles = 2 instead of 3 rounds on 256 CUs) and raises FLOP per LDS-fill byte.
template <typename AB, typename OT, int ABL = 0, int MODE = 0, int AFR = 4>
__global__ __launch_bounds__(512, 2) void grouped_gemm_pp256(
    const AB* __restrict__ A, const AB* __restrict__ W, const float* __restrict__ bias,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ group_expert, int E, int K, int N, int epilogue,
    const int64_t* __restrict__ row_map, const float* __restrict__ row_scale, const OT* residual, OT* out, int n_tiles_n,
    int group_m, int m_rows, const int64_t* __restrict__ a_gather, int a_div) {
  static_assert(sizeof(AB) == 2, "16-bit operands");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TBM = 64 * AFR, TBN = 256, NT = 512, NW = 8;
  constexpr int STAGE = (TBM + TBN) * BK_BYTES;  // 64 KiB (72 KiB for the 320-row tile)
  constexpr int ASLOTS = TBM / 8 / NW;           // 1-KiB DMA pieces per wave per K-tile: A (= AFR)
  constexpr int SLOTS = 4;                       //                                        W
  static_assert(AFR == 4 || AFR == 5, "tile height 256 or 320");
  static_assert(AFR == 4 || (ABL & 8) == 0, "the experimental schedule is written for the 256-row tile");

  int bid = blockIdx.x;
  if constexpr (MODE == 0) {
    if (!remap_balanced(offsets, E, TBM, group_m, n_tiles_n, bid)) return;
  } else {  // wgrad: the static grid is exact
    const int nwg = gridDim.x;
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int e, m0, m_end, n0, k_base = 0, nk = K / 64;
  if constexpr (MODE == 0) {
    const int per_group = group_m * n_tiles_n;
    const int g = bid / per_group, rem = bid % per_group;
    const int mt = g * group_m + rem % group_m;
    const int nt = rem / group_m;
    if (!find_tile(offsets, E, mt, e, m0, m_end, TBM)) return;
    if (group_expert) e = group_expert[e];
    n0 = nt * TBN;
  } else {
    // wgrad: static grid E x m-tiles x n-tiles (group_m carries the number of m-tiles)
    const int per_e = group_m * n_tiles_n;
    e = bid / per_e;
    const int rem = bid % per_e;
    m0 = (rem % group_m) * TBM;
    n0 = (rem / group_m) * TBN;
    m_end = m_rows;
    k_base = offsets[e];
    // MODE 2 takes separate row ranges [offsets[e], group_end[e]) through the `group_expert` argument, which a weight gradient has
    // no other use for (the slots of a static expert exchange: the padding rows behind group_end are never read)
    const int k_end = (MODE == 2 && group_expert) ? group_expert[e] : offsets[e + 1];
    nk = MODE == 2 ? (k_end - k_base + 63) / 64 : (k_end - k_base) / 64;
    out += (int64_t)e * m_rows * N;
  }
  const int k_hi = (MODE == 2) ? (group_expert ? group_expert[e] : offsets[e + 1]) : 0;  // MODE 2: first row past this expert's range

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int l_row = lane >> 3, l_pos = lane & 7;
  const AB* a_src[ASLOTS];
  const AB* w_src[SLOTS];
  // per K-tile advance of the source pointers (elements): 64 k-columns, or 64 token rows in MODE 2
  const int64_t a_step = (MODE == 2) ? (int64_t)64 * m_rows : 64, w_step = (MODE == 2) ? (int64_t)64 * N : 64;
  // MODE 2: 32-bit ELEMENT offsets from A / W instead of per-lane pointers (the launcher keeps both operands under 2^32
  // elements), and ONE running row offset + first column per operand -- slot s reads 64 s columns further, clamped at the
  // matrix edge when its piece is issued: the 320-row tile has no registers to spare for nine 64-bit running pointers
  uint32_t a_rowoff = 0, w_rowoff = 0;
  int a_col0 = 0, w_col0 = 0;
  const int t_row = 8 * wave + l_row;  // MODE 2: token row (inside the K-tile) this lane stages, the same in every slot
  if constexpr (MODE == 2) {
    // 1-KiB piece (s, wave): slab s (64 output columns), token rows 8 wave .. 8 wave + 7 of the K-tile
    const int ch = (((l_pos >> 1) ^ ((t_row >> 1) & 3)) << 1) | (l_pos & 1);   // 32-byte segment swizzle (source side)
    a_col0 = m0 + ch * 8;
    w_col0 = n0 + ch * 8;
    a_rowoff = (uint32_t)((int64_t)(k_base + t_row) * m_rows);
    w_rowoff = (uint32_t)((int64_t)(k_base + t_row) * N);
  }
#pragma unroll
  for (int s = 0; s < ASLOTS && MODE != 2; ++s) {
    const int r = 8 * (s * NW + wave) + l_row;
    int gr = m0 + r;
    if (gr >= m_end) gr = m_end - 1;
    // optional row gather (fused MOEScatter): tile row gr reads source row a_gather[gr] / a_div
    const int64_t arow = (MODE == 0 && a_gather) ? a_gather[gr] / a_div : (int64_t)gr;
    a_src[s] = A + arow * K + k_base + ((l_pos ^ ((r >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int s = 0; s < SLOTS && MODE != 2; ++s) {
    const int r = 8 * (s * NW + wave) + l_row;
    int gw = n0 + r;
    if (gw >= N) gw = N - 1;
    w_src[s] = W + ((int64_t)(MODE == 0 ? e : 0) * N + gw) * K + k_base + ((l_pos ^ ((r >> 1) & 7)) << 3);
  }
  // MODE 2: source of A piece s of K-tile kt -- the zero page once the token row is past the expert's range
  // (only the last K-tile of an expert can hold such rows: a wave-uniform test keeps the per-lane selects out of the
  // steady state)
  // (kt is wave-uniform: the K-tile advance is one scalar multiply and one vector add per DMA)
  auto a_ptr = [&](int s, int kt) -> const AB* {
    const AB* ptr = nullptr;
    if constexpr (MODE != 2) ptr = a_src[s] + kt * a_step;
    if constexpr (MODE == 2) {
      // columns past the matrix: any valid address (those outputs are not stored)
      int ca = a_col0 + 64 * s;
      ca = ca > m_rows - 8 ? m_rows - 8 : ca;
      ptr = A + (a_rowoff + (uint32_t)kt * (uint32_t)a_step + (uint32_t)ca);
      if (kt == nk - 1) {
        const uint64_t keep = (k_base + kt * 64 + t_row < k_hi) ? ~0ull : 0ull;
        ptr = reinterpret_cast<const AB*>((reinterpret_cast<uint64_t>(ptr) & keep) | (reinterpret_cast<uint64_t>(residual) & ~keep));
      }
    }
    return ptr;
  };
  // rows past the range read the zero page too (MODE 2 passes it in `residual`): 0 x 0, never 0 x Inf / NaN from
  // whatever bytes an arbitrary valid address happens to hold
  const AB* const w_safe = (MODE == 2) ? reinterpret_cast<const AB*>(residual) : W;
  auto w_ptr = [&](int s, int kt) -> const AB* {
    const AB* ptr = nullptr;
    if constexpr (MODE != 2) ptr = w_src[s] + kt * w_step;
    if constexpr (MODE == 2) {
      int cw = w_col0 + 64 * s;
      cw = cw > N - 8 ? N - 8 : cw;
      ptr = W + (w_rowoff + (uint32_t)kt * (uint32_t)w_step + (uint32_t)cw);
      if (kt == nk - 1) {
        const uint64_t keep = (k_base + kt * 64 + t_row < k_hi) ? ~0ull : 0ull;
        ptr = reinterpret_cast<const AB*>((reinterpret_cast<uint64_t>(ptr) & keep) | (reinterpret_cast<uint64_t>(w_safe) & ~keep));
      }
    }
    return ptr;
  };
  // DMA of A pieces of K-tile kt into buffer buf: "lo" (s0 == 0) = slots 0,1 (tile rows 0-127, read by wave group 0
  // only), "hi" = slots 2.. (the rest; for the 320-row tile slot 2 straddles the two groups' rows and is re-staged
  // with the group-1 region, after BOTH groups' last reads of it have retired)
  auto dma_a = [&](int kt, int buf, int s0) {
    if ((ABL & 1) && kt >= 2) return;
    char* sa = smem + buf * STAGE;
    if (s0 == 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr(s, kt)),
                                         (__attribute__((address_space(3))) void*)(sa + (s * NW + wave) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int s = 2; s < ASLOTS; ++s)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr(s, kt)),
                                         (__attribute__((address_space(3))) void*)(sa + (s * NW + wave) * 1024), 16, 0, 0);
    }
  };
  auto dma_a1 = [&](int kt, int buf, int s1) {  // one 1-KiB piece
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr(s1, kt)),
                                     (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (s1 * NW + wave) * 1024), 16, 0, 0);
  };
  auto dma_w1 = [&](int kt, int buf, int s1) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr(s1, kt)),
                                     (__attribute__((address_space(3))) void*)(smem + buf * STAGE + TBM * BK_BYTES + (s1 * NW + wave) * 1024), 16, 0, 0);
  };
  auto dma_w = [&](int kt, int buf, int s0) {
    if ((ABL & 1) && kt >= 2) return;
    char* sw = smem + buf * STAGE + TBM * BK_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr(s0 + s, kt)),
                                       (__attribute__((address_space(3))) void*)(sw + ((s0 + s) * NW + wave) * 1024), 16, 0, 0);
  };

  // ---- DEEP layout (ABL & 16): LDS regions = the four fragment halves a K-tile is read in (A-half h = rows
  //      wr*TBM/2 + h*16*AFR .. of both wave groups, B-half h = W rows wc*64 + h*32 .. of all four wave columns), so
  //      every region is read in ONE interval (B-half 0 in two) and can be re-staged for tile t+2 four intervals
  //      later: each half-tile DMA gets >= 6 intervals to land instead of 3, two half-tiles stay in flight across
  //      the tile boundary (cdna_hip_programming.md "256^2 8-phase template": counted vmcnt, never 0 in the loop).
  constexpr bool DEEP = (ABL & 16) != 0;
  constexpr int A_HALF_PIECES = TBM / 16;                      // 8-row DMA pieces per A-half (16 / 20)
  constexpr int A_HS = (A_HALF_PIECES + NW - 1) / NW;          // pieces per wave per A-half, rounded up (2 / 3)
  const AB* a_hsrc[2][A_HS];
  const AB* w_hsrc[2][2];
  if constexpr (DEEP) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int s2 = 0; s2 < A_HS; ++s2) {
        int pi = wave + NW * s2;
        if (pi >= A_HALF_PIECES) pi = A_HALF_PIECES - 1;       // never issued (guarded below); keep the pointer valid
        const int r = pi * 8 + l_row;                          // row inside the half
        const int trow = (r / (16 * AFR)) * (TBM / 2) + h * (16 * AFR) + r % (16 * AFR);
        int gr = m0 + trow;
        if (gr >= m_end) gr = m_end - 1;
        const int64_t arow = (MODE == 0 && a_gather) ? a_gather[gr] / a_div : (int64_t)gr;
        a_hsrc[h][s2] = A + arow * K + k_base + ((l_pos ^ ((r >> 1) & 7)) << 3);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int r = (wave + NW * s2) * 8 + l_row;            // row inside the half (0..127)
        int gw = n0 + (r / 32) * 64 + h * 32 + r % 32;
        if (gw >= N) gw = N - 1;
        w_hsrc[h][s2] = W + ((int64_t)(MODE == 0 ? e : 0) * N + gw) * K + k_base + ((l_pos ^ ((r >> 1) & 7)) << 3);
      }
    }
  }
  auto dma_ah = [&](int kt, int buf, int h) {
    char* sa = smem + buf * STAGE + h * (TBM / 2) * BK_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < A_HS; ++s2) {
      if (A_HALF_PIECES % NW == 0 || s2 + 1 < A_HS || wave < A_HALF_PIECES % NW)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_hsrc[h][s2] + kt * 64),
                                         (__attribute__((address_space(3))) void*)(sa + (wave + NW * s2) * 1024), 16, 0, 0);
    }
  };
  auto dma_wh = [&](int kt, int buf, int h) {
    char* sw = smem + buf * STAGE + (TBM + h * 128) * BK_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_hsrc[h][s2] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(sw + (wave + NW * s2) * 1024), 16, 0, 0);
  };
  // counted wait that leaves this wave's two newest half-tiles (one A-half, one B-half) in flight
  auto wait_keep2 = [&]() {
    if constexpr (A_HALF_PIECES % NW == 0) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      if (wave < A_HALF_PIECES % NW) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
  };

  f32x4 acc[2 * AFR][4];
#pragma unroll
  for (int i = 0; i < 2 * AFR; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 ar[AFR][2], br[2][2];  // current A-half (AFR row fragments x 2 k-steps), B-half (2 col fragments x 2 k-steps)

  const int fr = lane & 15, fq = lane >> 4;

  if (ABL & 2) {
#pragma unroll
    for (int i = 0; i < AFR; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) ar[i][kk] = u32x4{0x3c003800u + lane, 0xbc003400u, 0x38003c00u, 0x3400b800u + i};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) br[i][kk] = u32x4{0x2c002800u + lane, 0xac002400u, 0x28002c00u, 0x2400a800u + i};
  }
  // MODE 2: transposed fragment reads.  Lane (g = fq, qi = fr) addresses token row 32 kk + 4 g + (qi >> 2) (+16 for the
  // second half of the fragment), 8-byte chunk qi & 3 of the 16-column block; the hardware hands lane qi the four
  // tokens' values of column qi.
  // Issued as inline asm: behind the builtin hipcc puts an `s_waitcnt vmcnt(0)` in front of every group of
  // transposing reads (it cannot tell them from the LDS-DMA destinations in flight), which drains the operand pipeline
  // three times per K-tile (486 us per launch instead of ~330).  The price: the compiler does not count these reads
  // either, so the MFMA cluster that consumes them waits lgkmcnt(0) explicitly (PP_MFMA, MODE 2).
  // One address register per 16-column block: the four reads of a fragment pair (kk = 0 / 1, token rows r0 / r0 + 16) differ
  // by immediates (4096 kk + 2048), and the block's swizzle is an XOR into two otherwise unused address bits -- the lane part
  // is a single register, re-derived behind an opaque copy at every use so that the compiler does not keep one address per
  // (buffer, half, fragment, kk) alive through the main loop (the 320-row tile has no registers for that).
  const uint32_t tr_lane = (uint32_t)((4 * fq + (fr >> 2)) * 128 + (fr & 3) * 8) | ((uint32_t)(((4 * fq + (fr >> 2)) >> 1) & 3) << 5);
  auto read_tr2 = [&](uint32_t region, int seg, u32x4& k0, u32x4& k1) {   // region = LDS byte address of the slab
    uint32_t ln = tr_lane;
    asm volatile("" : "+v"(ln));
    const uint32_t a = region + (ln ^ ((uint32_t)seg << 5));
    s16x4 lo0, hi0, lo1, hi1;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo0) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi0) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(lo1) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(hi1) : "v"(a));
    s16x8 v0, v1;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) { v0[q4] = lo0[q4]; v0[4 + q4] = hi0[q4]; v1[q4] = lo1[q4]; v1[4 + q4] = hi1[q4]; }
    k0 = __builtin_bit_cast(u32x4, v0);
    k1 = __builtin_bit_cast(u32x4, v1);
  };
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
  auto read_a = [&](int buf, int half) {
    if constexpr (MODE == 2) {
      // 64 output rows = one [64 tokens][64 columns] slab; this wave's 16-row fragment i of the half is fragment
      // (2 wr + half) AFR + i of the tile: slab = that / 4, 16-column block = that % 4 (320-row tile: a half spans 1.25 slabs)
      const int f0 = (wr * 2 + half) * AFR;
#pragma unroll
      for (int i = 0; i < AFR; ++i)
        read_tr2(smem_lds + (uint32_t)(buf * STAGE + ((f0 + i) >> 2) * 8192), (f0 + i) & 3, ar[i][0], ar[i][1]);
      return;
    }
    if (ABL & 2) { asm volatile("" : "+v"(ar[0][0]), "+v"(ar[1][1])); return; }
    const char* sa = smem + buf * STAGE + (DEEP ? half * (TBM / 2) + wr * (16 * AFR) : wr * (TBM / 2) + half * (16 * AFR)) * BK_BYTES;
#pragma unroll
    for (int i = 0; i < AFR; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) ar[i][kk] = *reinterpret_cast<const u32x4*>(sa + swz(i * 16 + fr, kk * 4 + fq));
  };
  auto read_b = [&](int buf, int half) {
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        read_tr2(smem_lds + (uint32_t)(buf * STAGE + TBM * BK_BYTES + wc * 8192), half * 2 + i, br[i][0], br[i][1]);
      return;
    }
    if (ABL & 2) { asm volatile("" : "+v"(br[0][0]), "+v"(br[1][1])); return; }
    const char* sw = smem + buf * STAGE + (TBM + (DEEP ? half * 128 + wc * 32 : wc * 64 + half * 32)) * BK_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) br[i][kk] = *reinterpret_cast<const u32x4*>(sw + swz(i * 16 + fr, kk * 4 + fq));
  };
#define PP_MFMA(AH, BH)                                                                                              \
  do {                                                                                                               \
    if (ABL & 4) { asm volatile("" :: "v"(ar[0][0]), "v"(ar[AFR - 1][1]), "v"(br[0][0]), "v"(br[1][1])); break; }          \
    if constexpr (MODE == 2) {  /* the inline-asm fragment reads are invisible to the compiler's lgkmcnt bookkeeping */  \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < AFR; ++i)                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
      if constexpr (std::is_same<AB, f16>::value)                                                                    \
        acc[(AH)*AFR + i][(BH)*2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                        \
            __builtin_bit_cast(f16x8, br[j][kk]), __builtin_bit_cast(f16x8, ar[i][kk]), acc[(AH)*AFR + i][(BH)*2 + j], 0, 0, 0); \
      else                                                                                                           \
        acc[(AH)*AFR + i][(BH)*2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                       \
            __builtin_bit_cast(bf16x8_t, br[j][kk]), __builtin_bit_cast(bf16x8_t, ar[i][kk]), acc[(AH)*AFR + i][(BH)*2 + j], 0, 0, 0); \
    }                                                                                                                \
    __builtin_amdgcn_s_setprio(0);                                                                                   \
  } while (0)

  if (nk > 0) {  // block-uniform (an expert without rows has an empty k-range in wgrad mode)
  if constexpr (DEEP) {
    // ---- prologue: tile 0 -> buffer 0, the first two halves of tile 1 -> buffer 1 (they may stay in flight) ----
    dma_ah(0, 0, 0); dma_wh(0, 0, 0); dma_ah(0, 0, 1); dma_wh(0, 0, 1);
    if (nk > 1) {
      dma_ah(1, 1, 0); dma_wh(1, 1, 1);
      wait_keep2();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_BARRIER();
    if (wr == 1) PP_BARRIER();  // stagger: group 1 starts one interval late
    // Invariant at the head of tile t: tile t is in LDS; A-half 0 and B-half 1 of tile t+1 are in flight.
    //   R1(t): A-half 1 of t+1 (that region of the other buffer was last read in R3(t-1))
    //   R2(t): B-half 0 of t+1 (last read R4(t-1))      R3(t): A-half 0 of t+2 (this buffer, last read R1(t))
    //   R4(t): B-half 1 of t+2 (last read R2(t))
    // Every re-stage is >= 4 intervals after the region's last ds_read (3 would do: the reader's lgkmcnt(0) at the
    // head of its next interval plus one barrier for the staggered group); every half-tile has >= 6 intervals to
    // land; the wait that closes tile t keeps the two newest (both for tile t+2) in flight.
    for (int t = 0; t < nk; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      const bool n1 = (t + 1 < nk), n2 = (t + 2 < nk);
      read_b(cur, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_a(cur, 0);
      if (n1) dma_ah(t + 1, nxt, 1);
      PP_BARRIER();
      PP_MFMA(0, 0);
      PP_BARRIER();
      read_b(cur, 1);
      if (n1) dma_wh(t + 1, nxt, 0);
      PP_BARRIER();
      PP_MFMA(0, 1);
      PP_BARRIER();
      read_a(cur, 1);
      if (n2) dma_ah(t + 2, cur, 0);
      PP_BARRIER();
      PP_MFMA(1, 1);
      PP_BARRIER();
      read_b(cur, 0);
      if (n2) dma_wh(t + 2, cur, 1);
      if (wr == 1) {  // group 1: its program interval 6 is global interval 8t+7, the last one of tile t
        if (n2) wait_keep2();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PP_BARRIER();
      PP_MFMA(1, 0);
      if (wr == 0) {  // group 0: program interval 7 = global 8t+7
        if (n2) wait_keep2();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PP_BARRIER();
    }
  } else {
  // ---- prologue: tiles 0 and 1 -> buffers 0 and 1 ------------------------------------------------------
  dma_a(0, 0, 0); dma_a(0, 0, 2); dma_w(0, 0, 0); dma_w(0, 0, 2);
  if (nk > 1) {
    dma_a(1, 1, 0); dma_a(1, 1, 2); dma_w(1, 1, 0); dma_w(1, 1, 2);
    if constexpr (AFR == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 1's pieces may stay in flight
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  PP_BARRIER();
  if (wr == 1) PP_BARRIER();  // stagger: group 1 starts one interval late

  // DMA schedule (all issues sit in READ intervals, behind the ds_reads; MFMA intervals stay pure):
  //   R1(t): A rows 128-255 of tile t+1   R2(t): W rows 0-127 of t+1   R3(t): W rows 128-255 of t+1
  //   R4(t): A rows 0-127 of tile t+2 (that region of tile t's buffer was last read in R3(t))
  // Region reuse is safe because each region's last ds_read of the previous occupant is retired (lgkmcnt(0)
  // at the head of the reader's next MFMA interval) at least one barrier before the issue; the data is
  // needed >= 3 intervals later, and every wave drains all but its 2 newest DMAs before the barrier that
  // closes tile t (counted vmcnt: the newest two belong to tile t+2).
  if constexpr ((ABL & 8) != 0) {
    // experimental schedule: ONE DMA piece per interval per wave (read intervals: behind the ds_reads; MFMA
    // intervals: behind the MFMA cluster), so no wave ever queues two DMA issues back to back
    for (int t = 0; t < nk; ++t) {
      const int cur = t & 1, nxt = cur ^ 1;
      const bool pre1 = (t >= 1) && (t + 1 < nk);
      const bool pre2 = (t + 2 < nk);
      read_b(cur, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_a(cur, 0);
      if (pre1) dma_a1(t + 1, nxt, 2);
      PP_BARRIER();
      PP_MFMA(0, 0);
      if (pre1) dma_a1(t + 1, nxt, 3);
      PP_BARRIER();
      read_b(cur, 1);
      if (pre1) dma_w1(t + 1, nxt, 0);
      PP_BARRIER();
      PP_MFMA(0, 1);
      if (pre1) dma_w1(t + 1, nxt, 1);
      PP_BARRIER();
      read_a(cur, 1);
      if (pre1) dma_w1(t + 1, nxt, 2);
      PP_BARRIER();
      PP_MFMA(1, 1);
      if (pre1) dma_w1(t + 1, nxt, 3);
      PP_BARRIER();
      read_b(cur, 0);
      if (pre2) dma_a1(t + 2, cur, 0);
      if (wr == 1) {  // newest outstanding for group 1 here: only A_lo piece 0 of tile t+2
        if (pre2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PP_BARRIER();
      PP_MFMA(1, 0);
      if (pre2) dma_a1(t + 2, cur, 1);
      if (wr == 0) {
        if (pre2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PP_BARRIER();
    }
  } else {
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    const bool pre1 = (t >= 1) && (t + 1 < nk);
    const bool pre2 = (t + 2 < nk);
    // R1: B-half 0 then A-half 0
    read_b(cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(cur, 0);
    if (pre1) dma_a(t + 1, nxt, 2);
    PP_BARRIER();
    PP_MFMA(0, 0);
    PP_BARRIER();
    // R2: B-half 1
    read_b(cur, 1);
    if (pre1) dma_w(t + 1, nxt, 0);
    PP_BARRIER();
    PP_MFMA(0, 1);
    PP_BARRIER();
    // R3: A-half 1
    read_a(cur, 1);
    if (pre1) dma_w(t + 1, nxt, 2);
    PP_BARRIER();
    PP_MFMA(1, 1);
    PP_BARRIER();
    // R4: B-half 0 again
    read_b(cur, 0);
    if (pre2) dma_a(t + 2, cur, 0);
    if (wr == 1) {  // group 1: its program interval 6 is global interval 8t+7, the last one of tile t
      if (pre2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_BARRIER();
    PP_MFMA(1, 0);
    if (wr == 0) {  // group 0: program interval 7 = global 8t+7
      if (pre2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_BARRIER();
  }
  }
  }  // !DEEP
  if (wr == 0) PP_BARRIER();  // equalise barrier counts; after it every wave is done with LDS
  }
#undef PP_MFMA

  const OT* const resid_e = (MODE == 2) ? nullptr : residual;  // MODE 2 passes its zero page in `residual`
  // ---- epilogue in row passes through LDS (same as the glds variants) --------------------------------
  constexpr int TM = TBM / 2, TN = 64, MI = 2 * AFR, NI = 4;
  constexpr int OB = OutPack<OT>::bytes;
  constexpr int C_STRIDE = TBN * OB + C_PAD;
  constexpr int LDS_BYTES = 2 * STAGE;
  // passes: the smallest divisor of MI / 2 .. whose row block fits in LDS
  constexpr int NPASS = (TBM * C_STRIDE <= LDS_BYTES) ? 1 : ((TBM / 2) * C_STRIDE <= LDS_BYTES ? 2 : (AFR == 4 ? 4 : 5));
  constexpr int RP = TBM / NPASS;
  static_assert(RP * C_STRIDE <= LDS_BYTES, "epilogue pass does not fit in LDS");
  constexpr int CHUNKS = TBN * OB / 16;   // 16-B chunks per tile row
  constexpr int TPR = 16;                 // threads per output row: 16 consecutive threads = 256 contiguous bytes
  constexpr int CPT = CHUNKS / TPR;       // chunks per thread per row (strided by 256 B)
  constexpr int ROWS_PER_IT = NT / TPR;   // 32
  constexpr int ITS = RP / ROWS_PER_IT;
  static_assert(CHUNKS % TPR == 0 && RP % ROWS_PER_IT == 0, "epilogue thread map");
  // A pass stages RP tile rows: RP/2 from each wave group (wr = 0 / 1), i.e. MPP row fragments of EVERY wave, so all
  // eight waves share the bias / GELU / convert work of every pass (a pass made of consecutive tile rows would
  // leave one wave group idle).  LDS row r <-> tile row (r / HALF) * 128 + p * HALF + r % HALF.
  constexpr int HALF = RP / 2;
  constexpr int MPP = HALF / 16;
  static_assert(HALF % 16 == 0 && MPP * NPASS == MI, "epilogue pass split");
  const float* bias_e = bias ? bias + (int64_t)e * N : nullptr;
  f32x4 bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int nl = wc * TN + ni * 16 + fq * 4;
    bv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias_e && n0 + nl < N) bv[ni] = *reinterpret_cast<const f32x4*>(bias_e + n0 + nl);
  }
  const int trow = tid / TPR, tcol = tid % TPR;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    // (1) resolve this pass's output rows and start the resid_e loads: their latency hides under (2)
    int64_t orow[ITS];
    float oscale[ITS];
    u32x4 resv[ITS][CPT];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int r = trow + it * ROWS_PER_IT;
      const int m = m0 + (r / HALF) * TM + p * HALF + (r % HALF);
      orow[it] = -1;
      oscale[it] = 1.f;
      if (m < m_end) {
        orow[it] = row_map ? row_map[m] : (int64_t)m;
        if (row_map && row_scale) oscale[it] = row_scale[orow[it]];
      }
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        resv[it][j] = u32x4{0u, 0u, 0u, 0u};
        const int ncol = n0 + (tcol + j * TPR) * (16 / OB);
        if (resid_e && orow[it] >= 0 && ncol < N)
          resv[it][j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(resid_e) +
                                                        (orow[it] * (int64_t)N + ncol) * OB);
      }
    }
    // (2) bias (+GELU), convert, stage this pass's fragments in LDS
#pragma unroll
    for (int mm = 0; mm < MPP; ++mm) {
      const int mi = p * MPP + mm;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f32x4 v = acc[mi][ni] + bv[ni];
        if (epilogue == SMOE_EPI_GELU) {
          v = gelu_fast4(v);
        }
        const int nl = wc * TN + ni * 16 + fq * 4;
        OutPack<OT>::write4(smem + (wr * HALF + mm * 16 + fr) * C_STRIDE + nl * OB, v);
      }
    }
    __syncthreads();
    // (3) whole-row-segment stores (combine scale and resid_e / gelu' fused)
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      if (orow[it] >= 0) {
        const int r = trow + it * ROWS_PER_IT;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
          const int ch = tcol + j * TPR;
          const int ncol = n0 + ch * (16 / OB);
          if (ncol < N) {
            u32x4 v = *reinterpret_cast<const u32x4*>(smem + r * C_STRIDE + ch * 16);
            if (row_map && row_scale) v = scale16<OT>(v, oscale[it]);
            if (resid_e) v = fuse_aux<OT>(epilogue, resv[it][j], v);
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + (orow[it] * (int64_t)N + ncol) * OB) = v;
          }
        }
      }
    }
    if (p + 1 < NPASS) __syncthreads();
  }
}

template <typename AB, typename OT, int ABL = 0, int AFR = 4>
int launch_pp256(const void* A, const void* W, const float* bias, const int32_t* offsets, const int32_t* group_expert,
                 int E, int64_t m_rows_max, int K, int N, int epilogue, const int64_t* row_map, const float* row_scale,
                 const void* residual, void* out, int group_m, hipStream_t s, const int64_t* a_gather = nullptr,
                 int a_div = 1) {
  constexpr int TBM = 64 * AFR, TBN = 256;
  const int n_tiles_n = (N + TBN - 1) / TBN;
  const int max_m_tiles = (int)((m_rows_max + TBM - 1) / TBM) + E;
  const int m_groups = (max_m_tiles + group_m - 1) / group_m;
  const int grid = m_groups * group_m * n_tiles_n;
  const size_t smem = 2 * (size_t)(TBM + TBN) * BK_BYTES;
  auto kern = grouped_gemm_pp256<AB, OT, ABL, 0, AFR>;
  SMOE_ENSURE_SMEM(grouped_gemm_pp256<AB, OT, ABL, 0, AFR>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, (const AB*)A, (const AB*)W, bias, offsets, group_expert, E, K, N,
                     epilogue, row_map, row_scale, (const OT*)residual, (OT*)out, n_tiles_n, group_m, 0, a_gather, a_div);
  SMOE_CHECK_LAUNCH("smoe_grouped_gemm");
  return 0;
}

#include "gemm_persistent.h"

template <typename AB>
int launch_wgrad(const void* PT, const void* QT, const int32_t* offsets_pad, int E, int R1, int R2, int Lp, float* out,
                 hipStream_t s) {
  constexpr int TBM = 256, TBN = 256;
  const int tm = (R1 + TBM - 1) / TBM, tn = (R2 + TBN - 1) / TBN;
  const int grid = E * tm * tn;
  const size_t smem = 2 * (size_t)(TBM + TBN) * BK_BYTES;
  auto kern = grouped_gemm_pp256<AB, float, 0, 1>;
  SMOE_ENSURE_SMEM(grouped_gemm_pp256<AB, float, 0, 1>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, (const AB*)PT, (const AB*)QT, (const float*)nullptr, offsets_pad,
                     (const int32_t*)nullptr, E, Lp, R2, (int)SMOE_EPI_NONE, (const int64_t*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, out, tn, tm, R1, (const int64_t*)nullptr, 1);
  SMOE_CHECK_LAUNCH("smoe_grouped_wgrad");
  return 0;
}

// token-major weight gradient (MODE 2): P [n_rows, R1], Q [n_rows, R2] row-major, offsets = plain row ranges
template <typename AB>
int launch_wgrad_rows(const void* P, const void* Q, const int32_t* offsets, const int32_t* group_end, int E, int R1, int R2,
                      const void* zero16, float* out, hipStream_t s) {
  // Tile height 256 or 320 output rows, whichever needs fewer cost-weighted rounds of workgroups: the static grid has E x
  // tm x tn equal tiles, e.g. ViT-B's dW1 [3072, 768] x 8 experts = 288 tiles of 256 rows (two rounds on 256 CUs, the
  // second one an eighth full) or 240 of 320 rows (one round).
  constexpr int TBN = 256;
  const int tn = (R2 + TBN - 1) / TBN, tm4 = (R1 + 255) / 256, tm5 = (R1 + 319) / 320;
  const int64_t cus = smoe_num_cus();
  const double c4 = (double)(((int64_t)E * tm4 * tn + cus - 1) / cus), c5 = 1.25 * (double)(((int64_t)E * tm5 * tn + cus - 1) / cus);
  if (c5 < c4) {
    const size_t smem = 2 * (size_t)(320 + TBN) * BK_BYTES;
    auto kern = grouped_gemm_pp256<AB, float, 0, 2, 5>;
    SMOE_ENSURE_SMEM(grouped_gemm_pp256<AB, float, 0, 2, 5>);
    hipLaunchKernelGGL(kern, dim3(E * tm5 * tn), dim3(512), smem, s, (const AB*)P, (const AB*)Q, (const float*)nullptr, offsets,
                       group_end, E, 0, R2, (int)SMOE_EPI_NONE, (const int64_t*)nullptr, (const float*)nullptr,
                       (const float*)zero16, out, tn, tm5, R1, (const int64_t*)nullptr, 1);
  } else {
    const size_t smem = 2 * (size_t)(256 + TBN) * BK_BYTES;
    auto kern = grouped_gemm_pp256<AB, float, 0, 2>;
    SMOE_ENSURE_SMEM(grouped_gemm_pp256<AB, float, 0, 2>);
    hipLaunchKernelGGL(kern, dim3(E * tm4 * tn), dim3(512), smem, s, (const AB*)P, (const AB*)Q, (const float*)nullptr, offsets,
                       group_end, E, 0, R2, (int)SMOE_EPI_NONE, (const int64_t*)nullptr, (const float*)nullptr,
                       (const float*)zero16, out, tn, tm4, R1, (const int64_t*)nullptr, 1);
  }
  SMOE_CHECK_LAUNCH("smoe_grouped_wgrad_rows");
  return 0;
}

template <typename AB, typename OT>
int launch_variant(int variant, const void* A, const void* W, const float* bias, const int32_t* offsets,
                   const int32_t* group_expert, int E, int64_t m_rows_max, int K, int N, int epilogue,
                   const int64_t* row_map, const float* row_scale, const void* residual, void* out, hipStream_t s,
                   const int64_t* a_gather, int a_div, const int32_t* group_end, int64_t out_rows) {
  if constexpr (sizeof(AB) == 2) {
    switch (variant) {
      case 1: return launch_glds<AB, OT, 128, 128, 2, 2, 2>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 8, s);
      case 2: return launch_glds<AB, OT, 256, 128, 2, 2, 1>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
      case 3: return launch_glds<AB, OT, 256, 256, 2, 4, 2>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
      case 4: {  // auto: 256- or 320-row tiles, whichever needs fewer (cost-weighted) rounds of workgroups
        const int ntn = (N + 255) / 256;
        // expected tile counts (each group's last tile is half full on average); the kernel balances the real ones
        const int64_t t256 = ((m_rows_max + 255) / 256 + E / 2) * ntn, t320 = ((m_rows_max + 319) / 320 + E / 2) * ntn;
        const int cus = smoe_num_cus();
        const double c256 = (double)((t256 + cus - 1) / cus) * 1.0, c320 = (double)((t320 + cus - 1) / cus) * 1.25;
        // deep = half-organised LDS with two half-tiles in flight across the tile boundary (variants 7 / 8): +3-4 % on
        // long K loops (GEMM-2, 8192^3), -3 % at K = 768 where the uneven per-wave DMA split of the 320-row tile shows
        const bool deep = K >= 2048;
        if (c320 <= c256) {  // ties go to the taller tile (more FLOP per LDS-fill byte)
          if (deep) return launch_pp256<AB, OT, 16, 5>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div);
          return launch_pp256<AB, OT, 0, 5>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div);
        }
        if (deep) return launch_pp256<AB, OT, 16, 4>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div);
        return launch_pp256<AB, OT>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div);
      }
#ifdef SMOE_DIAG
      case 5: {
        const char* gm = getenv("SMOE_GROUP_M");
        return launch_pp256<AB, OT, 0, 5>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, gm ? atoi(gm) : 4, s, a_gather, a_div);
      }
#else
      case 5: return launch_pp256<AB, OT, 0, 5>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div);
#endif
      case 9: {  // as 4 (auto tile height / schedule), on the persistent kernel
        const int ntn = (N + 255) / 256;
        const int64_t t256 = ((m_rows_max + 255) / 256 + E / 2) * ntn, t320 = ((m_rows_max + 319) / 320 + E / 2) * ntn;
        const int cus = smoe_num_cus();
        const double c256 = (double)((t256 + cus - 1) / cus) * 1.0, c320 = (double)((t320 + cus - 1) / cus) * 1.25;
        const bool deep = K >= 2048;
        if (c320 <= c256) {
          if (deep) return launch_ps<AB, OT, 5, true>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows);
          return launch_ps<AB, OT, 5, false>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows);
        }
        if (deep) return launch_ps<AB, OT, 4, true>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows);
        return launch_ps<AB, OT, 4, false>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows);
      }
      case 14: {  // as 9 with the LDS-staged epilogue everywhere (A/B reference of the direct-store epilogue)
        const int ntn = (N + 255) / 256;
        const int64_t t256 = ((m_rows_max + 255) / 256 + E / 2) * ntn, t320 = ((m_rows_max + 319) / 320 + E / 2) * ntn;
        const int cus = smoe_num_cus();
        const double c256 = (double)((t256 + cus - 1) / cus) * 1.0, c320 = (double)((t320 + cus - 1) / cus) * 1.25;
        const bool deep = K >= 2048;
        if (c320 <= c256) {
          if (deep) return launch_ps<AB, OT, 5, true>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows, false);
          return launch_ps<AB, OT, 5, false>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows, false);
        }
        if (deep) return launch_ps<AB, OT, 4, true>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows, false);
        return launch_ps<AB, OT, 4, false>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows, false);
      }
      case 10: return launch_ps<AB, OT, 5, false>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows);
      case 11: return launch_ps<AB, OT, 4, false>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows);
      case 12: return launch_ps<AB, OT, 4, true>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows);
      case 13: return launch_ps<AB, OT, 5, true>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div, group_end, out_rows);
      case 6: return launch_pp256<AB, OT>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div);
      case 7: return launch_pp256<AB, OT, 16, 4>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div);
      case 8: return launch_pp256<AB, OT, 16, 5>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s, a_gather, a_div);
#ifdef SMOE_DIAG
      case 41: return launch_pp256<AB, OT, 1>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
      case 42: return launch_pp256<AB, OT, 2>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
      case 43: return launch_pp256<AB, OT, 3>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
      case 44: return launch_pp256<AB, OT, 4>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
      case 46: return launch_pp256<AB, OT, 6>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
      case 48: return launch_pp256<AB, OT, 8>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
      case 47: return launch_pp256<AB, OT, 7>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, 4, s);
#endif
      default: break;
    }
  }
  return launch_t128<AB, OT>(A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, s);
}

template <typename AB>
int dispatch_out(int variant, const void* A, const void* W, const float* bias, const int32_t* offsets, const int32_t* group_expert,
                 int E, int64_t m_rows_max, int K, int N, int epilogue, const int64_t* row_map, const float* row_scale,
                 const void* residual, void* out, int out_dtype, hipStream_t s, const int64_t* a_gather, int a_div,
                 const int32_t* group_end, int64_t out_rows) {
  switch (out_dtype) {
    case SMOE_F32: return launch_variant<AB, float>(variant, A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, s, a_gather, a_div, group_end, out_rows);
    case SMOE_F16: return launch_variant<AB, f16>(variant, A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, s, a_gather, a_div, group_end, out_rows);
    case SMOE_BF16: return launch_variant<AB, bf16_bits>(variant, A, W, bias, offsets, group_expert, E, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, s, a_gather, a_div, group_end, out_rows);
  }
  smoe_set_error("smoe_grouped_gemm: bad out_dtype %d", out_dtype);
  return 1;
}

}  // namespace

#ifdef SMOE_DIAG
extern "C" int smoe_diag_read_stamps(unsigned long long* host_out, size_t n) {
  const size_t cap = sizeof(unsigned long long) * 256 * 12 * 16;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(smoe_diag_stamps), n * 8 < cap ? n * 8 : cap);
}
extern "C" int smoe_diag_set_flags(int flags) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(smoe_diag_flags), &flags, sizeof(flags));
}
extern "C" int smoe_diag_clear_stamps() {
  static unsigned long long z[256 * 12 * 16];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(smoe_diag_stamps), z, sizeof(z));
}
#endif
#ifdef SMOE_CLOCK
// clock-probe build only: [1024 workgroups][start memtime, start memrealtime, end memtime, end memrealtime, fused: spins, GEMM-1
// tiles, GEMM-2 tiles, -] of the LAST persistent-GEMM launch (the stamps are overwritten, the counts accumulate until cleared)
extern "C" int smoe_clock_read_stamps(unsigned long long* host_out, size_t n) {
  const size_t cap = sizeof(unsigned long long) * 1024 * 8;
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(smoe_clock_stamps), n * 8 < cap ? n * 8 : cap);
}
extern "C" int smoe_clock_clear_stamps() {
  static unsigned long long z[1024 * 8];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(smoe_clock_stamps), z, sizeof(z));
}
#endif

extern "C" int smoe_grouped_gemm(const void* A, const void* W, const float* bias, const int32_t* offsets,
                                 const int32_t* group_expert, int G, int n_experts, int64_t m_rows_max, int K, int N,
                                 int ab_dtype, int epilogue, const int64_t* row_map, const float* row_scale,
                                 const void* residual, const int64_t* a_gather, int a_div, void* out, int64_t out_rows,
                                 int out_dtype, int variant, const int32_t* group_end, void* stream) {
  SMOE_REQUIRE(offsets && G >= 1 && G <= 65536, "smoe_grouped_gemm: bad G=%d / offsets", G);
  SMOE_REQUIRE(n_experts >= 1 && (group_expert || n_experts == G), "smoe_grouped_gemm: n_experts=%d != G=%d without a group map", n_experts, G);
  SMOE_REQUIRE(m_rows_max >= 0 && m_rows_max < (1ll << 31), "smoe_grouped_gemm: m_rows_max=%lld out of range",
               (long long)m_rows_max);
  SMOE_REQUIRE(K > 0 && N > 0, "smoe_grouped_gemm: bad K=%d N=%d", K, N);
  SMOE_REQUIRE(out_rows >= 0 && (out_rows == 0 || row_map || out_rows >= m_rows_max),
               "smoe_grouped_gemm: out_rows=%lld is smaller than m_rows_max=%lld", (long long)out_rows, (long long)m_rows_max);
  if (out_rows == 0 && !row_map) out_rows = m_rows_max;   // without a row map row r is stored to out[r]
  SMOE_REQUIRE(epilogue == SMOE_EPI_NONE || epilogue == SMOE_EPI_GELU || epilogue == SMOE_EPI_GELU_GRAD,
               "smoe_grouped_gemm: bad epilogue %d", epilogue);
  SMOE_REQUIRE(epilogue != SMOE_EPI_GELU_GRAD || (residual && !row_map),
               "smoe_grouped_gemm: SMOE_EPI_GELU_GRAD needs the pre-activations in `residual` and no row_map");
  SMOE_REQUIRE(smoe_dtype_ok(ab_dtype) && smoe_dtype_ok(out_dtype), "smoe_grouped_gemm: bad dtype");
  const int bke = BK_BYTES / smoe_dtype_size(ab_dtype);
  SMOE_REQUIRE(K % bke == 0, "smoe_grouped_gemm: K=%d must be a multiple of %d for this dtype", K, bke);
  SMOE_REQUIRE(N % 8 == 0, "smoe_grouped_gemm: N=%d must be a multiple of 8", N);
  if (m_rows_max == 0) return 0;
  SMOE_REQUIRE(A && W && out, "smoe_grouped_gemm: null pointer");
  if (K % 64 != 0 || smoe_dtype_size(ab_dtype) != 2) variant = 0;
  SMOE_REQUIRE(!a_gather || (variant >= 4 && variant <= 14 && a_div >= 1),
               "smoe_grouped_gemm: a_gather needs variant 4-14 (16-bit operands, K %% 64 == 0)");
  if (variant >= 9 && variant <= 14) {
    // the persistent kernel addresses both operands with 32-bit byte offsets; operands of 4 GiB and more take the
    // one-workgroup-per-tile kernel of the same tile height / schedule (9, 14 -> 4, 10 -> 5, 11 -> 6, 12 -> 7, 13 -> 8)
    const uint64_t a_bytes = (uint64_t)m_rows_max * (uint64_t)K * 2u, w_bytes = (uint64_t)n_experts * (uint64_t)N * (uint64_t)K * 2u;
    if (a_bytes >= (1ull << 32) || w_bytes >= (1ull << 32) || G > 63) variant = variant == 14 ? 4 : variant - 5;   // (its group table: 64 lanes)
  }
  SMOE_REQUIRE(!group_end || (variant >= 9 && variant <= 14),
               "smoe_grouped_gemm: group_end (separate row ranges per group) needs the persistent kernel: variant 9-14, 16-bit "
               "operands under 4 GiB, K %% 64 == 0, at most 63 groups");
  hipStream_t s = (hipStream_t)stream;
  switch (ab_dtype) {
    case SMOE_F32: return dispatch_out<float>(variant, A, W, bias, offsets, group_expert, G, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, out_dtype, s, a_gather, a_div, group_end, out_rows);
    case SMOE_F16: return dispatch_out<f16>(variant, A, W, bias, offsets, group_expert, G, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, out_dtype, s, a_gather, a_div, group_end, out_rows);
    case SMOE_BF16: return dispatch_out<bf16_bits>(variant, A, W, bias, offsets, group_expert, G, m_rows_max, K, N, epilogue, row_map, row_scale, residual, out, out_dtype, s, a_gather, a_div, group_end, out_rows);
  }
  return 1;
}

#ifdef SMOE_FFN_FUSED
// The expert FFN of one MoE layer (top-1 combine fused) as ONE persistent launch: see expert_ffn_fused in gemm_persistent.h.
// Returns -1 (no error set) for what that launch does not cover -- the caller then issues the two smoe_grouped_gemm launches.
extern "C" size_t smoe_expert_ffn_workspace_bytes(int64_t m_rows_max, int G) {
  return sizeof(int32_t) * (size_t)(FUSED_WS_HDR + (m_rows_max + 319) / 320 + (G > 0 ? G : 0) + 8);   // tickets, counters
}
extern "C" int smoe_expert_ffn(const void* X, const int64_t* a_gather, int a_div, const void* W1, const float* b1, void* H,
                               const void* W2, const float* b2, const int32_t* offsets, const int32_t* group_expert, int G,
                               int n_experts, int64_t m_rows_max, int d_in, int d_hidden, int d_out, int ab_dtype,
                               const int64_t* row_map, const float* row_scale, const void* residual, void* out, int64_t out_rows,
                               int out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
  SMOE_REQUIRE(offsets && G >= 1 && n_experts >= 1 && (group_expert || n_experts == G), "smoe_expert_ffn: bad groups");
  SMOE_REQUIRE(m_rows_max >= 0 && m_rows_max < (1ll << 31) && d_in > 0 && d_hidden > 0 && d_out > 0, "smoe_expert_ffn: bad sizes");
  SMOE_REQUIRE(a_div >= 1, "smoe_expert_ffn: a_div must be >= 1");
  if (m_rows_max == 0) return 0;
  SMOE_REQUIRE(X && W1 && H && W2 && out && workspace, "smoe_expert_ffn: null pointer");
  SMOE_REQUIRE(workspace_bytes >= smoe_expert_ffn_workspace_bytes(m_rows_max, G), "smoe_expert_ffn: workspace too small");
  if ((ab_dtype != SMOE_F16 && ab_dtype != SMOE_BF16) || out_dtype != SMOE_F32) return -1;
  if (d_in % 64 || d_hidden % 64 || d_out % 8 || G > 63) return -1;
  const uint64_t x_bytes = (uint64_t)m_rows_max * (uint64_t)(d_in > d_hidden ? d_in : d_hidden) * 2u;
  const uint64_t w_bytes = (uint64_t)n_experts * (uint64_t)d_hidden * (uint64_t)(d_in > d_out ? d_in : d_out) * 2u;
  if (x_bytes >= (1ull << 32) || w_bytes >= (1ull << 32)) return -1;            // 32-bit operand offsets
  if (out_rows <= 0 || out_rows * (int64_t)d_out * 4 >= (1ll << 31)) return -1;   // the buffer-addressed f32 epilogue's 2 GiB
  if ((int64_t)d_hidden * 320 * 2 >= (1ll << 31)) return -1;
  const int ntn1 = (d_hidden + 255) / 256, ntn2 = (d_out + 255) / 256;
  const int64_t max_mt = (m_rows_max + 319) / 320 + G;
  int grid = smoe_num_cus() & ~7;
  if (grid < 8) grid = 8;
  if (max_mt * ntn1 < grid) grid = (int)((max_mt * ntn1 + 7) & ~(int64_t)7);
  // tile costs in half K-tiles (main loop = 2 per K-tile) + the tile boundary as measured (profiles/r03_gemm*_tile_stamps.txt:
  // 14.7 k of 55.2 k cycles for the direct 16-bit epilogue, 47 k of 203 k for the f32 residual epilogue)
  static const int c1_env = getenv("SMOE_FUSED_C1") ? atoi(getenv("SMOE_FUSED_C1")) : 0;
  static const int c2_env = getenv("SMOE_FUSED_C2") ? atoi(getenv("SMOE_FUSED_C2")) : 0;
  // (SMOE_FUSED_C1=-1: the serial order, every GEMM-1 tile in front of every GEMM-2 tile, for A/B)
  const int c1 = c1_env != 0 ? c1_env : 2 * (d_in / 64) + 10, c2 = c2_env > 0 ? c2_env : 2 * (d_hidden / 64) + 29;
  hipStream_t s = (hipStream_t)stream;
  if (ab_dtype == SMOE_F16) {
    SMOE_ENSURE_SMEM(expert_ffn_fused<f16>);
    hipLaunchKernelGGL((expert_ffn_fused<f16>), dim3(grid), dim3(512), 160 * 1024, s, (const f16*)X, a_gather, a_div, (int)m_rows_max,
                       (const f16*)W1, b1, (f16*)H, (const f16*)W2, b2, offsets, group_expert, G, d_in, d_hidden, d_out, row_map,
                       row_scale, (const float*)residual, (float*)out, ntn1, ntn2, 4, c1, c2, (int32_t*)workspace);
  } else {
    SMOE_ENSURE_SMEM(expert_ffn_fused<bf16_bits>);
    hipLaunchKernelGGL((expert_ffn_fused<bf16_bits>), dim3(grid), dim3(512), 160 * 1024, s, (const bf16_bits*)X, a_gather, a_div,
                       (int)m_rows_max, (const bf16_bits*)W1, b1, (bf16_bits*)H, (const bf16_bits*)W2, b2, offsets, group_expert, G,
                       d_in, d_hidden, d_out, row_map, row_scale, (const float*)residual, (float*)out, ntn1, ntn2, 4, c1, c2,
                       (int32_t*)workspace);
  }
  SMOE_CHECK_LAUNCH("smoe_expert_ffn");
  return 0;
}

#endif  // SMOE_FFN_FUSED

// First expert linear of the TRAINING forward: pre_out = A W^T + bias (kept for gelu' in the backward) and out = gelu(pre_out),
// both in the operand dtype, from one epilogue of the persistent kernel.  Returns -1 when the shape is outside that kernel's
// reach (K % 64, operands of 4 GiB and more, more than 63 row groups): the caller then runs SMOE_EPI_NONE + smoe_gelu.
extern "C" int smoe_grouped_gemm_gelu_keep(const void* A, const void* W, const float* bias, const int32_t* offsets,
                                           const int32_t* group_expert, const int32_t* group_end, int G, int n_experts,
                                           int64_t m_rows_max, int K, int N, int ab_dtype, void* pre_out, void* out, void* stream) {
  SMOE_REQUIRE(offsets && G >= 1 && n_experts >= 1 && (group_expert || n_experts == G), "smoe_grouped_gemm_gelu_keep: bad groups");
  SMOE_REQUIRE(m_rows_max >= 0 && m_rows_max < (1ll << 31) && K > 0 && N > 0 && N % 8 == 0, "smoe_grouped_gemm_gelu_keep: bad sizes");
  SMOE_REQUIRE(ab_dtype == SMOE_F16 || ab_dtype == SMOE_BF16, "smoe_grouped_gemm_gelu_keep: 16-bit operands only");
  if (m_rows_max == 0) return 0;
  SMOE_REQUIRE(A && W && pre_out && out, "smoe_grouped_gemm_gelu_keep: null pointer");
  const uint64_t a_bytes = (uint64_t)m_rows_max * (uint64_t)K * 2u, w_bytes = (uint64_t)n_experts * (uint64_t)N * (uint64_t)K * 2u;
  if (K % 64 != 0 || G > 63 || a_bytes >= (1ull << 32) || w_bytes >= (1ull << 32)) return -1;
  hipStream_t s = (hipStream_t)stream;
  if (ab_dtype == SMOE_F16)
    return launch_ps<f16, f16, 5, false, true>(A, W, bias, offsets, group_expert, G, m_rows_max, K, N, SMOE_EPI_GELU, nullptr, nullptr, pre_out, out, 4, s, nullptr, 1, group_end, 0);
  return launch_ps<bf16_bits, bf16_bits, 5, false, true>(A, W, bias, offsets, group_expert, G, m_rows_max, K, N, SMOE_EPI_GELU, nullptr, nullptr, pre_out, out, 4, s, nullptr, 1, group_end, 0);
}

// Weight gradients of a grouped linear (fmoe_cuda.linear_backward's grad_W):
//   out[e] (f32 [R1,R2]) = PT[:, offsets_pad[e]:offsets_pad[e+1]] @ QT[:, same]^T
// PT [R1,Lp], QT [R2,Lp] are the K-major, 64-padded images made by smoe_transpose_pad (e.g. PT = dY^T, QT = A^T
// gives dW2[e] = dY_e^T A_e in W2's [d,h] layout).  16-bit operands, f32 result.
extern "C" int smoe_grouped_wgrad(const void* PT, const void* QT, int ab_dtype, const int32_t* offsets_pad, int E, int R1,
                                  int R2, int Lp, float* out, void* stream) {
  SMOE_REQUIRE(PT && QT && offsets_pad && out, "smoe_grouped_wgrad: null pointer");
  SMOE_REQUIRE(E >= 1 && R1 > 0 && R2 > 0 && Lp > 0 && Lp % 64 == 0 && R2 % 8 == 0,
               "smoe_grouped_wgrad: bad sizes E=%d R1=%d R2=%d Lp=%d", E, R1, R2, Lp);
  hipStream_t s = (hipStream_t)stream;
  if (ab_dtype == SMOE_F16) return launch_wgrad<f16>(PT, QT, offsets_pad, E, R1, R2, Lp, out, s);
  if (ab_dtype == SMOE_BF16) return launch_wgrad<bf16_bits>(PT, QT, offsets_pad, E, R1, R2, Lp, out, s);
  smoe_set_error("smoe_grouped_wgrad: operands must be f16 or bf16");
  return 1;
}

// Weight gradients straight from the token-major operands (no transposed copies):
//   out[e] (f32 [R1,R2]) = sum over rows r of expert e of P[r, :]^T Q[r, :]     P [n_rows,R1], Q [n_rows,R2], 16-bit
// offsets i32 [E+1] = the experts' row ranges (any lengths, empty allowed); zero16 = 16 bytes of zeros in device memory.
// group_end (optional, i32 [E]): expert e's rows are [offsets[e], group_end[e]) -- separate ranges, nothing behind group_end[e] is read.
extern "C" int smoe_grouped_wgrad_rows(const void* P, const void* Q, int ab_dtype, const int32_t* offsets,
                                       const int32_t* group_end, int E, int R1, int R2, const void* zero16, float* out,
                                       void* stream) {
  SMOE_REQUIRE(P && Q && offsets && out && zero16, "smoe_grouped_wgrad_rows: null pointer");
  SMOE_REQUIRE(E >= 1 && R1 >= 8 && R2 >= 8 && R1 % 8 == 0 && R2 % 8 == 0, "smoe_grouped_wgrad_rows: bad sizes E=%d R1=%d R2=%d", E, R1, R2);
  hipStream_t s = (hipStream_t)stream;
  if (ab_dtype == SMOE_F16) return launch_wgrad_rows<f16>(P, Q, offsets, group_end, E, R1, R2, zero16, out, s);
  if (ab_dtype == SMOE_BF16) return launch_wgrad_rows<bf16_bits>(P, Q, offsets, group_end, E, R1, R2, zero16, out, s);
  smoe_set_error("smoe_grouped_wgrad_rows: operands must be f16 or bf16");
  return 1;
}
