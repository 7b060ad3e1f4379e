// The embedding stage of the ViT forward and its last LayerNorm (models/vision_transformer.py:818-830), the caller-side glue in
// front of / behind the blocks, as three small HBM-bound kernels:
//   smoe_patchify_cast   images f32 [B, C, H, W] -> patch rows [B * gh * gw, C * ph * pw] in 16 bit: the operand of the per-patch
//                        projection GEMM (PatchEmbed's Conv2d with kernel == stride), read once, written once
//   smoe_embed_ln        x = cat(cls_token, tokens) + pos_embed (the f32 residual stream) AND LayerNorm(x) in 16 bit -- block 0's
//                        norm1 -- in one pass over the projected tokens: the broadcast add, the class-token row and a whole
//                        LayerNorm pass (154 MB read) become one kernel
//   smoe_layernorm_rows  LayerNorm of rows that are row_stride elements apart (the class-token rows x[:, 0] in front of the head)
#include "router16_kernel.h"

namespace {
using namespace r16;

// one thread per 4 consecutive pixels of a patch row (16-byte load, 8-byte store); consecutive threads walk a patch's
// (c, py, px) order = the output row, so stores are contiguous and loads come in 64-byte runs (one image row of a patch)
template <typename OT>
__global__ __launch_bounds__(256) void patchify_cast_kernel(const float* __restrict__ img, OT* __restrict__ out, int64_t n4, int C,
                                                            int H, int W, int ph, int pw, int gh, int gw) {
  const int row_len = C * ph * pw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t o = i * 4;
    const int64_t row = o / row_len;
    const int col = (int)(o - row * row_len);
    const int c = col / (ph * pw), rem = col - c * (ph * pw);
    const int py = rem / pw, px = rem - py * pw;
    const int64_t b = row / (gh * gw);
    const int pr = (int)(row - b * (gh * gw));
    const int gy = pr / gw, gx = pr - gy * gw;
    const float* src = img + ((b * C + c) * (int64_t)H + (gy * ph + py)) * W + gx * pw + px;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src);
    store4_16<OT>(out + o, v);
  }
}

// 16 lanes per token row (the router's layout): row (b, n) of the stream is tokens[b * P + n - 1] (n >= 1) or the class token
// (n == 0), plus pos[n]; LayerNorm over it with the arithmetic of layernorm16_kernel / router16_kernel<LN>.
// (launch bounds: 3 waves per SIMD -- every load of a row is issued before the first use; gamma / beta are loaded behind the statistics
//  (158 VGPRs at d = 768; with them hoisted the kernel needed 182 = 2 waves per SIMD, 96-100 us; now 89-92 us; 4 waves per SIMD = 128
//  VGPRs, 5 spills: the same 90 us))
template <typename TT, typename NT, int NJ>
__global__ __launch_bounds__(R16_THREADS, 3) void embed_ln_kernel(const TT* __restrict__ tok, const float* __restrict__ cls,
                                                                  const float* __restrict__ pos, const float* __restrict__ g,
                                                                  const float* __restrict__ be, float eps, int64_t B, int P,
                                                                  float* __restrict__ x32, NT* __restrict__ xn) {
  constexpr int d = 64 * NJ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, u = lane & 15;
  const int64_t T = B * (int64_t)(P + 1);
  const int64_t stride = (int64_t)gridDim.x * (R16_THREADS / 64) * 4;
  for (int64_t it0 = ((int64_t)blockIdx.x * (R16_THREADS / 64) + wave) * 4; it0 < T; it0 += stride) {
    const int64_t t = it0 + q;
    const bool live = t < T;
    const int64_t tc = live ? t : T - 1;
    const int64_t b = tc / (P + 1);
    const int n = (int)(tc - b * (P + 1));
    f32x4 xv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = u * 4 + 64 * j;
      f32x4 v;
      if (n == 0) {
        v = *reinterpret_cast<const f32x4*>(cls + c);
      } else {
        float tmp[4];
        load4(tok + (b * P + (n - 1)) * (int64_t)d + c, tmp);
        v = f32x4{tmp[0], tmp[1], tmp[2], tmp[3]};
      }
      xv[j] = v + *reinterpret_cast<const f32x4*>(pos + (int64_t)n * d + c);
    }
    if (live) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4*>(x32 + tc * (int64_t)d + u * 4 + 64 * j) = xv[j];
    }
    if (xn) {
      float s1 = 0.f;   // (the arithmetic of layernorm16_kernel, operation for operation: the same bits as smoe_layernorm)
#pragma unroll
      for (int j = 0; j < NJ; ++j) s1 += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
      const float mean = row16_sum(s1) / (float)d;
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float dv = xv[j][i] - mean; s2 = fmaf(dv, dv, s2); }
      const float rstd = rsqrtf(row16_sum(s2) / (float)d + eps);
      if (live) {
        // gamma / beta behind a per-row opaque zero: otherwise the loop-invariant loads are hoisted in front of the row's own loads
        // and held across the token loop (96 registers at d = 768: the difference between two and three waves per SIMD)
        int lz = 0;
        asm volatile("" : "+v"(lz));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = u * 4 + 64 * j;
          const f32x4 gg = g ? *reinterpret_cast<const f32x4*>(g + lz + c) : f32x4{1.f, 1.f, 1.f, 1.f};
          const f32x4 bb = be ? *reinterpret_cast<const f32x4*>(be + lz + c) : f32x4{0.f, 0.f, 0.f, 0.f};
          f32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = fmaf((xv[j][i] - mean) * rstd, gg[i], bb[i]);
          store4_16<NT>(xn + tc * (int64_t)d + c, o);
        }
      }
    }
  }
}

// The same pass with ONE WAVE PER ROW (d >= 768, where smoe_layernorm runs that layout: smoe_common.h smoe_ln_wave_layout) -- lane l
// holds the elements [8 l + 512 i, +8) of its row; the LayerNorm is wave_row_stats, operation for operation smoe_layernorm's.
template <typename TT, typename NT, int NI>
__global__ __launch_bounds__(256) void embed_ln_wave_kernel(const TT* __restrict__ tok, const float* __restrict__ cls,
                                                            const float* __restrict__ pos, const float* __restrict__ g,
                                                            const float* __restrict__ be, float eps, int64_t B, int P, int d,
                                                            float* __restrict__ x32, NT* __restrict__ xn) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t T = B * (int64_t)(P + 1);
  for (int64_t t = wave_gid; t < T; t += nwaves) {
    const int64_t b = t / (P + 1);
    const int n = (int)(t - b * (P + 1));
    float v[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 8 + 512 * i;
      if (c < d) {
        float a[8], pe[8];
        if (n == 0) load8(cls + c, a);
        else load8(tok + (b * P + (n - 1)) * (int64_t)d + c, a);
        load8(pos + (int64_t)n * d + c, pe);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[i][q] = a[q] + pe[q];
        store8(x32 + t * (int64_t)d + c, v[i]);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[i][q] = 0.f;
      }
    }
    if (xn) {
      float mean, rstd;
      wave_row_stats<NI>(v, d, lane, eps, mean, rstd);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = lane * 8 + 512 * i;
        if (c < d) {
          float gg[8], bb[8], o[8];
          if (g) load8(g + c, gg);
          if (be) load8(be + c, bb);
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = wave_row_affine(v[i][q], mean, rstd, g ? gg[q] : 1.f, be ? bb[q] : 0.f);
          store8(xn + t * (int64_t)d + c, o);
        }
      }
    }
  }
}

template <int NI>
__global__ __launch_bounds__(256) void layernorm_rows_wave_kernel(const float* __restrict__ x, int64_t row_stride,
                                                                  const float* __restrict__ g, const float* __restrict__ be, float eps,
                                                                  int64_t T, int d, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave_gid; t < T; t += nwaves) {
    float v[NI][8];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane * 8 + 512 * i;
      if (c < d) load8(x + t * row_stride + c, v[i]);
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
        if (be) load8(be + c, bb);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = wave_row_affine(v[i][q], mean, rstd, g ? gg[q] : 1.f, be ? bb[q] : 0.f);
        store8(out + t * (int64_t)d + c, o);
      }
    }
  }
}

// LayerNorm of T rows that start row_stride elements apart (f32 in, f32 out, contiguous [T, d] out)
template <int NJ>
__global__ __launch_bounds__(R16_THREADS, 2) void layernorm_rows_kernel(const float* __restrict__ x, int64_t row_stride,
                                                                        const float* __restrict__ g, const float* __restrict__ be,
                                                                        float eps, int64_t T, float* __restrict__ out) {
  constexpr int d = 64 * NJ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, u = lane & 15;
  const int64_t stride = (int64_t)gridDim.x * (R16_THREADS / 64) * 4;
  for (int64_t it0 = ((int64_t)blockIdx.x * (R16_THREADS / 64) + wave) * 4; it0 < T; it0 += stride) {
    const int64_t t = it0 + q;
    const bool live = t < T;
    const int64_t tc = live ? t : T - 1;
    f32x4 xv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) xv[j] = *reinterpret_cast<const f32x4*>(x + tc * row_stride + u * 4 + 64 * j);
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s1 += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
    const float mean = row16_sum(s1) / (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float dv = xv[j][i] - mean; s2 = fmaf(dv, dv, s2); }
    const float rstd = rsqrtf(row16_sum(s2) / (float)d + eps);
    if (live) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = u * 4 + 64 * j;
        const f32x4 gg = g ? *reinterpret_cast<const f32x4*>(g + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 bb = be ? *reinterpret_cast<const f32x4*>(be + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaf((xv[j][i] - mean) * rstd, gg[i], bb[i]);
        *reinterpret_cast<f32x4*>(out + tc * (int64_t)d + c) = o;
      }
    }
  }
}

int rows_grid16(int64_t T) {
  const int64_t need = (T + 15) / 16;
  return (int)(need < 1 ? 1 : (need > (1 << 20) ? (1 << 20) : need));
}

// stochastic depth's per-sample factor, expanded to the rows of the sample (timm DropPath: x / keep * mask, models/vision_transformer.py:308):
// factor[b] = mask[b] / keep (an IEEE division, as torch's div_), rows[b * N + n] = factor[b].  One launch for both.
__global__ __launch_bounds__(256) void depth_scale_rows_kernel(const float* __restrict__ mask, float keep, int64_t B, int N,
                                                               float* __restrict__ factor, float* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * N) return;
  const int64_t b = i / N;
  const float f = __fdiv_rn(mask[b], keep);
  rows[i] = f;
  if (i - b * N == 0) factor[b] = f;
}

}  // namespace

// mask f32 [B] (0 / 1 draws) -> factor f32 [B] = mask / keep and rows f32 [B * N] = factor of the row's sample
extern "C" int smoe_depth_scale_rows(const float* mask, float keep, int64_t B, int N, float* factor, float* rows, void* stream) {
  SMOE_REQUIRE(B >= 0 && N >= 1 && B * (int64_t)N < (1ll << 40) && keep > 0.f, "smoe_depth_scale_rows: bad sizes B=%lld N=%d keep=%g", (long long)B, N, (double)keep);
  if (B == 0) return 0;
  SMOE_REQUIRE(mask && factor && rows, "smoe_depth_scale_rows: null pointer");
  const int64_t blocks = (B * N + 255) / 256;
  hipLaunchKernelGGL(depth_scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, keep, B, N, factor, rows);
  SMOE_CHECK_LAUNCH("smoe_depth_scale_rows");
  return 0;
}

extern "C" int smoe_patchify_cast(const float* images, int64_t B, int C, int H, int W, int ph, int pw, void* out, int out_dtype,
                                  void* stream) {
  SMOE_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && H % ph == 0 && W % pw == 0 && pw % 4 == 0,
               "smoe_patchify_cast: bad sizes B=%lld C=%d H=%d W=%d patch %dx%d (patch width must be a multiple of 4)", (long long)B, C, H,
               W, ph, pw);
  if (B == 0) return 0;
  SMOE_REQUIRE(images && out, "smoe_patchify_cast: null pointer");
  SMOE_REQUIRE(out_dtype == SMOE_F16 || out_dtype == SMOE_BF16, "smoe_patchify_cast: out must be f16 or bf16");
  const int64_t n4 = B * C * (int64_t)H * W / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == SMOE_F16)
    hipLaunchKernelGGL((patchify_cast_kernel<f16>), dim3((int)blocks), dim3(256), 0, s, images, (f16*)out, n4, C, H, W, ph, pw, H / ph, W / pw);
  else
    hipLaunchKernelGGL((patchify_cast_kernel<bf16_bits>), dim3((int)blocks), dim3(256), 0, s, images, (bf16_bits*)out, n4, C, H, W, ph, pw, H / ph, W / pw);
  SMOE_CHECK_LAUNCH("smoe_patchify_cast");
  return 0;
}

extern "C" int smoe_embed_ln(const void* tokens, int tok_dtype, const float* cls_token, const float* pos_embed, const float* ln_gamma,
                             const float* ln_beta, float ln_eps, int64_t B, int P, int d, float* x32, void* xn, int xn_dtype,
                             void* stream) {
  SMOE_REQUIRE(B >= 0 && P >= 1 && (d == 192 || d == 384 || d == 768 || d == 1024), "smoe_embed_ln: unsupported shape B=%lld P=%d d=%d",
               (long long)B, P, d);
  if (B == 0) return 0;
  SMOE_REQUIRE(tokens && cls_token && pos_embed && x32, "smoe_embed_ln: null pointer");
  SMOE_REQUIRE(tok_dtype == SMOE_F16 || tok_dtype == SMOE_BF16, "smoe_embed_ln: tokens must be f16 or bf16");
  SMOE_REQUIRE(!xn || xn_dtype == SMOE_F16 || xn_dtype == SMOE_BF16, "smoe_embed_ln: xn must be f16 or bf16");
  hipStream_t s = (hipStream_t)stream;
  if (smoe_ln_wave_layout(d) && d > 512 && d <= 1024 && d % 8 == 0) {   // the layout smoe_layernorm uses at this width: the same bits
    const int64_t rows = B * (int64_t)(P + 1), blocks = (rows + 3) / 4;
    const int wg = (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
    const bool xb = xn && xn_dtype == SMOE_BF16;
#define ELW(TT, NT) hipLaunchKernelGGL((embed_ln_wave_kernel<TT, NT, 2>), dim3(wg), dim3(256), 0, s, (const TT*)tokens, cls_token, pos_embed, ln_gamma, ln_beta, ln_eps, B, P, d, x32, (NT*)xn)
    if (tok_dtype == SMOE_F16) { if (xb) ELW(f16, bf16_bits); else ELW(f16, f16); }
    else { if (xb) ELW(bf16_bits, bf16_bits); else ELW(bf16_bits, f16); }
#undef ELW
    SMOE_CHECK_LAUNCH("smoe_embed_ln/wave");
    return 0;
  }
  const int grid = rows_grid16(B * (int64_t)(P + 1));
#define EL(TT, NT, NJ) hipLaunchKernelGGL((embed_ln_kernel<TT, NT, NJ>), dim3(grid), dim3(R16_THREADS), 0, s, (const TT*)tokens, cls_token, pos_embed, ln_gamma, ln_beta, ln_eps, B, P, x32, (NT*)xn)
#define EL_D(TT, NT)                                                  \
  switch (d) {                                                        \
    case 192: EL(TT, NT, 3); break;                                   \
    case 384: EL(TT, NT, 6); break;                                   \
    case 768: EL(TT, NT, 12); break;                                  \
    default: EL(TT, NT, 16); break;                                   \
  }
  const bool xbf = xn && xn_dtype == SMOE_BF16;
  if (tok_dtype == SMOE_F16) { if (xbf) { EL_D(f16, bf16_bits) } else { EL_D(f16, f16) } }
  else { if (xbf) { EL_D(bf16_bits, bf16_bits) } else { EL_D(bf16_bits, f16) } }
#undef EL_D
#undef EL
  SMOE_CHECK_LAUNCH("smoe_embed_ln");
  return 0;
}

extern "C" int smoe_layernorm_rows(const float* x, int64_t row_stride, const float* gamma, const float* beta, float eps, int64_t T,
                                   int d, float* out, void* stream) {
  SMOE_REQUIRE(T >= 0 && (d == 192 || d == 384 || d == 768 || d == 1024) && row_stride >= d && row_stride % 4 == 0,
               "smoe_layernorm_rows: unsupported shape T=%lld d=%d row_stride=%lld", (long long)T, d, (long long)row_stride);
  if (T == 0) return 0;
  SMOE_REQUIRE(x && out, "smoe_layernorm_rows: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (smoe_ln_wave_layout(d) && d > 512 && row_stride % 8 == 0) {
    const int64_t blocks = (T + 3) / 4;
    const int wg = (int)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
    hipLaunchKernelGGL((layernorm_rows_wave_kernel<2>), dim3(wg), dim3(256), 0, s, x, row_stride, gamma, beta, eps, T, d, out);
    SMOE_CHECK_LAUNCH("smoe_layernorm_rows/wave");
    return 0;
  }
  const int grid = rows_grid16(T);
#define LR(NJ) hipLaunchKernelGGL((layernorm_rows_kernel<NJ>), dim3(grid), dim3(R16_THREADS), 0, s, x, row_stride, gamma, beta, eps, T, out)
  switch (d) {
    case 192: LR(3); break;
    case 384: LR(6); break;
    case 768: LR(12); break;
    default: LR(16); break;
  }
#undef LR
  SMOE_CHECK_LAUNCH("smoe_layernorm_rows");
  return 0;
}
