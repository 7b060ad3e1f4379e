// Optimizer side of the training step (engine.py:68-74 `loss_scaler(loss, optimizer, clip_grad=...,
// parameters=...)` = timm NativeScaler around torch.optim.AdamW, main.py:729-732, --opt adamw): the expert tensors are
// [E,h,d] / [E,d,h] f32 (2 x 75 MB per layer at ViT-B, E = 8) and every pass over them and their gradients is HBM-bound, so
// the step is TWO passes instead of the stock five (unscale, norm, clip-multiply, the optimizer's read and write):
//   smoe_grad_sumsq   : one read of a gradient -> sum of squares of (g * inv_scale) per 16K-element block (deterministic
//                       two-level sum; the caller adds the block partials) and the non-finite flag of GradScaler.unscale_
//   smoe_adamw_step   : decoupled-weight-decay Adam (torch.optim.AdamW arithmetic) reading the STILL-SCALED gradient times
//                       a device-side multiplier (inv_scale x clip coefficient); skipped entirely when found_inf is set;
//                       the step count lives on the device (no host sync anywhere in the step)
//   smoe_amp_update   : GradScaler.update() (growth / backoff of the loss scale) + the optimizer's step counter
#include "smoe_common.h"
#include <type_traits>

namespace {

constexpr int OPT_THREADS = 256;
constexpr int SUMSQ_BLOCK = 16384;  // elements per workgroup of the norm pass

// one 16K-element block of one gradient: sum of squares of (g * mul) -> partial[slot]; non-finite -> *found_inf = 1
template <typename GT>
__device__ __forceinline__ void sumsq_block(const GT* __restrict__ g, int64_t n, int64_t block, float mul, float* __restrict__ partial_slot,
                                            float* __restrict__ found_inf) {
  __shared__ float red[OPT_THREADS / 64];
  const int64_t base = block * SUMSQ_BLOCK;
  float acc = 0.f;
  bool bad = false;
  for (int i = threadIdx.x * 8; i < SUMSQ_BLOCK; i += OPT_THREADS * 8) {
    const int64_t at = base + i;
    if (at + 8 <= n) {
      float v[8];
      load8(g + at, v);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float u = v[q] * mul;
        bad |= !(fabsf(u) <= 3.4028234664e38f);   // inf or nan
        acc = fmaf(u, u, acc);
      }
    } else {
      for (int64_t j = at; j < n && j < at + 8; ++j) {
        float u;
        if constexpr (std::is_same<GT, float>::value) u = g[j] * mul;
        else if constexpr (std::is_same<GT, f16>::value) u = (float)g[j] * mul;
        else u = bf16_to_f32(g[j]) * mul;
        bad |= !(fabsf(u) <= 3.4028234664e38f);
        acc = fmaf(u, u, acc);
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  if (__ballot(bad) && (threadIdx.x & 63) == 0 && found_inf) *found_inf = 1.0f;
  __syncthreads();
  if (threadIdx.x == 0) *partial_slot = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename GT>
__global__ __launch_bounds__(OPT_THREADS) void grad_sumsq_kernel(const GT* __restrict__ g, int64_t n,
                                                                 const float* __restrict__ inv_scale,
                                                                 float* __restrict__ partial, float* __restrict__ found_inf) {
  sumsq_block<GT>(g, n, blockIdx.x, inv_scale ? *inv_scale : 1.0f, partial + blockIdx.x, found_inf);
}

// MULTI-TENSOR forms: one launch for every gradient / parameter of the step (a ViT-B/16 MoE has ~180 parameter tensors; one
// launch each left the optimizer step bound by launch overhead).  tab = int64 [5][n_t]: rows p, g, m, v (addresses) and n;
// blk = int32 [2][n_blocks]: tensor of workgroup b, 16K-element block inside it.  Same blocks, same arithmetic, same order of
// the partial sums as the single-tensor forms.
template <typename GT>
__global__ __launch_bounds__(OPT_THREADS) void grad_sumsq_multi_kernel(const int64_t* __restrict__ tab, int n_t,
                                                                       const int32_t* __restrict__ blk, int64_t n_blocks,
                                                                       const float* __restrict__ inv_scale,
                                                                       float* __restrict__ partial, float* __restrict__ found_inf) {
  const int t = blk[blockIdx.x];
  if (t < 0 || t >= n_t) return;   // the table is caller data
  sumsq_block<GT>(reinterpret_cast<const GT*>(tab[(int64_t)n_t + t]), tab[4ll * n_t + t], blk[n_blocks + blockIdx.x],
                  inv_scale ? *inv_scale : 1.0f, partial + blockIdx.x, found_inf);
}

// elements [begin, end) of one parameter, `stride` apart per pass (begin = this thread's first element, a multiple of 4)
// sh (optional): a 16-bit image of the parameter (the MFMA operand copy the forward reads), refreshed in the same pass -- the step
// then needs no separate f32 -> 16-bit cast of every weight (4 bytes read + 2 written per element) before the next forward
template <typename GT>
__device__ __forceinline__ void adamw_range(float* __restrict__ p, const GT* __restrict__ g, float* __restrict__ m,
                                            float* __restrict__ v, int64_t begin, int64_t end, int64_t stride, int64_t n, float lr,
                                            float b1, float b2, float eps, float wd, float t, float gm, void* sh = nullptr,
                                            int sh_dtype = SMOE_F16) {
  const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  const float step_size = lr / bc1, decay = 1.0f - lr * wd;
  for (int64_t i = begin; i < end; i += stride) {
    if (i + 4 <= n) {
      float gv[4];
      load4(g + i, gv);
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float gq = gv[q] * gm;
        pv[q] *= decay;
        mv[q] = mv[q] + (1.0f - b1) * (gq - mv[q]);          // exp_avg.lerp_(grad, 1 - beta1)
        vv[q] = fmaf(vv[q], b2, (1.0f - b2) * gq * gq);      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vv[q]) / bc2_sqrt + eps;
        pv[q] -= step_size * (mv[q] / denom);
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(m + i) = mv;
      *reinterpret_cast<f32x4*>(v + i) = vv;
      if (sh) {
        if (sh_dtype == SMOE_F16) {
          f16x4 h; h[0] = (f16)pv[0]; h[1] = (f16)pv[1]; h[2] = (f16)pv[2]; h[3] = (f16)pv[3];
          *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(sh) + i) = h;
        } else {
          s16x4 h; h[0] = (short)f32_to_bf16(pv[0]); h[1] = (short)f32_to_bf16(pv[1]); h[2] = (short)f32_to_bf16(pv[2]); h[3] = (short)f32_to_bf16(pv[3]);
          *reinterpret_cast<s16x4*>(reinterpret_cast<bf16_bits*>(sh) + i) = h;
        }
      }
    } else {
      for (int64_t j = i; j < n; ++j) {
        float gq;
        if constexpr (std::is_same<GT, float>::value) gq = g[j];
        else if constexpr (std::is_same<GT, f16>::value) gq = (float)g[j];
        else gq = bf16_to_f32(g[j]);
        gq *= gm;
        float pj = p[j] * decay, mj = m[j] + (1.0f - b1) * (gq - m[j]), vj = fmaf(v[j], b2, (1.0f - b2) * gq * gq);
        pj -= step_size * (mj / (sqrtf(vj) / bc2_sqrt + eps));
        p[j] = pj; m[j] = mj; v[j] = vj;
        if (sh) {
          if (sh_dtype == SMOE_F16) reinterpret_cast<f16*>(sh)[j] = (f16)pj;
          else reinterpret_cast<bf16_bits*>(sh)[j] = f32_to_bf16(pj);
        }
      }
    }
  }
}

template <typename GT>
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(float* __restrict__ p, const GT* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                            float b1, float b2, float eps, float wd,
                                                            const float* __restrict__ step, const float* __restrict__ grad_mult,
                                                            const float* __restrict__ found_inf) {
  if (found_inf && *found_inf != 0.f) return;   // GradScaler.step: a non-finite gradient skips the whole update
  // *step: already advanced for this update
  adamw_range<GT>(p, g, m, v, ((int64_t)blockIdx.x * OPT_THREADS + threadIdx.x) * 4, n, (int64_t)gridDim.x * OPT_THREADS * 4, n,
                  lr, b1, b2, eps, wd, *step, grad_mult ? *grad_mult : 1.0f);
}

// hyp = f32 [2][n_t]: lr, weight decay per tensor (parameter groups differ in both)
template <typename GT>
__global__ __launch_bounds__(OPT_THREADS) void adamw_multi_kernel(const int64_t* __restrict__ tab, const float* __restrict__ hyp,
                                                                  int n_t, const int32_t* __restrict__ blk, int64_t n_blocks,
                                                                  float b1, float b2, float eps, const float* __restrict__ step,
                                                                  const float* __restrict__ grad_mult,
                                                                  const float* __restrict__ found_inf,
                                                                  const int64_t* __restrict__ shadow) {
  if (found_inf && *found_inf != 0.f) return;
  const int t = blk[blockIdx.x];
  if (t < 0 || t >= n_t) return;
  const int64_t n = tab[4ll * n_t + t];
  const int64_t base = (int64_t)blk[n_blocks + blockIdx.x] * SUMSQ_BLOCK;
  const int64_t end = base + SUMSQ_BLOCK < n ? base + SUMSQ_BLOCK : n;
  adamw_range<GT>(reinterpret_cast<float*>(tab[t]), reinterpret_cast<const GT*>(tab[(int64_t)n_t + t]),
                  reinterpret_cast<float*>(tab[2ll * n_t + t]), reinterpret_cast<float*>(tab[3ll * n_t + t]),
                  base + threadIdx.x * 4, end, OPT_THREADS * 4, n, hyp[t], b1, b2, eps, hyp[n_t + t], *step,
                  grad_mult ? *grad_mult : 1.0f, shadow ? reinterpret_cast<void*>(shadow[t]) : nullptr,
                  shadow ? (int)shadow[(int64_t)n_t + t] : SMOE_F16);
}

__global__ void amp_update_kernel(float* scale, float* growth_tracker, const float* found_inf, float growth, float backoff,
                                  float interval) {
  if (found_inf && *found_inf != 0.f) {
    *scale *= backoff;
    *growth_tracker = 0.f;
  } else {
    const float tr = *growth_tracker + 1.f;
    if (tr >= interval) {
      *scale *= growth;
      *growth_tracker = 0.f;
    } else {
      *growth_tracker = tr;
    }
  }
}

__global__ void step_advance_kernel(float* step, const float* found_inf) {
  if (!(found_inf && *found_inf != 0.f)) *step += 1.f;
}

}  // namespace

extern "C" int64_t smoe_grad_sumsq_blocks(int64_t n) { return n > 0 ? (n + SUMSQ_BLOCK - 1) / SUMSQ_BLOCK : 0; }

extern "C" int smoe_grad_sumsq(const void* g, int g_dtype, int64_t n, const float* inv_scale, float* partial, float* found_inf,
                               void* stream) {
  SMOE_REQUIRE(n >= 0 && smoe_dtype_ok(g_dtype), "smoe_grad_sumsq: bad arguments");
  if (n == 0) return 0;
  SMOE_REQUIRE(g && partial, "smoe_grad_sumsq: null pointer");
  const int grid = (int)smoe_grad_sumsq_blocks(n);
  hipStream_t s = (hipStream_t)stream;
  switch (g_dtype) {
    case SMOE_F32: hipLaunchKernelGGL(grad_sumsq_kernel<float>, dim3(grid), dim3(OPT_THREADS), 0, s, (const float*)g, n, inv_scale, partial, found_inf); break;
    case SMOE_F16: hipLaunchKernelGGL(grad_sumsq_kernel<f16>, dim3(grid), dim3(OPT_THREADS), 0, s, (const f16*)g, n, inv_scale, partial, found_inf); break;
    default: hipLaunchKernelGGL(grad_sumsq_kernel<bf16_bits>, dim3(grid), dim3(OPT_THREADS), 0, s, (const bf16_bits*)g, n, inv_scale, partial, found_inf); break;
  }
  SMOE_CHECK_LAUNCH("smoe_grad_sumsq");
  return 0;
}

extern "C" int smoe_adamw_step(float* p, const void* g, int g_dtype, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, const float* step, const float* grad_mult,
                               const float* found_inf, void* stream) {
  SMOE_REQUIRE(n >= 0 && smoe_dtype_ok(g_dtype), "smoe_adamw_step: bad arguments");
  if (n == 0) return 0;
  SMOE_REQUIRE(p && g && m && v && step, "smoe_adamw_step: null pointer");
  int64_t blocks = (n / 4 + OPT_THREADS - 1) / OPT_THREADS;
  if (blocks < 1) blocks = 1;
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
  switch (g_dtype) {
    case SMOE_F32: hipLaunchKernelGGL(adamw_kernel<float>, dim3((int)blocks), dim3(OPT_THREADS), 0, s, p, (const float*)g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_mult, found_inf); break;
    case SMOE_F16: hipLaunchKernelGGL(adamw_kernel<f16>, dim3((int)blocks), dim3(OPT_THREADS), 0, s, p, (const f16*)g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_mult, found_inf); break;
    default: hipLaunchKernelGGL(adamw_kernel<bf16_bits>, dim3((int)blocks), dim3(OPT_THREADS), 0, s, p, (const bf16_bits*)g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_mult, found_inf); break;
  }
  SMOE_CHECK_LAUNCH("smoe_adamw_step");
  return 0;
}

extern "C" int smoe_amp_update(float* scale, float* growth_tracker, const float* found_inf, float growth_factor,
                               float backoff_factor, int growth_interval, void* stream) {
  SMOE_REQUIRE(scale && growth_tracker, "smoe_amp_update: null pointer");
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scale, growth_tracker, found_inf,
                     growth_factor, backoff_factor, (float)growth_interval);
  SMOE_CHECK_LAUNCH("smoe_amp_update");
  return 0;
}

extern "C" int smoe_step_advance(float* step, const float* found_inf, void* stream) {
  SMOE_REQUIRE(step, "smoe_step_advance: null pointer");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, found_inf);
  SMOE_CHECK_LAUNCH("smoe_step_advance");
  return 0;
}

extern "C" int smoe_grad_sumsq_multi(const int64_t* tab, int n_tensors, const int32_t* blk, int64_t n_blocks, int g_dtype,
                                     const float* inv_scale, float* partial, float* found_inf, void* stream) {
  SMOE_REQUIRE(n_tensors >= 0 && n_blocks >= 0 && n_blocks < (1ll << 31) && smoe_dtype_ok(g_dtype), "smoe_grad_sumsq_multi: bad arguments");
  if (n_blocks == 0) return 0;
  SMOE_REQUIRE(tab && blk && partial, "smoe_grad_sumsq_multi: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)n_blocks), block(OPT_THREADS);
  switch (g_dtype) {
    case SMOE_F32: hipLaunchKernelGGL(grad_sumsq_multi_kernel<float>, grid, block, 0, s, tab, n_tensors, blk, n_blocks, inv_scale, partial, found_inf); break;
    case SMOE_F16: hipLaunchKernelGGL(grad_sumsq_multi_kernel<f16>, grid, block, 0, s, tab, n_tensors, blk, n_blocks, inv_scale, partial, found_inf); break;
    default: hipLaunchKernelGGL(grad_sumsq_multi_kernel<bf16_bits>, grid, block, 0, s, tab, n_tensors, blk, n_blocks, inv_scale, partial, found_inf); break;
  }
  SMOE_CHECK_LAUNCH("smoe_grad_sumsq_multi");
  return 0;
}

extern "C" int smoe_adamw_step_multi(const int64_t* tab, const float* hyp, int n_tensors, const int32_t* blk, int64_t n_blocks,
                                     int g_dtype, float beta1, float beta2, float eps, const float* step, const float* grad_mult,
                                     const float* found_inf, const int64_t* shadow, void* stream) {
  SMOE_REQUIRE(n_tensors >= 0 && n_blocks >= 0 && n_blocks < (1ll << 31) && smoe_dtype_ok(g_dtype), "smoe_adamw_step_multi: bad arguments");
  if (n_blocks == 0) return 0;
  SMOE_REQUIRE(tab && hyp && blk && step, "smoe_adamw_step_multi: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)n_blocks), block(OPT_THREADS);
  switch (g_dtype) {
    case SMOE_F32: hipLaunchKernelGGL(adamw_multi_kernel<float>, grid, block, 0, s, tab, hyp, n_tensors, blk, n_blocks, beta1, beta2, eps, step, grad_mult, found_inf, shadow); break;
    case SMOE_F16: hipLaunchKernelGGL(adamw_multi_kernel<f16>, grid, block, 0, s, tab, hyp, n_tensors, blk, n_blocks, beta1, beta2, eps, step, grad_mult, found_inf, shadow); break;
    default: hipLaunchKernelGGL(adamw_multi_kernel<bf16_bits>, grid, block, 0, s, tab, hyp, n_tensors, blk, n_blocks, beta1, beta2, eps, step, grad_mult, found_inf, shadow); break;
  }
  SMOE_CHECK_LAUNCH("smoe_adamw_step_multi");
  return 0;
}
