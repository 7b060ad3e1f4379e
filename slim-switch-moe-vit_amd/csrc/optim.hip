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

template <typename GT>
__global__ __launch_bounds__(OPT_THREADS) void grad_sumsq_kernel(const GT* __restrict__ g, int64_t n,
                                                                 const float* __restrict__ inv_scale,
                                                                 float* __restrict__ partial, float* __restrict__ found_inf) {
  __shared__ float red[OPT_THREADS / 64];
  const float mul = inv_scale ? *inv_scale : 1.0f;
  const int64_t base = (int64_t)blockIdx.x * SUMSQ_BLOCK;
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
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename GT>
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(float* __restrict__ p, const GT* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                            float b1, float b2, float eps, float wd,
                                                            const float* __restrict__ step, const float* __restrict__ grad_mult,
                                                            const float* __restrict__ found_inf) {
  if (found_inf && *found_inf != 0.f) return;   // GradScaler.step: a non-finite gradient skips the whole update
  const float t = *step;                        // already advanced for this update
  const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  const float step_size = lr / bc1, decay = 1.0f - lr * wd, gm = grad_mult ? *grad_mult : 1.0f;
  const int64_t stride = (int64_t)gridDim.x * OPT_THREADS * 4;
  for (int64_t i = ((int64_t)blockIdx.x * OPT_THREADS + threadIdx.x) * 4; i < n; i += stride) {
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
      }
    }
  }
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
