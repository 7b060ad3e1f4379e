// Shared device/host helpers for libslimmoe_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/slimmoe.h"

#define SMOE_WAVE 64

void smoe_set_error(const char* fmt, ...);

#define SMOE_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      smoe_set_error(__VA_ARGS__);         \
      return 1;                            \
    }                                      \
  } while (0)

#define SMOE_CHECK_LAUNCH(name)                                                  \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      smoe_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return (int)e__;                                                           \
    }                                                                            \
  } while (0)

typedef _Float16 f16;
typedef unsigned short bf16_bits;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
// round-to-nearest-even; NaN stays NaN (plain cast path, see MI355X_MICROARCH 'Correctness boundaries')
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  const unsigned int u = __float_as_uint(f);
  const unsigned int quiet = (u >> 16) | 0x40u, rounded = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  return (unsigned short)(((u & 0x7fffffffu) > 0x7f800000u) ? quiet : rounded);  // a select, not a branch
}

// dtype tags
struct TagF32 { typedef float T; static constexpr int code = SMOE_F32; };
struct TagF16 { typedef f16 T; static constexpr int code = SMOE_F16; };
struct TagBF16 { typedef bf16_bits T; static constexpr int code = SMOE_BF16; };

// load 4 consecutive elements as float
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
__device__ __forceinline__ void load4(const f16* p, float (&v)[4]) {
  f16x4 t = *reinterpret_cast<const f16x4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
__device__ __forceinline__ void load4(const bf16_bits* p, float (&v)[4]) {
  s16x4 t = *reinterpret_cast<const s16x4*>(p);
  v[0] = bf16_to_f32((unsigned short)t[0]); v[1] = bf16_to_f32((unsigned short)t[1]);
  v[2] = bf16_to_f32((unsigned short)t[2]); v[3] = bf16_to_f32((unsigned short)t[3]);
}
// load / store 8 consecutive elements
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void load8(const f16* p, float (&v)[8]) {
  f16x8 t = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void load8(const bf16_bits* p, float (&v)[8]) {
  s16x8 t = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = bf16_to_f32((unsigned short)t[i]);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(f16* p, const float (&v)[8]) {
  f16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (f16)v[i];
  *reinterpret_cast<f16x8*>(p) = t;
}
__device__ __forceinline__ void store8(bf16_bits* p, const float (&v)[8]) {
  s16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (short)f32_to_bf16(v[i]);
  *reinterpret_cast<s16x8*>(p) = t;
}

// ---- per-(kernel, device) launch attributes -------------------------------------------------------------------------
// Kernels that need more dynamic LDS than the default limit register themselves at library load (a static object per
// instantiation); smoe_init() raises the limit for every registered kernel on the CURRENT device, and every launcher
// calls SMOE_ENSURE_SMEM as a thread-safe, per-device fallback (an atomic bit per device; setting the attribute twice
// is harmless).  After smoe_init() no launcher touches function attributes any more, which is what makes the launches
// graph-capturable.
#include <atomic>
struct SmoeKernelEntry {
  const void* kern;
  std::atomic<uint32_t> done;  // bit i: attribute set on device i (devices >= 32 are not cached)
  SmoeKernelEntry* next;
  explicit SmoeKernelEntry(const void* k);
};
int smoe_kernel_ensure(SmoeKernelEntry& e);  // api.hip
int smoe_num_cus();                          // CU count of the current device (cached per device)
int smoe_reserved_cus();                     // CUs the persistent GEMM leaves to other streams' kernels (smoe_set_reserved_cus)
// Zero `words` 32-bit words at p (4-byte aligned) with a kernel on stream s.  Used for the few device-side counters instead of
// hipMemsetAsync: a plain kernel node orders like every other launch when the caller captures the stream into a graph.
hipError_t smoe_zero_words(void* p, int64_t words, hipStream_t s);  // api.hip
template <auto KERN> struct SmoeKernelReg { static SmoeKernelEntry entry; };
template <auto KERN> SmoeKernelEntry SmoeKernelReg<KERN>::entry{reinterpret_cast<const void*>(KERN)};
#define SMOE_ENSURE_SMEM(...)                                                \
  do {                                                                       \
    const int rc__ = smoe_kernel_ensure(SmoeKernelReg<__VA_ARGS__>::entry);  \
    if (rc__ != 0) return rc__;                                              \
  } while (0)

static inline int smoe_dtype_size(int code) { return code == SMOE_F32 ? 4 : 2; }
static inline bool smoe_dtype_ok(int code) { return code == SMOE_F32 || code == SMOE_F16 || code == SMOE_BF16; }

// Append token t to a device-side list behind an atomic counter.  The list holds `cap` entries; a slot outside it (a counter
// that was not zeroed, or corrupted) is dropped instead of becoming a wild store.
__device__ __forceinline__ void list_push(int32_t* count, int32_t* list, int64_t cap, int64_t t) {
  const int32_t slot = atomicAdd(count, 1);
  if (slot >= 0 && (int64_t)slot < cap) list[slot] = (int32_t)t;
}

// ---- LayerNorm of one row held by ONE WAVE -----------------------------------------------------------------------------------
// Lane l holds v[i][0..8) = the row's elements [8 l + 512 i, 8 l + 512 i + 8) (zeros past d).  THE arithmetic of every wave-per-row
// LayerNorm of the library -- smoe_layernorm / smoe_embed_ln / smoe_layernorm_rows at d >= 768, smoe_gather_combine_ln at every d --
// so that each of them produces the same bits for the same row (tests/test_gpu_dense.py).  Two passes in registers.
template <int NI>
__device__ __forceinline__ void wave_row_stats(const float (&v)[NI][8], int d, int lane, float eps, float& mean, float& rstd) {
  const float inv_d = 1.0f / (float)d;
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int q = 0; q < 8; ++q) s1 += v[i][q];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s1 += __shfl_xor(s1, m, 64);
  mean = s1 * inv_d;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (lane * 8 + 512 * i < d) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float dv = v[i][q] - mean; s2 = fmaf(dv, dv, s2); }
    }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s2 += __shfl_xor(s2, m, 64);
  rstd = rsqrtf(s2 * inv_d + eps);
}

// ... and the affine output of one element.  The f32 value is the result: the empty asm keeps the compiler from fusing the FMA with a
// following f32 -> f16 conversion (v_fma_mixlo / mixhi_f16 round ONCE; a kernel whose store converts in an instruction of its own
// rounds TWICE -- the two differ in ~1 of 2^13 elements, which is how smoe_gather_combine_ln and smoe_layernorm disagreed at first).
__device__ __forceinline__ float wave_row_affine(float v, float mean, float rstd, float g, float b) {
  float o = fmaf((v - mean) * rstd, g, b);
  asm volatile("" : "+v"(o));
  return o;
}

// which LayerNorm layout serves a width (one answer per width for EVERY kernel that promises smoe_layernorm's bits): a wave per row
// from d = 768 up (in the model: d 768 48.5 -> 44.0 us, d 1024 74.8 -> 39.1 us per launch), 16 lanes per token below (d 384: 38.4
// vs 39.5 us, d 192: 12.9 vs 16.4).  SMOE_LN_WAVE=0 / 1 forces one (A/B, tools/ln_ab.py).
inline bool smoe_ln_wave_layout(int d) {
  static const int forced = [] { const char* e = getenv("SMOE_LN_WAVE"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced != 0;
  return d >= 768;
}
