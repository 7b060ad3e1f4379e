// Self-attention forward for ViT token counts (N <= 256, head dim 64): softmax(q k^T * scale) v, models/
// vision_transformer.py:248-280 -- the caller-side kernel next to the MoE hot path.
//
// One workgroup (4 waves) per (image, head): K and V of that head (N x 64, 16-bit) sit in LDS, every wave walks
// 16-query tiles.  Per tile, "key on the lane":
//   S^T = K Q^T      MFMA 16x16x32 with the K fragment as the A operand and the Q fragment as B: lane (g, q) ends up
//                    with the scores of query q against keys 16 kt + 4 g + r  (kt = key tile, r = 0..3)
//   softmax          per-lane over its 4*NT scores, then across the 4 lanes that share a query (xor 16, xor 32)
//   O^T = V^T P^T    the S^T accumulators, rounded to 16 bit, ARE the B operand (k-slot (g, j) <-> key
//                    32 ks + 16 (j>>2) + 4 g + (j&3)); the matching A operand is read from row-major V with
//                    ds_read_b64_tr_b16 (hardware transpose read), two reads per fragment
// so P never touches LDS and V is never transposed in memory.  Output rows leave as 8-byte pieces
// (4 consecutive head-dim elements per lane).
#include "smoe_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int ATT_D = 64;
constexpr int ATT_THREADS = 256;

__device__ __forceinline__ int k_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// V: 32-byte segment index (= 16-column block dt) XOR-swizzled by (row>>1)&3 -> the transposed reads of 8
// consecutive rows hit 8 different 32-byte slots of the 256-byte bank row
__device__ __forceinline__ int v_lds_off(int row, int dt) { return row * 128 + ((dt ^ ((row >> 1) & 3)) << 5); }

template <typename HT, int NT>  // NT = key tiles of 16 (even, padded)
__global__ __launch_bounds__(ATT_THREADS, 3) void attn_fwd_kernel(const HT* __restrict__ qkv, HT* __restrict__ out, int N,
                                                                  int H, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nkt = (N + 15) >> 4;  // real key tiles (<= NT); LDS holds exactly nkt * 16 rows of K and of V
  char* Ks = smem;
  char* Vs = smem + nkt * 16 * 128;
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tok_stride = (int64_t)3 * H * ATT_D;  // elements between consecutive tokens
  const HT* base = qkv + (int64_t)b * N * tok_stride + h * ATT_D;

  // ---- stage K and V of this head: rows >= N are zero ------------------------------------------------
  for (int i = tid; i < nkt * 16 * 8; i += ATT_THREADS) {
    const int row = i >> 3, c = i & 7;
    u32x4 kv = u32x4{0u, 0u, 0u, 0u}, vv = u32x4{0u, 0u, 0u, 0u};
    if (row < N) {
      const HT* p = base + (int64_t)row * tok_stride + c * 8;
      kv = *reinterpret_cast<const u32x4*>(p + H * ATT_D);
      vv = *reinterpret_cast<const u32x4*>(p + 2 * H * ATT_D);
    }
    *reinterpret_cast<u32x4*>(Ks + k_lds_off(row, c)) = kv;
    *reinterpret_cast<u32x4*>(Vs + v_lds_off(row, c >> 1) + (c & 1) * 16) = vv;
  }
  __syncthreads();

  const int g = lane >> 4, qi = lane & 15;
  const int nqt = (N + 15) >> 4;
  for (int qt = wave; qt < nqt; qt += ATT_THREADS / 64) {
    const int q0 = qt * 16;
    int qrow = q0 + qi;
    if (qrow >= N) qrow = N - 1;
    const HT* qp = base + (int64_t)qrow * tok_stride + g * 8;
    const u32x4 qf0 = *reinterpret_cast<const u32x4*>(qp);
    const u32x4 qf1 = *reinterpret_cast<const u32x4*>(qp + 32);

    // ---- S^T tiles ---------------------------------------------------------------------------------------
    f32x4 sacc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      sacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
        const u32x4 kf0 = *reinterpret_cast<const u32x4*>(Ks + k_lds_off(kt * 16 + qi, g));
        const u32x4 kf1 = *reinterpret_cast<const u32x4*>(Ks + k_lds_off(kt * 16 + qi, 4 + g));
        if constexpr (std::is_same<HT, f16>::value) {
          sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf0), __builtin_bit_cast(f16x8, qf0), sacc[kt], 0, 0, 0);
          sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf1), __builtin_bit_cast(f16x8, qf1), sacc[kt], 0, 0, 0);
        } else {
          sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf0), __builtin_bit_cast(bf16x8_t, qf0), sacc[kt], 0, 0, 0);
          sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf1), __builtin_bit_cast(bf16x8_t, qf1), sacc[kt], 0, 0, 0);
        }
      }
    }
    // ---- softmax over keys (scores of one query live in the 4 lanes g = 0..3 with equal qi) -------------------
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + g * 4 + r;
        const float s = (key < N) ? sacc[kt][r] * scale_log2e : -INFINITY;
        sacc[kt][r] = s;
        m = fmaxf(m, s);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(sacc[kt][r] - m);
        sacc[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv_l = 1.0f / l;

    // ---- O^T = V^T P^T --------------------------------------------------------------------------------------
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = qi >> 2, tp = qi & 3;  // transposed-read address roles inside the 16-lane group
#pragma unroll
    for (int ks = 0; ks < NT / 2; ++ks) {
      u32x4 pf;
      if constexpr (std::is_same<HT, f16>::value) {
        f16x8 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = (f16)sacc[2 * ks][r]; t[4 + r] = (f16)sacc[2 * ks + 1][r]; }
        pf = __builtin_bit_cast(u32x4, t);
      } else {
        s16x8 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = (short)f32_to_bf16(sacc[2 * ks][r]); t[4 + r] = (short)f32_to_bf16(sacc[2 * ks + 1][r]); }
        pf = __builtin_bit_cast(u32x4, t);
      }
      const int row1 = 32 * ks + 4 * g + tq;  // row this lane addresses in block 1; block 2 is 16 rows further
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        s16x4 v0 = s16x4{0, 0, 0, 0};
        if (2 * ks < nkt)  // wave-uniform: tiles past the real key count have no rows in LDS
          v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(Vs + v_lds_off(row1, dt) + tp * 8));
        s16x4 v1 = s16x4{0, 0, 0, 0};
        if (2 * ks + 1 < nkt)  // wave-uniform: the padding tile has no rows in LDS (its P entries are zero anyway)
          v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(Vs + v_lds_off(row1 + 16, dt) + tp * 8));
        s16x8 vf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { vf[r] = v0[r]; vf[4 + r] = v1[r]; }
        if constexpr (std::is_same<HT, f16>::value)
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf), __builtin_bit_cast(f16x8, pf), oacc[dt], 0, 0, 0);
        else
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf), oacc[dt], 0, 0, 0);
      }
    }
    // ---- store: lane (g, qi) holds head-dim elements 16 dt + 4 g + r of query q0 + qi ----------------------------
    if (q0 + qi < N) {
      HT* op = out + ((int64_t)b * N + q0 + qi) * (H * ATT_D) + h * ATT_D + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if constexpr (std::is_same<HT, f16>::value) {
          f16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (f16)(oacc[dt][r] * inv_l);
          *reinterpret_cast<f16x4*>(op + dt * 16) = o;
        } else {
          s16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (short)f32_to_bf16(oacc[dt][r] * inv_l);
          *reinterpret_cast<s16x4*>(op + dt * 16) = o;
        }
      }
    }
  }
}

template <typename HT, int NT>
int launch_attn(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t s) {
  const size_t smem = 2 * (size_t)((N + 15) / 16) * 16 * 128;
  auto kern = attn_fwd_kernel<HT, NT>;
  static bool attr_done = false;
  if (!attr_done && smem > 48 * 1024) {
    hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (ae != hipSuccess) {
      smoe_set_error("smoe_attention_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(ae));
      return (int)ae;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(ATT_THREADS), smem, s, (const HT*)qkv, (HT*)out, N, H,
                     scale * 1.4426950408889634f);
  SMOE_CHECK_LAUNCH("smoe_attention_fwd");
  return 0;
}

template <typename HT>
int attn_dispatch(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t s) {
  const int nt = ((N + 15) / 16 + 1) & ~1;
  switch (nt) {
    case 2: case 4: case 6: case 8: return launch_attn<HT, 8>(qkv, out, B, N, H, scale, s);
    case 10: case 12: case 14: return launch_attn<HT, 14>(qkv, out, B, N, H, scale, s);
    case 16: return launch_attn<HT, 16>(qkv, out, B, N, H, scale, s);
  }
  smoe_set_error("smoe_attention_fwd: N=%d unsupported (N <= 256)", N);
  return 1;
}

}  // namespace

// qkv [B, N, 3, H, 64] (the output layout of the fused qkv projection), out [B, N, H*64]; f16 or bf16.
extern "C" int smoe_attention_supported(int N, int head_dim) { return (N >= 1 && N <= 256 && head_dim == ATT_D) ? 1 : 0; }

extern "C" int smoe_attention_fwd(const void* qkv, void* out, int dtype, int B, int N, int H, int head_dim, float scale,
                                  void* stream) {
  SMOE_REQUIRE(smoe_attention_supported(N, head_dim), "smoe_attention_fwd: unsupported N=%d head_dim=%d", N, head_dim);
  SMOE_REQUIRE(B >= 0 && H >= 1, "smoe_attention_fwd: bad B=%d H=%d", B, H);
  if (B == 0) return 0;
  SMOE_REQUIRE(qkv && out, "smoe_attention_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SMOE_F16) return attn_dispatch<f16>(qkv, out, B, N, H, scale, s);
  if (dtype == SMOE_BF16) return attn_dispatch<bf16_bits>(qkv, out, B, N, H, scale, s);
  smoe_set_error("smoe_attention_fwd: dtype must be f16 or bf16");
  return 1;
}
