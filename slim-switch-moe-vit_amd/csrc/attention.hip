// Self-attention forward for ViT token counts (N <= 256 in one pass, N <= 640 with an online softmax; head dim 64): softmax(q k^T * scale) v, models/
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

// NKT = key tiles of 16 held in LDS (compile-time, >= ceil(N/16): every loop below is static, so the compiler can
// batch the LDS reads ahead of the MFMAs); NT = NKT rounded up to even (k-steps of 32 keys).  EXACT = (NKT ==
// ceil(N/16)): only the last key tile can hold keys >= N, so only its 4 scores per lane carry mask code (the general
// form costs a compare-select pair on all 4 NKT scores: 150 of the ~650 VALU issues of a query tile, and this
// kernel is VALU-issue bound -- 54 MFMAs against ~600 VALU per tile).
template <typename HT, int NKT, bool EXACT>
__global__ __launch_bounds__(ATT_THREADS, 3) void attn_fwd_kernel(const HT* __restrict__ qkv, HT* __restrict__ out, int N,
                                                                  int H, float scale_log2e, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NKT + (NKT & 1);
  constexpr int nkt = NKT;
  char* Ks = smem;
  char* Vs = smem + nkt * 16 * 128;
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tok_stride = (int64_t)3 * H * ATT_D;  // elements between consecutive tokens
  const HT* base = qkv + (int64_t)b * N * tok_stride + h * ATT_D;

  // ---- stage K and V of this head by LDS DMA (global_load_lds, 1 KiB = 8 rows per wave-instruction; the source
  //      address carries the swizzle, cdna_hip_programming.md rule 21).  Rows >= N duplicate row N-1: their scores
  //      are masked and their probabilities are exactly 0, so they never contribute. ------------------------------
  {
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int l_row = lane >> 3, l_pos = lane & 7;
    const int npieces = nkt * 2;  // 8-row pieces per operand
    for (int pc = wave_u; pc < npieces; pc += ATT_THREADS / 64) {
      const int row = pc * 8 + l_row;
      const int srow = row < N ? row : N - 1;
      const HT* p = base + (int64_t)srow * tok_stride;
      const int kc = l_pos ^ ((row >> 1) & 7);                           // K: 16-byte chunk swizzle
      const int vc = (((l_pos >> 1) ^ ((row >> 1) & 3)) << 1) | (l_pos & 1);  // V: 32-byte segment swizzle
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + H * ATT_D + kc * 8),
                                       (__attribute__((address_space(3))) void*)(Ks + pc * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 2 * H * ATT_D + vc * 8),
                                       (__attribute__((address_space(3))) void*)(Vs + pc * 1024), 16, 0, 0);
    }
  }
  const int g = lane >> 4, qi = lane & 15;
  const int nqt = (N + 15) >> 4;
  auto load_q = [&](int qt, u32x4& f0, u32x4& f1) {
    int qrow = qt * 16 + qi;
    if (qrow >= N) qrow = N - 1;
    const HT* qp = base + (int64_t)qrow * tok_stride + g * 8;
    f0 = *reinterpret_cast<const u32x4*>(qp);
    f1 = *reinterpret_cast<const u32x4*>(qp + 32);
  };
  u32x4 qn0 = u32x4{0u, 0u, 0u, 0u}, qn1 = qn0;
  if (wave < nqt) load_q(wave, qn0, qn1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int qt = wave; qt < nqt; qt += ATT_THREADS / 64) {
    const int q0 = qt * 16;
    const u32x4 qf0 = qn0, qf1 = qn1;
    if (qt + ATT_THREADS / 64 < nqt) load_q(qt + ATT_THREADS / 64, qn0, qn1);  // next tile's Q under this tile's math

    // ---- S^T tiles: fragment reads in batches of 4 key tiles (8 ds_read_b128) ahead of their 8 MFMAs; the
    //      sched_barrier keeps hipcc from either sinking each read next to its MFMA (one exposed LDS latency per
    //      MFMA) or hoisting all 26 reads (104 VGPRs, spills) --------------------------------------------------
    f32x4 sacc[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) sacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NKG = (nkt + 3) / 4;
    u32x4 kf[2][4][2];  // double-buffered: group gi+1 is read while group gi feeds the MFMAs
    auto read_k = [&](int buf, int kg) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (kg + i < nkt) {
          kf[buf][i][0] = *reinterpret_cast<const u32x4*>(Ks + k_lds_off((kg + i) * 16 + qi, g));
          kf[buf][i][1] = *reinterpret_cast<const u32x4*>(Ks + k_lds_off((kg + i) * 16 + qi, 4 + g));
        }
      }
    };
    read_k(0, 0);
#pragma unroll
    for (int gi = 0; gi < NKG; ++gi) {
      const int kg = gi * 4;
      if (gi + 1 < NKG) read_k((gi + 1) & 1, kg + 4);
      __builtin_amdgcn_sched_barrier(0);
      // first halves of the 4 tiles, then the second halves: the two MFMAs of one accumulator are 3 issues apart
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (kg + i < nkt) {
            if constexpr (std::is_same<HT, f16>::value)
              sacc[kg + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[gi & 1][i][hf]), __builtin_bit_cast(f16x8, hf ? qf1 : qf0), sacc[kg + i], 0, 0, 0);
            else
              sacc[kg + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[gi & 1][i][hf]), __builtin_bit_cast(bf16x8_t, hf ? qf1 : qf0), sacc[kg + i], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // keys >= N: -inf (their probability is exactly 0).  The odd padding tile NKT (when NT > NKT) has no scores at
    // all: it is left out of max / exp and packed as zeros below.
    if constexpr (EXACT) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if ((NKT - 1) * 16 + g * 4 + r >= N) sacc[NKT - 1][r] = -INFINITY;
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 16 + g * 4 + r >= N) sacc[kt][r] = -INFINITY;
    }
    float mm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // four independent v_max3 chains, two per tile
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      mm[(2 * kt) & 3] = fmaxf(fmaxf(mm[(2 * kt) & 3], sacc[kt][0]), sacc[kt][1]);
      mm[(2 * kt + 1) & 3] = fmaxf(fmaxf(mm[(2 * kt + 1) & 3], sacc[kt][2]), sacc[kt][3]);
    }
    float m = fmaxf(fmaxf(mm[0], mm[1]), fmaxf(mm[2], mm[3]));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float nmc = -m * scale_log2e;  // exp2(s*c - m*c): scale folded into one FMA per score
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sacc[kt][r] = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], scale_log2e, nmc));

    // ---- O^T = V^T P^T: P packed to 16 bit first (frees the f32 scores), then per k-step 8 transposed reads issued
    //      together ahead of their 4 MFMAs --------------------------------------------------------------------------
    u32x4 pf[NT / 2];
#pragma unroll
    for (int ks = 0; ks < NT / 2; ++ks) {
      if constexpr (std::is_same<HT, f16>::value) {
        f16x8 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = (f16)sacc[2 * ks][r]; t[4 + r] = (2 * ks + 1 < NKT) ? (f16)sacc[2 * ks + 1][r] : (f16)0.f; }
        pf[ks] = __builtin_bit_cast(u32x4, t);
      } else {
        s16x8 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = (short)f32_to_bf16(sacc[2 * ks][r]); t[4 + r] = (2 * ks + 1 < NKT) ? (short)f32_to_bf16(sacc[2 * ks + 1][r]) : (short)0; }
        pf[ks] = __builtin_bit_cast(u32x4, t);
      }
    }
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // softmax denominator on the matrix pipe: an all-ones "V^T" fragment sums the (16-bit rounded) probabilities of
    // every query over the keys -- 7 MFMAs instead of 56 v_add + 2 cross-lane adds, and the normaliser is the sum
    // of exactly the values that multiply V
    f32x4 lacc = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 ones = std::is_same<HT, f16>::value ? u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u}
                                                    : u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const int tq = qi >> 2, tp = qi & 3;  // transposed-read address roles inside the 16-lane group
    s16x4 v0[2][4], v1[2][4];  // double-buffered transposed V fragments
    auto read_v = [&](int buf, int ks) {
      const int row1 = 32 * ks + 4 * g + tq;  // row this lane addresses in block 1; block 2 is 16 rows further
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        v0[buf][dt] = s16x4{0, 0, 0, 0};
        v1[buf][dt] = s16x4{0, 0, 0, 0};
        if (2 * ks < nkt)  // static
          v0[buf][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(Vs + v_lds_off(row1, dt) + tp * 8));
        if (2 * ks + 1 < nkt)  // static: the padding tile has no rows in LDS (its P entries are zero anyway)
          v1[buf][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(Vs + v_lds_off(row1 + 16, dt) + tp * 8));
      }
    };
    read_v(0, 0);
#pragma unroll
    for (int ks = 0; ks < NT / 2; ++ks) {
      if (ks + 1 < NT / 2) read_v((ks + 1) & 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        s16x8 vf;
#pragma unroll
        for (int r = 0; r < 4; ++r) { vf[r] = v0[ks & 1][dt][r]; vf[4 + r] = v1[ks & 1][dt][r]; }
        if constexpr (std::is_same<HT, f16>::value)
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf), __builtin_bit_cast(f16x8, pf[ks]), oacc[dt], 0, 0, 0);
        else
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[ks]), oacc[dt], 0, 0, 0);
      }
      if constexpr (std::is_same<HT, f16>::value)
        lacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ones), __builtin_bit_cast(f16x8, pf[ks]), lacc, 0, 0, 0);
      else
        lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones), __builtin_bit_cast(bf16x8_t, pf[ks]), lacc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float inv_l = 1.0f / lacc[0];
    // training: log2 of the softmax normaliser in the scaled-score domain, p = exp2(s c - lse) (the backward recomputes p from it)
    if (lse && g == 0 && q0 + qi < N) lse[((int64_t)b * H + h) * N + q0 + qi] = fmaf(m, scale_log2e, __log2f(lacc[0]));
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

// ---- long sequences (256 < N <= 640; ViT-L/16 @384: N = 577, models/vision_transformer.py:1227-1236) ------------------
// Same layout and fragment roles as attn_fwd_kernel; K and V of the head still sit in LDS whole (N = 577: 2 x 74 KB), but
// the scores of a query tile no longer fit the registers, so the keys are walked in CHUNKS of CH tiles with the online
// softmax: running maximum m, running denominator l (kept on the matrix pipe, as above) and running output O are
// rescaled by exp2((m_old - m_new) c) at every chunk -- unconditionally (no threshold, no data-dependent branch).
// 8 waves per workgroup (one workgroup per CU: the LDS is full), each wave walks query tiles wave, wave + 8, ...
constexpr int ATTL_THREADS = 512;
template <typename HT, int NKT, int CH>
__global__ __launch_bounds__(ATTL_THREADS, 2) void attn_fwd_long_kernel(const HT* __restrict__ qkv, HT* __restrict__ out, int N,
                                                                        int H, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(CH % 2 == 0, "chunks are whole 32-key k-steps");
  constexpr int NCHUNK = (NKT + CH - 1) / CH;
  char* Ks = smem;
  char* Vs = smem + NKT * 16 * 128;
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tok_stride = (int64_t)3 * H * ATT_D;
  const HT* base = qkv + (int64_t)b * N * tok_stride + h * ATT_D;
  {
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int l_row = lane >> 3, l_pos = lane & 7;
    for (int pc = wave_u; pc < NKT * 2; pc += ATTL_THREADS / 64) {
      const int row = pc * 8 + l_row;
      const int srow = row < N ? row : N - 1;
      const HT* p = base + (int64_t)srow * tok_stride;
      const int kc = l_pos ^ ((row >> 1) & 7);
      const int vc = (((l_pos >> 1) ^ ((row >> 1) & 3)) << 1) | (l_pos & 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + H * ATT_D + kc * 8),
                                       (__attribute__((address_space(3))) void*)(Ks + pc * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 2 * H * ATT_D + vc * 8),
                                       (__attribute__((address_space(3))) void*)(Vs + pc * 1024), 16, 0, 0);
    }
  }
  const int g = lane >> 4, qi = lane & 15;
  const int nqt = (N + 15) >> 4;
  auto load_q = [&](int qt, u32x4& f0, u32x4& f1) {
    int qrow = qt * 16 + qi;
    if (qrow >= N) qrow = N - 1;
    const HT* qp = base + (int64_t)qrow * tok_stride + g * 8;
    f0 = *reinterpret_cast<const u32x4*>(qp);
    f1 = *reinterpret_cast<const u32x4*>(qp + 32);
  };
  u32x4 qn0 = u32x4{0u, 0u, 0u, 0u}, qn1 = qn0;
  if (wave < nqt) load_q(wave, qn0, qn1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const u32x4 ones = std::is_same<HT, f16>::value ? u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u}
                                                  : u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const int tq = qi >> 2, tp = qi & 3;
  for (int qt = wave; qt < nqt; qt += ATTL_THREADS / 64) {
    const int q0 = qt * 16;
    const u32x4 qf0 = qn0, qf1 = qn1;
    if (qt + ATTL_THREADS / 64 < nqt) load_q(qt + ATTL_THREADS / 64, qn0, qn1);
    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 lacc = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY;
    // the chunk loop is NOT unrolled (one chunk's scores in registers at a time); every chunk has CH full tile slots,
    // slots past the last tile held in LDS re-read tile NKT - 1 and are masked like every key >= N
#pragma unroll 1
    for (int c = 0; c < NCHUNK; ++c) {
      const int t0 = __builtin_amdgcn_readfirstlane(c * CH);
      f32x4 sacc[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) sacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      // S^T tiles of the chunk, two tiles (4 fragment reads) ahead of their 4 MFMAs
#pragma unroll
      for (int i0 = 0; i0 < CH; i0 += 2) {
        u32x4 kf[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int tl = (t0 + i0 + i < NKT) ? (t0 + i0 + i) : (NKT - 1);
          kf[i][0] = *reinterpret_cast<const u32x4*>(Ks + k_lds_off(tl * 16 + qi, g));
          kf[i][1] = *reinterpret_cast<const u32x4*>(Ks + k_lds_off(tl * 16 + qi, 4 + g));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (std::is_same<HT, f16>::value)
              sacc[i0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[i][hf]), __builtin_bit_cast(f16x8, hf ? qf1 : qf0), sacc[i0 + i], 0, 0, 0);
            else
              sacc[i0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[i][hf]), __builtin_bit_cast(bf16x8_t, hf ? qf1 : qf0), sacc[i0 + i], 0, 0, 0);
          }
        }
      }
      if ((t0 + CH) * 16 > N) {  // wave-uniform: only the chunk(s) reaching past key N - 1 carry mask code
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((t0 + i) * 16 + g * 4 + r >= N) sacc[i][r] = -INFINITY;
      }
      float mc = -INFINITY;
#pragma unroll
      for (int i = 0; i < CH; ++i) mc = fmaxf(fmaxf(mc, fmaxf(sacc[i][0], sacc[i][1])), fmaxf(sacc[i][2], sacc[i][3]));
      mc = fmaxf(mc, __shfl_xor(mc, 16, 64));
      mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
      const float m_new = fmaxf(m_run, mc);                  // finite from the first chunk on (it holds key 0 < N)
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);   // first chunk: exp2(-inf) = 0
      m_run = m_new;
      const float nmc = -m_new * scale_log2e;
#pragma unroll
      for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[i][r] = __builtin_amdgcn_exp2f(fmaf(sacc[i][r], scale_log2e, nmc));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
      lacc *= alpha;
      // O^T += V^T P^T, l += 1^T P^T over the chunk's k-steps of 32 keys (masked slots carry P = 0 exactly)
#pragma unroll
      for (int ks = 0; ks < CH / 2; ++ks) {
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
        const int ta = (t0 + 2 * ks < NKT) ? (t0 + 2 * ks) : (NKT - 1);
        const int tb = (t0 + 2 * ks + 1 < NKT) ? (t0 + 2 * ks + 1) : (NKT - 1);
        s16x4 v0[4], v1[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          v0[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + v_lds_off(ta * 16 + 4 * g + tq, dt) + tp * 8));
          v1[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + v_lds_off(tb * 16 + 4 * g + tq, dt) + tp * 8));
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          s16x8 vf;
#pragma unroll
          for (int r = 0; r < 4; ++r) { vf[r] = v0[dt][r]; vf[4 + r] = v1[dt][r]; }
          if constexpr (std::is_same<HT, f16>::value)
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf), __builtin_bit_cast(f16x8, pf), oacc[dt], 0, 0, 0);
          else
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf), oacc[dt], 0, 0, 0);
        }
        if constexpr (std::is_same<HT, f16>::value)
          lacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ones), __builtin_bit_cast(f16x8, pf), lacc, 0, 0, 0);
        else
          lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones), __builtin_bit_cast(bf16x8_t, pf), lacc, 0, 0, 0);
      }
    }
    const float inv_l = 1.0f / lacc[0];
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

template <typename HT, int NKT>
int launch_attn_long(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t s) {
  constexpr int CH = 10;
  const size_t smem = 2 * (size_t)NKT * 16 * 128;
  SMOE_ENSURE_SMEM(attn_fwd_long_kernel<HT, NKT, CH>);
  hipLaunchKernelGGL((attn_fwd_long_kernel<HT, NKT, CH>), dim3(B * H), dim3(ATTL_THREADS), smem, s, (const HT*)qkv, (HT*)out,
                     N, H, scale * 1.4426950408889634f);
  SMOE_CHECK_LAUNCH("smoe_attention_fwd/long");
  return 0;
}

template <typename HT, int NKT, bool EXACT>
int launch_attn(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t s, float* lse) {
  const size_t smem = 2 * (size_t)NKT * 16 * 128;
  auto kern = attn_fwd_kernel<HT, NKT, EXACT>;
  SMOE_ENSURE_SMEM(attn_fwd_kernel<HT, NKT, EXACT>);
  hipLaunchKernelGGL(kern, dim3(B * H), dim3(ATT_THREADS), smem, s, (const HT*)qkv, (HT*)out, N, H,
                     scale * 1.4426950408889634f, lse);
  SMOE_CHECK_LAUNCH("smoe_attention_fwd");
  return 0;
}

template <typename HT>
int attn_dispatch(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t s, float* lse) {
  const int nkt = (N + 15) / 16;
  if (lse && nkt > 16) {
    smoe_set_error("smoe_attention_fwd: the log-sum-exp output (training) is kept for N <= 256 only, N=%d", N);
    return 1;
  }
  if (nkt == 13) return launch_attn<HT, 13, true>(qkv, out, B, N, H, scale, s, lse);   // N = 197 / 198 (ViT @224 + cls)
  if (nkt == 4) return launch_attn<HT, 4, true>(qkv, out, B, N, H, scale, s, lse);
  if (nkt == 8) return launch_attn<HT, 8, true>(qkv, out, B, N, H, scale, s, lse);
  if (nkt == 16) return launch_attn<HT, 16, true>(qkv, out, B, N, H, scale, s, lse);
  if (nkt < 4) return launch_attn<HT, 4, false>(qkv, out, B, N, H, scale, s, lse);
  if (nkt < 8) return launch_attn<HT, 8, false>(qkv, out, B, N, H, scale, s, lse);
  if (nkt < 13) return launch_attn<HT, 13, false>(qkv, out, B, N, H, scale, s, lse);
  if (nkt < 16) return launch_attn<HT, 16, false>(qkv, out, B, N, H, scale, s, lse);
  // long sequences: K / V tiles held in LDS rounded up to the next instantiated size (rows >= N duplicate row N - 1)
  if (nkt <= 20) return launch_attn_long<HT, 20>(qkv, out, B, N, H, scale, s);
  if (nkt <= 30) return launch_attn_long<HT, 30>(qkv, out, B, N, H, scale, s);
  if (nkt <= 37) return launch_attn_long<HT, 37>(qkv, out, B, N, H, scale, s);   // N = 577: ViT-L/16 @384 + cls
  if (nkt <= 40) return launch_attn_long<HT, 40>(qkv, out, B, N, H, scale, s);
  smoe_set_error("smoe_attention_fwd: N=%d unsupported (N <= 640)", N);
  return 1;
}

}  // namespace

// qkv [B, N, 3, H, 64] (the output layout of the fused qkv projection), out [B, N, H*64]; f16 or bf16.
extern "C" int smoe_attention_supported(int N, int head_dim) { return (N >= 1 && N <= 640 && head_dim == ATT_D) ? 1 : 0; }

extern "C" int smoe_attention_fwd(const void* qkv, void* out, int dtype, int B, int N, int H, int head_dim, float scale,
                                  float* lse, void* stream) {
  SMOE_REQUIRE(smoe_attention_supported(N, head_dim), "smoe_attention_fwd: unsupported N=%d head_dim=%d", N, head_dim);
  SMOE_REQUIRE(B >= 0 && H >= 1, "smoe_attention_fwd: bad B=%d H=%d", B, H);
  if (B == 0) return 0;
  SMOE_REQUIRE(qkv && out, "smoe_attention_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SMOE_F16) return attn_dispatch<f16>(qkv, out, B, N, H, scale, s, lse);
  if (dtype == SMOE_BF16) return attn_dispatch<bf16_bits>(qkv, out, B, N, H, scale, s, lse);
  smoe_set_error("smoe_attention_fwd: dtype must be f16 or bf16");
  return 1;
}
