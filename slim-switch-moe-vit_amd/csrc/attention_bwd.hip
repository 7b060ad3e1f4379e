// Self-attention backward for ViT token counts (N <= 256, head dim 64): the training step's counterpart of attention.hip
// (models/vision_transformer.py:248-280 under engine.py:52-74's forward + backward).
//
//   S = scale q k^T,  P = softmax(S),  O = P v          (forward; P is never stored, only lse = log2 sum exp2(S log2 e))
//   dV = P^T dO,  dP = dO V^T,  dS = P (dP - delta) scale,  delta = rowsum(dO O),  dQ = dS K,  dK = dS^T Q
//
// One workgroup (4 waves, one per SIMD, up to 512 registers each) per (image, head); Q, K, V, dO of the head sit in LDS
// whole (row-major [token][64], 16-byte chunks XOR-swizzled: conflict-free ds_read_b128 row reads, 2-way transposed
// reads), queries are walked in steps of 32:
//   phase A  "key on the lane": wave w owns key tiles w, w + 4, ... and keeps their dK^T / dV^T ([64 x 16] each) in
//            accumulators over the whole sweep.  Per owned tile and step: S and dP as  A = Q / dO rows, B = K / V rows
//            (lane (g, key) ends up with queries 4 g + r), P = exp2(S c - lse), dS; then the accumulators of S / dP ARE
//            the B operands of  dV^T += dO^T P  and  dK^T += Q^T dS  (contraction over the 32 queries of the step; the
//            A operands are transposed reads of dO / Q) -- P never leaves the registers.
//   phase B  dQ contracts over KEYS, which sit on the lanes: dS (16 bit) crosses LDS once, as a [32 query][key] image;
//            wave w then computes  dQ^T[16 w .. 16 w + 15][32 queries] = K^T dS^T  over ALL keys (A = transposed reads
//            of K, B = row reads of the image), so dQ needs no reduction across waves or steps and is stored at once.
// Rows / keys >= N duplicate row N - 1 in LDS and carry P = dS = 0.  10 N^2 64 FLOP per (image, head) on the matrix cores.
#include "smoe_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int AB_D = 64;
constexpr int DS_ROW = 512;                 // bytes per query row of the dS image (up to 256 keys x 2 B)
constexpr int DS_BYTES = 32 * DS_ROW;

__device__ __forceinline__ int img_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename HT>
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, const f32x4& c) {
  if constexpr (std::is_same<HT, f16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <typename HT> __device__ __forceinline__ unsigned short to16(float v) {
  if constexpr (std::is_same<HT, f16>::value) return __builtin_bit_cast(unsigned short, (f16)v);
  else return f32_to_bf16(v);
}
template <typename HT> __device__ __forceinline__ float from16(unsigned short v) {
  if constexpr (std::is_same<HT, f16>::value) return (float)__builtin_bit_cast(f16, v);
  else return bf16_to_f32(v);
}
template <typename HT> __device__ __forceinline__ u32x4 pack8(const f32x4& lo, const f32x4& hi) {
  u32x4 r;
  r[0] = (uint32_t)to16<HT>(lo[0]) | ((uint32_t)to16<HT>(lo[1]) << 16);
  r[1] = (uint32_t)to16<HT>(lo[2]) | ((uint32_t)to16<HT>(lo[3]) << 16);
  r[2] = (uint32_t)to16<HT>(hi[0]) | ((uint32_t)to16<HT>(hi[1]) << 16);
  r[3] = (uint32_t)to16<HT>(hi[2]) | ((uint32_t)to16<HT>(hi[3]) << 16);
  return r;
}

// two transposed reads -> the A operand [row = column idx of the 16-column block `cb`][k-slot (g, j)]: j < 4 <-> image rows
// r_lo + j, j >= 4 <-> r_hi + (j - 4); the lane supplies row (r + tq), columns 16 cb + 4 tp .. + 3
__device__ __forceinline__ u32x4 tr_pair(const char* img, int r_lo, int r_hi, int cb, int tq, int tp) {
  const int chunk = 2 * cb + (tp >> 1), sub = (tp & 1) * 8;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + img_off(r_lo + tq, chunk) + sub));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + img_off(r_hi + tq, chunk) + sub));
  s16x8 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) { v[r] = a[r]; v[4 + r] = b[r]; }
  return __builtin_bit_cast(u32x4, v);
}

template <typename HT, int NKT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void attn_bwd_kernel(const HT* __restrict__ qkv, const HT* __restrict__ o,
                                                                 const HT* __restrict__ dout, const float* __restrict__ lse,
                                                                 HT* __restrict__ dqkv, int N, int H, float scale, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NKT + (NKT & 1);      // key tiles rounded to whole 32-key steps
  constexpr int NR = NT * 16;              // rows of every image
  constexpr int KS = (NKT + NW - 1) / NW;  // key tiles a wave may own
  constexpr int AB_THREADS = 64 * NW;      // NW = 4: one wave per SIMD; NW = 8: two (the same LDS images, half the tiles per wave)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  char* Qs = smem;
  char* Ks = Qs + NR * 128;
  char* Vs = Ks + NR * 128;
  char* Ds = Vs + NR * 128;                // dO
  char* Si = Ds + NR * 128;                // dS image [32 queries][DS_ROW bytes]
  float* L2 = reinterpret_cast<float*>(Si + DS_BYTES);
  float* Dl = L2 + NR;

  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int64_t ts3 = (int64_t)3 * H * AB_D, ts1 = (int64_t)H * AB_D;   // elements between tokens: qkv / o, dout
  const HT* qbase = qkv + (int64_t)b * N * ts3 + h * AB_D;
  const HT* obase = o + (int64_t)b * N * ts1 + h * AB_D;
  const HT* dbase = dout + (int64_t)b * N * ts1 + h * AB_D;
  HT* gbase = dqkv + (int64_t)b * N * ts3 + h * AB_D;

  // ---- stage Q, K, V, dO (LDS DMA, 8 rows per wave-instruction, swizzle on the source address) ------------------------------
  {
    const int l_row = lane >> 3, l_pos = lane & 7;
    for (int pc = wave; pc < NR / 8; pc += NW) {
      const int row = pc * 8 + l_row;
      const int srow = row < N ? row : N - 1;
      const int kc = l_pos ^ ((row >> 1) & 7);
      const HT* p3 = qbase + (int64_t)srow * ts3 + kc * 8;
      const HT* p1 = dbase + (int64_t)srow * ts1 + kc * 8;
#define AB_DMA(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
      AB_DMA(p3, Qs + pc * 1024);
      AB_DMA(p3 + H * AB_D, Ks + pc * 1024);
      AB_DMA(p3 + 2 * H * AB_D, Vs + pc * 1024);
      AB_DMA(p1, Ds + pc * 1024);
#undef AB_DMA
    }
  }
  // the dS image starts out zero: the columns of the padding tile (keys NKT*16 .. NT*16) are never written
  for (int i = tid; i < DS_BYTES / 16; i += AB_THREADS) reinterpret_cast<u32x4*>(Si)[i] = u32x4{0u, 0u, 0u, 0u};
  // per-query constants: lse and delta = rowsum(dO O)
  for (int q = tid; q < NR; q += AB_THREADS) {
    const int sq = q < N ? q : N - 1;
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4 ov = *reinterpret_cast<const u32x4*>(obase + (int64_t)sq * ts1 + c * 8);
      const u32x4 dv = *reinterpret_cast<const u32x4*>(dbase + (int64_t)sq * ts1 + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dl = fmaf(from16<HT>((unsigned short)(ov[e] & 0xffffu)), from16<HT>((unsigned short)(dv[e] & 0xffffu)), dl);
        dl = fmaf(from16<HT>((unsigned short)(ov[e] >> 16)), from16<HT>((unsigned short)(dv[e] >> 16)), dl);
      }
    }
    Dl[q] = dl;
    L2[q] = lse[((int64_t)b * H + h) * N + sq];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 dVa[KS][4], dKa[KS][4];
#pragma unroll
  for (int i = 0; i < KS; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dVa[i][dt] = dKa[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = (N + 31) >> 5;
  for (int qs = 0; qs < nsteps; ++qs) {
    const int q0 = qs * 32;
    // ---------------------------------------------------------------- phase A: own key tiles, all 32 queries of the step
    u32x4 qf[2][2], df[2][2], dOT[4], QT[4];
    f32x4 l2v[2], dlv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        qf[t][kk] = *reinterpret_cast<const u32x4*>(Qs + img_off(q0 + 16 * t + li, 4 * kk + g));
        df[t][kk] = *reinterpret_cast<const u32x4*>(Ds + img_off(q0 + 16 * t + li, 4 * kk + g));
      }
      l2v[t] = *reinterpret_cast<const f32x4*>(L2 + q0 + 16 * t + 4 * g);
      dlv[t] = *reinterpret_cast<const f32x4*>(Dl + q0 + 16 * t + 4 * g);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dOT[dt] = tr_pair(Ds, q0 + 4 * g, q0 + 16 + 4 * g, dt, tq, tp);
      QT[dt] = tr_pair(Qs, q0 + 4 * g, q0 + 16 + 4 * g, dt, tq, tp);
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int kt = wave + NW * i;
      if (kt < NKT) {        // wave-uniform
        u32x4 kf[2], vf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          kf[kk] = *reinterpret_cast<const u32x4*>(Ks + img_off(kt * 16 + li, 4 * kk + g));
          vf[kk] = *reinterpret_cast<const u32x4*>(Vs + img_off(kt * 16 + li, 4 * kk + g));
        }
        f32x4 S[2], dP[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          S[t] = mma<HT>(qf[t][0], kf[0], f32x4{0.f, 0.f, 0.f, 0.f});
          dP[t] = mma<HT>(df[t][0], vf[0], f32x4{0.f, 0.f, 0.f, 0.f});
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          S[t] = mma<HT>(qf[t][1], kf[1], S[t]);
          dP[t] = mma<HT>(df[t][1], vf[1], dP[t]);
        }
        const int key = kt * 16 + li;
        const bool kvalid = key < N;
        f32x4 P[2], dS[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool valid = kvalid && (q0 + 16 * t + 4 * g + r < N);
            const float p = valid ? __builtin_amdgcn_exp2f(fmaf(S[t][r], scale_log2e, -l2v[t][r])) : 0.f;
            P[t][r] = p;
            dS[t][r] = p * (dP[t][r] - dlv[t][r]) * scale;
          }
        const u32x4 pb = pack8<HT>(P[0], P[1]), sb = pack8<HT>(dS[0], dS[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dVa[i][dt] = mma<HT>(dOT[dt], pb, dVa[i][dt]);
          dKa[i][dt] = mma<HT>(QT[dt], sb, dKa[i][dt]);
        }
        // dS -> LDS image [query row][key], 16-bit: phase B's contraction runs over the keys
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * g + r;
            const unsigned short v = (unsigned short)((r & 1) ? (sb[2 * t + (r >> 1)] >> 16) : (sb[2 * t + (r >> 1)] & 0xffffu));
            *reinterpret_cast<unsigned short*>(Si + row * DS_ROW + ((((key >> 3) ^ (row & 15))) << 4) + (key & 7) * 2) = v;
          }
      }
    }
    __syncthreads();
    // ---------------------------------------------------------------- phase B: dQ^T rows 16 w .. 16 w + 15, all keys
    // (NW = 8: wave w takes head-dim block w & 3 for the 16 queries of sub-tile w >> 2 only)
    {
      constexpr int TB = NW == 4 ? 2 : 1;                 // query sub-tiles per wave
      const int db = wave & 3, t0 = NW == 4 ? 0 : (wave >> 2);
      f32x4 dQ[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) dQ[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NT / 2; ++ks) {
        const u32x4 ka = tr_pair(Ks, 32 * ks + 8 * g, 32 * ks + 8 * g + 4, db, tq, tp);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const int row = 16 * (t0 + t) + li;
          const u32x4 sv = *reinterpret_cast<const u32x4*>(Si + row * DS_ROW + (((4 * ks + g) ^ (row & 15)) << 4));
          dQ[t] = mma<HT>(ka, sv, dQ[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const int q = q0 + 16 * (t0 + t) + li;
        if (q < N) {
          u32x2 pk;
          pk[0] = (uint32_t)to16<HT>(dQ[t][0]) | ((uint32_t)to16<HT>(dQ[t][1]) << 16);
          pk[1] = (uint32_t)to16<HT>(dQ[t][2]) | ((uint32_t)to16<HT>(dQ[t][3]) << 16);
          *reinterpret_cast<u32x2*>(gbase + (int64_t)q * ts3 + 16 * db + 4 * g) = pk;
        }
      }
    }
    __syncthreads();   // the image is rewritten by the next step's phase A
  }
  // ---- dK, dV of the owned key tiles: lane (g, key li) holds head-dim elements 16 dt + 4 g + r -----------------------------
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int kt = wave + NW * i;
    const int key = kt * 16 + li;
    if (kt < NKT && key < N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        u32x2 pk, pv;
        pk[0] = (uint32_t)to16<HT>(dKa[i][dt][0]) | ((uint32_t)to16<HT>(dKa[i][dt][1]) << 16);
        pk[1] = (uint32_t)to16<HT>(dKa[i][dt][2]) | ((uint32_t)to16<HT>(dKa[i][dt][3]) << 16);
        pv[0] = (uint32_t)to16<HT>(dVa[i][dt][0]) | ((uint32_t)to16<HT>(dVa[i][dt][1]) << 16);
        pv[1] = (uint32_t)to16<HT>(dVa[i][dt][2]) | ((uint32_t)to16<HT>(dVa[i][dt][3]) << 16);
        HT* kp = gbase + (int64_t)key * ts3 + H * AB_D + 16 * dt + 4 * g;
        *reinterpret_cast<u32x2*>(kp) = pk;
        *reinterpret_cast<u32x2*>(kp + H * AB_D) = pv;
      }
    }
  }
}

template <typename HT, int NKT, int NW>
int launch_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int B, int N, int H, float scale,
               hipStream_t s) {
  constexpr int NT = NKT + (NKT & 1);
  const size_t smem = 4 * (size_t)NT * 16 * 128 + DS_BYTES + 2 * (size_t)NT * 16 * sizeof(float);
  SMOE_ENSURE_SMEM((attn_bwd_kernel<HT, NKT, NW>));
  hipLaunchKernelGGL((attn_bwd_kernel<HT, NKT, NW>), dim3(B * H), dim3(64 * NW), smem, s, (const HT*)qkv, (const HT*)o,
                     (const HT*)dout, lse, (HT*)dqkv, N, H, scale, scale * 1.4426950408889634f);
  SMOE_CHECK_LAUNCH("smoe_attention_bwd");
  return 0;
}

// waves per workgroup: the images fill the LDS (one workgroup per CU), so 8 waves = two per SIMD is what hides the latencies between
// a step's dependent phases (LDS reads -> MFMA -> exp / pack -> MFMA); SMOE_ATTN_BWD_WAVES=4 keeps one per SIMD (A/B)
inline int bwd_waves() {
  static const int w = [] { const char* e = getenv("SMOE_ATTN_BWD_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
  return w;
}

template <typename HT>
int bwd_dispatch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int B, int N, int H, float scale,
                 hipStream_t s) {
  const int nkt = (N + 15) / 16;
  if (nkt <= 4) return launch_bwd<HT, 4, 4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
  if (nkt <= 8) return bwd_waves() == 8 ? launch_bwd<HT, 8, 8>(qkv, o, dout, lse, dqkv, B, N, H, scale, s)
                                        : launch_bwd<HT, 8, 4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
  if (nkt <= 13) return bwd_waves() == 8 ? launch_bwd<HT, 13, 8>(qkv, o, dout, lse, dqkv, B, N, H, scale, s)   // N = 197 / 198
                                         : launch_bwd<HT, 13, 4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
  return bwd_waves() == 8 ? launch_bwd<HT, 16, 8>(qkv, o, dout, lse, dqkv, B, N, H, scale, s)
                          : launch_bwd<HT, 16, 4>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
}

}  // namespace

extern "C" int smoe_attention_bwd_supported(int N, int head_dim) { return (N >= 1 && N <= 256 && head_dim == AB_D) ? 1 : 0; }

// qkv, dqkv [B, N, 3, H, 64]; o, dout [B, N, H*64]; lse [B, H, N] f32 from smoe_attention_fwd; f16 or bf16
extern "C" int smoe_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int dtype, int B,
                                  int N, int H, int head_dim, float scale, void* stream) {
  SMOE_REQUIRE(smoe_attention_bwd_supported(N, head_dim), "smoe_attention_bwd: unsupported N=%d head_dim=%d (N <= 256, head dim 64)",
               N, head_dim);
  SMOE_REQUIRE(B >= 0 && H >= 1, "smoe_attention_bwd: bad B=%d H=%d", B, H);
  if (B == 0) return 0;
  SMOE_REQUIRE(qkv && o && dout && lse && dqkv, "smoe_attention_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SMOE_F16) return bwd_dispatch<f16>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
  if (dtype == SMOE_BF16) return bwd_dispatch<bf16_bits>(qkv, o, dout, lse, dqkv, B, N, H, scale, s);
  smoe_set_error("smoe_attention_bwd: dtype must be f16 or bf16");
  return 1;
}
