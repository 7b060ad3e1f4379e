// Two-workgroups-per-CU grouped GEMM for plain 16-bit outputs (included by gemm.hip; uses its helpers).
//
// What the persistent 8-wave kernel cannot do: overlap a tile's stores -- ~10.5 B/clk per CU, 15 k of a
// 55 k-cycle GEMM-1 tile -- with MFMAs; a store costs its issuing wave the same time wherever it is issued, and that kernel's waves
// own the whole register file and LDS.  Here a workgroup is HALF a CU: four waves (one per SIMD) at 256 registers, a 128 x 256 tile,
// 32-deep K-steps in a three-stage LDS ring (72 KiB), one workgroup per tile, two workgroups resident per CU.  The two are not
// synchronised with each other, so one's prologue (tile lookup, gather addresses, first fetches) and epilogue (bias, GELU, pack, 16
// stores per wave) run under the other's main loop, and inside the main loops each SIMD holds one wave of each workgroup: the DMA
// issue and LDS latency of one hide behind the MFMAs of the other with no barrier between them.
//   wave (wr, wc) of 2 x 2: rows wr * 64 .. + 63 (4 row fragments), columns wc * 128 .. + 127 (8 column fragments): 32 accumulator
//   fragments = 128 registers; per K-step 8 + 4 ds_read_b128 for 32 MFMAs.
//   LDS stage: A rows 0-127, W rows 0-255, 64 B per row (32 k-elements); the row's four 16-byte chunks are XOR-swizzled with
//   {0, 2, 3, 1}[(row >> 2) & 3], which makes the b128 fragment reads conflict-free under the LDS's 16-lane read groups.
//   DMA piece = 16 rows x 64 B; 8 A + 16 W pieces per K-step = 6 per wave.
// Same MFMA, same K order as every other variant: bit-identical results.
template <typename AB, typename OT>
__global__ __launch_bounds__(256, 2) void grouped_gemm_w2(
    const AB* __restrict__ A, const AB* __restrict__ W, const float* __restrict__ bias, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ group_expert, int E, int K, int N, int epilogue, OT* __restrict__ out, int n_tiles_n, int group_m,
    const int64_t* __restrict__ a_gather, int a_div) {
  static_assert(sizeof(AB) == 2 && sizeof(OT) == 2, "16-bit operands and outputs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TBM = 128, TBN = 256, NW = 4, KB = 64;          // KB: bytes of a row per K-step (32 elements)
  constexpr int STAGE = (TBM + TBN) * KB;                        // 24 KiB
  constexpr int NSTAGE = 3;

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
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = K / 32;

  // ---- operand sources: this wave's pieces are A pieces wave, wave + 4 and W pieces wave, wave + 4, + 8, + 12 ------------------
  auto swz4 = [](int row) -> int { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; };   // {0, 2, 3, 1}[(row >> 2) & 3], packed 2 bits each
  const int l_row = lane >> 2, l_pos = lane & 3;
  const char* a_src[2];
  const char* w_src[4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int r = (wave + NW * s) * 16 + l_row;                  // tile row
    int gr = m0 + r;
    if (gr >= m_end) gr = m_end - 1;
    int64_t arow = gr;
    if (a_gather) {
      const int64_t v = a_gather[gr];
      arow = a_div == 1 ? v : (int64_t)((uint32_t)v / (uint32_t)a_div);
    }
    a_src[s] = reinterpret_cast<const char*>(A) + (arow * K + ((l_pos ^ swz4(r)) << 3)) * 2;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int r = (wave + NW * s) * 16 + l_row;
    int gw = n0 + r;
    if (gw >= N) gw = N - 1;
    w_src[s] = reinterpret_cast<const char*>(W) + (((int64_t)e * N + gw) * K + ((l_pos ^ swz4(r)) << 3)) * 2;
  }
#define W2_DMA(SRC, DST)                                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),                 \
                                   (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
  auto dma_step = [&](int t) {          // the 6 pieces of K-step t -> stage t % NSTAGE
    char* sa = smem + (t % NSTAGE) * STAGE;
    char* sw = sa + TBM * KB;
#pragma unroll
    for (int s = 0; s < 2; ++s) W2_DMA(a_src[s] + (int64_t)t * KB, sa + (wave + NW * s) * 1024);
#pragma unroll
    for (int s = 0; s < 4; ++s) W2_DMA(w_src[s] + (int64_t)t * KB, sw + (wave + NW * s) * 1024);
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  dma_step(0);
  if (nk > 1) dma_step(1);
  // fragment read offsets inside a stage (the swizzle depends on the row only)
  int a_off[4], b_off[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wr * 64 + i * 16 + fr;
    a_off[i] = r * KB + ((fq ^ swz4(r)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = wc * 128 + j * 16 + fr;
    b_off[j] = TBM * KB + r * KB + ((fq ^ swz4(r)) << 4);
  }

  for (int t = 0; t < nk; ++t) {
    // K-step t's pieces have landed for this wave (the 6 of step t + 1 may stay in flight); the barrier makes them visible and
    // tells that every wave is done with step t - 1, whose stage step t + 2 takes over
    if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nk) dma_step(t + 2);
    const char* st = smem + (t % NSTAGE) * STAGE;
    u32x4 b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const u32x4*>(st + b_off[j]);
    u32x4 a0 = *reinterpret_cast<const u32x4*>(st + a_off[0]);
    u32x4 a1 = *reinterpret_cast<const u32x4*>(st + a_off[1]);
    __builtin_amdgcn_sched_barrier(0);
#define W2_ROW(I, AF)                                                                                                  \
  _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                      \
    if constexpr (std::is_same<AB, f16>::value)                                                                        \
      acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[j]), __builtin_bit_cast(f16x8, AF), \
                                                         acc[I][j], 0, 0, 0);                                          \
    else                                                                                                               \
      acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[j]),                            \
                                                          __builtin_bit_cast(bf16x8_t, AF), acc[I][j], 0, 0, 0);        \
  }
    const u32x4 a2 = *reinterpret_cast<const u32x4*>(st + a_off[2]);
    W2_ROW(0, a0);
    __builtin_amdgcn_sched_barrier(0);
    const u32x4 a3 = *reinterpret_cast<const u32x4*>(st + a_off[3]);
    W2_ROW(1, a1);
    __builtin_amdgcn_sched_barrier(0);
    W2_ROW(2, a2);
    W2_ROW(3, a3);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef W2_ROW
#undef W2_DMA

  // ---- epilogue: bias, GELU, pack, lane swaps, 16 sixteen-byte stores per wave (as grouped_gemm_ps<..., DIRECT>) ---------------
  f32x4 bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = n0 + wc * 128 + j * 16 + fq * 4;
    bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias && col < N) bv[j] = *reinterpret_cast<const f32x4*>(bias + (int64_t)e * N + col);   // N % 8 == 0: four columns exist together
  }
  const int rows_here = (m_end - m0 < TBM) ? (m_end - m0) : TBM;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + (int64_t)m0 * N * 2, 0,
                                                                       rows_here * N * 2, 0x00020000);
  const uint32_t OOR = 0x80000000u;
  // after the swaps lane (fr, fq) holds columns  (2 q + (fq & 1)) * 16 + (fq >> 1) * 8 ... + 7  of fragment pair q
  const int cq = wc * 128 + (fq & 1) * 16 + (fq >> 1) * 8;
  uint32_t off0 = (uint32_t)(((wr * 64 + fr) * N + n0 + cq) * 2);
  const uint32_t row16 = (uint32_t)(16 * N * 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 va = acc[i][2 * q] + bv[2 * q], vb = acc[i][2 * q + 1] + bv[2 * q + 1];
      if (epilogue == SMOE_EPI_GELU) { va = gelu_fast4(va); vb = gelu_fast4(vb); }
      uint32_t a0, a1, b0, b1;
      pack4<OT>(va, a0, a1);
      pack4<OT>(vb, b0, b1);
      const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
      const u32x4 v = u32x4{s0[0], s1[0], s0[1], s1[1]};
      const bool ok = n0 + cq + q * 32 < N;
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(ok ? off0 + (uint32_t)(q * 64) : OOR), 0, 0);
    }
    off0 += row16;
  }
}

template <typename AB, typename OT>
int launch_w2(const void* A, const void* W, const float* bias, const int32_t* offsets, const int32_t* group_expert, int E,
              int64_t m_rows_max, int K, int N, int epilogue, void* out, hipStream_t s, const int64_t* a_gather, int a_div) {
  constexpr int TBM = 128, TBN = 256;
  const int n_tiles_n = (N + TBN - 1) / TBN;
  const int group_m = 8;
  const int max_m_tiles = (int)((m_rows_max + TBM - 1) / TBM) + E;
  const int m_groups = (max_m_tiles + group_m - 1) / group_m;
  const int grid = m_groups * group_m * n_tiles_n;
  const size_t lds = 3 * (size_t)(TBM + TBN) * 64;
  SMOE_ENSURE_SMEM((grouped_gemm_w2<AB, OT>));
  hipLaunchKernelGGL((grouped_gemm_w2<AB, OT>), dim3(grid), dim3(256), lds, s, (const AB*)A, (const AB*)W, bias, offsets, group_expert, E,
                     K, N, epilogue, (OT*)out, n_tiles_n, group_m, a_gather, a_div);
  SMOE_CHECK_LAUNCH("smoe_grouped_gemm/w2");
  return 0;
}
