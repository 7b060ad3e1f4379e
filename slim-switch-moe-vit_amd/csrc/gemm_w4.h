// One-wave-per-SIMD persistent grouped GEMM for plain 16-bit outputs (included by gemm.hip behind gemm_persistent.h; uses its
// helpers).  Why a second kernel: gemm_persistent.h's stamps show what holds the 8-wave ping-pong
// kernel at 0.37-0.40 of the MFMA peak -- a CU moves its output at ~10.5 B/clk and nothing overlaps that with MFMAs (15 k of a
// 55 k-cycle GEMM-1 tile), because eight waves at 256 registers hold exactly one tile's accumulators.  Here a workgroup is FOUR
// waves, one per SIMD, 512 registers each:
//   * 256 x 256 tile, 2 x 2 waves of 128 x 128: the 64 accumulator fragments are 256 registers (the AGPR half of the file); the
//     LDS reads per MFMA drop from 0.40 KB to 0.25 KB; one barrier per K-tile instead of sixteen hand-overs;
//   * SPREAD: the finished tile is converted (bias, GELU, pack, lane swaps) into 32 sixteen-byte registers per lane and its stores
//     are ISSUED four per K-tile under the next tile's MFMAs, each group right behind that K-tile's operand DMA: `vmcnt` counts in
//     issue order, so the wait that publishes K-tile t + 1 leaves exactly those four stores in flight, and a CU drains ~5 stores per
//     wave and K-tile, so none is ever waited for.
// Same arithmetic per element as grouped_gemm_ps<..., DIRECT> (same MFMA, same K order): bit-identical results.
//
// LDS: two 64-KiB operand stages (A rows 0-255, W rows 0-255, 128 B per row, chunk-swizzled as everywhere), the gather-row buffer
// and the two bias tiles at the top.  The K-tile loop keeps a parity: tile i's K-tile t sits in stage (t + par) & 1, and par flips
// between tiles when the number of K-tiles is odd, so the next tile's first K-tile always goes to the stage that was free longest.
template <typename AB, typename OT, bool SPREAD>
__global__ __launch_bounds__(256, 1) void grouped_gemm_w4(
    const AB* __restrict__ A, const AB* __restrict__ W, const float* __restrict__ bias, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ group_expert, int E, int K, int N, int epilogue, OT* out, int n_tiles_n, int group_m,
    const int64_t* __restrict__ a_gather, int a_div, const int32_t* __restrict__ group_end) {
  static_assert(sizeof(AB) == 2 && sizeof(OT) == 2, "16-bit operands and outputs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TBM = 256, TBN = 256, NW = 4;
  constexpr int STAGE = (TBM + TBN) * BK_BYTES;  // 64 KiB
  constexpr int LDS_TOTAL = 160 * 1024;
  constexpr int NS = 32;                         // 16-byte stores per lane per tile (8 row fragments x 4 column-fragment pairs)
  constexpr int SPK = 4;                         // stores issued per K-tile (SPREAD)
  static_assert(NS % SPK == 0 && SPK == 4, "the counted wait below says vmcnt(4)");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nk = K / 64;

  // ---- tile enumeration (the lane table of gemm_persistent.h; strided order) ------------------------------------------------
  int t_off = 0, t_end = 0, t_tb = 0, t_ge = lane;
  {
    const int li = lane < E ? lane : (group_end ? E - 1 : E);
    t_off = offsets[li];
    t_end = group_end ? group_end[li] : offsets[li < E ? li + 1 : E];
    int cnt = lane < E ? (t_end - t_off + TBM - 1) / TBM : 0;
    int incl = cnt;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const int up = __shfl_up(incl, sft, 64);
      if (lane >= sft) incl += up;
    }
    t_tb = incl - cnt;
    if (group_expert && lane < E) t_ge = group_expert[lane];
  }
  const int total_mt = __builtin_amdgcn_readlane(t_tb, 63);
  const int per_group = group_m * n_tiles_n;
  const int n_tiles = ((total_mt + group_m - 1) / group_m) * per_group;
  const int G = gridDim.x, per_xcd = G >> 3;
  int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  int e = 0, m0 = 0, m_end = 0, n0 = 0;
  auto advance = [&]() -> bool {
    while (tile < n_tiles) {
      const int g = tile / per_group, rem = tile % per_group;
      const int mt = g * group_m + rem % group_m;
      if (mt < total_mt) {
        const unsigned long long msk = __ballot(lane < E && t_tb <= mt);
        const int gi = 63 - __builtin_clzll(msk);
        m0 = __builtin_amdgcn_readlane(t_off, gi) + (mt - __builtin_amdgcn_readlane(t_tb, gi)) * TBM;
        m_end = __builtin_amdgcn_readlane(t_end, gi);
        e = __builtin_amdgcn_readlane(t_ge, gi);
        n0 = (rem / group_m) * TBN;
        return true;
      }
      tile += G;
    }
    return false;
  };

  // ---- operand sources: 32-bit byte offsets (the launcher keeps both operands under 4 GiB) ------------------------------------
  uint32_t a_src[8], w_src[8];
  const char* const Ab = reinterpret_cast<const char*>(A);
  const char* const Wb = reinterpret_cast<const char*>(W);
  const int* const rowbuf = reinterpret_cast<const int*>(smem + LDS_TOTAL - 2048 - TBM * 4);
  char* const bias_lds = smem + LDS_TOTAL - 2048;
  bool rows_in_lds = false;
  auto gathered_row = [&](int gr) -> int64_t {
    if (!a_gather) return gr;
    const int64_t v = rows_in_lds ? (int64_t)rowbuf[gr - m0] : a_gather[gr];
    return a_div == 1 ? v : (int64_t)((uint32_t)v / (uint32_t)a_div);
  };
  auto setup = [&](int lo) {
    const int l_row = lo >> 3, l_pos = lo & 7;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int r = 8 * (s * NW + wave) + l_row;
      const uint32_t sw8 = (uint32_t)((l_pos ^ ((r >> 1) & 7)) << 3);
      int gr = m0 + r;
      if (gr >= m_end) gr = m_end - 1;
      a_src[s] = (uint32_t)((gathered_row(gr) * K + sw8) * 2);
      int gw = n0 + r;
      if (gw >= N) gw = N - 1;
      w_src[s] = (uint32_t)((((int64_t)e * N + gw) * K + sw8) * 2);
    }
  };
#define W4_DMA(SRC, DST)                                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),                 \
                                   (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
  auto dma_tile = [&](int kt, int buf) {   // 16 one-KiB pieces per wave: the whole K-tile kt of the tile set up -> stage buf
    char* sa = smem + buf * STAGE;
    char* sw = sa + TBM * BK_BYTES;
#pragma unroll
    for (int s = 0; s < 8; ++s) W4_DMA(Ab + (a_src[s] + (uint32_t)kt * 128u), sa + (s * NW + wave) * 1024);
#pragma unroll
    for (int s = 0; s < 8; ++s) W4_DMA(Wb + (w_src[s] + (uint32_t)kt * 128u), sw + (s * NW + wave) * 1024);
  };
  auto prefetch_rows = [&]() {   // the NEXT tile's gather rows (low words), 64 per wave
    if (a_gather) {
      int gr = m0 + wave * 64 + lane;
      if (gr >= m_end) gr = m_end - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_gather + gr),
                                       (__attribute__((address_space(3))) void*)(smem + LDS_TOTAL - 2048 - TBM * 4 + wave * 256),
                                       4, 0, 0);
    }
  };
  auto issue_bias = [&](int parity) {
    if (bias && wave == 0) {
      int col = n0 + lane * 4;
      if (col > N - 4) col = N - 4;
      W4_DMA(bias + (int64_t)e * N + col, bias_lds + parity * 1024);
    }
  };

  f32x4 acc[8][8];
#define W4_MFMA_ROW(I, AF, BF)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                      \
    if constexpr (std::is_same<AB, f16>::value)                                                                        \
      acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, BF[j]), __builtin_bit_cast(f16x8, AF), \
                                                         acc[I][j], 0, 0, 0);                                          \
    else                                                                                                               \
      acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, BF[j]),                           \
                                                          __builtin_bit_cast(bf16x8_t, AF), acc[I][j], 0, 0, 0);        \
  }
  // One K-tile: 2 k-steps x 8 row fragments x 8 MFMAs.  One wave per SIMD has nobody to hide its LDS latency: every group of 8
  // MFMAs (128 cycles) is preceded by the read of the NEXT row fragment, the second k-step's 8 column fragments are read two per
  // group under the first k-step's last four groups, and sched_barriers keep hipcc from undoing the order (left to itself it read
  // each row fragment right in front of its MFMAs: a full LDS round trip exposed sixteen times per K-tile).
  auto compute = [&](int buf) {
    const char* sa = smem + buf * STAGE;
    const char* sw = sa + TBM * BK_BYTES;
    const int fr = lane & 15, fq = lane >> 4;
    u32x4 b0[8], b1[8], a0, a1;
    auto rd_a = [&](int i, int kk) { return *reinterpret_cast<const u32x4*>(sa + swz(wr * 128 + i * 16 + fr, kk * 4 + fq)); };
    auto rd_b = [&](int j, int kk) { return *reinterpret_cast<const u32x4*>(sw + swz(wc * 128 + j * 16 + fr, kk * 4 + fq)); };
#pragma unroll
    for (int j = 0; j < 8; ++j) b0[j] = rd_b(j, 0);
    a0 = rd_a(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      a1 = rd_a(i + 1, 0);
      if (i >= 4) { b1[(i - 4) * 2] = rd_b((i - 4) * 2, 1); b1[(i - 4) * 2 + 1] = rd_b((i - 4) * 2 + 1, 1); }
      W4_MFMA_ROW(i, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      a0 = i + 2 < 8 ? rd_a(i + 2, 0) : rd_a(0, 1);
      if (i >= 4) { b1[(i - 4) * 2 + 2] = rd_b((i - 4) * 2 + 2, 1); b1[(i - 4) * 2 + 3] = rd_b((i - 4) * 2 + 3, 1); }
      W4_MFMA_ROW(i + 1, a1, b0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      a1 = rd_a(i + 1, 1);
      W4_MFMA_ROW(i, a0, b1);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 < 8) a0 = rd_a(i + 2, 1);
      W4_MFMA_ROW(i + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // the finished tile's packed output, waiting to be stored (SPREAD), and where it goes
  u32x4 packed[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) packed[k] = u32x4{0u, 0u, 0u, 0u};
  int st_left = 0;                     // stores of the previous tile not issued yet (wave-uniform)
  int st_next = 0;                     // index of the next one
  __amdgpu_buffer_rsrc_t st_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);
  uint32_t st_off = 0;                 // lane's byte offset of store 0 inside the tile's row window
  uint32_t st_row16 = 0;
  int st_ok = 0;                       // bit q: the lane's 8 columns of fragment pair q exist
  constexpr uint32_t OOR = 0x80000000u;
  auto issue_stores = [&](int n) {     // the next n (<= SPK) stores of the pending tile; packed[] is shifted down by n afterwards
#pragma unroll
    for (int s = 0; s < SPK; ++s) {
      if (s < n) {
        const int k = st_next + s;
        const uint32_t off = st_off + (uint32_t)(k >> 2) * st_row16 + (uint32_t)(k & 3) * 64u;
        __builtin_amdgcn_raw_buffer_store_b128(packed[s], st_rs, (int)(((st_ok >> (k & 3)) & 1) ? off : OOR), 0, 0);
      }
    }
    if (n == SPK) {
#pragma unroll
      for (int k = 0; k + SPK < NS; ++k) packed[k] = packed[k + SPK];
    }
    st_next += n;
    st_left -= n;
  };

  if (!advance()) return;
  {
    int oz = 0;
    asm volatile("" : "+v"(oz));
    setup(lane + oz);
  }
  int par = 0;                          // stage of this tile's K-tile 0
  int bias_par = 0;
  issue_bias(0);
  dma_tile(0, 0);

  for (;;) {
    const int cm0 = m0, cm_end = m_end, cn0 = n0;   // the tile computed in this iteration
    tile += G;
    const bool more = advance();                    // (e, m0, m_end, n0) = the next tile from here on
    if (more) prefetch_rows();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int younger = 0;                                // vector-memory operations this wave issued AFTER the pieces of K-tile t
    for (int t = 0; t < nk; ++t) {
      const int cur = (t + par) & 1;
      // K-tile t's pieces have landed for this wave (everything older too); the barrier makes them visible and tells that every
      // wave is done with the other stage
      if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (= SPK: the stores issued behind those pieces)
      PP_BARRIER();
      younger = 0;
      if (t + 1 < nk) dma_tile(t + 1, cur ^ 1);
      if (SPREAD && st_left > 0) {        // NS % SPK == 0: always a full group
        issue_stores(SPK);
        younger = SPK;
      }
      compute(cur);
    }
    par = (par + nk) & 1;

    // ---- boundary: every wave is done with this tile's stages once it passes the barrier -------------------------------------
    if (SPREAD && st_left > 0) {          // a tile shorter than NS / SPK K-tiles: the rest of the previous tile's stores leave now
      while (st_left > 0) issue_stores(st_left < SPK ? st_left : SPK);
    }
    PP_BARRIER();
    int oz = 0;
    asm volatile("" : "+v"(oz));
    const int lane_e = (tid + oz) & 63;
    if (more) {
      rows_in_lds = a_gather != nullptr;
      setup(lane_e);
      issue_bias(bias_par ^ 1);
      dma_tile(0, par);                   // under the conversion below
    }
    // ---- conversion of (cm0, cm_end, cn0): bias, GELU, pack, lane swaps -> 32 sixteen-byte registers ----------------------------
    {
      const int fr = lane_e & 15, fq = lane_e >> 4;
      f32x4 bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int nl = wc * 128 + j * 16 + fq * 4;
        bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias) bv[j] = *reinterpret_cast<const f32x4*>(bias_lds + bias_par * 1024 + nl * 4);
      }
      bias_par ^= 1;
      const int rows_here = (cm_end - cm0 < TBM) ? (cm_end - cm0) : TBM;
      st_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(out) + (int64_t)cm0 * N * 2, 0, rows_here * N * 2, 0x00020000);
      // after the swaps lane (fr, fq) holds columns  (2 q + (fq & 1)) * 16 + (fq >> 1) * 8 ... + 7  of fragment pair q
      const int cq = wc * 128 + (fq & 1) * 16 + (fq >> 1) * 8;
      st_off = (uint32_t)(((wr * 128 + fr) * N + cn0 + cq) * 2);
      st_row16 = (uint32_t)(16 * N * 2);
      st_ok = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) st_ok |= (cn0 + cq + q * 32 < N) ? (1 << q) : 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 va = acc[i][2 * q] + bv[2 * q], vb = acc[i][2 * q + 1] + bv[2 * q + 1];
          if (epilogue == SMOE_EPI_GELU) { va = gelu_fast4(va); vb = gelu_fast4(vb); }
          uint32_t a0, a1, b0, b1;
          pack4<OT>(va, a0, a1);
          pack4<OT>(vb, b0, b1);
          const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          packed[i * 4 + q] = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      }
      st_left = NS;
      st_next = 0;
    }
    if (!SPREAD || !more) {               // no next main loop to hide them under: all stores now
      while (st_left > 0) issue_stores(st_left < SPK ? st_left : SPK);
    }
    if (!more) break;
  }
#undef W4_DMA
#undef W4_MFMA_ROW
}

template <typename AB, typename OT>
int launch_w4(const void* A, const void* W, const float* bias, const int32_t* offsets, const int32_t* group_expert, int E,
              int64_t m_rows_max, int K, int N, int epilogue, void* out, hipStream_t s, const int64_t* a_gather, int a_div,
              const int32_t* group_end, bool spread) {
  constexpr int TBM = 256, TBN = 256;
  const int n_tiles_n = (N + TBN - 1) / TBN;
  const int64_t max_tiles = ((m_rows_max + TBM - 1) / TBM + E) * n_tiles_n;
  int grid = smoe_num_cus() & ~7;
  if (grid < 8) grid = 8;
  if (max_tiles < grid) grid = (int)((max_tiles + 7) & ~(int64_t)7);
  const size_t lds = 160 * 1024;
  if (spread) {
    SMOE_ENSURE_SMEM((grouped_gemm_w4<AB, OT, true>));
    hipLaunchKernelGGL((grouped_gemm_w4<AB, OT, true>), dim3(grid), dim3(256), lds, s, (const AB*)A, (const AB*)W, bias, offsets,
                       group_expert, E, K, N, epilogue, (OT*)out, n_tiles_n, 4, a_gather, a_div, group_end);
  } else {
    SMOE_ENSURE_SMEM((grouped_gemm_w4<AB, OT, false>));
    hipLaunchKernelGGL((grouped_gemm_w4<AB, OT, false>), dim3(grid), dim3(256), lds, s, (const AB*)A, (const AB*)W, bias, offsets,
                       group_expert, E, K, N, epilogue, (OT*)out, n_tiles_n, 4, a_gather, a_div, group_end);
  }
  SMOE_CHECK_LAUNCH("smoe_grouped_gemm/w4");
  return 0;
}
