// Persistent form of the 8-wave ping-pong grouped GEMM (included by gemm.hip behind grouped_gemm_pp256; uses its helpers).
//
// One workgroup per CU walks tiles  slot, slot + G, slot + 2 G, ...  (G = grid size; inside a round the slots are
// XCD-contiguous: workgroups that share an XCD take neighbouring tiles = shared operand panels in its L2).  What the
// loop buys over one workgroup per tile:
//   * the NEXT tile is located, its (gathered) row addresses are loaded and its first K-tile is on its way into LDS
//     buffer 0 BEFORE the current tile's epilogue starts (the epilogue stages the output tile through buffer 1 and
//     the 16 KiB of LDS behind the two buffers), so the ~3 us of tile lookup + gather-address latency + first operand
//     fetch that opened every workgroup now run under the GELU / convert / store work;
//   * no workgroup launch, LDS allocation and kernel-argument fetch per tile;
//   * the epilogue's passes are separated by raw barriers (LDS ordering only): a pass no longer waits for the
//     previous pass's global stores to be acknowledged (__syncthreads() drains the vector-memory counter).
// Ordering of the operand DMA against the stores: vmcnt counts in issue order, so a counted wait that covers K-tile 0
// of the next tile would also cover every store issued after it.  Instead each wave drains vmcnt(0) right before
// the barrier of the LAST epilogue pass -- K-tile 0 was issued a whole epilogue earlier and has landed, the earlier
// passes' stores are one GELU phase old -- and the last pass's stores then leave with nothing waiting on them: the
// main loop's first counted wait comes a full K-tile (8 intervals) later.
//
// Main loop, LDS layout, fragment roles and both DMA schedules (DEEP = half-organised regions, two half-tiles in flight
// across the tile boundary) are those of grouped_gemm_pp256; results are bit-identical to it (same accumulation order).
#ifndef PS_STORE_AUX
#define PS_STORE_AUX 0   // cache-policy bits of the direct epilogue's stores (diagnostic builds: -DPS_STORE_AUX=2 = nt)
#endif
#ifdef SMOE_DIAG
// diagnostic build only: s_memtime stamps of wave 0 / lane 0 of every workgroup, 16 stamps per tile, first 12 tiles
__device__ unsigned long long smoe_diag_stamps[256 * 12 * 16];
__device__ int smoe_diag_flags;   // bit 0: the main loop issues no operand DMA (garbage results: what do MFMA + LDS reads alone cost?)
#define PS_NODMA (diag_nodma)
// bit 1: the main loop's interval barriers are skipped (garbage results: what do the ping-pong hand-overs cost?)
#define PS_IBAR() do { if (!diag_nobar) PP_BARRIER(); } while (0)
#define PS_STAMP(i)                                                                                              \
  do {                                                                                                           \
    if (wave == 0 && lane == 0 && tile_no < 12 && blockIdx.x < 256)                                              \
      smoe_diag_stamps[(blockIdx.x * 12 + tile_no) * 16 + (i)] = __builtin_amdgcn_s_memtime();                   \
  } while (0)
#else
#define PS_STAMP(i) do {} while (0)
#define PS_NODMA false
#define PS_IBAR() PP_BARRIER()
#endif

#ifdef SMOE_CLOCK
// clock-probe build only (libslimmoe_hip_clock.so, `make clock`; MI355X_MICROARCH.md 'DVFS give-back' item 6): wave 0 / lane 0 of
// every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) once before its first tile and once behind its
// last; the clock the chip held over the launch is d(memtime) / d(memrealtime) x 100 MHz.  The stamps go to a buffer of their own
// that nothing else reads; no output depends on them.  The production library executes no stamp.
__device__ unsigned long long smoe_clock_stamps[1024 * 8];   // per workgroup: 4 stamps + (fused launch) spins, GEMM-1 / GEMM-2 tiles, cycles in GEMM-2 runs
#define PS_CLOCK(i)                                                                                             \
  do {                                                                                                          \
    if (wave == 0 && lane == 0 && blockIdx.x < 1024) {                                                          \
      unsigned long long t_sh, t_rt;                                                                            \
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_sh), "=s"(t_rt)::"memory"); \
      smoe_clock_stamps[blockIdx.x * 8 + (i)] = t_sh;                                                           \
      smoe_clock_stamps[blockIdx.x * 8 + (i) + 1] = t_rt;                                                       \
    }                                                                                                           \
  } while (0)
#define PS_COUNT(slot, n)                                                                                       \
  do {                                                                                                          \
    if (wave == 0 && lane == 0 && blockIdx.x < 1024) smoe_clock_stamps[blockIdx.x * 8 + (slot)] += (n);          \
  } while (0)
#else
#define PS_CLOCK(i) do {} while (0)
#define PS_COUNT(slot, n) do {} while (0)
#endif

// KEEP (training forward of the first expert linear): the epilogue stores BOTH the pre-activation H = A W^T + b and gelu(H) --
// the backward needs H for gelu' and gelu(H) as the operand of the second linear's weight gradient; a separate GELU pass over
// H was 620 MB of HBM traffic (110 us at ViT-B).  `residual` carries the address that RECEIVES H; no row map / residual fusion.
// Its own instantiation (320-row tile, 5 staging passes of 2 x 64 rows): the inference kernels' registers are untouched.
//
// DIRECT (16-bit outputs without row map / residual: GEMM-1, qkv, patch embedding): the output tile does not pass through
// LDS at all.  The swapped MFMA leaves every lane with 4 consecutive columns of one row per 16 x 16 fragment; after the
// conversion two v_permlane16_swap per pair of neighbouring fragments give it 8 consecutive columns = one 16-byte store
// (a wave-instruction writes a 64-byte segment of 16 rows; the next one completes their 128-byte lines).  The stores are
// asynchronous and leave between the GELU arithmetic of the following fragments, so the staged form's store phase (two
// LDS read + store sweeps of ~4.5 k cycles per tile behind two barriers) overlaps the VALU phase, no epilogue barrier is left
// besides the one that publishes the next tile's first K-tile, and -- the staging region being unused -- the next tile's
// K-tile 1 streams under the epilogue as well.  Rows past the group's end are dropped by the buffer descriptor's range check
// (num_records = the rows of this tile that exist), so every wave issues the same number of stores and the wait that
// publishes K-tile 0 can be a counted one (the stores are younger than the operand pieces).  Same arithmetic per element
// as the staged epilogue: bit-identical results.
// ---- the fused expert-FFN launch (expert_ffn_fused below): GEMM-1 and GEMM-2 tiles of one MoE layer in ONE persistent launch ----
// Separate launches end in a partly filled round of workgroups each (ViT-B, 256 images: 1,920 GEMM-1 tiles = 7.5 rounds on 256
// CUs, 480 GEMM-2 tiles = 1.9 rounds: ~7 % of both kernels is tail).  Here every workgroup computes tiles of BOTH GEMMs; a
// GEMM-2 tile (m-tile i, any n) may start once the 12 (= h / 256) GEMM-1 tiles of m-tile i have stored their rows of H, which a
// per-m-tile counter says.
// ONE LIST PER XCD: the workgroups with blockIdx % 8 == k (they share an XCD: an affinity label, nothing depends on it) own the
// m-tiles [total_mt k / 8, total_mt (k + 1) / 8) -- their GEMM-1 AND GEMM-2 tiles -- so tiles that share operand panels meet in
// ONE L2 and H is consumed on the XCD that produced it.  The list is cut into SLOTS, two per workgroup, each a STATIC tile sequence
// (closed form of slot number and list geometry, fused_plan):
//   stage 1, slot w:  GEMM-1 tiles  w, w + G, ..., w + (x - 1) G              (G = workgroups per list; x G = A tiles)
//   stage 2, slot w:  w <  B2: GEMM-2 tile w (the early batch: its producers lie in [0, A)),
//                     w >= B2: the GEMM-1 tiles left over, (w - B2) + j (G - B2) behind A;
//                     then the remaining GEMM-2 tiles  B2 + w + j G.
// x balances the two kinds of stage-2 slots (x c1 + (q2 + 1) c2 against y c1 + q2 c2): every CU ends within about one GEMM-1
// tile of the others, and neither GEMM has a partly filled round of its own.
// Slots are CLAIMED through two tickets per list (one atomic each per workgroup and stage -- not per tile: a per-tile ticket put
// an atomic's latency into the in-order vmcnt queue in front of every tile's first operand wait, +4 us per GEMM-1 tile, and bound
// tiles to workgroups two tiles ahead of time; profiles/r04_fused_ffn.md).  Stage-2 slots are claimed light-first (the slots
// with GEMM-1 work), and a workgroup starts a stage-2 slot only after it has seen the stage-1 ticket exhausted (it computes any
// stage-1 slot it still gets).  No deadlock, whatever the number of resident workgroups: a GEMM-2 tile waits only for GEMM-1
// tiles of slots that were claimed before its own slot could start -- claimed by RUNNING workgroups, which reach those tiles
// without waiting for anything; workgroups that start late (another launch holds their CU) find the slots nobody claimed.
// A bounded spin (FUSED_SPIN_LIMIT polls) turns a broken counter into a wrong result plus an error word, not a hung GPU.
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility / cdna_hip_programming.md G16): the producer stores H
// write-through (sc1), every storing wave drains vmcnt(0), a workgroup barrier, then ONE lane adds to the m-tile's counter
// (agent scope); the consumer polls the counter with sc1 loads, then an agent-scope acquire (buffer_inv sc1), then its
// plain LDS-DMA loads -- every wave for itself, so no further barrier is needed.  Co-location is never relied on.
constexpr int FUSED_WS_HDR = 32;            // ws[k] / ws[8 + k]: stage-1 / stage-2 ticket of list k, [16] workgroups that left, [17] error word
constexpr int FUSED_WS_LEFT = 16, FUSED_WS_ERR = 17;
constexpr int FUSED_SPIN_LIMIT = 1 << 22;   // x s_sleep(16) ~ 1 k cycles each: seconds, far beyond any legitimate wait
struct FusedCtl {
  int32_t* ws;          // tickets, [FUSED_WS_HDR + mt] = GEMM-1 tiles of m-tile mt done
  int mt_lo, cnt;       // this workgroup's list: m-tiles [mt_lo, mt_lo + cnt)
  int n1, n2, A, B2, G; // its GEMM-1 / GEMM-2 tile counts, the plan (A = x G), workgroups per list
  int ntn1, ntn2, gm;   // n-tiles of the two GEMMs (ntn1 = the counter value that releases an m-tile), m-tiles per numbering group
  int stage, slot, j;   // the slot being walked: 1 = stage 1; 2 = stage 2, first part; 3 = stage 2, remaining GEMM-2 tiles
  int nk, nid;          // the next unprocessed tile: kind (1 = GEMM-1, 2 = GEMM-2, 0 = the slot is finished), index in its GEMM's order
};
// advance (nk, nid) to the tile behind the current one of the slot
__device__ __forceinline__ void fused_advance(FusedCtl& fc) {
  if (fc.stage == 1) {
    const int id = fc.slot + fc.G * fc.j;
    if (id < fc.A) { fc.nk = 1; fc.nid = id; ++fc.j; return; }
    fc.nk = 0;
    return;
  }
  if (fc.stage == 2) {
    if (fc.slot < fc.B2) {
      if (fc.j == 0) { fc.nk = 2; fc.nid = fc.slot; fc.j = 1; return; }
    } else {
      const int id = fc.A + (fc.slot - fc.B2) + (fc.G - fc.B2) * fc.j;
      if (id < fc.n1) { fc.nk = 1; fc.nid = id; ++fc.j; return; }
    }
    fc.stage = 3;
    fc.j = 0;
  }
  const int f = fc.B2 + fc.slot + fc.G * fc.j;
  if (f < fc.n2) { fc.nk = 2; fc.nid = f; ++fc.j; return; }
  fc.nk = 0;
}

template <typename AB, typename OT, int AFR, bool DEEP, bool KEEP, bool DIRECT, bool BUF, int FUSED>
__device__ __forceinline__ void ps_body(
    const AB* __restrict__ A, const AB* __restrict__ W, const float* __restrict__ bias,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ group_expert, int E, int K, int N, int epilogue,
    const int64_t* __restrict__ row_map, const float* __restrict__ row_scale, const OT* residual, OT* out, int n_tiles_n,
    int group_m, const int64_t* __restrict__ a_gather, int a_div, int gather_len, const int32_t* __restrict__ group_end,
    FusedCtl* fc) {
  static_assert(sizeof(AB) == 2, "16-bit operands");
  static_assert(FUSED == 0 || (!KEEP && ((FUSED == 1 && DIRECT) || (FUSED == 2 && BUF))), "fused roles: 1 = producer (direct 16-bit stores), 2 = consumer");
  static_assert(!DIRECT || (sizeof(OT) == 2 && !KEEP), "the direct epilogue stores 16-bit outputs, one output per tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TBM = 64 * AFR, TBN = 256, NT = 512, NW = 8;
  constexpr int STAGE = (TBM + TBN) * BK_BYTES;  // 64 KiB (72 KiB for the 320-row tile)
  constexpr int ASLOTS = TBM / 8 / NW;           // 1-KiB DMA pieces per wave per K-tile: A (= AFR)
  constexpr int SLOTS = 4;                       //                                        W
  constexpr int LDS_TOTAL = 160 * 1024;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = K / 64;
#ifdef SMOE_DIAG
  const bool diag_nodma = (__builtin_amdgcn_readfirstlane(smoe_diag_flags) & 1) != 0;
  const bool diag_nobar = (__builtin_amdgcn_readfirstlane(smoe_diag_flags) & 2) != 0;
#endif

  // ---- tile enumeration ------------------------------------------------------------------------------------------------
  // Row-group table in two registers (E <= 63 groups, the launcher checks): lane l holds offsets[l] and the number of
  // m-tiles in front of group l (lane E: offsets[E] and the total).  Locating a tile is then a ballot, a count-leading-
  // zeros and three v_readlane -- no memory access; the scalar scan over `offsets` it replaces cost ~2,300 cycles per
  // tile (stamps: tools/gemm_stamps.py).
  // group_end != NULL (the padded exchange buffers of a capacity gate): group l is the rows [offsets[l], group_end[l]) and
  // `offsets` has E entries; otherwise the groups are contiguous, [offsets[l], offsets[l + 1]).
  int t_off = 0, t_end = 0, t_tb = 0, t_ge = lane;
  {
    const int li = lane < E ? lane : (group_end ? E - 1 : E);
    t_off = offsets[li];
    t_end = group_end ? group_end[li] : offsets[li < E ? li + 1 : E];
    int cnt = lane < E ? (t_end - t_off + TBM - 1) / TBM : 0;
    int incl = cnt;                                 // inclusive scan over the wave
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const int up = __shfl_up(incl, sft, 64);
      if (lane >= sft) incl += up;
    }
    t_tb = incl - cnt;                              // exclusive prefix; lanes >= E hold the total
    if (group_expert && lane < E) t_ge = group_expert[lane];
  }
  const int total_mt = __builtin_amdgcn_readlane(t_tb, 63);
  // Tile order.  group_m > 0: tiles are numbered in groups of group_m m-tiles x all n-tiles; in a round the 8 XCDs take 8
  // neighbouring runs of per_xcd tiles and the whole grid moves on by G.
  // group_m < 0 (-(quad | n_block << 8)): n-block-major numbering -- for each block of n_block n-tiles, for each quad of m-tiles,
  // quad x n_block tiles -- and every XCD owns ONE CONTIGUOUS eighth of that order, walked per_xcd tiles at a time: round after
  // round its workgroups meet the same n_block weight panels (until the expert changes) and only the A panels move on.
  // FUSED: the tile comes from the launch's list (FusedCtl); numbering = groups of gm m-tiles x all n-tiles without padding (the
  // last group holds the m-tiles that are left), so that the list positions are exactly the tiles that exist.
  const bool xcd_runs = group_m < 0;
  const int gm = FUSED ? fc->gm : (xcd_runs ? ((-group_m) & 0xff) : group_m);
  const int n_block = xcd_runs ? ((-group_m) >> 8) : n_tiles_n;          // the launcher makes it a divisor of n_tiles_n
  const int per_group = gm * n_block;
  const int tiles_per_nb = ((total_mt + gm - 1) / gm) * per_group;
  const int n_tiles = tiles_per_nb * (n_tiles_n / n_block);
  const int G = gridDim.x, per_xcd = G >> 3;     // the launcher keeps G a multiple of 8
  const int xcd_share = (n_tiles + 7) >> 3;
  const int t_step = xcd_runs ? per_xcd : G;
  const int xcd_end = ((int)(blockIdx.x & 7) + 1) * xcd_share;
  const int t_last = xcd_runs ? (xcd_end < n_tiles ? xcd_end : n_tiles) : n_tiles;
  int tile = (blockIdx.x & 7) * (xcd_runs ? xcd_share : per_xcd) + (blockIdx.x >> 3);
  int e = 0, m0 = 0, m_end = 0, n0 = 0;          // the tile whose operands are being set up / streamed
  int mt_next = 0;                               // FUSED: its m-tile (the index of its ready counter)
  auto locate = [&](int mt) {                    // (e, m0, m_end) of m-tile mt
    // groups in front of or holding m-tile mt form a lane prefix; the owner is its last lane (an empty group has the
    // same prefix count as its successor, so it is never last)
    const unsigned long long msk = __ballot(lane < E && t_tb <= mt);
    const int gi = 63 - __builtin_clzll(msk);
    m0 = __builtin_amdgcn_readlane(t_off, gi) + (mt - __builtin_amdgcn_readlane(t_tb, gi)) * TBM;
    m_end = __builtin_amdgcn_readlane(t_end, gi);
    e = __builtin_amdgcn_readlane(t_ge, gi);
  };
  auto locate_fused = [&](int id) {              // tile `id` of THIS GEMM's order within the list -> (e, m0, m_end, n0, mt_next)
    const int g = id / per_group, rem = id % per_group;   // numbering: groups of gm m-tiles x all n-tiles, no padding
    int gme = fc->cnt - g * gm;
    if (gme > gm) gme = gm;
    const int mt = fc->mt_lo + g * gm + rem % gme;
    locate(mt);
    n0 = (rem / gme) * TBN;
    mt_next = mt;
  };
  auto advance = [&]() -> bool {                 // first existing tile at or after `tile` on this workgroup's stride
    while (tile < t_last) {
      const int nb = tile / tiles_per_nb, r = tile % tiles_per_nb;
      const int g = r / per_group, rem = r % per_group;
      const int mt = g * gm + rem % gm;
      if (mt < total_mt) {
        locate(mt);
        n0 = (nb * n_block + rem / gm) * TBN;
        return true;
      }
      tile += t_step;
    }
    return false;
  };
  // ---- fused launch: readiness, completion --------------------------------------------------------------------------------
  int polled = 0;         // consumer: the next tile's ready counter as read at the top of the current tile
  int mt_done = -1;       // producer: m-tile of the tile whose stores are still draining (signalled one tile later)
  auto fused_signal = [&](int mt) {   // every wave has drained its stores of that tile and passed a barrier since
    if constexpr (FUSED == 1) {
      if (wave == 0 && lane == 0)
        __hip_atomic_fetch_add(fc->ws + FUSED_WS_HDR + mt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto fused_poll = [&](int mt) -> int {
    return __hip_atomic_load(fc->ws + FUSED_WS_HDR + mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto fused_wait = [&](int mt, int seen) {   // until all GEMM-1 tiles of m-tile mt have stored; then this wave may load H
    if constexpr (FUSED == 2) {
      int v = __builtin_amdgcn_readfirstlane(seen);
      int spins = 0;
      while (v < fc->ntn1) {
        __builtin_amdgcn_s_sleep(16);
        v = __builtin_amdgcn_readfirstlane(fused_poll(mt));
        if (++spins > FUSED_SPIN_LIMIT) {        // never in a healthy launch: report, then go on (wrong rows, no hang)
          if (lane == 0) fc->ws[FUSED_WS_ERR] = 1;
          break;
        }
      }
      PS_COUNT(4, spins);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  };

  // ---- operand source pointers of the current (e, m0, m_end, n0) -----------------------------------------------------
  constexpr int A_HALF_PIECES = TBM / 16;                      // 8-row DMA pieces per A-half (16 / 20)
  constexpr int A_HS = (A_HALF_PIECES + NW - 1) / NW;          // pieces per wave per A-half, rounded up (2 / 3)
  // 32-bit BYTE offsets from A / W (the launcher guarantees both operands span < 4 GiB): half the registers of
  // per-lane 64-bit pointers -- the 320-row tile has none to spare -- and the DMA takes (uniform base, 32-bit offset)
  uint32_t a_src[DEEP ? 2 * A_HS : ASLOTS];
  uint32_t w_src[SLOTS];
  const char* const Ab = reinterpret_cast<const char*>(A);
  const char* const Wb = reinterpret_cast<const char*>(W);
  // `lo` = the lane id behind a per-call opaque zero: everything derived from it (per-slot swizzle constants, row
  // numbers) is recomputed per tile instead of being hoisted out of the tile loop and kept in registers through the
  // main loop, where there are none to spare
  const int* const rowbuf = reinterpret_cast<const int*>(smem + LDS_TOTAL - 2048 - TBM * 4);
  bool rows_in_lds = false;               // true once a tile's gather rows were prefetched into rowbuf
  auto gathered_row = [&](int gr) -> int64_t {   // source row of tile row gr (already clamped to the group's range)
    if (!a_gather) return gr;
    const int64_t v = rows_in_lds ? (int64_t)rowbuf[gr - m0] : a_gather[gr];
    return a_div == 1 ? v : (int64_t)((uint32_t)v / (uint32_t)a_div);
  };
  auto setup = [&](int lo) {
    const int l_row = lo >> 3, l_pos = lo & 7;
    if constexpr (!DEEP) {
#pragma unroll
      for (int s = 0; s < ASLOTS; ++s) {
        const int r = 8 * (s * NW + wave) + l_row;
        int gr = m0 + r;
        if (gr >= m_end) gr = m_end - 1;
        const int64_t arow = gathered_row(gr);   // fused MOEScatter
        a_src[s] = (uint32_t)((arow * K + ((l_pos ^ ((r >> 1) & 7)) << 3)) * 2);
      }
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        const int r = 8 * (s * NW + wave) + l_row;
        int gw = n0 + r;
        if (gw >= N) gw = N - 1;
        w_src[s] = (uint32_t)((((int64_t)e * N + gw) * K + ((l_pos ^ ((r >> 1) & 7)) << 3)) * 2);
      }
    } else {
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
          const int64_t arow = gathered_row(gr);
          a_src[h * A_HS + s2] = (uint32_t)((arow * K + ((l_pos ^ ((r >> 1) & 7)) << 3)) * 2);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int r = (wave + NW * s2) * 8 + l_row;            // row inside the half (0..127)
          int gw = n0 + (r / 32) * 64 + h * 32 + r % 32;
          if (gw >= N) gw = N - 1;
          w_src[h * 2 + s2] = (uint32_t)((((int64_t)e * N + gw) * K + ((l_pos ^ ((r >> 1) & 7)) << 3)) * 2);
        }
      }
    }
  };
#define PS_DMA(SRC, DST)                                                                             \
  do {                                                                                               \
    if (!PS_NODMA)                                                                                   \
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),         \
                                       (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);    \
  } while (0)
  // non-DEEP pieces: "lo" (s0 == 0) = slots 0,1 (tile rows 0-127, read by wave group 0 only), "hi" = the rest
  auto dma_a = [&](int kt, int buf, int s0) {
    char* sa = smem + buf * STAGE;
    if (s0 == 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s) PS_DMA(Ab + (a_src[s] + (uint32_t)kt * 128u), sa + (s * NW + wave) * 1024);
    } else {
#pragma unroll
      for (int s = 2; s < ASLOTS; ++s) PS_DMA(Ab + (a_src[s] + (uint32_t)kt * 128u), sa + (s * NW + wave) * 1024);
    }
  };
  auto dma_w = [&](int kt, int buf, int s0) {
    char* sw = smem + buf * STAGE + TBM * BK_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) PS_DMA(Wb + (w_src[s0 + s] + (uint32_t)kt * 128u), sw + ((s0 + s) * NW + wave) * 1024);
  };
  // DEEP half-tiles
  auto dma_ah = [&](int kt, int buf, int h) {
    char* sa = smem + buf * STAGE + h * (TBM / 2) * BK_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < A_HS; ++s2) {
      if (A_HALF_PIECES % NW == 0 || s2 + 1 < A_HS || wave < A_HALF_PIECES % NW)
        PS_DMA(Ab + (a_src[h * A_HS + s2] + (uint32_t)kt * 128u), sa + (wave + NW * s2) * 1024);
    }
  };
  auto dma_wh = [&](int kt, int buf, int h) {
    char* sw = smem + buf * STAGE + (TBM + h * 128) * BK_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) PS_DMA(Wb + (w_src[h * 2 + s2] + (uint32_t)kt * 128u), sw + (wave + NW * s2) * 1024);
  };
  auto wait_keep2 = [&]() {  // leaves this wave's two newest half-tiles (one A-half, one B-half) in flight
    if constexpr (A_HALF_PIECES % NW == 0) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      if (wave < A_HALF_PIECES % NW) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
  };
  auto issue_kt0 = [&]() {   // the whole first K-tile of the tile just set up -> buffer 0
    if constexpr (DEEP) { dma_ah(0, 0, 0); dma_wh(0, 0, 0); dma_ah(0, 0, 1); dma_wh(0, 0, 1); }
    else { dma_a(0, 0, 0); dma_a(0, 0, 2); dma_w(0, 0, 0); dma_w(0, 0, 2); }
  };
  auto issue_kt1 = [&]() {   // what the main loop expects in flight for K-tile 1 at its head -> buffer 1
    if (nk > 1) {
      if constexpr (DEEP) { dma_ah(1, 1, 0); dma_wh(1, 1, 1); }
      else { dma_a(1, 1, 0); dma_a(1, 1, 2); dma_w(1, 1, 0); dma_w(1, 1, 2); }
    }
  };

  f32x4 acc[2 * AFR][4];
  u32x4 ar[AFR][2], br[2][2];  // current A-half (AFR row fragments x 2 k-steps), B-half (2 col fragments x 2 k-steps)
  auto read_a = [&](int buf, int half) {
    const char* sa = smem + buf * STAGE + (DEEP ? half * (TBM / 2) + wr * (16 * AFR) : wr * (TBM / 2) + half * (16 * AFR)) * BK_BYTES;
#pragma unroll
    for (int i = 0; i < AFR; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) ar[i][kk] = *reinterpret_cast<const u32x4*>(sa + swz(i * 16 + fr, kk * 4 + fq));
  };
  auto read_b = [&](int buf, int half) {
    const char* sw = smem + buf * STAGE + (TBM + (DEEP ? half * 128 + wc * 32 : wc * 64 + half * 32)) * BK_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) br[i][kk] = *reinterpret_cast<const u32x4*>(sw + swz(i * 16 + fr, kk * 4 + fq));
  };
#define PS_MFMA(AH, BH)                                                                                              \
  do {                                                                                                               \
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

  // ---- epilogue geometry (as grouped_gemm_pp256; the staging region is buffer 1 + the LDS behind the buffers) -----------
  constexpr int TM = TBM / 2, TN = 64, MI = 2 * AFR, NI = 4;
  constexpr int OB = OutPack<OT>::bytes;
  constexpr int C_STRIDE = TBN * OB + C_PAD;
  constexpr int BIAS_BYTES = 2048;        // two 1-KiB bias tiles (256 f32) at the top of the LDS, alternating per tile
  constexpr int ROWBUF_BYTES = TBM * 4;   // below them: the next tile's gathered row numbers (i32), prefetched by DMA
  constexpr int EPI_BYTES = LDS_TOTAL - STAGE - BIAS_BYTES - ROWBUF_BYTES;
  static_assert(!KEEP || (AFR == 5 && !DEEP && OB == 2), "the two-output epilogue is instantiated for the 320-row tile, 16-bit outputs");
  constexpr int NPASS = KEEP ? 5 : ((TBM * C_STRIDE <= EPI_BYTES) ? 1 : ((TBM / 2) * C_STRIDE <= EPI_BYTES ? 2 : (AFR == 4 ? 4 : 5)));
  constexpr int RP = TBM / NPASS;
  static_assert((KEEP ? 2 : 1) * RP * C_STRIDE <= EPI_BYTES, "epilogue pass does not fit behind buffer 0");
  constexpr int CHUNKS = TBN * OB / 16;   // 16-B chunks per tile row
  constexpr int TPR = 16;                 // threads per output row: 16 consecutive threads = 256 contiguous bytes
  constexpr int CPT = CHUNKS / TPR;       // chunks per thread per row (strided by 256 B)
  constexpr int ROWS_PER_IT = NT / TPR;   // 32
  constexpr int ITS = RP / ROWS_PER_IT;
  static_assert(CHUNKS % TPR == 0 && RP % ROWS_PER_IT == 0, "epilogue thread map");
  constexpr int HALF = RP / 2;
  constexpr int MPP = HALF / 16;
  static_assert(HALF % 16 == 0 && MPP * NPASS == MI, "epilogue pass split");
  char* const cst = smem + STAGE;         // output staging
  char* const cst2 = cst + RP * C_STRIDE; // KEEP: the pre-activations of the same rows
  // Row-mapped stores (the fused combine): the tile's output rows and combine scales are resolved ONCE per tile into LDS behind
  // the staging rows, where there is room for them -- resolved per pass, the chain row_map -> row_scale -> residual put two
  // exposed global-load latencies in front of every pass's stores.
  constexpr bool OMAP = (KEEP ? 2 : 1) * RP * C_STRIDE + 2 * TBM * 4 <= EPI_BYTES;
  int* const omap = reinterpret_cast<int*>(cst + (KEEP ? 2 : 1) * RP * C_STRIDE);
  float* const oscl = reinterpret_cast<float*>(omap + TBM);
  const OT* const resid = KEEP ? nullptr : residual;
  OT* const pre_out = KEEP ? const_cast<OT*>(residual) : nullptr;
  // The tile's bias row reaches the epilogue through LDS by DMA (one 1-KiB piece, wave 0), issued with the tile's first
  // operand pieces: an ordinary load in the epilogue would make hipcc drain the whole vector-memory queue -- the next
  // tile's operand DMAs included -- at its first use (cdna_hip_programming.md section 5, "Pipelining across barriers").
  char* const bias_lds = smem + LDS_TOTAL - BIAS_BYTES;
  int bias_par = 0;
  // the NEXT tile's gather rows: low words of a_gather[min(m0 + r, m_end - 1)], 64 rows per DMA piece (waves 0 .. TBM/64-1),
  // issued at the head of the current tile's main loop and read by setup() a whole main loop later
  auto prefetch_rows = [&]() {
    if (a_gather && wave < TBM / 64) {
      int gr = m0 + wave * 64 + lane;
      if (gr >= m_end) gr = m_end - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_gather + gr),
                                       (__attribute__((address_space(3))) void*)(smem + LDS_TOTAL - 2048 - TBM * 4 + wave * 256),
                                       4, 0, 0);
    }
  };
  auto issue_bias = [&](int par) {
    if (bias && wave == 0) {
      int col = n0 + lane * 4;
      if (col > N - 4) col = N - 4;       // columns past N are never stored
      PS_DMA(bias + (int64_t)e * N + col, bias_lds + par * 1024);
    }
  };

  if constexpr (FUSED == 0) {
    if (!advance()) return;
  } else {
    locate_fused(fc->nid);                                              // the caller made sure (nk, nid) is a tile of this GEMM
    if constexpr (FUSED == 2) fused_wait(mt_next, fused_poll(mt_next));   // before the first operand fetch from H
  }
  if constexpr (FUSED == 0) PS_CLOCK(0);
  {
    int oz = 0;
    asm volatile("" : "+v"(oz));
    setup(lane + oz);
  }
  // ---- prologue of this workgroup's first tile ------------------------------------------------------------------------
  issue_bias(0);
  issue_kt0();
  if (nk > 1) {
    issue_kt1();
    if constexpr (DEEP) wait_keep2();
    else if constexpr (AFR == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 1's pieces may stay in flight
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  PP_BARRIER();

  int tile_no = 0;
  (void)tile_no;
  int cmt = 0;                                              // FUSED: m-tile of the tile computed in this iteration
  for (;; ++tile_no) {
    PS_STAMP(0);
    // ---- look ahead: where this workgroup goes next; that tile's gather rows start streaming into LDS now ----------------
    const int ce = e, cm0 = m0, cm_end = m_end, cn0 = n0;   // the tile computed in this iteration
    cmt = mt_next;
    if constexpr (FUSED != 0) PS_COUNT(4 + FUSED, 1);
    bool more;
    if constexpr (FUSED == 0) {
      tile += t_step;
      more = advance();                                     // (e, m0, m_end, n0) = the next tile from here on
    } else {
      fused_advance(*fc);      // (nk, nid) = the slot's tile behind the one computed in this iteration
      more = fc->nk == FUSED;  // a tile of the other GEMM, or the end of the slot, ends this run
      if (more) {
        locate_fused(fc->nid);
        if constexpr (FUSED == 2) polled = fused_poll(mt_next);   // usually released long ago: checked at the tile boundary
      }
    }
    if (more) prefetch_rows();
#pragma unroll
    for (int i = 0; i < 2 * AFR; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wr == 1) PP_BARRIER();  // stagger: group 1 runs one interval behind group 0
    if constexpr (DEEP) {
      // Invariant at the head of tile t: tile t is in LDS; A-half 0 and B-half 1 of tile t+1 are in flight.
      for (int t = 0; t < nk; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const bool n1 = (t + 1 < nk), n2 = (t + 2 < nk);
        read_b(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(cur, 0);
        if (n1) dma_ah(t + 1, nxt, 1);
        PS_IBAR();
        PS_MFMA(0, 0);
        PS_IBAR();
        read_b(cur, 1);
        if (n1) dma_wh(t + 1, nxt, 0);
        PS_IBAR();
        PS_MFMA(0, 1);
        PS_IBAR();
        read_a(cur, 1);
        if (n2) dma_ah(t + 2, cur, 0);
        PS_IBAR();
        PS_MFMA(1, 1);
        PS_IBAR();
        read_b(cur, 0);
        if (n2) dma_wh(t + 2, cur, 1);
        if (wr == 1) {  // group 1: its program interval 6 is global interval 8t+7, the last one of tile t
          if (n2) wait_keep2();
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PS_IBAR();
        PS_MFMA(1, 0);
        if (wr == 0) {  // group 0: program interval 7 = global 8t+7
          if (n2) wait_keep2();
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PS_IBAR();
      }
    } else {
      // R1(t): A rows 128.. of tile t+1   R2(t): W rows 0-127 of t+1   R3(t): W rows 128-255 of t+1
      // R4(t): A rows 0-127 of tile t+2 (that region of tile t's buffer was last read in R3(t))
      for (int t = 0; t < nk; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        const bool pre1 = (t >= 1) && (t + 1 < nk);
        const bool pre2 = (t + 2 < nk);
        read_b(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(cur, 0);
        if (pre1) dma_a(t + 1, nxt, 2);
        PS_IBAR();
        PS_MFMA(0, 0);
        PS_IBAR();
        read_b(cur, 1);
        if (pre1) dma_w(t + 1, nxt, 0);
        PS_IBAR();
        PS_MFMA(0, 1);
        PS_IBAR();
        read_a(cur, 1);
        if (pre1) dma_w(t + 1, nxt, 2);
        PS_IBAR();
        PS_MFMA(1, 1);
        PS_IBAR();
        read_b(cur, 0);
        if (pre2) dma_a(t + 2, cur, 0);
        if (wr == 1) {
          if (pre2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PS_IBAR();
        PS_MFMA(1, 0);
        if (wr == 0) {
          if (pre2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PS_IBAR();
      }
    }
    if (wr == 0) PP_BARRIER();  // equalise barrier counts; after it every wave is done with the operand buffers
    PS_STAMP(1);
    if constexpr (FUSED != 0) {
      // every wave drained its vector-memory queue at the last K-tile (vmcnt(0)) and passed a barrier since: the PREVIOUS tile's
      // write-through stores of H are complete -> release its m-tile's count
      if (mt_done >= 0) fused_signal(mt_done);
    }

    // ---- tile boundary: the next tile's operands start streaming, then this tile's epilogue -----------------------------
    PS_STAMP(2);
    int oz = 0;
    asm volatile("" : "+v"(oz));          // per-tile opaque zero (see setup)
    const int tid_e = tid + oz, lane_e = tid_e & 63;
    if (more) {
      if constexpr (FUSED == 2) fused_wait(mt_next, polled);   // H rows of the next tile: all 12 producer tiles have stored
      rows_in_lds = a_gather != nullptr;
      setup(lane_e);  // the next tile's row numbers (from LDS) and operand offsets
      issue_bias(bias_par ^ 1);
      PS_STAMP(3);
      issue_kt0();    // its first K-tile streams into buffer 0 under the epilogue below
      // DIRECT: nothing is staged through buffer 1, so K-tile 1 could follow at once -- and did at first: the tile boundary then
      // pushes 144 KB of operand pieces + 160 KB of output through the CU's vector-memory path in one burst, ~17 k cycles at
      // ~20 B/clk whatever the epilogue's arithmetic costs (stamps: 2.1 k issue + 9.1 k epilogue + 6.6 k wait; the same sum with
      // the GELU switched off).  The plain schedule issues K-tile 1 BEHIND the stores instead (below); the deep schedule's
      // K-tile-1 pieces differ per wave (no exact count for the wait), so it keeps the early issue.
      if constexpr (DIRECT && DEEP) issue_kt1();
    }
    PS_STAMP(4);

    if constexpr (DIRECT) {
      // ---- direct epilogue of (ce, cm0, cm_end, cn0): registers -> global memory ------------------------------------------
      (void)ce;
      const int fr = lane_e & 15, fq = lane_e >> 4;
      f32x4 bv[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int nl = wc * 64 + ni * 16 + fq * 4;
        bv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias) bv[ni] = *reinterpret_cast<const f32x4*>(bias_lds + bias_par * 1024 + nl * 4);
      }
      bias_par ^= 1;
      // this tile's rows as a buffer: rows >= cm_end fall outside num_records and are dropped by the hardware
      const int rows_here = (cm_end - cm0 < TBM) ? (cm_end - cm0) : TBM;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char*>(out) + (int64_t)cm0 * N * 2, 0, rows_here * N * 2, 0x00020000);
      // after the swaps lane (fr, fq) holds columns  (2 q + (fq & 1)) * 16 + (fq >> 1) * 8 ... + 7  of fragment pair q
      const int cq = wc * 64 + (fq & 1) * 16 + (fq >> 1) * 8;
      const uint32_t OOR = 0x80000000u;                   // beyond any tile's num_records (320 rows x N x 2 B < 2 GiB)
      uint32_t off0 = (uint32_t)(((wr * (TBM / 2) + fr) * N + cn0 + cq) * 2);
      const bool ok0 = cn0 + cq < N, ok1 = cn0 + cq + 32 < N;   // N % 8 == 0: a lane's 8 columns exist together
      const uint32_t row16 = (uint32_t)(16 * N * 2);
#pragma unroll
      for (int mi = 0; mi < 2 * AFR; ++mi) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f32x4 va = acc[mi][2 * q] + bv[2 * q], vb = acc[mi][2 * q + 1] + bv[2 * q + 1];
          if (epilogue == SMOE_EPI_GELU) { va = gelu_fast4(va); vb = gelu_fast4(vb); }
          uint32_t a0, a1, b0, b1;
          pack4<OT>(va, a0, a1);
          pack4<OT>(vb, b0, b1);
          // odd 16-lane rows of the first operand <-> even rows of the second: (a, b) -> 8 consecutive columns per lane
          const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          const u32x4 v = u32x4{s0[0], s1[0], s0[1], s1[1]};
          const uint32_t off = off0 + (uint32_t)(q * 64);
          // FUSED producer: write-through (sc1 = aux 16), so that the consumer's XCD finds the rows in memory
#ifdef FUSED_EXP_PLAIN_STORES   // timing experiment only (clock build): H stored like the two-launch kernel stores it
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)((q ? ok1 : ok0) ? off : OOR), 0, PS_STORE_AUX);
#else
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)((q ? ok1 : ok0) ? off : OOR), 0, FUSED == 1 ? 16 : PS_STORE_AUX);
#endif
        }
        off0 += row16;
      }
      PS_STAMP(5);
      if (!more) break;
      // The next tile's K-tile 0 was issued BEFORE these stores: a counted wait that leaves the stores (and, on the plain
      // schedule, the K-tile-1 pieces issued behind them: AFR + 4 per wave) in flight publishes it without waiting for a store.
      if constexpr (DEEP) {
        if constexpr (AFR == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else {
        issue_kt1();
        if (nk > 1) {
          if constexpr (AFR == 5) asm volatile("s_waitcnt vmcnt(29)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        } else {
          if constexpr (AFR == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
      }
      PP_BARRIER();
      PS_STAMP(12);
      if constexpr (FUSED == 1) mt_done = cmt;   // its stores drain under the next tile's main loop
      continue;
    }

    // ---- epilogue of (ce, cm0, cm_end, cn0) in row passes through LDS ------------------------------------------------
    (void)ce;
    const int fr = lane_e & 15, fq = lane_e >> 4;          // shadow the main loop's copies: epilogue-only address math
    const int trow = tid_e / TPR, tcol = tid_e % TPR;
    f32x4 bv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nl = wc * TN + ni * 16 + fq * 4;
      bv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bias) bv[ni] = *reinterpret_cast<const f32x4*>(bias_lds + bias_par * 1024 + nl * 4);
    }
    bias_par ^= 1;
    const bool use_omap = OMAP && row_map != nullptr;   // workgroup-uniform
    if (use_omap && tid_e < TBM) {   // tile row tid_e -> (output row, scale); visible to everybody after pass 0's barrier
      const int m = cm0 + tid_e;
      int o = -1;
      float sc = 1.f;
      if (m < cm_end) {
        o = (int)row_map[m];
        if (row_scale) sc = row_scale[o];
      }
      omap[tid_e] = o;
      oscl[tid_e] = sc;
    }
    // BUF: `out` / `residual` as raw buffers over the first 2 GiB (the launcher checks that every output row ends below that).
    // Built per tile from opaque copies of the pointers: hoisted out of the tile loop, the two descriptors would sit in 8
    // scalar registers through the main loop, which has none to spare -- the spills reach the vector registers.
    constexpr uint32_t BUF_OOR = 0x80000000u;
    OT* out_t = out;
    const OT* res_t = resid ? resid : out;
    if constexpr (BUF) asm volatile("" : "+s"(out_t), "+s"(res_t));
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_t, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<OT*>(res_t), 0, 0x7FFFFFFF, 0x00020000);
    (void)rs_out; (void)rs_res; (void)BUF_OOR;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      // (1) resolve this pass's output rows (32-bit: a row index, not an address) and their combine scales
      int orow[ITS];
      float oscale[ITS];
      if (!use_omap) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int r = trow + it * ROWS_PER_IT;
          const int m = cm0 + (r / HALF) * TM + p * HALF + (r % HALF);
          orow[it] = -1;
          oscale[it] = 1.f;
          if (m < cm_end) {
            orow[it] = row_map ? (int)row_map[m] : m;
            if (row_map && row_scale) oscale[it] = row_scale[orow[it]];
          }
        }
      }
      if (use_omap && p > 0) {   // the table is visible since pass 0's barrier
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int r = trow + it * ROWS_PER_IT;
          const int tr = (r / HALF) * TM + p * HALF + (r % HALF);
          orow[it] = omap[tr];
          oscale[it] = oscl[tr];
        }
      }
      if constexpr (BUF) {
        // ---- buffer-addressed sweep: every residual load and output store is issued unconditionally (rows past the group
        // and columns past N get an offset beyond num_records, which the hardware drops), so hipcc's wait counts are exact:
        // the residual of a whole pass travels under the staging arithmetic and no use of it waits for a store.  With flat
        // accesses under `if (row valid)` the counts collapse to vmcnt(0) -- every row iteration then waited for the previous
        // iteration's stores to be acknowledged.  GEMM-2's tile boundary: 52.5 k -> 47.2 k cycles
        // (profiles/r03_gemm2_tile_stamps.txt); the rest is the CU's store path (~10 B/clk, profiles/r03_mainloop_floor.txt).
        uint32_t boff[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) boff[it] = BUF_OOR;   // (defined on every path: see `res`)
        auto resolve_off = [&]() {
#pragma unroll
          for (int it = 0; it < ITS; ++it)
            boff[it] = orow[it] >= 0 ? (uint32_t)((orow[it] * N + cn0 + tcol * (16 / OB)) * OB) : BUF_OOR;
        };
        auto off_of = [&](int it, int j) -> int {
          const int ncol = cn0 + (tcol + j * TPR) * (16 / OB);
          return (int)(ncol < N ? boff[it] + (uint32_t)(j * TPR * 16) : BUF_OOR);
        };
        // The residual loads are issued whether there is a residual or not (without one: out of range, answered with zeros
        // by the address unit): a branch around them would leave hipcc two paths to merge and its counts conservative again.
        // Zero-initialised: left undefined on some path, the registers count as live around the whole tile loop and spill.
        u32x4 res[ITS][CPT];
#pragma unroll
        for (int it = 0; it < ITS; ++it)
#pragma unroll
          for (int j = 0; j < CPT; ++j) res[it][j] = u32x4{0u, 0u, 0u, 0u};
        auto fetch = [&](int it0, int it1) {
#pragma unroll
          for (int it = it0; it < it1; ++it)
#pragma unroll
            for (int j = 0; j < CPT; ++j)
              res[it][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rs_res, resid ? off_of(it, j) : (int)BUF_OOR, 0, 0));
        };
        const bool rows_known = !(use_omap && p == 0);
        if (rows_known) resolve_off();
        // registers: the accumulators of the passes already stored are dead, 16 more per staged fragment row.  From pass 2
        // on the whole pass's residual is requested before the staging arithmetic; pass 1 has room for the first row
        // iteration only, pass 0 for none (the rest follows behind the staging, still in front of the barrier)
        const int n_early = p >= 2 ? ITS : (p == 1 ? 1 : 0);   // rows_known whenever p > 0
        fetch(0, n_early);
#pragma unroll
        for (int mm = 0; mm < MPP; ++mm) {
          const int mi = p * MPP + mm;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x4 v = acc[mi][ni] + bv[ni];
            const int nl = wc * TN + ni * 16 + fq * 4;
            if (epilogue == SMOE_EPI_GELU) v = gelu_fast4(v);
            OutPack<OT>::write4(cst + (wr * HALF + mm * 16 + fr) * C_STRIDE + nl * OB, v);
          }
        }
        if (rows_known) fetch(n_early, ITS);
        if (p == 0) PS_STAMP(5);
        if (p == NPASS - 1) PS_STAMP(8);
        if (p == NPASS - 1 && more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (p == NPASS - 1) PS_STAMP(9);
        PP_BARRIER();
        if (p == 0) PS_STAMP(6);
        if (!rows_known) {
#pragma unroll
          for (int it = 0; it < ITS; ++it) {
            const int r = trow + it * ROWS_PER_IT;
            const int tr = (r / HALF) * TM + p * HALF + (r % HALF);
            orow[it] = omap[tr];
            oscale[it] = oscl[tr];
          }
          resolve_off();
          fetch(0, ITS);
        }
        if (p == NPASS - 1) PS_STAMP(10);
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int r = trow + it * ROWS_PER_IT;
#pragma unroll
          for (int j = 0; j < CPT; ++j) {
            u32x4 v = *reinterpret_cast<const u32x4*>(cst + r * C_STRIDE + (tcol + j * TPR) * 16);
            if (row_map && row_scale) v = scale16<OT>(v, oscale[it]);
            if (resid) v = add16<OT>(res[it][j], v);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, off_of(it, j), 0, 0);
          }
        }
        if (p == 0) PS_STAMP(7);
        if (p == NPASS - 1) PS_STAMP(11);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_BARRIER();
        continue;
      }
      auto res_fetch = [&](int it, u32x4 (&dst)[CPT]) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
          const int ncol = cn0 + (tcol + j * TPR) * (16 / OB);
          dst[j] = u32x4{0u, 0u, 0u, 0u};
          if (resid && orow[it] >= 0 && ncol < N)
            dst[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(resid) + ((int64_t)orow[it] * N + ncol) * OB);
        }
      };
      u32x4 resv[CPT], resn[CPT];
      // passes after the first know their rows before staging (and the accumulators of the passes already stored are dead, so
      // there are registers for it): the first residual segments travel under the staging arithmetic instead of behind the barrier
      const bool early_res = p > 0 && (use_omap || !row_map);
      if (early_res) res_fetch(0, resv);
      // (2) bias (+GELU), convert, stage this pass's fragments in LDS
#pragma unroll
      for (int mm = 0; mm < MPP; ++mm) {
        const int mi = p * MPP + mm;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          f32x4 v = acc[mi][ni] + bv[ni];
          const int nl = wc * TN + ni * 16 + fq * 4;
          if constexpr (KEEP) OutPack<OT>::write4(cst2 + (wr * HALF + mm * 16 + fr) * C_STRIDE + nl * OB, v);
          if (KEEP || epilogue == SMOE_EPI_GELU) v = gelu_fast4(v);
          OutPack<OT>::write4(cst + (wr * HALF + mm * 16 + fr) * C_STRIDE + nl * OB, v);
        }
      }
      // LDS ordering only (no vector-memory drain): every wave's staging writes done, then the barrier.  On the last
      // pass each wave also retires its own outstanding vector-memory operations -- the next tile's K-tile 0 (issued
      // before pass 0) and the earlier passes' stores -- so that after this barrier buffer 0 is valid for everybody.
      if (p == 0) PS_STAMP(5);
      if (p == NPASS - 1) PS_STAMP(8);
      if (p == NPASS - 1 && more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (p == NPASS - 1) PS_STAMP(9);
      PP_BARRIER();
      if (p == 0) PS_STAMP(6);
      if (use_omap && p == 0) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int r = trow + it * ROWS_PER_IT;
          const int tr = (r / HALF) * TM + p * HALF + (r % HALF);
          orow[it] = omap[tr];
          oscale[it] = oscl[tr];
        }
      }
      if (p == NPASS - 1) PS_STAMP(10);
      // (3) whole-row-segment stores (combine scale and residual / gelu' fused).  Residual segments are fetched one row
      //     iteration ahead, and every use of a row's segments comes BEFORE that row's first store: with LDS-DMA in flight
      //     hipcc waits vmcnt(0) at any use of an ordinary load, so a use behind a store would wait for that store's
      //     acknowledgement (that was 4.2 k cycles per row iteration; the loads of row it + 1 now travel under the stores of it)
      if (!early_res) res_fetch(0, resv);
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int r = trow + it * ROWS_PER_IT;
        const int64_t rbase = (int64_t)orow[it] * N;
        u32x4 v[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
          v[j] = *reinterpret_cast<const u32x4*>(cst + r * C_STRIDE + (tcol + j * TPR) * 16);
          if (row_map && row_scale) v[j] = scale16<OT>(v[j], oscale[it]);
          if (resid) v[j] = fuse_aux<OT>(epilogue, resv[j], v[j]);
        }
        if (it + 1 < ITS) res_fetch(it + 1, resn);
        if (orow[it] >= 0) {
#pragma unroll
          for (int j = 0; j < CPT; ++j) {
            const int ncol = cn0 + (tcol + j * TPR) * (16 / OB);
            if (ncol < N) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + (rbase + ncol) * OB) = v[j];
          }
          if constexpr (KEEP) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
              const int ncol = cn0 + (tcol + j * TPR) * (16 / OB);
              const u32x4 h2 = *reinterpret_cast<const u32x4*>(cst2 + r * C_STRIDE + (tcol + j * TPR) * 16);
              if (ncol < N) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(pre_out) + (rbase + ncol) * OB) = h2;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) resv[j] = resn[j];
      }
      // the staging reads above are complete when their stores have issued; the next pass (or the next tile's K-tile 1)
      // may overwrite the region once every wave is here
      if (p == 0) PS_STAMP(7);
      if (p == NPASS - 1) PS_STAMP(11);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BARRIER();
    }
    PS_STAMP(12);
    if (!more) break;
    issue_kt1();   // into buffer 1 = the staging region just released; lands during the first K-tile's 8 intervals
  }
  if constexpr (FUSED == 0) PS_CLOCK(2);
  if constexpr (FUSED != 0) {
    // end of this run (the next position is a tile of the other GEMM, or the list is exhausted): drain this run's last stores,
    // release its m-tile, hand the two known positions back to the caller; behind the barrier the LDS is free for the other body
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PP_BARRIER();
    if constexpr (FUSED == 1) fused_signal(cmt);
  }
#undef PS_MFMA
#undef PS_DMA
}

template <typename AB, typename OT, int AFR, bool DEEP, bool KEEP = false, bool DIRECT = false, bool BUF = false>
__global__ __launch_bounds__(512, 2) void grouped_gemm_ps(
    const AB* __restrict__ A, const AB* __restrict__ W, const float* __restrict__ bias,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ group_expert, int E, int K, int N, int epilogue,
    const int64_t* __restrict__ row_map, const float* __restrict__ row_scale, const OT* residual, OT* out, int n_tiles_n,
    int group_m, const int64_t* __restrict__ a_gather, int a_div, int gather_len, const int32_t* __restrict__ group_end) {
  ps_body<AB, OT, AFR, DEEP, KEEP, DIRECT, BUF, 0>(A, W, bias, offsets, group_expert, E, K, N, epilogue, row_map, row_scale, residual,
                                                   out, n_tiles_n, group_m, a_gather, a_div, gather_len, group_end, nullptr);
}

// ---- the fused expert FFN: H = gelu(X[gather] W1^T + b1) (16-bit, write-through) and out[row_map] = residual + scale (H W2^T + b2)
//      (f32) from one persistent launch; both GEMMs on 320-row tiles (shared m-tiles), GEMM-1 = the direct-store body, GEMM-2 = the
//      deep-schedule body with the buffer-addressed f32 epilogue -- the same code, tile for tile, as the two separate launches ------
struct FusedPlan { int A, B2; };
// Closed-form balance for a list of `cnt` m-tiles drawn by G workgroups (every workgroup computes it the same way): q2 = n2 / G
// GEMM-2 tiles for every CU and one more for rem2 of them ("heavy").  All CUs first compute x GEMM-1 tiles (A = x G), then the
// heavy ones start their extra GEMM-2 tile while the light ones take the GEMM-1 tiles that are left (y each); x minimises
// max(x c1 + (q2 + 1) c2, y c1 + q2 c2).  The early GEMM-2 batch may only hold tiles whose producers lie in [0, A): whole
// numbering groups of gm m-tiles.
__device__ __forceinline__ FusedPlan fused_plan(int cnt, int ntn1, int ntn2, int G, int gm, int c1, int c2) {
  FusedPlan p;
  const int n1 = cnt * ntn1, n2 = cnt * ntn2;
  const int q2 = n2 / G, rem2 = n2 % G;
  p.A = n1;
  p.B2 = 0;
  if (c1 > 0 && rem2 > 0 && rem2 < G) {      // (c1 <= 0: every GEMM-1 tile in front of every GEMM-2 tile, for A/B)
    int best_x = -1, best_cost = 0x7fffffff;
    const int x_hi = n1 / G;
    for (int x = x_hi; x >= 0 && x >= x_hi - 8; --x) {
      const int left = n1 - x * G;                                     // GEMM-1 tiles for the light CUs after the common part
      const int y = x + (left + (G - rem2) - 1) / (G - rem2);
      const int heavy = x * c1 + (q2 + 1) * c2, light = y * c1 + q2 * c2;
      const int cost = heavy > light ? heavy : light;
      if (cost < best_cost) { best_cost = cost; best_x = x; }
    }
    const int A = best_x * G;
    const int early_cap = (A / (gm * ntn1)) * gm * ntn2;               // GEMM-2 tiles whose m-tiles are complete within [0, A)
    const int B2 = rem2 < early_cap ? rem2 : early_cap;
    if (A < n1 && B2 > 0) { p.A = A; p.B2 = B2; }
  }
  return p;
}

#ifdef SMOE_FFN_FUSED   // the fused GEMM-1 + GEMM-2 launch: measured slower than the two launches (profiles/r04_fused_ffn.md); not in the
                        // default build since round 5 -- `make FFN=-DSMOE_FFN_FUSED` compiles it (and smoe_expert_ffn) back in
template <typename AB>
__global__ __launch_bounds__(512, 2) void expert_ffn_fused(
    const AB* __restrict__ X, const int64_t* __restrict__ a_gather, int a_div, int gather_len, const AB* __restrict__ W1,
    const float* __restrict__ b1, AB* H, const AB* __restrict__ W2, const float* __restrict__ b2,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ group_expert, int E, int K1, int N1, int N2,
    const int64_t* __restrict__ row_map, const float* __restrict__ row_scale, const float* residual, float* out, int ntn1, int ntn2,
    int gm, int c1, int c2, int32_t* ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TBM = 320;
  const int tid = threadIdx.x, lane = tid & 63;
  // m-tiles of all row groups (the bodies build the same table again; it is a load and a wave scan)
  int total_mt;
  {
    const int li = lane < E ? lane : E;
    const int off = offsets[li], end = offsets[li < E ? li + 1 : E];
    int cnt = lane < E ? (end - off + TBM - 1) / TBM : 0;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const int up = __shfl_up(cnt, sft, 64);
      if (lane >= sft) cnt += up;
    }
    total_mt = __builtin_amdgcn_readlane(cnt, 63);
  }
  const int wave = tid >> 6;
  (void)wave;
  PS_CLOCK(0);
  const int home = (int)blockIdx.x & 7;               // the list of this workgroup (blocks b and b + 8 share an XCD)
  FusedCtl fc;
  fc.ws = ws;
  fc.G = (int)gridDim.x >> 3;                         // workgroups per list (the launcher keeps the grid a multiple of 8)
  fc.mt_lo = (total_mt * home) >> 3;
  fc.cnt = ((total_mt * (home + 1)) >> 3) - fc.mt_lo;
  fc.n1 = fc.cnt * ntn1;
  fc.n2 = fc.cnt * ntn2;
  {
    const FusedPlan pl = fused_plan(fc.cnt, ntn1, ntn2, fc.G, gm, c1, c2);
    fc.A = pl.A;
    fc.B2 = pl.B2;
  }
  fc.ntn1 = ntn1; fc.ntn2 = ntn2;
  fc.gm = gm;
  int* const slot2 = reinterpret_cast<int*>(smem);
  // one ticket of each stage, drawn together (lane 0 / lane 1 of one instruction: one round trip).  The stage-2 ticket is kept
  // until this workgroup has seen stage 1 exhausted.
  if (tid < 2) slot2[tid] = __hip_atomic_fetch_add(ws + home + 8 * tid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int t1 = __builtin_amdgcn_readfirstlane(slot2[0]);
  const int t2 = __builtin_amdgcn_readfirstlane(slot2[1]);
  __syncthreads();
  auto run_slot = [&]() {   // walk the slot (fc.stage, fc.slot): runs of GEMM-1 / GEMM-2 tiles alternate between the two bodies
    fc.j = 0;
    fused_advance(fc);
    while (fc.nk != 0) {
      if (fc.nk == 1) {
        ps_body<AB, AB, 5, false, false, true, false, 1>(X, W1, b1, offsets, group_expert, E, K1, N1, SMOE_EPI_GELU, nullptr, nullptr,
                                                         nullptr, H, ntn1, gm, a_gather, a_div, gather_len, nullptr, &fc);
      } else {
#ifdef SMOE_CLOCK
        const unsigned long long t_in = __builtin_amdgcn_s_memtime();
#endif
        ps_body<AB, float, 5, true, false, false, true, 2>(H, W2, b2, offsets, group_expert, E, N1, N2, SMOE_EPI_NONE, row_map, row_scale,
                                                           residual, out, ntn2, gm, nullptr, 1, gather_len, nullptr, &fc);
#ifdef SMOE_CLOCK
        PS_COUNT(7, __builtin_amdgcn_s_memtime() - t_in);
#endif
      }
    }
  };
  // ---- stage 1: the slot drawn, then any slot nobody else has drawn (only when fewer workgroups run than the grid holds) -------
  while (t1 < fc.G) {
    fc.stage = 1;
    fc.slot = t1;
    run_slot();
    if (tid == 0) slot2[0] = __hip_atomic_fetch_add(ws + home, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    t1 = __builtin_amdgcn_readfirstlane(slot2[0]);
    __syncthreads();
  }
  // ---- stage 2 (every stage-1 slot of this list has been drawn by a running workgroup): light slots first ---------------------
  if (t2 < fc.G) {
    const int L = fc.G - fc.B2;
    fc.stage = 2;
    fc.slot = t2 < L ? fc.B2 + t2 : t2 - L;
    run_slot();
  }
  PS_CLOCK(2);
  // the last workgroup to leave puts the workspace back to zero for the next launch (kept-zero workspace, as the router's).
  // This workgroup's own counter adds (issued by thread 0) must have been performed before its leave-count is: they go to other
  // addresses, and an add that landed after the zeroing would release an m-tile of the NEXT launch early.
  __syncthreads();
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    slot2[0] = __hip_atomic_fetch_add(ws + FUSED_WS_LEFT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (slot2[0] == (int)gridDim.x - 1) {
    for (int i = tid; i < FUSED_WS_HDR + total_mt; i += 512)
      if (i != FUSED_WS_ERR) __hip_atomic_store(ws + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the error word stays)
  }
}

#endif  // SMOE_FFN_FUSED

template <typename AB, typename OT, int AFR, bool DEEP, bool KEEP = false>
int launch_ps(const void* A, const void* W, const float* bias, const int32_t* offsets, const int32_t* group_expert, int E,
              int64_t m_rows_max, int K, int N, int epilogue, const int64_t* row_map, const float* row_scale,
              const void* residual, void* out, int group_m, hipStream_t s, const int64_t* a_gather, int a_div,
              const int32_t* group_end, int64_t out_rows, bool allow_direct = true) {
  constexpr int TBM = 64 * AFR, TBN = 256;
  const int n_tiles_n = (N + TBN - 1) / TBN;
  const int64_t max_tiles = ((m_rows_max + TBM - 1) / TBM + E) * n_tiles_n;
  int grid = (smoe_num_cus() - smoe_reserved_cus()) & ~7;   // one workgroup per CU (the LDS is full); a multiple of 8 (XCD slots);
                                                            // minus the CUs left to other streams (smoe_set_reserved_cus)
  if (grid < 8) grid = 8;
  if (max_tiles < grid) grid = (int)((max_tiles + 7) & ~(int64_t)7);
  // tile order (see the kernel): XCD-contiguous runs over blocks of n_block n-tiles.  Measured on the bench model's shapes,
  // strided order = 1.00.  Alone, back to back / behind a 512-MB write (profiles/r03_tile_order_ab.txt, _cold.txt): GEMM-1
  // (12 n-tiles) blocks of 4: 0.937 / 0.958, 6: 0.953 / 0.953, 12: 0.953 / 0.956; GEMM-2 (3 n-tiles) 3: 1.00 / 0.983;
  // attention projection (3 n-tiles) 3: 0.979 / 1.00; qkv (9 n-tiles) 3: 1.005 / 1.023.  INSIDE the model (bench.py's
  // per-kernel times, both orders on one box, profiles/r03_bench_order_ab.txt): projection 0.969, GEMM-2 0.993, but GEMM-1
  // 1.012 -- its A rows were written by the LayerNorm + router kernel a moment ago and come from the Infinity Cache either
  // way; what the isolated runs measured was the cold A stream.  The L2 counters do not move under any order (TCC hit
  // 71.7 %, profiles/r03_pmc_grouped_gemm.txt).  Hence: one block of all n-tiles when there are at most 4, else strided.
  int n_block = n_tiles_n <= 4 ? n_tiles_n : 0;
#ifdef SMOE_DIAG
  if (const char* gcap = getenv("SMOE_PS_GRID")) grid = atoi(gcap) & ~7;   // diagnostic: fewer CUs (is a phase chip- or CU-bound?)
  if (const char* ord = getenv("SMOE_PS_NBLOCK")) n_block = atoi(ord);    // diagnostic: 0 = the strided order, else the n-block width
  if (n_block && n_tiles_n % n_block) n_block = 0;
#endif
  if (n_block) group_m = -(group_m | (n_block << 8));
  if constexpr (!KEEP && sizeof(OT) == 2) {
    // plain 16-bit outputs (GEMM-1, qkv, patch embedding): the epilogue that stores from the registers
    if (allow_direct && !row_map && !residual && (epilogue == SMOE_EPI_NONE || epilogue == SMOE_EPI_GELU) && (int64_t)N * TBM * 2 < (1ll << 31)) {
      SMOE_ENSURE_SMEM(grouped_gemm_ps<AB, OT, AFR, DEEP, false, true>);
      hipLaunchKernelGGL((grouped_gemm_ps<AB, OT, AFR, DEEP, false, true>), dim3(grid), dim3(512), 160 * 1024, s, (const AB*)A,
                         (const AB*)W, bias, offsets, group_expert, E, K, N, epilogue, row_map, row_scale, (const OT*)residual,
                         (OT*)out, n_tiles_n, group_m, a_gather, a_div, (int)m_rows_max, group_end);
      SMOE_CHECK_LAUNCH("smoe_grouped_gemm/persistent-direct");
      return 0;
    }
  }
  if constexpr (!KEEP && sizeof(OT) == 4) {
    // f32 outputs (GEMM-2 with the fused combine + residual, the attention projection): the staged epilogue whose residual
    // loads and output stores go through buffer descriptors; every output row must end inside their 2 GiB
    if (allow_direct && epilogue != SMOE_EPI_GELU_GRAD && out_rows > 0 && out_rows * (int64_t)N * 4 < (1ll << 31)) {
      SMOE_ENSURE_SMEM(grouped_gemm_ps<AB, OT, AFR, DEEP, false, false, true>);
      hipLaunchKernelGGL((grouped_gemm_ps<AB, OT, AFR, DEEP, false, false, true>), dim3(grid), dim3(512), 160 * 1024, s,
                         (const AB*)A, (const AB*)W, bias, offsets, group_expert, E, K, N, epilogue, row_map, row_scale,
                         (const OT*)residual, (OT*)out, n_tiles_n, group_m, a_gather, a_div, (int)m_rows_max, group_end);
      SMOE_CHECK_LAUNCH("smoe_grouped_gemm/persistent-buffer");
      return 0;
    }
  }
  SMOE_ENSURE_SMEM(grouped_gemm_ps<AB, OT, AFR, DEEP, KEEP>);
  hipLaunchKernelGGL((grouped_gemm_ps<AB, OT, AFR, DEEP, KEEP>), dim3(grid), dim3(512), 160 * 1024, s, (const AB*)A, (const AB*)W,
                     bias, offsets, group_expert, E, K, N, epilogue, row_map, row_scale, (const OT*)residual, (OT*)out,
                     n_tiles_n, group_m, a_gather, a_div, (int)m_rows_max, group_end);
  SMOE_CHECK_LAUNCH("smoe_grouped_gemm/persistent");
  return 0;
}
