// Dispatch plan (histogram + stable counting sort + capacity prune), token scatter, gather+combine.
// Replaces fmoe_cuda.expert_count / assign_pos / limit_by_capacity / prune_gate_by_capacity and the
// index_select / index_copy_ / bmm of MOEScatter / MOEGather (N1-N3, N6-N9).
//
// Plan over n = T*k flat entries, chunked CH entries per workgroup:
//   plan_count : per-chunk LDS histogram                    -> blockcnt[nblk][E]
//   plan_scan  : per expert, exclusive scan over chunks      -> rawbase[nblk][E], counts, offsets
//   plan_assign: per chunk, stable rank of each entry among same-expert entries with lower flat
//                index (wave ballot + LDS running counters) -> pos / inv_pos / idx_pruned
//   plan_tail  : pos[kept .. n) = -1
// When the chunk histogram table is small (nblk * E <= PLAN_FUSED_MAX, E <= 64: every ViT shape) the scan and the
// tail are folded into the assign launch -- every workgroup re-derives its own chunk's base ranks, the totals and
// the offsets from the whole table in LDS (a few KB) -- so the plan is two launches instead of four.
// Upstream orders slots by atomicSub race; here slot order is ascending flat index (deterministic),
// and with a capacity an entry is kept iff its raw rank in its expert is < capacity.
#include "smoe_common.h"
#include <type_traits>

namespace {

constexpr int PLAN_THREADS = 256;
constexpr int PLAN_WAVES = PLAN_THREADS / 64;
constexpr int PLAN_ITERS = 4;                                   // 64-entry steps per wave
constexpr int PLAN_CH = PLAN_THREADS * PLAN_ITERS;              // entries per workgroup
constexpr int PLAN_FUSED_MAX = 8192;                            // table entries the fused assign re-reads per workgroup
constexpr int PLAN_FUSED_E = 64;

__global__ __launch_bounds__(PLAN_THREADS) void plan_count_kernel(const int64_t* __restrict__ idx, int64_t n, int E,
                                                                  int32_t* __restrict__ blockcnt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* hist = reinterpret_cast<int32_t*>(smem);
  for (int e = threadIdx.x; e < E; e += PLAN_THREADS) hist[e] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * PLAN_CH;
#pragma unroll
  for (int it = 0; it < PLAN_ITERS; ++it) {
    const int64_t i = base + it * PLAN_THREADS + threadIdx.x;
    if (i < n) {
      const int64_t e = idx[i];
      if (e >= 0 && e < E) atomicAdd(&hist[(int)e], 1);  // ids outside [0, E) count as dropped (-1)
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += PLAN_THREADS) blockcnt[(int64_t)blockIdx.x * E + e] = hist[e];
}

// single workgroup.  rawbase[b][e] = sum_{b'<b} blockcnt[b'][e]; counts[e] = min(total, cap); offsets = prefix.
__global__ __launch_bounds__(1024) void plan_scan_kernel(const int32_t* __restrict__ blockcnt, int nblk, int E,
                                                         int64_t capacity, int32_t* __restrict__ rawbase,
                                                         int32_t* __restrict__ counts, int32_t* __restrict__ offsets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* cnt = reinterpret_cast<int32_t*>(smem);  // [E]
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    int32_t run = 0;
    for (int b = 0; b < nblk; ++b) {
      const int32_t c = blockcnt[(int64_t)b * E + e];
      rawbase[(int64_t)b * E + e] = run;
      run += c;
    }
    if (capacity >= 0 && (int64_t)run > capacity) run = (int32_t)capacity;
    cnt[e] = run;
    counts[e] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int e = 0; e < E; ++e) {
      offsets[e] = run;
      run += cnt[e];
    }
    offsets[E] = run;
  }
}

// FUSED: rawbase_or_cnt is blockcnt[nblk][E]; the scan (base ranks of this chunk, totals, capacity clamp, offsets)
// happens here in LDS, workgroup 0 publishes counts / offsets, and every workgroup clears its share of the pos tail.
// PADDED layout (slot_stride > 0; the static exchange buffers of a capacity gate under expert parallelism): expert e owns the
// slots [e * slot_stride, (e + 1) * slot_stride) whatever the counts are, slot = e * slot_stride + rank; pos has E * slot_stride
// entries (unused ones -1) and group_end_out[e] = e * slot_stride + counts[e] closes expert e's row range for the grouped GEMM.
template <bool FUSED>
__global__ __launch_bounds__(PLAN_THREADS) void plan_assign_kernel(
    const int64_t* __restrict__ idx, int64_t n, int E, int64_t capacity, const int32_t* __restrict__ rawbase_or_cnt,
    const int32_t* __restrict__ offsets_in, int64_t* __restrict__ pos, int64_t* __restrict__ inv_pos,
    int64_t* __restrict__ idx_pruned, int nblk, int32_t* __restrict__ counts_out, int32_t* __restrict__ offsets_out,
    int64_t slot_stride, int32_t* __restrict__ group_end_out, int tab_rows, int tab_ratio,
    int32_t* __restrict__ raw_out = nullptr, const int32_t* __restrict__ slot_base = nullptr, int hdr_rows = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // run[w][e]: raw rank of the next entry of expert e seen by wave w (wave w owns a contiguous quarter of the chunk)
  int32_t* run = reinterpret_cast<int32_t*>(smem);  // [PLAN_WAVES][E]
  int32_t* sbase = run + PLAN_WAVES * E;            // FUSED: [E] base rank of this chunk, [E] totals, [E+1] offsets
  int32_t* stot = sbase + E;
  int32_t* soff = stot + E;
  // SLOT TABLE layout (slot_base != NULL; the speculative static exchange's per-expert slots): expert e owns the slots
  // [slot_base[e], slot_base[e + 1]), the last hdr_rows of which are not payload; it keeps at most that many entries (and at most
  // `capacity` when that is >= 0).  ssb = the table in LDS, scap = every expert's payload capacity.
  int32_t* ssb = soff + E + 1;                      // [E+1]
  int32_t* scap = ssb + E + 1;                      // [E]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t chunk_base = (int64_t)blockIdx.x * PLAN_CH;
  const int64_t wave_base = chunk_base + (int64_t)wave * (64 * PLAN_ITERS);

  for (int i = tid; i < PLAN_WAVES * E; i += PLAN_THREADS) run[i] = 0;
  if constexpr (FUSED) {
    for (int i = tid; i < 2 * E; i += PLAN_THREADS) sbase[i] = 0;
    if (slot_base) {
      for (int i = tid; i <= E; i += PLAN_THREADS) ssb[i] = slot_base[i];
    }
    __syncthreads();
    if (slot_base) {
      for (int e = tid; e < E; e += PLAN_THREADS) {
        int32_t c = ssb[e + 1] - ssb[e] - hdr_rows;
        c = c < 0 ? 0 : c;
        if (capacity >= 0 && (int64_t)c > capacity) c = (int32_t)capacity;
        scap[e] = c;
      }
    }
    // the count table: tab_rows rows of E, row b = the entries [b CH / tab_ratio, (b + 1) CH / tab_ratio) -- this kernel's own
    // chunks (tab_ratio 1, smoe_dispatch_plan's counting launch) or the finer chunks the fused router counted on its way
    // (smoe_dispatch_plan_hist): the rows in front of this workgroup's chunk are those below me * tab_ratio
    const int me = (int)blockIdx.x;
    if (PLAN_THREADS % E == 0) {
      // every thread meets ONE expert on its stride (j = tid + i * 256, 256 % E == 0): private sums, two LDS atomics per thread
      // (an atomic per table entry serialised thousands of adds on E words: the router's 64-token rows make the table 16 x longer)
      const int e = tid % E;
      int32_t tot = 0, base = 0;
      const int first = me * tab_ratio * E;          // entries below `first` belong to rows in front of this chunk
      for (int j = tid; j < tab_rows * E; j += PLAN_THREADS) {
        const int32_t c = rawbase_or_cnt[j];
        tot += c;
        base += j < first ? c : 0;
      }
      if (tot) atomicAdd(&stot[e], tot);
      if (base) atomicAdd(&sbase[e], base);
    } else {
      for (int j = tid; j < tab_rows * E; j += PLAN_THREADS) {
        const int32_t c = rawbase_or_cnt[j];
        if (c) {
          const int b = j / E, e = j - b * E;
          atomicAdd(&stot[e], c);
          if (b < me * tab_ratio) atomicAdd(&sbase[e], c);
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      int32_t acc = 0;
      for (int e = 0; e < E; ++e) {
        int32_t c = stot[e];
        if (me == 0 && raw_out) raw_out[e] = c;   // the un-clamped total: what the speculative exchange's overflow test reads
        if (slot_base) { if (c > scap[e]) c = scap[e]; }
        else if (capacity >= 0 && (int64_t)c > capacity) c = (int32_t)capacity;
        stot[e] = c;
        soff[e] = acc;
        acc += c;
      }
      soff[E] = acc;
    }
    __syncthreads();
    if (me == 0) {
      for (int e = tid; e < E; e += PLAN_THREADS) counts_out[e] = stot[e];
      for (int e = tid; e <= E; e += PLAN_THREADS) offsets_out[e] = soff[e];
      if (group_end_out)
        for (int e = tid; e < E; e += PLAN_THREADS)
          group_end_out[e] = (slot_base ? ssb[e] : (int32_t)(e * slot_stride)) + stot[e];
    }
    if (slot_base) {
      // unused payload slots (and the header rows) of every expert's range hold no entry
      for (int e = 0; e < E; ++e)
        for (int64_t sl = (int64_t)ssb[e] + stot[e] + (int64_t)me * PLAN_THREADS + tid; sl < ssb[e + 1]; sl += (int64_t)nblk * PLAN_THREADS)
          pos[sl] = -1;
    } else if (slot_stride > 0) {
      // unused slots of every expert's fixed range hold no entry; the workgroups share the E * slot_stride slots evenly
      const int64_t total = (int64_t)E * slot_stride, share = (total + nblk - 1) / nblk;
      for (int64_t sl = (int64_t)me * share + tid; sl < (int64_t)(me + 1) * share && sl < total; sl += PLAN_THREADS) {
        const int e = (int)(sl / slot_stride);
        if (sl - (int64_t)e * slot_stride >= stot[e]) pos[sl] = -1;
      }
    } else {
      // tail: slots [kept, n) hold no entry
      const int64_t kept = soff[E];
      for (int64_t i = chunk_base + tid; i < chunk_base + PLAN_CH && i < n; i += PLAN_THREADS)
        if (i >= kept) pos[i] = -1;
    }
  }
  const int32_t* offsets = FUSED ? soff : offsets_in;
  __syncthreads();
  // pass 1: per-wave histogram of its quarter
  int64_t myidx[PLAN_ITERS];
#pragma unroll
  for (int it = 0; it < PLAN_ITERS; ++it) {
    const int64_t i = wave_base + it * 64 + lane;
    int64_t e = -1;
    if (i < n) e = idx[i];
    if (e >= E) e = -1;
    myidx[it] = e;
    if (e >= 0) atomicAdd(&run[wave * E + (int)e], 1);
  }
  __syncthreads();
  // turn per-wave counts into per-wave starting raw ranks: rawbase[b][e] + sum_{w'<w} cnt[w'][e]
  for (int e = tid; e < E; e += PLAN_THREADS) {
    int32_t acc = FUSED ? sbase[e] : rawbase_or_cnt[(int64_t)blockIdx.x * E + e];
#pragma unroll
    for (int w = 0; w < PLAN_WAVES; ++w) {
      const int32_t c = run[w * E + e];
      run[w * E + e] = acc;
      acc += c;
    }
  }
  __syncthreads();
  // pass 2: in flat-index order inside the wave's quarter
  int32_t* myrun = run + wave * E;
#pragma unroll
  for (int it = 0; it < PLAN_ITERS; ++it) {
    const int64_t i = wave_base + it * 64 + lane;
    const int e = (int)myidx[it];
    // rank among lanes of this step with the same expert: peel one distinct expert per round
    int rank_in_step = 0, group = 0;
    unsigned long long todo = __ballot(e >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int e0 = __shfl(e, leader, 64);
      const unsigned long long m = __ballot(e == e0);
      if (e == e0) {
        rank_in_step = __popcll(m & ((1ull << lane) - 1ull));
        group = __popcll(m);
      }
      todo &= ~m;
    }
    if (e >= 0) {
      const int32_t raw = myrun[e] + rank_in_step;
      const bool keep = FUSED && slot_base ? raw < scap[e] : ((capacity < 0) || ((int64_t)raw < capacity));
      if (keep) {
        const int64_t slot = FUSED && slot_base ? (int64_t)ssb[e] + raw
                             : (slot_stride > 0 ? (int64_t)e * slot_stride + raw : (int64_t)offsets[e] + raw);
        pos[slot] = i;
        inv_pos[i] = slot;
      } else {
        inv_pos[i] = -1;
      }
      if (idx_pruned) idx_pruned[i] = keep ? (int64_t)e : -1;
    } else if (i < n) {
      inv_pos[i] = -1;
      if (idx_pruned) idx_pruned[i] = -1;
    }
    __builtin_amdgcn_wave_barrier();
    // the first lane of each expert group advances that expert's running rank
    if (e >= 0 && rank_in_step == 0) myrun[e] += group;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// fill the tail pos[kept .. n) with -1
__global__ void plan_tail_kernel(const int32_t* __restrict__ offsets, int E, int64_t n, int64_t* __restrict__ pos) {
  const int64_t kept = offsets[E];
  for (int64_t i = kept + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    pos[i] = -1;
}

// ------------------------------------------------------------------------------------------------ rows
// one wave per slot; 8 elements per lane per step
template <typename XT, typename BT>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const XT* __restrict__ x, const int64_t* __restrict__ pos,
                                                           const float* __restrict__ scale, int64_t n_slots, int k, int d,
                                                           BT* __restrict__ buf, int zero_unmapped) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t s = wave_gid; s < n_slots; s += nwaves) {
    const int64_t p = pos[s];
    BT* dst = buf + s * (int64_t)d;
    if (p < 0) {  // wave-uniform: a slot no token maps to (past the kept count) -- left alone, or cleared in the same pass
      if (zero_unmapped) {
        float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = lane * 8; c < d; c += 512) store8(dst + c, z);
      }
      continue;
    }
    const XT* src = x + (p / k) * (int64_t)d;
    const float sc = scale ? scale[p] : 1.0f;
    for (int c = lane * 8; c < d; c += 512) {
      float v[8];
      load8(src + c, v);
      if (scale) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= sc;
      }
      store8(dst + c, v);
    }
  }
}

template <typename YT, typename OT, int KMAX>
__global__ __launch_bounds__(256) void gather_combine_kernel(const YT* __restrict__ y,
                                                             const int64_t* __restrict__ inv_pos,
                                                             const float* __restrict__ score, int64_t T, int k, int d,
                                                             const OT* __restrict__ residual, OT* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave_gid; t < T; t += nwaves) {
    int64_t slot[KMAX];
    float sc[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      slot[j] = (j < k) ? inv_pos[t * k + j] : -1;
      sc[j] = (j < k) ? score[t * k + j] : 0.f;
    }
    for (int c = lane * 8; c < d; c += 512) {
      float acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        if (slot[j] >= 0) {
          float v[8];
          load8(y + slot[j] * (int64_t)d + c, v);
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[q] = fmaf(sc[j], v[q], acc[q]);
        }
      }
      if (residual) {
        float r[8];
        load8(residual + t * (int64_t)d + c, r);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += r[q];
      }
      store8(out + t * (int64_t)d + c, acc);
    }
  }
}

// gather + combine + residual AND the LayerNorm that reads the result next (the expert-parallel path's return side: the block's
// `x + mlp(...)` is produced here row by row, and the NEXT block starts with norm1 of exactly that row, models/
// vision_transformer.py:320): the wave that owns a row holds it whole in registers, so mean / variance (two-pass, as
// smoe_layernorm) and the normalised 16-bit row cost no further HBM read -- one 155-MB pass per layer less.
template <typename YT, typename NT, int KMAX, int NJ>
__global__ __launch_bounds__(256) void gather_combine_ln_kernel(const YT* __restrict__ y, const int64_t* __restrict__ inv_pos,
                                                                const float* __restrict__ score, int64_t T, int k, int d,
                                                                const float* __restrict__ residual, float* __restrict__ out,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float eps, NT* __restrict__ xn) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave_gid; t < T; t += nwaves) {
    int64_t slot[KMAX];
    float sc[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      slot[j] = (j < k) ? inv_pos[t * k + j] : -1;
      sc[j] = (j < k) ? score[t * k + j] : 0.f;
    }
    float acc[NJ][8];
#pragma unroll
    for (int it = 0; it < NJ; ++it) {
      const int c = lane * 8 + 512 * it;
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[it][q] = 0.f;
      if (c < d) {
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          if (slot[j] >= 0) {
            float v[8];
            load8(y + slot[j] * (int64_t)d + c, v);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[it][q] = fmaf(sc[j], v[q], acc[it][q]);
          }
        }
        if (residual) {
          float r[8];
          load8(residual + t * (int64_t)d + c, r);
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[it][q] += r[q];
        }
        store8(out + t * (int64_t)d + c, acc[it]);
      }
    }
    float mean, rstd;      // (smoe_common.h: THE wave-per-row LayerNorm arithmetic -- the bits of smoe_layernorm at d >= 768)
    wave_row_stats<NJ>(acc, d, lane, eps, mean, rstd);
#pragma unroll
    for (int it = 0; it < NJ; ++it) {
      const int c = lane * 8 + 512 * it;
      if (c < d) {
        float g[8], b[8], o[8];
        load8(gamma + c, g);
        load8(beta + c, b);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = wave_row_affine(acc[it][q], mean, rstd, g[q], b[q]);
        store8(xn + t * (int64_t)d + c, o);
      }
    }
  }
}

template <typename ST, typename DT>
__global__ __launch_bounds__(256) void cast_kernel(const ST* __restrict__ src, DT* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      float v[8];
      load8(src + i, v);
      store8(dst + i, v);
    } else {
      for (int64_t j = i; j < n; ++j) {
        float v[4];
        // scalar tail
        if constexpr (sizeof(ST) == 4) v[0] = ((const float*)src)[j];
        else if constexpr (std::is_same<ST, f16>::value) v[0] = (float)((const f16*)src)[j];
        else v[0] = bf16_to_f32(((const bf16_bits*)src)[j]);
        if constexpr (sizeof(DT) == 4) ((float*)dst)[j] = v[0];
        else if constexpr (std::is_same<DT, f16>::value) ((f16*)dst)[j] = (f16)v[0];
        else ((bf16_bits*)dst)[j] = f32_to_bf16(v[0]);
      }
    }
  }
}

inline int rows_grid(int64_t rows) {
  int64_t blocks = (rows + 3) / 4;  // 4 waves per workgroup
  if (blocks < 1) blocks = 1;
  if (blocks > 8192) blocks = 8192;
  return (int)blocks;
}

template <typename XT>
int scatter_dispatch_b(const void* x, const int64_t* pos, const float* scale, int64_t n_slots, int k, int d, void* buf,
                       int buf_dtype, int zero_unmapped, hipStream_t s) {
  const int grid = rows_grid(n_slots);
  switch (buf_dtype) {
    case SMOE_F32:
      hipLaunchKernelGGL((scatter_rows_kernel<XT, float>), dim3(grid), dim3(256), 0, s, (const XT*)x, pos, scale, n_slots, k, d, (float*)buf, zero_unmapped);
      break;
    case SMOE_F16:
      hipLaunchKernelGGL((scatter_rows_kernel<XT, f16>), dim3(grid), dim3(256), 0, s, (const XT*)x, pos, scale, n_slots, k, d, (f16*)buf, zero_unmapped);
      break;
    case SMOE_BF16:
      hipLaunchKernelGGL((scatter_rows_kernel<XT, bf16_bits>), dim3(grid), dim3(256), 0, s, (const XT*)x, pos, scale, n_slots, k, d, (bf16_bits*)buf, zero_unmapped);
      break;
    default:
      smoe_set_error("smoe_scatter_rows: bad buf_dtype %d", buf_dtype);
      return 1;
  }
  SMOE_CHECK_LAUNCH("smoe_scatter_rows");
  return 0;
}

template <typename YT, typename OT>
int combine_launch(const void* y, const int64_t* inv_pos, const float* score, int64_t T, int k, int d,
                   const void* residual, void* out, hipStream_t s) {
  const int grid = rows_grid(T);
  if (k == 1)
    hipLaunchKernelGGL((gather_combine_kernel<YT, OT, 1>), dim3(grid), dim3(256), 0, s, (const YT*)y, inv_pos, score, T, k, d, (const OT*)residual, (OT*)out);
  else if (k == 2)
    hipLaunchKernelGGL((gather_combine_kernel<YT, OT, 2>), dim3(grid), dim3(256), 0, s, (const YT*)y, inv_pos, score, T, k, d, (const OT*)residual, (OT*)out);
  else if (k <= 4)
    hipLaunchKernelGGL((gather_combine_kernel<YT, OT, 4>), dim3(grid), dim3(256), 0, s, (const YT*)y, inv_pos, score, T, k, d, (const OT*)residual, (OT*)out);
  else
    hipLaunchKernelGGL((gather_combine_kernel<YT, OT, 8>), dim3(grid), dim3(256), 0, s, (const YT*)y, inv_pos, score, T, k, d, (const OT*)residual, (OT*)out);
  SMOE_CHECK_LAUNCH("smoe_gather_combine");
  return 0;
}

template <typename YT>
int combine_dispatch_o(const void* y, const int64_t* inv_pos, const float* score, int64_t T, int k, int d,
                       const void* residual, void* out, int out_dtype, hipStream_t s) {
  switch (out_dtype) {
    case SMOE_F32: return combine_launch<YT, float>(y, inv_pos, score, T, k, d, residual, out, s);
    case SMOE_F16: return combine_launch<YT, f16>(y, inv_pos, score, T, k, d, residual, out, s);
    case SMOE_BF16: return combine_launch<YT, bf16_bits>(y, inv_pos, score, T, k, d, residual, out, s);
  }
  smoe_set_error("smoe_gather_combine: bad out_dtype %d", out_dtype);
  return 1;
}

template <typename ST>
int cast_dispatch_d(const void* src, void* dst, int dst_dtype, int64_t n, hipStream_t s) {
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  switch (dst_dtype) {
    case SMOE_F32: hipLaunchKernelGGL((cast_kernel<ST, float>), dim3((int)blocks), dim3(256), 0, s, (const ST*)src, (float*)dst, n); break;
    case SMOE_F16: hipLaunchKernelGGL((cast_kernel<ST, f16>), dim3((int)blocks), dim3(256), 0, s, (const ST*)src, (f16*)dst, n); break;
    case SMOE_BF16: hipLaunchKernelGGL((cast_kernel<ST, bf16_bits>), dim3((int)blocks), dim3(256), 0, s, (const ST*)src, (bf16_bits*)dst, n); break;
    default: smoe_set_error("smoe_cast: bad dst_dtype %d", dst_dtype); return 1;
  }
  SMOE_CHECK_LAUNCH("smoe_cast");
  return 0;
}

inline int64_t plan_nblk(int64_t n) { return n > 0 ? (n + PLAN_CH - 1) / PLAN_CH : 1; }

}  // namespace

// workspace layout: [16 B reserved][blockcnt nblk*E i32][rawbase nblk*E i32]
extern "C" size_t smoe_dispatch_plan_workspace_bytes(int64_t n, int E) {
  if (n < 0 || E <= 0) return 0;
  const size_t per = (((size_t)plan_nblk(n) * (size_t)E * 4) + 15) & ~(size_t)15;
  return 16 + 2 * per;
}

static int plan_impl(const int64_t* idx, int64_t n, int E, int64_t capacity, int32_t* counts, int32_t* offsets, int64_t* pos,
                     int64_t* inv_pos, int64_t* idx_pruned, void* workspace, size_t workspace_bytes, void* stream,
                     int64_t slot_stride, int32_t* group_end, int32_t* raw_counts = nullptr, const int32_t* slot_base = nullptr,
                     int hdr_rows = 0) {
  SMOE_REQUIRE(counts && offsets && workspace, "smoe_dispatch_plan: null pointer");
  SMOE_REQUIRE(n >= 0 && n < (1ll << 31), "smoe_dispatch_plan: n=%lld out of range", (long long)n);
  SMOE_REQUIRE(E >= 1 && E <= 8192, "smoe_dispatch_plan: E=%d out of range [1, 8192]", E);
  SMOE_REQUIRE(n == 0 || (idx && pos && inv_pos), "smoe_dispatch_plan: null pointer");
  SMOE_REQUIRE(workspace_bytes >= smoe_dispatch_plan_workspace_bytes(n, E),
               "smoe_dispatch_plan: workspace too small (%zu < %zu)", workspace_bytes,
               smoe_dispatch_plan_workspace_bytes(n, E));
  hipStream_t s = (hipStream_t)stream;
  const int64_t nblk = plan_nblk(n);
  const size_t per = (((size_t)nblk * (size_t)E * 4) + 15) & ~(size_t)15;
  int32_t* blockcnt = reinterpret_cast<int32_t*>((char*)workspace + 16);
  int32_t* rawbase = reinterpret_cast<int32_t*>((char*)workspace + 16 + per);
  hipLaunchKernelGGL(plan_count_kernel, dim3((int)nblk), dim3(PLAN_THREADS), (size_t)E * 4, s, idx, n, E, blockcnt);
  SMOE_CHECK_LAUNCH("smoe_dispatch_plan/count");
  if (n > 0 && E <= PLAN_FUSED_E && nblk * E <= PLAN_FUSED_MAX) {
    hipLaunchKernelGGL(plan_assign_kernel<true>, dim3((int)nblk), dim3(PLAN_THREADS), (size_t)(PLAN_WAVES * E + 5 * E + 2) * 4, s,
                       idx, n, E, capacity, blockcnt, nullptr, pos, inv_pos, idx_pruned, (int)nblk, counts, offsets, slot_stride,
                       group_end, (int)nblk, 1, raw_counts, slot_base, hdr_rows);
    SMOE_CHECK_LAUNCH("smoe_dispatch_plan/assign_fused");
    return 0;
  }
  SMOE_REQUIRE(slot_stride == 0 && !slot_base, "smoe_dispatch_plan_padded / _slots: the padded layouts need E <= %d and ceil(n / %d) * E <= %d",
               PLAN_FUSED_E, PLAN_CH, PLAN_FUSED_MAX);
  const int scan_threads = E < 64 ? 64 : (E > 1024 ? 1024 : ((E + 63) / 64) * 64);
  hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(scan_threads), (size_t)E * 4, s, blockcnt, (int)nblk, E, capacity, rawbase, counts, offsets);
  SMOE_CHECK_LAUNCH("smoe_dispatch_plan/scan");
  if (n > 0) {
    hipLaunchKernelGGL(plan_assign_kernel<false>, dim3((int)nblk), dim3(PLAN_THREADS), (size_t)PLAN_WAVES * E * 4, s, idx, n, E, capacity, rawbase, offsets, pos, inv_pos, idx_pruned, (int)nblk, nullptr, nullptr, (int64_t)0, nullptr, 0, 1);
    SMOE_CHECK_LAUNCH("smoe_dispatch_plan/assign");
    int tb = (int)((n + 255) / 256);
    if (tb > 1024) tb = 1024;
    hipLaunchKernelGGL(plan_tail_kernel, dim3(tb), dim3(256), 0, s, offsets, E, n, pos);
    SMOE_CHECK_LAUNCH("smoe_dispatch_plan/tail");
  }
  return 0;
}

extern "C" int smoe_dispatch_plan(const int64_t* idx, int64_t n, int E, int64_t capacity, int32_t* counts,
                                  int32_t* offsets, int64_t* pos, int64_t* inv_pos, int64_t* idx_pruned,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return plan_impl(idx, n, E, capacity, counts, offsets, pos, inv_pos, idx_pruned, workspace, workspace_bytes, stream, 0, nullptr);
}

// The plan from a chunk histogram the router already made (smoe_ln_router_topk / smoe_gate_ln_router `chunk_hist`): hist[c][e] =
// entries of expert e among the flat entries [c * hist_chunk, (c + 1) * hist_chunk) of idx (entries outside [0, E) uncounted).
// ONE launch (the fused assign) instead of two, and no second pass over idx for counting.  Falls back to smoe_dispatch_plan
// (counting launch included) when the table does not fit the fused kernel (E > 64, rows x E > 8192, hist_chunk not a divisor of 1024).
extern "C" int smoe_dispatch_plan_hist(const int64_t* idx, int64_t n, int E, int64_t capacity, const int32_t* hist,
                                       int hist_chunk, int32_t* counts, int32_t* offsets, int64_t* pos, int64_t* inv_pos,
                                       int64_t* idx_pruned, void* workspace, size_t workspace_bytes, void* stream) {
  const int64_t rows = hist_chunk > 0 ? (n + hist_chunk - 1) / hist_chunk : 0;
  if (!hist || n <= 0 || hist_chunk <= 0 || PLAN_CH % hist_chunk != 0 || E > PLAN_FUSED_E || rows * E > PLAN_FUSED_MAX)
    return plan_impl(idx, n, E, capacity, counts, offsets, pos, inv_pos, idx_pruned, workspace, workspace_bytes, stream, 0, nullptr);
  SMOE_REQUIRE(counts && offsets && idx && pos && inv_pos, "smoe_dispatch_plan_hist: null pointer");
  SMOE_REQUIRE(n < (1ll << 31) && E >= 1, "smoe_dispatch_plan_hist: bad sizes");
  const int64_t nblk = plan_nblk(n);
  hipLaunchKernelGGL(plan_assign_kernel<true>, dim3((int)nblk), dim3(PLAN_THREADS), (size_t)(PLAN_WAVES * E + 5 * E + 2) * 4,
                     (hipStream_t)stream, idx, n, E, capacity, hist, nullptr, pos, inv_pos, idx_pruned, (int)nblk, counts, offsets,
                     (int64_t)0, nullptr, (int)rows, PLAN_CH / hist_chunk);
  SMOE_CHECK_LAUNCH("smoe_dispatch_plan_hist/assign_fused");
  return 0;
}

extern "C" int smoe_dispatch_plan_padded(const int64_t* idx, int64_t n, int E, int64_t capacity, int64_t slot_rows,
                                         int32_t* counts, int32_t* offsets, int32_t* group_end, int64_t* pos_padded,
                                         int64_t* inv_pos, int64_t* idx_pruned, int32_t* raw_counts, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  SMOE_REQUIRE(capacity >= 1 && n >= 1, "smoe_dispatch_plan_padded: needs a capacity >= 1 and n >= 1");
  SMOE_REQUIRE(slot_rows >= capacity, "smoe_dispatch_plan_padded: slot_rows=%lld < capacity=%lld", (long long)slot_rows,
               (long long)capacity);
  SMOE_REQUIRE(group_end, "smoe_dispatch_plan_padded: null pointer");
  SMOE_REQUIRE((int64_t)E * slot_rows < (1ll << 31), "smoe_dispatch_plan_padded: E * slot_rows out of range");
  return plan_impl(idx, n, E, capacity, counts, offsets, pos_padded, inv_pos, idx_pruned, workspace, workspace_bytes, stream,
                   slot_rows, group_end, raw_counts);
}

// The same plan over a SLOT TABLE (the static expert exchange, ep.py): expert e owns the slots [slot_base[e], slot_base[e + 1]) of
// the send buffer, the last hdr_rows of them reserved (the in-band header row); it keeps min(region - hdr_rows, capacity if >= 0)
// entries.  pos has slot_base[E] entries (unused ones -1); group_end[e] = slot_base[e] + counts[e]; raw_counts as above.
extern "C" int smoe_dispatch_plan_slots(const int64_t* idx, int64_t n, int E, int64_t capacity, const int32_t* slot_base,
                                        int hdr_rows, int32_t* counts, int32_t* offsets, int32_t* group_end, int64_t* pos_slots,
                                        int64_t* inv_pos, int64_t* idx_pruned, int32_t* raw_counts, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  SMOE_REQUIRE(n >= 1 && slot_base && group_end, "smoe_dispatch_plan_slots: needs n >= 1, a slot table and group_end");
  SMOE_REQUIRE(hdr_rows >= 0 && hdr_rows <= 1, "smoe_dispatch_plan_slots: hdr_rows must be 0 or 1");
  return plan_impl(idx, n, E, capacity, counts, offsets, pos_slots, inv_pos, idx_pruned, workspace, workspace_bytes, stream, 0,
                   group_end, raw_counts, slot_base, hdr_rows);
}

// ---- in-band headers of the static expert exchange (SURVEY.md N10 folded into N11) ---------------------------------------------
// The static exchange's send buffer holds one region per global expert g, rows [slot_base[g], slot_base[g + 1]): payload rows
// followed by ONE header row (the region's last).  The header's leading int32 words are
//   {rows kept for this group, rows routed to it before the clamp, the source rank's row count T, E, raw[0], ..., raw[E - 1]}
// -- raw = the SOURCE's pre-clamp counts for ALL E global experts.  The counts thereby travel WITH the rows: no count all-to-all
// per layer (upstream's expert_exchange), and every receiver learns every source's T and whole routing histogram -- the same
// [W, E] matrix on all ranks, which lets all ranks decide "somebody overflowed" and re-size the slots IDENTICALLY without a
// collective (ep.py).
namespace {
__global__ void ep_pack_headers_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ raw,
                                       const int32_t* __restrict__ slot_base, int G, int64_t row_bytes, int32_t t_rows,
                                       char* __restrict__ send) {
  // one workgroup; thread (g, j): header g, word j of the raw vector
  for (int i = threadIdx.x; i < G * (G + 4); i += blockDim.x) {
    const int g = i / (G + 4), j = i - g * (G + 4);
    int32_t* h = reinterpret_cast<int32_t*>(send + ((int64_t)slot_base[g + 1] - 1) * row_bytes);
    int32_t v;
    if (j == 0) v = counts ? counts[g] : 0;
    else if (j == 1) v = raw ? raw[g] : (counts ? counts[g] : 0);
    else if (j == 2) v = t_rows;
    else if (j == 3) v = G;
    else v = raw ? raw[j - 4] : (counts ? counts[j - 4] : 0);
    h[j] = v;
  }
}

// receiver: source w's block = rows [w * block_rows, (w + 1) * block_rows), block_rows = lbase[E_local]; inside it local expert e'
// owns [lbase[e'], lbase[e' + 1]) (header = the last row).  group l = w * E_local + e'.
__global__ void ep_unpack_headers_kernel(const char* __restrict__ recv, int W, int E_local, const int32_t* __restrict__ lbase,
                                         int64_t row_bytes, int E_tot, int32_t* __restrict__ starts, int32_t* __restrict__ ends,
                                         int32_t* __restrict__ stats) {
  const int G = W * E_local;
  const int64_t block_rows = lbase[E_local];
  for (int l = threadIdx.x; l < G; l += blockDim.x) {
    const int w = l / E_local, e = l - w * E_local;
    const int64_t r0 = (int64_t)w * block_rows + lbase[e], r1 = (int64_t)w * block_rows + lbase[e + 1];
    const int32_t* h = reinterpret_cast<const int32_t*>(recv + (r1 - 1) * row_bytes);
    int32_t c = h[0];
    const int32_t cap = (int32_t)(r1 - r0 - 1);
    c = c < 0 ? 0 : (c > cap ? cap : c);
    starts[l] = (int32_t)r0;
    ends[l] = (int32_t)r0 + c;
  }
  if (stats) {
    // stats[w] = {T of source w, raw[w][0 .. E_tot)} from the header of that source's first group
    for (int i = threadIdx.x; i < W * (E_tot + 1); i += blockDim.x) {
      const int w = i / (E_tot + 1), j = i - w * (E_tot + 1);
      const int32_t* h = reinterpret_cast<const int32_t*>(recv + ((int64_t)w * block_rows + lbase[1] - 1) * row_bytes);
      stats[i] = j == 0 ? h[2] : (h[3] == E_tot ? h[3 + j] : -1);
    }
  }
}
}  // namespace

extern "C" int smoe_ep_pack_headers(const int32_t* counts, const int32_t* raw_counts, const int32_t* slot_base, int G,
                                    int64_t row_bytes, int64_t t_rows, void* send, void* stream) {
  SMOE_REQUIRE(G >= 1 && G <= 1024 && row_bytes % 4 == 0 && row_bytes >= 16 + 4 * (int64_t)G,
               "smoe_ep_pack_headers: bad sizes G=%d row_bytes=%lld (a header holds 4 + G int32 words)", G, (long long)row_bytes);
  SMOE_REQUIRE(send && slot_base, "smoe_ep_pack_headers: null pointer");
  SMOE_REQUIRE(t_rows >= 0 && t_rows < (1ll << 31), "smoe_ep_pack_headers: t_rows out of range");
  hipLaunchKernelGGL(ep_pack_headers_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, raw_counts, slot_base, G,
                     row_bytes, (int32_t)t_rows, (char*)send);
  SMOE_CHECK_LAUNCH("smoe_ep_pack_headers");
  return 0;
}

extern "C" int smoe_ep_unpack_headers(const void* recv, int W, int E_local, const int32_t* local_base, int64_t row_bytes, int E_total,
                                      int32_t* starts, int32_t* ends, int32_t* stats, void* stream) {
  SMOE_REQUIRE(W >= 1 && E_local >= 1 && (int64_t)W * E_local <= 8192 && E_total >= 1 && E_total <= 1024 && row_bytes % 4 == 0 &&
               row_bytes >= 16 + 4 * (int64_t)E_total,
               "smoe_ep_unpack_headers: bad sizes W=%d E_local=%d E_total=%d row_bytes=%lld", W, E_local, E_total, (long long)row_bytes);
  SMOE_REQUIRE(recv && local_base && starts && ends, "smoe_ep_unpack_headers: null pointer");
  hipLaunchKernelGGL(ep_unpack_headers_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const char*)recv, W, E_local, local_base,
                     row_bytes, E_total, starts, ends, stats);
  SMOE_CHECK_LAUNCH("smoe_ep_unpack_headers");
  return 0;
}

static int scatter_rows_impl(const void* x, int x_dtype, const int64_t* pos, const float* scale, int64_t n_slots,
                             int k, int d, void* buf, int buf_dtype, int zero_unmapped, void* stream) {
  SMOE_REQUIRE(n_slots >= 0 && k >= 1 && d > 0 && d % 8 == 0, "smoe_scatter_rows: bad sizes n_slots=%lld k=%d d=%d",
               (long long)n_slots, k, d);
  if (n_slots == 0) return 0;
  SMOE_REQUIRE(x && pos && buf, "smoe_scatter_rows: null pointer");
  hipStream_t s = (hipStream_t)stream;
  switch (x_dtype) {
    case SMOE_F32: return scatter_dispatch_b<float>(x, pos, scale, n_slots, k, d, buf, buf_dtype, zero_unmapped, s);
    case SMOE_F16: return scatter_dispatch_b<f16>(x, pos, scale, n_slots, k, d, buf, buf_dtype, zero_unmapped, s);
    case SMOE_BF16: return scatter_dispatch_b<bf16_bits>(x, pos, scale, n_slots, k, d, buf, buf_dtype, zero_unmapped, s);
  }
  smoe_set_error("smoe_scatter_rows: bad x_dtype %d", x_dtype);
  return 1;
}

extern "C" int smoe_scatter_rows(const void* x, int x_dtype, const int64_t* pos, const float* scale, int64_t n_slots,
                                 int k, int d, void* buf, int buf_dtype, void* stream) {
  return scatter_rows_impl(x, x_dtype, pos, scale, n_slots, k, d, buf, buf_dtype, 0, stream);
}

// the same, and slots with pos[s] < 0 (no token: past the kept count under a capacity) are written as zero rows in the same pass
extern "C" int smoe_scatter_rows_fill(const void* x, int x_dtype, const int64_t* pos, const float* scale, int64_t n_slots,
                                      int k, int d, void* buf, int buf_dtype, void* stream) {
  return scatter_rows_impl(x, x_dtype, pos, scale, n_slots, k, d, buf, buf_dtype, 1, stream);
}

extern "C" int smoe_gather_combine(const void* y, int y_dtype, const int64_t* inv_pos, const float* score, int64_t T,
                                   int k, int d, const void* residual, void* out, int out_dtype, void* stream) {
  SMOE_REQUIRE(T >= 0 && k >= 1 && k <= 8 && d > 0 && d % 8 == 0, "smoe_gather_combine: bad sizes T=%lld k=%d d=%d",
               (long long)T, k, d);
  if (T == 0) return 0;
  SMOE_REQUIRE(y && inv_pos && score && out, "smoe_gather_combine: null pointer");
  hipStream_t s = (hipStream_t)stream;
  switch (y_dtype) {
    case SMOE_F32: return combine_dispatch_o<float>(y, inv_pos, score, T, k, d, residual, out, out_dtype, s);
    case SMOE_F16: return combine_dispatch_o<f16>(y, inv_pos, score, T, k, d, residual, out, out_dtype, s);
    case SMOE_BF16: return combine_dispatch_o<bf16_bits>(y, inv_pos, score, T, k, d, residual, out, out_dtype, s);
  }
  smoe_set_error("smoe_gather_combine: bad y_dtype %d", y_dtype);
  return 1;
}

namespace {
template <typename YT, typename NT>
int combine_ln_launch(const void* y, const int64_t* inv_pos, const float* score, int64_t T, int k, int d, const float* residual,
                      float* out, const float* gamma, const float* beta, float eps, void* xn, hipStream_t s) {
  const int grid = rows_grid(T);
#define CLN(KMAX, NJ) hipLaunchKernelGGL((gather_combine_ln_kernel<YT, NT, KMAX, NJ>), dim3(grid), dim3(256), 0, s, (const YT*)y, inv_pos, score, T, k, d, residual, out, gamma, beta, eps, (NT*)xn)
  const int nj = (d + 511) / 512;
  if (nj == 1) { if (k == 1) CLN(1, 1); else if (k == 2) CLN(2, 1); else CLN(4, 1); }
  else { if (k == 1) CLN(1, 2); else if (k == 2) CLN(2, 2); else CLN(4, 2); }
#undef CLN
  SMOE_CHECK_LAUNCH("smoe_gather_combine_ln");
  return 0;
}
}  // namespace

extern "C" int smoe_gather_combine_ln(const void* y, int y_dtype, const int64_t* inv_pos, const float* score, int64_t T, int k, int d,
                                      const float* residual, float* out, const float* gamma, const float* beta, float eps,
                                      void* xn, int xn_dtype, void* stream) {
  SMOE_REQUIRE(T >= 0 && k >= 1 && k <= 4 && d > 0 && d % 8 == 0 && d <= 1024, "smoe_gather_combine_ln: bad sizes T=%lld k=%d d=%d",
               (long long)T, k, d);
  if (T == 0) return 0;
  SMOE_REQUIRE(y && inv_pos && score && out && gamma && beta && xn, "smoe_gather_combine_ln: null pointer");
  SMOE_REQUIRE((y_dtype == SMOE_F16 || y_dtype == SMOE_BF16) && (xn_dtype == SMOE_F16 || xn_dtype == SMOE_BF16),
               "smoe_gather_combine_ln: y and xn must be f16 / bf16");
  hipStream_t s = (hipStream_t)stream;
  if (y_dtype == SMOE_F16) {
    if (xn_dtype == SMOE_F16) return combine_ln_launch<f16, f16>(y, inv_pos, score, T, k, d, residual, out, gamma, beta, eps, xn, s);
    return combine_ln_launch<f16, bf16_bits>(y, inv_pos, score, T, k, d, residual, out, gamma, beta, eps, xn, s);
  }
  if (xn_dtype == SMOE_F16) return combine_ln_launch<bf16_bits, f16>(y, inv_pos, score, T, k, d, residual, out, gamma, beta, eps, xn, s);
  return combine_ln_launch<bf16_bits, bf16_bits>(y, inv_pos, score, T, k, d, residual, out, gamma, beta, eps, xn, s);
}

extern "C" int smoe_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
  SMOE_REQUIRE(n >= 0, "smoe_cast: n < 0");
  if (n == 0) return 0;
  SMOE_REQUIRE(src && dst, "smoe_cast: null pointer");
  hipStream_t s = (hipStream_t)stream;
  switch (src_dtype) {
    case SMOE_F32: return cast_dispatch_d<float>(src, dst, dst_dtype, n, s);
    case SMOE_F16: return cast_dispatch_d<f16>(src, dst, dst_dtype, n, s);
    case SMOE_BF16: return cast_dispatch_d<bf16_bits>(src, dst, dst_dtype, n, s);
  }
  smoe_set_error("smoe_cast: bad src_dtype %d", src_dtype);
  return 1;
}
