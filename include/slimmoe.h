/* slimmoe.h -- C-ABI of libslimmoe_hip.so: MI355X (gfx950) kernels for the Switch-MoE ViT hot path.
 *
 * Drop-in boundary.  The reference (d0-rb/slim-switch-moe-vit) has no FFI of its own: its MoE
 * operator is `from fmoe import FMoETransformerMLP` (models/resMoE.py:6, ctor at 27-29, forward at
 * 121/143 and models/vision_transformer.py:321), and FastMoE in turn binds a pybind module
 * `fmoe_cuda` (expert_count, assign_pos, linear_forward, limit_by_capacity, prune_gate_by_capacity,
 * global_scatter/gather ...).  The entry points below are what a binding
 * for that op set would bind; each comment names the upstream op it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *    (torch tensors), contiguous, 16-byte aligned; nothing here allocates, frees or synchronises;
 *  - process-global state is limited to (i) the thread-local error string and (ii) one atomic bit per
 *    (kernel, device) recording that the kernel's dynamic-LDS limit has been raised on that device, plus the
 *    cached CU count per device.  smoe_init() sets all of (ii) for the CURRENT device; call it once per device
 *    before capturing launches into a hipGraph (after it no entry point touches function attributes).  Without
 *    it the first launch of a kernel on a device sets its attribute itself (thread-safe: the update is an
 *    atomic OR and setting the attribute twice is harmless);
 *  - entry points may be called concurrently from several host threads and on several streams; buffers handed
 *    to concurrent calls (workspaces, outputs) must be distinct;
 *  - all launches go to `stream` (a hipStream_t passed as void*);
 *  - return 0 on success, non-zero on error; smoe_last_error() gives the thread-local message;
 *  - dtype codes: SMOE_F32 / SMOE_F16 / SMOE_BF16.
 */
#ifndef SLIMMOE_H
#define SLIMMOE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SMOE_F32 = 0, SMOE_F16 = 1, SMOE_BF16 = 2 };
enum { SMOE_GATE_NAIVE = 0, SMOE_GATE_SWITCH = 1 };
enum { SMOE_EPI_NONE = 0, SMOE_EPI_GELU = 1, SMOE_EPI_GELU_GRAD = 2 };

/* smoe_set_reserved_cus(n): the persistent grouped GEMM (variants 9-14: one workgroup per CU for the whole launch, with all of the
 * CU's LDS and registers) leaves n CUs free from now on (process-wide; 0 <= n <= 128; default 0).  A kernel on ANOTHER stream cannot
 * share a CU with it, so RCCL's all-to-all under the expert-parallel micro-batch pipeline ("overlapped with the
 * local GEMM on a second HIP stream") only overlaps the GEMMs if some CUs are left to it; results do not depend on n. */
int smoe_set_reserved_cus(int n);

/* library ABI version (bumped on any signature change) */
int smoe_abi_version(void);
/* first 16 hex digits of the sha256 over the sources this binary was built from (csrc/Makefile HASHED): lets the loader
 * refuse, and build() replace, a shipped binary that does not belong to the shipped sources */
const char* smoe_build_id(void);
/* raise the dynamic-LDS limit of every kernel of the library on the current device (see Conventions); idempotent */
int smoe_init(void);
/* message for the last non-zero return on this thread */
const char* smoe_last_error(void);

/* ---- router ------------------------------------------------------------------------------------
 * Replaces fmoe NaiveGate.forward / SwitchGate.forward (gate = nn.Linear(d,E): evidenced by
 * models/resmoe_flop_hook.py:7-8; selected at models/resMoE.py:26-29).
 *   logits = x @ wg^T + bg           routing decided as if accumulated in f64 and rounded once to f32:
 *                                    f32 fast path with a rigorous error bound, f64 re-do of a token when a
 *                                    deciding gap is inside the bound (gate_kind | 0x100 forces f64 for all)
 *   NAIVE : (val, idx) = top-k(logits) [ties: lowest expert id; descending logit]; score = softmax(val)
 *   SWITCH: k == 1; p = softmax(logits + noise) over all E; idx = argmax; score = p[idx];
 *           probs (may be NULL) receives p [T,E] for the aux loss.
 * x [T,d] (x_dtype), wg [E,d] f32, bg [E] f32 or NULL, noise [T,E] f32 or NULL.
 * idx [T,k] i64, score [T,k] f32, logits_out [T,E] f32 or NULL.
 * Requires d % 8 == 0, d <= 2048, 1 <= k <= E, k <= 4.  workspace: smoe_router_workspace_bytes(T) bytes
 * (redo counter + list of tokens handed to the f64 pass).  The calls clear the counter words with a small launch of
 * their own -- unless the caller keeps ONE workspace per stream and says so (gate_kind | 0x200; smoe_gate_ln_router:
 * with_ln | 2): every redo pass leaves the counter words zero on its way out, so a workspace that was zero before its
 * first use stays ready, and the clearing launch in front of every router call disappears.                          */
size_t smoe_router_workspace_bytes(int64_t T);
int smoe_router_topk(const void* x, int x_dtype, const float* wg, const float* bg, const float* noise,
                     int64_t T, int d, int E, int k, int gate_kind,
                     int64_t* idx, float* score, float* logits_out, float* probs,
                     void* workspace, size_t workspace_bytes, void* stream);

/* LayerNorm + router fused (block glue, models/vision_transformer.py:321 `mlp(norm2(x))`): xn = LN(x)*gamma+beta
 * is written as a 16-bit image (xn16: f16/bf16, may be NULL) and / or f32 (xn32, may be NULL) and routed exactly
 * as smoe_router_topk routes xn.  Shapes covered: smoe_ln_router_supported(d, E, k) (k <= 4; E <= 8 with
 * d in {192, 384, 768, 1024}, E <= 32 with d in {768, 1024}); workspace as smoe_router_workspace_bytes(T).
 * chunk_hist (may be NULL; i32 [ceil(T / tok)][E], tok = smoe_router_chunk_hist_tokens(d, E, k) > 0): the per-chunk expert
 * histogram of the routing (count_by_gate folded into the router pass: the f32 pass walks contiguous chunks of `tok` tokens and
 * counts what it decides, the redo pass adds the tokens it decides) -- smoe_dispatch_plan_hist then builds the plan without its
 * own counting launch.  Shapes whose router writes none report tok = 0 and ignore the pointer.                             */
int smoe_ln_router_supported(int d, int E, int k);
int smoe_router_chunk_hist_tokens(int d, int E, int k);
int smoe_ln_router_topk(const void* x, int x_dtype, const float* ln_gamma, const float* ln_beta, float ln_eps,
                        void* xn16, int xn16_dtype, float* xn32, const float* wg, const float* bg, const float* noise,
                        int64_t T, int d, int E, int k, int gate_kind, int64_t* idx, float* score,
                        float* logits_out, float* probs, int32_t* chunk_hist, void* workspace, size_t workspace_bytes,
                        void* stream);

/* Token-skip gate of the residual-MoE block (models/resMoE.py:32-85 `Gate`; used at 126-145), fused with the LayerNorm
 * in front of it and -- for the MoE half -- with the router behind it: ONE pass over the activations replaces
 * `x = norm(x); mask = gate(x); skip_tk = x * mask[..,0]; tk = x * mask[..,1]` (+ the router's own pass over tk).
 *   xn   = with_ln ? LN(x) * gamma + beta : x                                  x [T,d] f32
 *   z    = <xn, gate_w> + gate_b;   skipped <=> sigmoid(z) > *threshold  <=>  z > logit(*threshold)
 *          (decided as if z were accumulated in f64: f32 pass with an error bound + f64 redo, like the router;
 *           threshold is read from DEVICE memory -- the module's buffer -- so no host sync; NULL = gate disabled)
 *   xn16 [T,d] (f16/bf16, may be NULL): the operand image of `tk`: xn, zeros for skipped tokens
 *   xn32 [T,d] (f32, may be NULL): the residual image `tk + skip_tk` = xn, plus zero_out[d] (may be NULL) on skipped
 *          rows -- the MoE output every all-zero row receives (smoe_zero_row_output), so the caller never dispatches them
 *   mask [T,2] f32 (skip, keep) (may be NULL);  *skip_count += skipped tokens (may be NULL)
 *   E > 0: NaiveGate routing of `tk` (a skipped token routes as the zero row it is: by the gate bias):
 *          idx / score as smoe_router_topk; idx_plan [T,k] = idx, or -1 for skipped tokens (feed THIS to
 *          smoe_dispatch_plan).  E == 0: gate only (the attention half).
 * Shapes: smoe_gate_ln_router_supported(d, E, k): d in {192, 384, 768, 1024}, E <= 8, k <= 4.
 * workspace as smoe_router_workspace_bytes(T).                                                              */
int smoe_gate_ln_router_supported(int d, int E, int k);
int smoe_gate_ln_router(const void* x, int x_dtype, int with_ln, const float* ln_gamma, const float* ln_beta, float ln_eps,
                        const float* gate_w, const float* gate_b, const float* threshold, void* xn16, int xn16_dtype,
                        float* xn32, const float* zero_out, const float* wg, const float* bg, int64_t T, int d, int E,
                        int k, int64_t* idx, int64_t* idx_plan, float* score, float* mask, int32_t* skip_count,
                        int32_t* chunk_hist /* as smoe_ln_router_topk's; counts idx_plan's dispatched entries; may be NULL */,
                        float* tk32 /* [T,d] f32 or NULL: the (normed) row, ZEROS for a skipped token -- `x * mask[..., 1:]` of
                                       models/resMoE.py:141, the masked f32 image the training path's router and scatter read */,
                        void* workspace, size_t workspace_bytes, void* stream);
/* What the MoE returns for an all-zero input row (every token the skip gate masks, resMoE.py:140-143):
 *   out[d] = sum_j score_j (W2[e_j] gelu(b1[e_j]) + b2[e_j]),  (e_j, score_j) = NaiveGate top-k of the gate bias bg.
 * w2 [E,d,h], b1 [E,h], b2 [E,d], bg [E] f32 (b1 / b2 / bg may be NULL = zeros).  Depends on parameters only.  */
/* Under expert parallelism bg covers all E global experts while w2 / b1 / b2 hold the E_local experts [e_base, e_base + E_local)
 * of this rank: out is then this rank's PARTIAL sum (chosen experts that live elsewhere contribute nothing); the ranks' partial
 * sums add up to the full row (one all-reduce per parameter version).  Single rank: e_base = 0, E_local = E.                  */
int smoe_zero_row_output(const float* bg, int E, int k, const float* w2, const float* b1, const float* b2, int d, int h,
                         int e_base, int E_local, float* out, void* stream);

/* smoe_skip_gate_bwd: backward of one gated half of the residual-MoE block in TRAINING (models/resMoE.py:68-77: hard masks with
 * the straight-through estimator; 131-143: tk = xn * m1, skip_tk = xn * m0, out = f(tk) + tk + skip_tk).  xn [T,d] f32 = the normed
 * activations, g_f [T,d] (f32 / f16 / bf16) = d loss / d tk as the operator's input, g_out [T,d] f32 = d loss / d out (or NULL),
 * mask [T,2] f32 = the forward's decisions (skip, keep: smoe_gate_ln_router), gate_w [d], gate_b [1] or NULL.  One pass, p recomputed:
 *     dz[t]  = -<g_f[t], xn[t]> p (1 - p),  p = sigmoid(<xn[t], gate_w> + gate_b)          (d loss / d gate logit; f32 [T] or NULL)
 *     dxn[t] = g_f[t] * keep[t] + g_out[t] + dz[t] * gate_w                                 (f32 [T,d]: into the LayerNorm backward)
 * gate_on = 0 (Gate.disable: constant masks, no gradient): dxn = g_f + g_out, dz = 0.  d in {192, 384, 768, 1024}.
 * The gate's parameter gradients follow from dz: dW = dz^T xn (smoe_gate_wgrad with E = 1), db = sum dz.                      */
int smoe_skip_gate_bwd(const float* xn, const void* g_f, int g_f_dtype, const float* g_out, const float* gate_w,
                       const float* gate_b, const float* mask, int gate_on, int64_t T, int d, float* dxn, float* dz, void* stream);
/* smoe_gate_ln_bwd: smoe_skip_gate_bwd, the LayerNorm backward in front of it and the gate's parameter gradients as ONE pass over the
 * half's INPUT x [T,d] f32 (the normed activations, p and the gate logit are recomputed from x in registers; nothing but x and the
 * forward's decisions is read):  dx [T,d] f32 = LayerNorm backward of dxn;  out f32 [3 d + 4] = dgamma [d] | dbeta [d] | dgate_w [d] |
 * dgate_b, 0, 0, 0;  dz f32 [T] or NULL.  gamma / beta f32 [d] (NULL = 1 / 0); the other operands as in smoe_skip_gate_bwd; d % 4 == 0,
 * d <= 1024.  Deterministic (per-workgroup partial rows, two-stage ordered reduction).                                              */
size_t smoe_gate_ln_bwd_workspace_bytes(int64_t T, int d);
int smoe_gate_ln_bwd(const float* x, const void* g_f, int g_f_dtype, const float* g_out, const float* gamma, const float* beta,
                     float eps, const float* gate_w, const float* gate_b, const float* mask, int gate_on, int64_t T, int d, float* dx,
                     float* dz, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* smoe_depth_scale_rows: stochastic depth's per-sample factor (timm DropPath as models/vision_transformer.py:308 uses it: x / keep * mask)
 * expanded to the sample's rows, the form the fused stores take it in (the GEMM's per-row combine scale): mask f32 [B] (0 / 1 draws)
 * -> factor f32 [B] = mask / keep (IEEE division), rows f32 [B * N] = factor[row / N].                                                 */
int smoe_depth_scale_rows(const float* mask, float keep, int64_t B, int N, float* factor, float* rows, void* stream);
/* ---- the embedding stage and the last LayerNorm of the ViT forward (models/vision_transformer.py:818-830) ----
 * smoe_patchify_cast:  images f32 [B, C, H, W] -> patch rows [B * (H/ph) * (W/pw), C * ph * pw] (f16 / bf16), row (b, gy, gx) =
 *                      images[b, :, gy*ph:(gy+1)*ph, gx*pw:(gx+1)*pw] flattened (c, py, px) -- the operand of the per-patch projection
 *                      GEMM (timm PatchEmbed: Conv2d with kernel == stride); pw % 4 == 0
 * smoe_embed_ln:       x32[b, 0] = cls_token + pos_embed[0],  x32[b, n] = tokens[b * P + n - 1] + pos_embed[n]  (n = 1 .. P)  -- the
 *                      f32 residual stream [B, P + 1, d] -- and, when xn != NULL, xn = LayerNorm(x32) in f16 / bf16 (block 0's norm1)
 *                      from the same pass.  tokens [B * P, d] f16 / bf16 (the projection GEMM's output), cls_token [d], pos_embed
 *                      [P + 1, d], gamma / beta [d] f32; d in {192, 384, 768, 1024}
 * smoe_layernorm_rows: LayerNorm of T f32 rows that start row_stride elements apart (the class-token rows x[:, 0] behind the last
 *                      block: row_stride = (P + 1) * d); out f32 [T, d]                                                              */
int smoe_patchify_cast(const float* images, int64_t B, int C, int H, int W, int ph, int pw, void* out, int out_dtype, void* stream);
int smoe_embed_ln(const void* tokens, int tok_dtype, const float* cls_token, const float* pos_embed, const float* ln_gamma,
                  const float* ln_beta, float ln_eps, int64_t B, int P, int d, float* x32, void* xn, int xn_dtype, void* stream);
int smoe_layernorm_rows(const float* x, int64_t row_stride, const float* gamma, const float* beta, float eps, int64_t T, int d,
                        float* out, void* stream);

/* LayerNorm alone (same arithmetic as the fused form; the `norm1` of models/vision_transformer.py:320 feeding the
 * attention GEMMs in 16 bit): d in {192, 384, 768, 1024}.                                              */
int smoe_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, float eps, int64_t T, int d,
                   void* out, int out_dtype, void* stream);

/* Self-attention forward of the block's attention half (models/vision_transformer.py:248-280), N <= 640 tokens
 * (one pass for N <= 256; above that -- ViT-L/16 @384: N = 577 -- key chunks with an online softmax),
 * head dim 64: out[b,n,h*64+:] = softmax(q k^T * scale) v with qkv [B,N,3,H,64] as the fused qkv projection
 * writes it; f16 / bf16.  Caller-side kernel, not part of the MoE operator.       */
int smoe_attention_supported(int N, int head_dim);
int smoe_attention_fwd(const void* qkv, void* out, int dtype, int B, int N, int H, int head_dim, float scale,
                       float* lse, void* stream);
/* lse (may be NULL; N <= 256): [B, H, N] f32, log2 of the softmax normaliser in the scaled-score domain, p = exp2(s scale
 * log2(e) - lse) -- what the backward recomputes the probabilities from.
 * smoe_attention_bwd: dqkv [B,N,3,H,64] (same fused layout) from qkv, the forward's out and lse, and dout [B,N,H*64]:
 * dV = P^T dO, dS = P (dO V^T - rowsum(dO O)) scale, dQ = dS K, dK = dS^T Q; scores recomputed per (image, head), N <= 256. */
int smoe_attention_bwd_supported(int N, int head_dim);
int smoe_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int dtype, int B, int N,
                       int H, int head_dim, float scale, void* stream);

/* ---- dispatch plan --------------------------------------------------------------------------------
 * Replaces fmoe_cuda.expert_count + cumsum + assign_pos (+ limit_by_capacity /
 * prune_gate_by_capacity) = fmoe count_by_gate / prepare_forward.
 * Deterministic stable counting sort of the n = T*k flat entries by expert id:
 *   counts[e]  = kept entries on e (min(raw, capacity) when capacity >= 0)
 *   offsets    = exclusive prefix of counts, offsets[E] = kept total
 *   pos[s]     = flat index t*k+j of slot s (ascending flat index inside an expert); -1 for s >= kept
 *   inv_pos[i] = slot of flat entry i, -1 if dropped (idx < 0 or over capacity)
 *   idx_pruned = idx with dropped entries set to -1 (may be NULL when capacity < 0)
 * idx values must lie in [-1, E).  workspace: smoe_dispatch_plan_workspace_bytes(n, E) bytes.   */
size_t smoe_dispatch_plan_workspace_bytes(int64_t n, int E);
int smoe_dispatch_plan(const int64_t* idx, int64_t n, int E, int64_t capacity,
                       int32_t* counts, int32_t* offsets, int64_t* pos, int64_t* inv_pos,
                       int64_t* idx_pruned, void* workspace, size_t workspace_bytes, void* stream);
/* smoe_dispatch_plan_hist: the same plan from a chunk histogram the fused router wrote (smoe_ln_router_topk / smoe_gate_ln_router
 * `chunk_hist`; hist_chunk = tokens per row x k flat entries): one launch, no counting pass over idx.  Falls back to
 * smoe_dispatch_plan when the table does not fit the fused kernel.  idx here is what the plan is built over (for the gated MoE
 * half: idx_plan, -1 = not dispatched -- exactly the entries the router left out of the histogram).                          */
int smoe_dispatch_plan_hist(const int64_t* idx, int64_t n, int E, int64_t capacity, const int32_t* hist, int hist_chunk,
                            int32_t* counts, int32_t* offsets, int64_t* pos, int64_t* inv_pos, int64_t* idx_pruned,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same plan in the PADDED layout of a capacity gate's static expert-parallel exchange (SURVEY.md section 8e, Appendix B
 * `cap` note: "[W, E_local, cap, d] exchange buffers, no count exchange / host sync needed"): expert e owns the slots
 * [e * slot_rows, (e + 1) * slot_rows) whatever its count, slot = e * slot_rows + rank (slot_rows >= capacity: the slot
 * size every rank agreed on, e.g. the capacity of the largest local batch; `capacity` is what THIS rank keeps).  pos_padded
 * has E * slot_rows entries (unused slots -1), inv_pos[i] is the padded slot, group_end[e] = e * slot_rows + counts[e] closes
 * expert e's row range (smoe_grouped_gemm's `group_end`); counts / offsets as above.  capacity >= 1; E <= 64. */
int smoe_dispatch_plan_padded(const int64_t* idx, int64_t n, int E, int64_t capacity, int64_t slot_rows, int32_t* counts,
                              int32_t* offsets, int32_t* group_end, int64_t* pos_padded, int64_t* inv_pos,
                              int64_t* idx_pruned,
                              int32_t* raw_counts /* i32 [E] or NULL: entries routed to each expert BEFORE the capacity clamp */,
                              void* workspace, size_t workspace_bytes, void* stream);

/* The same plan over a SLOT TABLE (the static expert exchange's send layout): expert e owns the slots [slot_base[e],
 * slot_base[e + 1]) (slot_base i32 [E + 1], device memory), the last hdr_rows (0 / 1) of them reserved for the in-band header; it
 * keeps min(region - hdr_rows, capacity if capacity >= 0) entries.  pos_slots has slot_base[E] entries (unused ones -1), inv_pos[i]
 * is the slot, group_end[e] = slot_base[e] + counts[e]; raw_counts as above.  E <= 64. */
int smoe_dispatch_plan_slots(const int64_t* idx, int64_t n, int E, int64_t capacity, const int32_t* slot_base, int hdr_rows,
                             int32_t* counts, int32_t* offsets, int32_t* group_end, int64_t* pos_slots, int64_t* inv_pos,
                             int64_t* idx_pruned, int32_t* raw_counts, void* workspace, size_t workspace_bytes, void* stream);

/* ---- in-band headers of the static expert exchange ------------------------------------------------------------------------
 * Replaces fmoe_cuda.expert_exchange (the count all-to-all in front of global_scatter) for the static exchange:
 * the counts travel INSIDE the token all-to-all.  Send buffer = one region per global expert g, rows [slot_base[g],
 * slot_base[g + 1]) (blocks of E_local consecutive regions go to one peer): payload rows + ONE header row (the region's last),
 * whose leading int32 words are {kept rows of the group, rows routed to it before the clamp, the source's row count, G, the
 * source's pre-clamp counts of ALL G experts}.  row_bytes >= 16 + 4 G.
 * smoe_ep_pack_headers writes the G header rows of a send buffer (counts / raw_counts i32 [G] from smoe_dispatch_plan_slots;
 * raw_counts NULL = counts; counts NULL = a rank without rows);
 * smoe_ep_unpack_headers reads the W * E_local headers of a RECEIVED buffer ([source rank][local expert] order; local_base i32
 * [E_local + 1] = this rank's experts' region offsets inside one source's block, local_base[E_local] = rows per block):
 *   starts[l], ends[l] (i32 [W * E_local])  -> smoe_grouped_gemm's offsets / group_end
 *   stats (may be NULL) i32 [W][1 + E_total] = {source w's row count, source w's pre-clamp counts of all experts} -- every rank
 *   receives the same matrix, so all ranks can decide "somebody overflowed" (and re-size the slots) identically, without a collective */
int smoe_ep_pack_headers(const int32_t* counts, const int32_t* raw_counts, const int32_t* slot_base, int G, int64_t row_bytes,
                         int64_t t_rows, void* send, void* stream);
int smoe_ep_unpack_headers(const void* recv, int W, int E_local, const int32_t* local_base, int64_t row_bytes, int E_total,
                           int32_t* starts, int32_t* ends, int32_t* stats, void* stream);

/* ---- token scatter (MOEScatter.forward local part: index_select(x, 0, pos // k)) ----
 * buf[s,:] = cast(x[pos[s] / k, :]) for every slot s < n_slots with pos[s] >= 0; other rows untouched.
 * x [T,d], buf [n_slots,d]; d % 8 == 0.  scale (f32 [T*k], may be NULL) multiplies row s by scale[pos[s]]
 * (backward of the combine: dY[s] = score * dout[token]).                                           */
int smoe_scatter_rows(const void* x, int x_dtype, const int64_t* pos, const float* scale, int64_t n_slots, int k, int d,
                      void* buf, int buf_dtype, void* stream);
/* the same, and every slot with pos[s] < 0 (no token: past the kept count under a capacity) is written as a zero row in the same
 * pass (the training path's buffers: instead of clearing the whole buffer first).                                             */
int smoe_scatter_rows_fill(const void* x, int x_dtype, const int64_t* pos, const float* scale, int64_t n_slots, int k, int d,
                           void* buf, int buf_dtype, void* stream);

/* ---- gather + combine (MOEGather.forward + bmm(gate_score, y)) -------------------
 * out[t,:] = sum_j score[t,j] * y[inv_pos[t*k+j], :]   (a dropped entry contributes 0)
 * optional fused residual add (the `+ tk + skip_tk` of models/resMoE.py:143): out[t,:] += residual[t,:]
 * when residual != NULL (same dtype as out).  y [n_slots,d] (y_dtype), out [T,d] (out_dtype).       */
int smoe_gather_combine(const void* y, int y_dtype, const int64_t* inv_pos, const float* score,
                        int64_t T, int k, int d, const void* residual, void* out, int out_dtype,
                        void* stream);
/* smoe_gather_combine_ln: the same combine (+ residual, f32 out [T,d]) AND LayerNorm(out row) -> xn [T,d] (f16 / bf16) in one pass:
 * the next block's `norm1` of exactly the row this call produces (models/vision_transformer.py:320) -- the expert-parallel
 * return path saves a 155-MB read per layer.  y f16 / bf16; k <= 4; d % 8 == 0, d <= 1024; gamma / beta f32 [d].             */
int smoe_gather_combine_ln(const void* y, int y_dtype, const int64_t* inv_pos, const float* score, int64_t T, int k, int d,
                           const float* residual, float* out, const float* gamma, const float* beta, float eps, void* xn,
                           int xn_dtype, void* stream);

/* ---- grouped (variable-batch) expert GEMM on MFMA ----------------------------------------------------
 * Replaces fmoe_cuda.linear_forward = MOELinear / FMoELinear: for each local
 * expert e, rows r in [offsets[e], offsets[e+1]):
 *     out[r, :] = epilogue( A[r, :] @ W[e]^T + bias[e] )          W [E,N,K] row-major ([out,in])
 * One launch for all experts; the tile -> (expert, m-tile) map is derived on device from `offsets`
 * (no host sync; empty experts are fine).  m_rows_max = upper bound on offsets[E] (rows allocated in A/out).
 * ab_dtype: SMOE_F16 / SMOE_BF16 (MFMA 16-bit inputs, f32 accumulate) or SMOE_F32 (exact f32 MFMA).
 * bias f32 [E,N] or NULL.  epilogue SMOE_EPI_GELU = exact-erf GELU (models/resMoE.py:23-25).
 * Optional fused combine for top-1 (row_map != NULL): row r is stored to out[row_map[r], :]
 * multiplied by row_scale[row_map[r]] if row_scale != NULL  (MOEGather + bmm for k = 1).
 * Optional fused residual (residual != NULL, same dtype / shape as out): the stored value is
 * residual[orow, :] + value -- the `x + mlp(norm2(x))` add of models/vision_transformer.py:321.
 * Optional fused scatter (a_gather != NULL; variants 4-8): row r of the GEMM reads A[a_gather[r] / a_div, :]
 * instead of A[r, :] -- MOEScatter's index_select(x, pos // k) folded into the operand DMA (A = the token matrix).
 * SMOE_EPI_GELU_GRAD (backward of the activation, fused into the dgrad GEMM): `residual` then holds the saved
 * pre-activations H and the stored value is value * gelu'(H[r, :]).
 * Optional group -> expert map (group_expert != NULL, i32 [G]): `offsets` then delimits G row groups and
 * group g uses W[group_expert[g]] / bias[group_expert[g]] (expert-parallel receive layout: one group per
 * (source rank, local expert)); n_experts = leading dimension of W / bias.
 * Optional separate row ranges (group_end != NULL, i32 [G]; persistent kernel only): group g is the rows
 * [offsets[g], group_end[g]) and `offsets` holds G entries -- the padded [W, E_local, cap] receive buffer of a capacity
 * gate's static exchange, whose slots are only partly filled (smoe_dispatch_plan_padded); tiles past a group's end are
 * never scheduled, so the padding costs no MFMA work.
 * out_rows = rows allocated in `out` (and `residual`): every row_map value must be below it.  0 = not stated (taken as
 * m_rows_max without a row map); the persistent kernel then keeps flat addressing for f32 outputs instead of the
 * buffer-addressed epilogue (whose descriptors need the bound: out_rows * N * 4 < 2 GiB).
 * Requires K*sizeof(ab) % 128 == 0 and N % 8 == 0.
 * variant: 9 = production choice (the 4 below as a PERSISTENT kernel, one workgroup per CU walking tiles, with the
 * direct-store epilogue for plain 16-bit outputs; 14 = the same with the LDS-staged epilogue everywhere, A/B reference;
 * 10-13 force tile height / schedule); 4 = production choice (8-wave ping-pong kernel, LDS-DMA staging; picks the 256- or 320-row tile by the
 * number of workgroup rounds and, for K >= 2048, the deep prefetch schedule); 5 / 6 force the 320- / 256-row tile,
 * 7 / 8 the deep schedule on the 256- / 320-row tile; 1-3 earlier LDS-DMA structures and 0 the register-staged
 * kernel (the only one for f32 operands or K % 64 != 0; chosen automatically then) are kept as A/B references.
 * All variants compute the same function.                                                              */
int smoe_grouped_gemm(const void* A, const void* W, const float* bias, const int32_t* offsets,
                      const int32_t* group_expert, int G, int n_experts, int64_t m_rows_max, int K, int N,
                      int ab_dtype, int epilogue, const int64_t* row_map, const float* row_scale,
                      const void* residual, const int64_t* a_gather, int a_div,
                      void* out, int64_t out_rows, int out_dtype, int variant, const int32_t* group_end, void* stream);

/* OPTIONAL -- only in a library built with `make FFN=-DSMOE_FFN_FUSED` (it measured slower than the two smoe_grouped_gemm launches in
 * every scheduler design, profiles/r04_fused_ffn.md, and left the default build in round 5; bit-identical to them).
 * smoe_expert_ffn: the whole expert FFN of one MoE layer with the top-1 combine -- fmoe_cuda.linear_forward x 2 with the activation
 * between them, MOEGather and the combine (models/resMoE.py:143 -> FastMoE `_Expert.forward`) -- as ONE persistent
 * launch:   H[r]   = gelu(X[a_gather[r] / a_div] W1[e]^T + b1[e])                      (16-bit, [m_rows_max, d_hidden]; caller's buffer)
 *           out[row_map[r]] = residual[row_map[r]] + row_scale[row_map[r]] (H[r] W2[e]^T + b2[e])       (f32, [out_rows, d_out])
 * for r in row group g's range [offsets[g], offsets[g+1]), e = group_expert[g] (or g).  Every number is computed exactly as by the
 * two smoe_grouped_gemm launches (variant 9: 320-row tiles, same accumulation order): bit-identical results.  What the single
 * launch buys: the workgroups draw tiles of BOTH GEMMs from one list (a GEMM-2 tile starts when the GEMM-1 tiles of its rows
 * have stored, counted per m-tile in `workspace`), so neither GEMM ends in a partly filled round of workgroups.
 * a_gather / row_map / row_scale / residual / b1 / b2 / group_expert may be NULL.  `workspace`: smoe_expert_ffn_workspace_bytes
 * bytes, ZERO before the first launch; every launch leaves it zero again (word 17 = error flag: set if a wait on a row counter ran
 * out -- never in a healthy launch).  Returns -1 (no error set) for shapes outside this launch's reach (operands not f16 / bf16,
 * out not f32, d_in or d_hidden % 64, > 63 groups, operands >= 4 GiB, out >= 2 GiB): issue the two smoe_grouped_gemm then. */
size_t smoe_expert_ffn_workspace_bytes(int64_t m_rows_max, int G);
int smoe_expert_ffn(const void* X, const int64_t* a_gather, int a_div, const void* W1, const float* b1, void* H, const void* W2,
                    const float* b2, const int32_t* offsets, const int32_t* group_expert, int G, int n_experts,
                    int64_t m_rows_max, int d_in, int d_hidden, int d_out, int ab_dtype, const int64_t* row_map,
                    const float* row_scale, const void* residual, void* out, int64_t out_rows, int out_dtype, void* workspace,
                    size_t workspace_bytes, void* stream);

/* smoe_grouped_gemm_gelu_keep: the first expert linear of the TRAINING forward (fmoe_cuda.linear_forward + the activation, whose
 * input autograd keeps): pre_out = A W^T + bias and out = gelu(pre_out), both [m_rows, N] in the operand dtype (f16 / bf16), from one
 * epilogue.  Returns -1 (no error set) for shapes outside the persistent kernel's reach (K % 64 != 0, operands >= 4 GiB, G > 63):
 * use smoe_grouped_gemm(SMOE_EPI_NONE) + smoe_gelu then.  group_end (optional, i32 [G]): separate row ranges [offsets[g], group_end[g])
 * as in smoe_grouped_gemm -- the slots of the static expert exchange in training; rows outside the ranges are neither read nor written. */
int smoe_grouped_gemm_gelu_keep(const void* A, const void* W, const float* bias, const int32_t* offsets,
                                const int32_t* group_expert, const int32_t* group_end, int G, int n_experts, int64_t m_rows_max,
                                int K, int N, int ab_dtype, void* pre_out, void* out, void* stream);

/* ---- backward pieces (fmoe_cuda.linear_backward and the adjoints of scatter / gather) ---------
 * smoe_gelu:            dst = gelu_erf(src), n % 8 == 0 (training forward keeps the pre-activations)
 * smoe_rowdot:          dscore[i] = <dout[i / k, :], y[inv_pos[i], :]>, 0 for dropped entries (i < n = T*k)
 * smoe_pad_offsets:     offsets_pad[e] = sum_{e'<e} round_up(count[e'], 64)
 * smoe_transpose_pad:   src [n_rows, C] expert-sorted -> dst [C, Lp] K-major, per-expert ranges at offsets_pad, zero pad
 * smoe_grouped_wgrad:   out[e] (f32 [R1,R2]) = PT[:, range e] @ QT[:, range e]^T   (PT [R1,Lp], QT [R2,Lp], 16-bit)
 * smoe_group_colsum:    out[e, c] = sum over expert e's rows of src[:, c]  (bias gradients); C % 4 == 0; two passes
 *                       (512-row chunk partials in `workspace`, then a per-expert sum in chunk order: deterministic);
 *                       n_rows_max >= offsets[E] sizes the launch; group_end (optional, i32 [E]): expert e's rows are
 *                       [offsets[e], group_end[e]) -- separate ranges (slots of the static expert exchange), n_rows_max >=
 *                       the sum of their lengths                                                                        */
int smoe_gelu(const void* src, void* dst, int dtype, int64_t n, void* stream);
int smoe_rowdot(const void* dout, int dout_dtype, const void* y, int y_dtype, const int64_t* inv_pos,
                int64_t n, int k, int d, float* dscore, void* stream);
int smoe_pad_offsets(const int32_t* offsets, int E, int32_t* offsets_pad, void* stream);
/* smoe_split_offsets: every row group [offsets[g], offsets[g + 1]) cut into S pseudo-groups of whole 64-row chunks (the last ones may be
 * empty): out i32 [G * S + 1].  smoe_grouped_wgrad_rows' grid is groups x output tiles, so a weight gradient over few long groups
 * (fmoe_cuda.linear_backward's grad_W at DeiT-Tiny's dims: 24 tiles) is computed as S partial products per expert, summed in piece
 * order by the caller (deterministic). */
int smoe_split_offsets(const int32_t* offsets, int G, int S, int32_t* out, void* stream);
int smoe_transpose_pad(const void* src, int dtype, const int32_t* offsets, const int32_t* offsets_pad, int E,
                       int64_t n_rows, int C, int Lp, void* dst, void* stream);
/* smoe_switch_gate_bwd: gradient of the SwitchGate's score and load-balance loss w.r.t. the router logits (fmoe.gates.SwitchGate:
 * score = softmax(logits)[idx], aux = E sum_e frac_e prob_e), one pass over [T, E]:  g[t,e] = coef[e] + (e == idx[t] ? dscore[t] : 0),
 * dlogits[t,e] = probs[t,e] (g[t,e] - sum_j probs[t,j] g[t,j]).  coef [E] f32 (device; = E * frac_e / kept, smoe_switch_aux's) or
 * NULL, times *coef_scale (device scalar: d loss / d aux; NULL = 1); dscore [T] f32 or NULL, idx [T] int64 (entries outside
 * [0, E) select nothing).                                                                                                       */
int smoe_switch_gate_bwd(const float* probs, const int64_t* idx, const float* dscore, const float* coef, const float* coef_scale,
                         int64_t T, int E, float* dlogits, void* stream);
/* smoe_switch_aux: the SwitchGate's load-balance loss (fmoe.gates.SwitchGate), forward: aux [1] = E sum_e frac_e prob_e,
 * frac_e = counts[e] / kept, prob_e = sum_t probs[t,e] / kept, kept = max(sum_e counts[e], 1); coef [E] = E frac_e / kept
 * (= d aux / d probs[t,e] for every t: what smoe_switch_gate_bwd takes).  probs f32 [T, E], counts i32 [E] (the plan's kept counts);
 * E <= 256; two deterministic launches (chunk column sums, then one workgroup).                                                 */
size_t smoe_switch_aux_workspace_bytes(int64_t T, int E);
int smoe_switch_aux(const float* probs, const int32_t* counts, int64_t T, int E, float* aux, float* coef, void* workspace,
                    size_t workspace_bytes, void* stream);
/* smoe_transpose_cast: dst[b][c][r] = (dst_dtype) src[b][r][c] for b < B; R % 64 == 0, C % 64 == 0.  The backward pass reads every
 * expert weight [E, out, in] a second time as [E, in, out] (FastMoE: `MOELinear.backward` -> fmoe_cuda.linear_backward contracts
 * grad_out with the weight over `out`); this makes that 16-bit image straight from the f32 master in one pass.               */
int smoe_transpose_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int R, int C, void* stream);
int smoe_grouped_wgrad(const void* PT, const void* QT, int ab_dtype, const int32_t* offsets_pad, int E, int R1,
                       int R2, int Lp, float* out, void* stream);
/* smoe_grouped_wgrad_rows: the same weight gradients as smoe_grouped_wgrad, straight from the token-major operands
 * (P [n_rows,R1], Q [n_rows,R2], f16 / bf16, expert-sorted rows; offsets i32 [E+1] = plain row ranges, any lengths):
 * out[e] (f32 [R1,R2]) = sum_{r in expert e} P[r,:]^T Q[r,:].  No transposed copies: the MFMA fragments are read
 * from LDS with transposing reads.  R1, R2 multiples of 8; zero16 = 16 zero bytes in device memory.  group_end (optional, i32 [E]):
 * expert e's rows are [offsets[e], group_end[e]) -- separate ranges; nothing behind group_end[e] is read. */
int smoe_grouped_wgrad_rows(const void* P, const void* Q, int ab_dtype, const int32_t* offsets, const int32_t* group_end, int E,
                            int R1, int R2, const void* zero16, float* out, void* stream);
/* smoe_gate_wgrad: router weight gradient dWg [E, C] f32 = dl^T x, dl [n_rows, E] f32 (d loss / d logits), x [n_rows, C]
 * f32 / f16 / bf16; E <= 16, C % 4 == 0; HBM-bound two-pass weighted column sum (what torch's matmul backward of the gate
 * nn.Linear computes, models/resmoe_flop_hook.py:7-8 names that layer).  db (f32 [E], may be NULL) = column sums of dl, the gate
 * bias' gradient, from the same pass (x's "ones column"). */
size_t smoe_gate_wgrad_workspace_bytes(int64_t n_rows, int E, int C);
int smoe_gate_wgrad(const float* dl, const void* x, int x_dtype, int64_t n_rows, int E, int C, float* out, float* db,
                    void* workspace, size_t workspace_bytes, void* stream);
/* smoe_zero_group_fold: the training path's zero-row groups (tokens the token-skip gate masked, models/resMoE.py:141 `x * mask`: all-zero
 * rows, every one routed by the gate bias alone to expert gmap[E + j]) need no weight-gradient GEMM: group E + j adds the rank-1 term
 * colsum(dY_g) (x) A_row to dW2[gmap[E + j]] (the rows of A = gelu(b1[e]) are identical), and an expert's bias gradients are the column sums
 * of its own group plus those of the zero groups that use it.  cs2 f32 [E + Z, d], cs1 f32 [E + Z, h] = smoe_group_colsum of dY / dH over
 * all E + Z groups; A [n_rows, h] f16 / bf16 / f32; offsets i32 [E + Z + 1]; gmap i32 [E + Z]; dW2 f32 [E, d, h] updated IN PLACE;
 * db2 f32 [E, d], db1 f32 [E, h] written (either may be NULL).  h % 4 == 0.  Groups are folded in index order.                        */
int smoe_zero_group_fold(const float* cs2, const float* cs1, const void* A, int a_dtype, const int32_t* offsets, const int32_t* gmap,
                         int E, int Z, int d, int h, int64_t n_rows, float* dW2, float* db2, float* db1, void* stream);
size_t smoe_group_colsum_workspace_bytes(int64_t n_rows_max, int E, int C);
int smoe_group_colsum(const void* src, int dtype, const int32_t* offsets, const int32_t* group_end, int E, int64_t n_rows_max, int C,
                      float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- expert exchange (expert parallelism over RCCL / xGMI) ----------------------------------------------------------
 * Replaces fmoe_cuda.ensure_nccl / expert_exchange / global_scatter / global_gather (reached from
 * models/resMoE.py:27-29 when world_size > 1).  A context owns ONE RCCL communicator (built from a unique-id blob that
 * rank 0 creates with smoe_unique_id and the caller distributes by whatever means it has -- MPI, a file, torch's store),
 * ONE communication stream and a ring of completion events.  Every exchange is enqueued on the context's stream behind an
 * event recorded on the caller's `stream` (so it sees the send buffer the compute stream produced) and ends with an event
 * the caller's stream waits on: immediately when wait != 0, or later -- in between the compute stream is free to run other
 * work (the expert GEMMs of another micro-batch) while rows move over xGMI.  Every exchange has a ticket (1, 2, ...:
 * smoe_a2a_last_ticket right after posting it); smoe_a2a_wait_ticket waits for THAT exchange (the ring holds the last 16;
 * an older ticket waits for the exchange that re-used its slot, which is later on the same in-order stream), so exchanges
 * of several micro-batches in flight can be waited for in any order; smoe_a2a_wait = the latest one.  Arguments are
 * checked before the RCCL group opens and a failing call inside the group closes it before returning.  All peers' transfers of one
 * call are posted as one RCCL group: on the point-to-point xGMI mesh every link then carries its pair's rows concurrently.
 * RCCL is resolved at run time from the librccl.so already in the process (or the system's): no link-time dependency.
 *   smoe_a2a_counts : send_counts[w*E_local + e] = rows this rank routes to rank w's local expert e  ->  recv_counts[w*E_local
 *                     + e] = rows rank w routes to this rank's local expert e   (i32 [W*E_local], device memory)
 *   smoe_a2a_tokens : all-to-all-v of d-element rows; `send` holds the rows for rank 0, 1, ... (send_rows[w] each: the
 *                     expert-sorted send buffer has that order), `recv` receives recv_rows[w] rows from rank w, rank-major
 *                     (the [source rank][local expert] layout smoe_grouped_gemm takes through group_expert);
 *                     send_rows / recv_rows are HOST i64 [W]
 * wait = SMOE_A2A_INLINE posts the exchange on the caller's `stream` itself (no communication stream, no events, no ticket):
 * in stream order with the kernels either side of it, for a caller that has nothing to run beside the exchange.             */
#define SMOE_A2A_INLINE 2
typedef struct smoe_ctx smoe_ctx;
int smoe_unique_id_bytes(void);
int smoe_unique_id(void* out_id);
int smoe_ctx_create(const void* unique_id, int world_size, int rank, smoe_ctx** out);   /* on the current device */
int smoe_ctx_destroy(smoe_ctx* ctx);
void* smoe_ctx_comm_stream(smoe_ctx* ctx);
int smoe_ctx_world_size(smoe_ctx* ctx);
int smoe_ctx_rank(smoe_ctx* ctx);
int smoe_a2a_counts(smoe_ctx* ctx, const int32_t* send_counts, int32_t* recv_counts, int E_local, void* stream, int wait);
int smoe_a2a_tokens(smoe_ctx* ctx, const void* send, const int64_t* send_rows, void* recv, const int64_t* recv_rows, int d,
                    int dtype, void* stream, int wait);
int smoe_a2a_wait(smoe_ctx* ctx, void* stream);
int64_t smoe_a2a_last_ticket(smoe_ctx* ctx);
int smoe_a2a_wait_ticket(smoe_ctx* ctx, int64_t ticket, void* stream);

/* ---- backward of the dense half of the block (training step, engine.py:52-74) -------------------------
 * The linears' backward needs no kernel of its own: dgrad = smoe_grouped_gemm with one row group on the transposed weight
 * image (smoe_transpose_cast), wgrad = smoe_grouped_wgrad_rows with one group, bias gradient = smoe_group_colsum.
 * smoe_layernorm_bwd: backward of nn.LayerNorm(d, eps) (models/vision_transformer.py:303-311) from x alone (statistics
 * recomputed): dx = rstd (g - mean(g) - xhat mean(g xhat)) [+ dres: the gradient arriving over the residual connection that
 * bypasses the norm, fused], g = dy gamma; dgamma_dbeta [2 d] = (sum dy xhat, sum dy), deterministic (partial rows per
 * workgroup in `workspace`, added in order).  x / dx / dres f32 [T, d]; dy f32 / f16 / bf16; d % 4 == 0, d <= 1024.   */
/* smoe_gate_dgrad: the router linear's input gradient dx [T, d] = dl [T, E] W [E, d] (f32 in, out f32 / f16 / bf16): K = E is too
 * thin for the matrix cores, the kernel is bound by its [T, d] store.                                                            */
int smoe_gate_dgrad(const float* dl, const float* w, int64_t T, int E, int d, void* out, int out_dtype, void* stream);
size_t smoe_layernorm_bwd_workspace_bytes(int64_t T, int d);
int smoe_layernorm_bwd(const float* x, const void* dy, int dy_dtype, const float* gamma, const float* dres, float eps, int64_t T,
                       int d, float* dx, float* dgamma_dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* ---- optimizer side of the training step (engine.py:68-74: timm NativeScaler around torch.optim.AdamW; SURVEY.md 8f
 * rank 3).  Everything stays on the device -- loss scale, non-finite flag, clip coefficient, step count -- so a step has
 * no host sync, and a gradient is read twice in all (norm pass, update pass) instead of four times.
 * smoe_grad_sumsq : partial[b] = sum over block b (smoe_grad_sumsq_blocks(n) blocks of 16384 elements) of
 *                   (g * *inv_scale)^2, deterministic; *found_inf = 1 when any scaled-back element is inf / nan
 *                   (GradScaler.unscale_'s check; the caller zeroes found_inf once per step); inv_scale may be NULL (= 1)
 * smoe_adamw_step : torch.optim.AdamW arithmetic on f32 p / m / v with the gradient g (f32 / f16 / bf16) multiplied by
 *                   *grad_mult (NULL = 1: e.g. inv_scale x clip coefficient): p *= 1 - lr wd; m = lerp(m, g, 1 - b1);
 *                   v = b2 v + (1 - b2) g^2; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), t = *step
 *                   (device f32, already advanced); the whole update is skipped when *found_inf != 0 (found_inf may be NULL)
 * smoe_amp_update : GradScaler.update(): found_inf ? (scale *= backoff, tracker = 0)
 *                                                 : (++tracker == interval ? (scale *= growth, tracker = 0) : -)
 * smoe_step_advance: *step += 1 unless *found_inf != 0                                                                  */
int64_t smoe_grad_sumsq_blocks(int64_t n);
int smoe_grad_sumsq(const void* g, int g_dtype, int64_t n, const float* inv_scale, float* partial, float* found_inf,
                    void* stream);
int smoe_adamw_step(float* p, const void* g, int g_dtype, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, const float* step, const float* grad_mult, const float* found_inf,
                    void* stream);
/* Multi-tensor forms of the two passes: ONE launch for every gradient / parameter of the step (torch's `foreach` / fused
 * AdamW play this role upstream).  tab = int64 [5][n_tensors] in device memory: rows p, g, exp_avg, exp_avg_sq (addresses) and
 * n (elements); the sumsq form reads rows g and n only.  hyp = f32 [2][n_tensors]: lr, weight_decay.  blk = int32 [2][n_blocks]:
 * the tensor of workgroup b and the 16,384-element block inside it, blocks in tensor order; partial[b] receives block b's sum.
 * Same arithmetic and the same partial sums as the single-tensor forms.  shadow (may be NULL) = int64 [2][n_tensors]: the address of
 * a 16-bit image of parameter t (0 = none) and its dtype code (SMOE_F16 / SMOE_BF16): the step writes the updated parameter there
 * as well, so the MFMA operand copies the next forward reads need no cast pass of their own (autocast re-casts every weight on
 * every forward upstream; engine.py:52-53).  A step skipped by found_inf leaves parameter and image untouched.               */
int smoe_grad_sumsq_multi(const int64_t* tab, int n_tensors, const int32_t* blk, int64_t n_blocks, int g_dtype,
                          const float* inv_scale, float* partial, float* found_inf, void* stream);
int smoe_adamw_step_multi(const int64_t* tab, const float* hyp, int n_tensors, const int32_t* blk, int64_t n_blocks,
                          int g_dtype, float beta1, float beta2, float eps, const float* step, const float* grad_mult,
                          const float* found_inf, const int64_t* shadow, void* stream);
int smoe_amp_update(float* scale, float* growth_tracker, const float* found_inf, float growth_factor, float backoff_factor,
                    int growth_interval, void* stream);
int smoe_step_advance(float* step, const float* found_inf, void* stream);

/* ---- small helpers ------------------------------------------------------------------------------------
 * elementwise cast between dtypes (weight shadow copies; not on the per-step path)                  */
int smoe_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLIMMOE_H */
