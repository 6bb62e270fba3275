/*
 * ufm_hip.h -- C ABI of libufm_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * UFM dense-correspondence inference hot path.
 *
 * The reference (labrat97/UFM) is pure Python; its FFI for this path is "torch ops called from
 * nn.Module.forward".  Each entry point below replaces the torch op sequence at the cited
 * reference call site (paths relative to /root/reference/uniflowmatch; lines marked [U] are the
 * absent third-party `uniception` blocks whose only in-reference trace is the call site).
 * The reference-side binding a maintainer would add is the ctypes stub shown in INTEGRATION.md
 * (ufm_amd/hip.py is that stub).
 *
 * Conventions
 *   - plain device pointers + sizes; no torch types.  All pointers are DEVICE pointers unless noted.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - every call is asynchronous on `stream`, allocates nothing, never synchronises, and is
 *     hipGraph-capturable; workspaces are passed in by the caller.
 *   - return value: 0 on success, negative UFM_ERR_* otherwise; ufm_last_error() gives the text
 *     (thread-local, host pointer).
 *   - "rows" are tokens or pixels; activations are row-major [rows][channels] (tokens-major /
 *     NHWC), so transformer outputs feed the DPT head without any transpose.
 *   - bf16 = raw uint16_t bit pattern (round-to-nearest-even from fp32).
 */
#ifndef UFM_HIP_H
#define UFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UFM_OK 0
#define UFM_ERR_ARG (-1)    /* shape / alignment contract violated (nothing was launched) */
#define UFM_ERR_LAUNCH (-2) /* hipLaunch reported an error */

#define UFM_ABI_VERSION 2 /* 2: relu_in of the conv entry points became a flag word (bit 1 = replicate padding) */

int ufm_abi_version(void);
const char* ufm_last_error(void);
/* Returns the gfx arch string of device 0's code object this library was built for ("gfx950"). */
const char* ufm_built_arch(void);

/* ---- data types for out_dtype / in_dtype arguments ---- */
#define UFM_F32 0
#define UFM_BF16 1
/* "split" fp32-class format: two bf16 planes [2][rows][C]; plane 0 = hi = bf16(x), plane 1 = lo =
 * bf16(x - hi); the pointer addresses plane 0 and plane 1 follows at rows*C elements. */
#define UFM_BF16X2 2
#define UFM_BF16X2_IL 3 /* split bf16, INTERLEAVED per 32-channel chunk: [rows][C / 32][hi 32 | lo 32] (round 6; ufm_layernorm output, ufm_gemm_bf16x3_il operand) */

/* ---- activation codes ---- */
#define UFM_ACT_NONE 0
#define UFM_ACT_GELU 1 /* exact erf GELU ([U] Mlp.act, encoder + info-sharing blocks) */
#define UFM_ACT_RELU 2

/* =====================================================================================
 * Pre-processing.  Replaces models/base.py:215-229 (uint8 -> float/255 -> (x-mean)/std) fused
 * with the im2col view of the 14x14/14 patch-embed conv ([U] DINOv2 PatchEmbed.proj, call site
 * models/ufm.py:308-309).  Output row r = (img, py, px) patch; column c*P*P + i*P + j (the
 * flattened Conv2d weight order), zero padded to `kpad` columns.
 *   in_layout: 0 = [B][H][W][3] (BHWC), 1 = [B][3][H][W] (BCHW)
 *   in_dtype : 0 = uint8 (normalise with mean/std), 1 = float32 (affine a*x+b per channel:
 *              identity or the re-normalisation of base.py:212-213)
 *   scale3[3], shift3[3]: HOST floats.  uint8 input:  value = (x/255 - shift3[c]) / scale3[c]  with
 *              shift3 = mean, scale3 = std (the exact expression of base.py:228-229);
 *              float32 input: value = x*scale3[c] + shift3[c].
 * ===================================================================================== */
int ufm_patchify(const void* img, int in_dtype, int in_layout, int B, int H, int W, int patch,
                 const float* scale3, const float* shift3, void* out, int out_dtype, int kpad,
                 void* stream);

/* Antialiased bilinear resize, align_corners=False (utils/flow_resizing.py:313-326,
 * F.interpolate(..., mode="bilinear", antialias=True)); input is normalised like ufm_patchify,
 * output float32 [B][3][Ho][Wo].  Exact identity when (H,W)==(Ho,Wo). */
int ufm_resize_antialias(const void* img, int in_dtype, int in_layout, int B, int H, int W,
                         const float* scale3, const float* shift3, float* out, int Ho, int Wo,
                         float* tmp /* B*3*H*Wo floats */, void* stream);

/* =====================================================================================
 * bf16 MFMA GEMM, C[M,N] = A[M,K] . W[N,K]^T with fused epilogue.  Replaces every nn.Linear of
 * the encoder / info-sharing transformer blocks under the reference's bf16 autocast
 * (models/base.py:273; [U] Attention.qkv/proj, Mlp.fc1/fc2, proj_embed; patch-embed conv as GEMM).
 *   v   = acc + bias[n]                       (bias may be NULL)
 *   v   = act(v)                              (UFM_ACT_*)
 *   v   = v * gamma[n]                        (LayerScale; gamma may be NULL)
 *   v  += res[(row % res_row_mod) * ldres + n] (fp32; res may be NULL; res_row_mod<=0 -> row)
 *   out[orow * ldo + n] = v  as out_dtype, where orow = row, or, if out_row_group>0,
 *          (row / out_row_group) * (out_row_group + 1) + 1 + row % out_row_group   (cls slot skipped)
 * res may alias out (in-place residual update).  Requirements: K % 64 == 0, N % 128 == 0,
 * lda/ldw % 8 == 0, A/W 16-byte aligned.  M arbitrary (tail rows masked).
 * ===================================================================================== */
int ufm_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K,
                  const float* bias, int act, const float* gamma, const float* res, int ldres,
                  int res_row_mod, void* out, int out_dtype, int ldo, int out_row_group,
                  void* stream);

/* ufm_gemm_bf16 with RoPE-2D fused into the epilogue ("fused QKV+RoPE"): columns [0, rope_cols) of the bf16 output -- the q and k
 * heads of a QKV / Q / K|V projection of the cross-attention info-sharing variant ([U] custom_positional_encoding = CroCo RoPE2D,
 * call site models/ufm.py:193) -- are rotated per 64-wide head on the fp32 accumulator, after bias and gamma and before the bf16
 * rounding:  out[j] = v[j] cos[t][j] + v[j ^ 16] sin[t][j],  t = row % rope_mod, tables fp32 [rope_mod][64] (sin carries the sign;
 * built by the host from the token grid, see ufm_rope2d).  Needs out_dtype UFM_BF16, no residual / activation / row re-mapping. */
int ufm_gemm_bf16_rope(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K,
                       const float* bias, int act, const float* gamma, const float* res, int ldres,
                       int res_row_mod, void* out, int out_dtype, int ldo, int out_row_group,
                       const float* rope_cos, const float* rope_sin, int rope_mod, int rope_cols, void* stream);
/* The standalone, in-place form for every activation format (numerics "parity": UFM_F32, "precise": UFM_BF16X2 with the lo plane
 * at rows*ld elements; UFM_BF16: the reference point of the fused form): columns [col0, col0 + ncols) of x[rows][ld]. */
int ufm_rope2d(void* x, int dtype, int rows, int ld, int col0, int ncols, const float* cos_table, const float* sin_table,
               int mod, void* stream);

/* Two-source attention (cross-attention): softmax(scale * q k^T) v with queries from one buffer and keys / values from
 * another ([U] CrossAttention of the cross-attention info-sharing variant; also self-attention on any column layout).
 *   q: [B*Nq][ldq], k / v: [B*Nk][ldkv], out: [B*Nq][ldo]; head h = columns [64 h, 64 h + 64) of each pointer; Nq != Nk allowed.
 * _bf16: bf16 operands (fast), _f32: exact-fp32 MFMA (parity), _bf16x3: UFM_BF16X2 planes (precise; the lo planes follow at
 * B*Nq*ldq, B*Nk*ldkv and B*Nq*ldo elements).
 * ufm_cross_attention_bf16, contract by `scale` (it forwards to ufm_attention_bf16_strided since round 4):
 *   scale > 0: the register-staged kernel applies it; ldq / ldkv % 8 == 0, ldo % 4 == 0, q / k / v 16-byte and out 8-byte aligned.
 *   scale == 0 is NOT an error: it means "q is already multiplied by softmax_scale * log2(e)" (the projection GEMM's epilogue
 *   does that) and selects the persistent LDS-DMA kernel, which needs ldo % 8 == 0 and a 16-byte aligned out as well.  A caller
 *   that passes 0 by mistake gets an UNSCALED softmax, not UFM_ERR_ARG.  scale < 0 is rejected. */
int ufm_cross_attention_bf16(const uint16_t* q, int ldq, const uint16_t* k, const uint16_t* v, int ldkv, uint16_t* out, int ldo,
                             int B, int Nq, int Nk, int H, float scale, void* stream);
/* The general form of the bf16 kernel: batch item b's queries start at row b * q_batch_rows of q, its keys / values at row
 * b * kv_batch_rows of k / v, its output at row b * out_batch_rows of out (ufm_cross_attention_bf16 = the densely packed case
 * q_batch_rows = out_batch_rows = Nq, kv_batch_rows = Nk).  This is how the LAST joint-attention layer runs its view-1 queries
 * only against both views' keys (the reference decodes view 1 only, /root/reference/uniflowmatch/models/ufm.py:637-641): q = the
 * pair's first Np rows of the qkv buffer (q_batch_rows = 2 Np), k / v = all 2 Np rows.
 * scale == 0: q is pre-scaled by softmax_scale * log2(e) (the projection GEMM's epilogue does it) and the call runs on the
 * persistent LDS-DMA kernel of ufm_attention_bf16 (csrc/attention_bf16_pw.hip); scale > 0: the register-staged kernel applies it
 * (densely packed batch items only).  ldq / ldkv / ldo % 8 == 0, 16-byte aligned pointers. */
int ufm_attention_bf16_strided(const uint16_t* q, int ldq, int q_batch_rows, const uint16_t* k, const uint16_t* v, int ldkv,
                               int kv_batch_rows, uint16_t* out, int ldo, int out_batch_rows, int B, int Nq, int Nk, int H,
                               float scale, void* stream);
int ufm_cross_attention_f32(const float* q, int ldq, const float* k, const float* v, int ldkv, float* out, int ldo, int B,
                            int Nq, int Nk, int H, float scale, void* stream);
int ufm_cross_attention_bf16x3(const uint16_t* q, int ldq, const uint16_t* k, const uint16_t* v, int ldkv, uint16_t* out, int ldo,
                               int B, int Nq, int Nk, int H, float scale, void* stream);

/* Launch objective of the tile-height choice in ufm_gemm_bf16 / ufm_gemm_bf16x3 / ufm_conv2d_nhwc_bf16x3 (round 5).  By default a launch is
 * dispatched as if it ran alone on the chip (lowest latency).  A caller that drives SEVERAL streams concurrently -- the engine's two micro-batch
 * streams, the way the reference's users run independent pair batches side by side -- flags those streams once: launches on a flagged stream of
 * 8192 rows or more then use full-height 8-phase tiles only (fewer, more efficient workgroups: less CU time taken from the neighbour stream;
 * +1...+2.4 % pairs/s "fast", +2.2...+2.8 % "precise" in the two-stream pipeline, -0.8 % if the stream in fact runs alone).  Results are bitwise
 * the same either way.  Flags are REFERENCE-COUNTED per handle (round 6): every successful on = 1 is paired with one on = 0 by its
 * holder (the engine un-flags its streams when it is destroyed), the flag goes with the last reference; on = 0 for an unknown handle is a
 * no-op.  At most 32 distinct streams (UFM_ERR_ARG beyond: the caller should say so, the launch policy silently stays "latency" on that
 * stream); the null stream cannot be flagged.  A stale flag -- a holder that never gives it back, or a destroyed stream whose handle the
 * runtime hands out again -- changes the tile policy of launches on that handle only, never a result bit.  Thread-safe. */
int ufm_hint_concurrent_stream(void* stream, int on);
/* Tuning hooks (tests / tools only; the product path never calls them).
 * variant: 0 = auto (cost model per shape), 1 = 128x128 kernel, 4 = 256x256 8-phase kernel, 5 = hybrid (8-phase on the rows
 * that fill whole rounds of the chip + 128x128 on the rest).  flags (timing diagnostics, results are wrong): 2 = no DMA,
 * 4 = no epilogue traffic, 8 = direct (un-staged) epilogue of the 128x128 kernel (results right), 16 / 32 = lda / ldw taken
 * as 0 (cache-hit probe), bits 8..15 = grouped-rasterization height of the 8-phase kernel (results right).
 * tile_rows: 0 = cost model, 160 / 192 / 224 / 256 = pin the row height of the 8-phase tiles (results are bitwise the same).
 * Round 5: variant 6 = the 256x128 four-wave kernel with two resident workgroups per CU (gemm_bf16_pair.hip; any N % 128 == 0,
 * K >= 128; bitwise the other kernels); flags bits 16..22 = first-round start delay of a CU's second resident workgroup in
 * that kernel (units of s_sleep(64); results right); variant 7 = the persistent 8-phase kernel (gemm_bf16_8ph_persist.hip) on every
 * bf16-output launch of whole 256-row tiles with a compile-time epilogue (auto uses it on whole rounds of the chip only); flags bit 28 =
 * never the persistent kernel in auto; flags bits 24..27 = flip the auto dispatch's four pair-kernel rules (bit 24: the K = 768 bf16-output
 * shapes, 25: N = 768 read-modify-write, 26: the encoder's QKV below 16 000 rows -- these three ON by default --, 27: everywhere, off);
 * flags bit 23 = the serial read-modify-write read-out of rounds 1-4 (results right); flags bits 29 / 30 = the latency / the CU-time objective of
 * the tile-height choice on every stream, whatever ufm_hint_concurrent_stream says. */
int ufm_debug_set_gemm_variant(int variant);
int ufm_debug_set_gemm_flags(int flags);
int ufm_debug_set_gemm_tile_rows(int rows);
/* Lab (round 6): deterministic 2-way split-K of the read-modify-write launches of ufm_gemm_bf16 (fp32 residual in place: proj / fc2) on `stream`:
 * while the stream has a workspace, such launches with K >= min_k (K % 128 == 0, N % 256 == 0) run two workgroups per 256 x 256 tile, one per K
 * half; the second to arrive (agent-scope arrival counter) adds the two fp32 partial tiles in half order and runs the epilogue.  The cut is at
 * K / 2 -- a function of the layer alone, never of M -- so a row's bits do not depend on its batch neighbours; they differ from the unsplit sum's
 * last bits.  ws: >= 64 KiB of ZEROED counters + tiles x 512 KiB; ws = NULL removes the stream's entry.  An A/B arm (tools/lab/gemm_splitk_ab.py,
 * profiles/r06/gemm_splitk_ab.log), not a dispatch rule. */
int ufm_debug_set_gemm_splitk(void* stream, void* ws, long long bytes, int min_k);
/* The lab flag words' field tables (ufm_amd/csrc/lab_flags.h: ONE table of {name, shift, width} per word, pairwise disjoint at compile
 * time; every consumer reads its field masked to its width; the two setters return UFM_ERR_ARG for any bit outside the table).
 * word 0 = ufm_debug_set_gemm_flags, word 1 = ufm_debug_set_conv_variant; index 0.. until UFM_ERR_ARG.  Host only, no GPU call. */
int ufm_debug_lab_field(int word, int index, const char** name, int* shift, int* width);
/* In-kernel stamps of the 8-phase and pair GEMM kernels (diagnostic instantiations; the shipped kernels execute no stamp):
 * while `buf` is set, launches with the fc1 (bias + GELU -> bf16) or proj / fc2 (bias + LayerScale + fp32 residual) epilogue
 * write 8 x uint64 per workgroup b < rows: {b | HW_ID << 32, XCC_ID | LDS_ALLOC << 32, s_memtime at entry, after the K loop,
 * after the last store completed, s_memrealtime (100 MHz) at entry, at the end, s_memtime after the prologue}.  The buffer
 * is read by no kernel.  buf = NULL, rows = 0 turns it off.  tools/lab/gemm_stamps.py. */
int ufm_debug_set_gemm_stamps(unsigned long long* buf, int rows);
/* The same for the 256 px x 256 cout 8-phase kernel of ufm_conv2d_nhwc_bf16x3 / ufm_gemm_bf16x3 (every launch of it while set). */
int ufm_debug_set_conv_stamps(unsigned long long* buf, int rows);
/* Tuning hook, bit mask: bit 0 -- ufm_attention_bf16 (scale == 0 form) with 2 waves per workgroup instead of 4 (default);
 * bit 1 -- ufm_attention_bf16x3 / ufm_cross_attention_bf16x3 on the round-1 kernel (attention_bf16x3.hip) instead of round 5's
 * LDS-DMA kernel (attention_bf16x3_pw.hip); bit 2 -- that kernel with 8 waves per workgroup (A/B); bit 3 -- that kernel with the per-tile
 * running maximum of rounds 1-5 (bitwise the round-1 kernel) instead of round 6's FIXED softmax reference (the row's maximum over its first key
 * tile, -m_ref as the C operand of the first QK^T MFMA, a cold 2^-64 shift if a row sum passes 2^64: the same softmax with other roundings,
 * tested against fp64; -1.2 of 7.1 VALU instructions per MFMA in an issue-bound kernel). */
int ufm_debug_set_attn_variant(int v);
/* Tuning hook for ufm_conv2d_nhwc_bf16x3: 0 = auto, 1 = 128-row kernels only, 2 = 256x256 8-phase kernel wherever applicable,
 * 4 = the 256 px x 128 cout two-resident-workgroups kernel of round 5 (conv_bf16x3_pair.hip) wherever applicable,
 * 3 = 128-row kernels only and never the deep (4-stage) ring; + 16 = the serial per-pass residual read-out of rounds 1-4 in the
 * epilogue instead of round 5's grouped loads (bitwise the same results; A/B); bits 8..11 = nf in 5..8: pin the 8-phase kernel's
 * tile height to 32 nf rows (with variant 0 or 2; bitwise the same results). */
int ufm_debug_set_conv_variant(int v);
/* Tuning hook, bit mask (default 1): bit 0 -- ufm_upsample_bilinear_nhwc (split format) on the LDS-tiled kernel where applicable;
 * bit 1 -- ufm_dpt_tail_fused with the plain stage-A tile of rounds 1-4 (2-way LDS bank conflicts in stage B) instead of round 5's
 * half-swapped one (bitwise the same results; A/B). */
int ufm_debug_set_upsample_variant(int tiled);

/* `ufm infer` post-processing: warp the target image into the source frame with the predicted
 * flow -- [R] utils/viz.py:11-59 warp_image_with_flow: F.grid_sample(bilinear, align_corners=False, zeros padding)
 * at clip(x + flow_x, 0, Wt-1), clip(y + flow_y, 0, Ht-1).  target: HWC uint8 (tgt_dtype 0) or fp32 (1), 3 channels;
 * flow: fp32 planar [2][H][W] (the layout of UFMFlowFieldOutput.flow_output[b]); out: fp32 [H][W][3].
 * mask_mode 0: none; 1: out *= (mask > 0.5) (viz.py:56-57); 2: out = mask*out + (1-mask)*fill (cli.py:141-143). */
int ufm_warp_bilinear(const void* target, int tgt_dtype, int Ht, int Wt, const float* flow, int H, int W,
                      const float* mask, int mask_mode, float fill, float* out, void* stream);

/* =====================================================================================
 * LayerNorm over the channel dim, eps inside the sqrt ([U] Block.norm1/norm2, encoder .norm,
 * info-sharing .norm; nn.LayerNorm(eps=1e-6)).  x: fp32 [*, D] rows of stride ldx.
 * Output row i is computed from input row (row_index ? row_index[i] : i): this is how the cls
 * token is dropped and views are re-ordered (models/ufm.py:313, :596-615) without a copy.
 * ===================================================================================== */
/* out_dtype: UFM_F32, UFM_BF16, UFM_BF16X2 (lo plane at out + rows_out*ldo elements) or UFM_BF16X2_IL (ufm_layernorm only, D % 256 == 0:
 * [rows_out][ldo / 32][hi 32 | lo 32], i.e. 2 ldo elements per row; the operand format of ufm_gemm_bf16x3_il). */
int ufm_layernorm(const float* x, int ldx, const int32_t* row_index, int rows_out, int D,
                  const float* weight, const float* bias, float eps, void* out, int out_dtype,
                  int ldo, void* stream);
/* The same, writing a row slice of a larger UFM_BF16X2 buffer: the lo plane is out_plane elements behind the hi plane
 * (>= rows_out*ldo; ignored for the other output types). */
int ufm_layernorm_slice(const float* x, int ldx, const int32_t* row_index, int rows_out, int D,
                        const float* weight, const float* bias, float eps, void* out, int out_dtype,
                        int ldo, long long out_plane, void* stream);

/* Residual update fused in front of the LayerNorm:  x[r, :] += gamma[:] * branch[r, :]  (fp32 math; branch = the bf16
 * output of the preceding Attention.proj / Mlp.fc2 Linear, gamma = LayerScale or NULL), x is written back, then
 * out[r, :] = LayerNorm(x[r, :]) as ufm_layernorm.  This is the reference's own GPU arithmetic for
 * `x = x + ls(branch)` under bf16 autocast (models/base.py:273: the Linear returns bf16, the LayerScale multiply and the
 * add run in fp32 on the fp32 residual stream; [U] Block.forward), and it moves the residual stream's fp32
 * read-modify-write out of the GEMM epilogue into this HBM-bound kernel. */
int ufm_add_layernorm(float* x, int ldx, const uint16_t* branch, int ldb, const float* gamma, int rows, int D,
                      const float* weight, const float* bias, float eps, void* out, int out_dtype, int ldo,
                      void* stream);

/* out[r, :] = src[(r % src_rows), :]  for the rows listed: fills the cls-token rows
 * (cls_token + pos_embed[0]) of the token buffer: out row = g*(group)+0.  ([U] DINOv2 prepare_tokens) */
int ufm_fill_rows(float* out, int ldo, int n_groups, int group_stride_rows, const float* src, int D,
                  void* stream);

/* out[r, :] = x[row_index[r], :]: compacts selected fp32 rows of the residual stream.  Used by the LAST joint-attention layer,
 * which only needs its view-1 rows (the reference decodes view 1 only, /root/reference/uniflowmatch/models/ufm.py:637-641). */
int ufm_gather_rows_f32(const float* x, int ldx, const int32_t* row_index, int rows, int D, float* out, int ldo, void* stream);

/* out[orow(r), :] = a[r, :] + tab[r % tab_mod, :] with the same orow() as ufm_gemm_bf16: fp32 token
 * assembly (patch tokens + pos-embed, cls slot skipped; view positional encoding) for the
 * "parity" numerics mode where the GEMM runs on the fp32 conv kernel.  tab may be NULL. */
int ufm_add_rows(const float* a, int lda, const float* tab, int ldtab, int tab_mod, float* out,
                 int ldo, int out_row_group, int rows, int D, void* stream);

/* =====================================================================================
 * Fused multi-head attention forward, non-causal, head_dim 64 ([U] Attention.forward:
 * softmax(q k^T / sqrt(d)) v; under autocast the reference runs SDPA in bf16).
 * qkv: bf16 [B*N][3*H*64] laid out (which, head, d) per row = the raw output of the qkv Linear.
 * out: bf16 [B*N][H*64].  Flash-style: K/V tiles staged in LDS, online softmax in registers,
 * S = Q K^T and O = P V on MFMA (32x32x16 bf16, fp32 accumulate); N arbitrary (tail masked).
 * scale > 0: softmax scale applied in-kernel.  scale == 0: the Q columns are PRE-SCALED by
 * softmax_scale*log2(e) (ufm_gemm_bf16's per-column gamma on the QKV projection does it before the
 * bf16 rounding) and the faster log2-domain kernel with deferred rescaling runs.
 * ===================================================================================== */
int ufm_attention_bf16(const uint16_t* qkv, uint16_t* out, int B, int N, int H, float scale,
                       void* stream);

/* Diagnostics for the default ufm_attention_bf16 kernel (scale == 0 form): the same kernel with s_memtime stamps
 * around the four MFMA/softmax slots of every key tile (cdna_hip_programming.md section 7, "In-kernel stamps").
 * diag: 16 x uint64 per workgroup = {sync + DMA issue, slot 0, slot 1, slot 2, slot 3, whole kernel (shader cycles),
 * s_memrealtime ticks (100 MHz) of the whole kernel, key tiles, units, and per unit seam: state init, prologue wait + K
 * fragment reads, next-unit DMA/Q issue, drain, epilogue compute, store issue, 0}.  waves = 2 or 4 per workgroup.
 * Never on the product path. */
int ufm_debug_attention_stamps(const uint16_t* qkv, uint16_t* out, int B, int N, int H, int waves,
                               unsigned long long* diag, void* stream);

/* fp32 variant (numerics mode "parity": exact-fp32 MFMA, same tiling). qkv/out are float. */
int ufm_attention_f32(const float* qkv, float* out, int B, int N, int H, float scale, void* stream);

/* =====================================================================================
 * fp32 implicit-GEMM convolution on NHWC activations (exact-fp32 MFMA 32x32x2), the fp32
 * "island" of the DPT heads (models/ufm.py:635-642; [U] DPTFeature / DPTRegressionProcessor:
 * 1x1, 3x3/s1, 3x3/s2 Conv2d, ConvTranspose2d with kernel==stride).  Also serves as the fp32
 * dense GEMM (H=1, W=rows, 1x1) for the "parity" numerics mode.
 *   in  : fp32 [B][H][W][Cin]         weight: fp32 [Cout][KH][KW][Cin]   (pre-packed)
 *   out : fp32 [B][Ho][Wo][Cout], Ho = (H + 2*pad - KH)/stride + 1
 *   v = conv(relu_in ? relu(in) : in) + bias ; v = act(v) ; v *= gamma[n] ; v += res1 + res2
 *   relu_in is a flag word: bit 0 = ReLU on the input, bit 1 (value 2) = padding_mode "replicate" (out-of-range taps read the
 *   nearest edge pixel; [U] MoGeConvFeature's convolutions) instead of zero padding; the same in ufm_conv2d_nhwc_bf16x3.
 *   (res1/res2: fp32, same shape as out, may be NULL, may alias out)
 *   shuffle > 0: ConvTranspose(k=s=shuffle) mode: weight is [(kh,kw,co)][Cin] with
 *   Cout = shuffle*shuffle*Co; element (b,y,x,(kh,kw,co)) is stored at
 *   out[b][y*s+kh][x*s+kw][co]  (KH=KW=1, stride 1).
 * Requirements: Cin % 32 == 0, Cout % 32 == 0 (Co % 4 == 0 in shuffle mode).
 * ===================================================================================== */
int ufm_conv2d_nhwc_f32(const float* in, int B, int H, int W, int Cin, const float* weight,
                        int Cout, int KH, int KW, int stride, int pad, int relu_in,
                        const float* bias, int act, const float* gamma, const float* res1,
                        const float* res2, int shuffle, float* out, int out_dtype_unused,
                        const float* zero_page /* >=128 B of zeros */, void* stream);

/* Same convolution with fp32-class accuracy on the bf16 matrix cores (numerics mode "fast"): in /
 * weight / res / out are in the UFM_BF16X2 split format and every product is evaluated as
 * hi*hi + hi*lo + lo*hi (3 bf16 MFMAs, fp32 accumulate; ~2^-17 relative error per dot product --
 * tighter than the TF32 cuDNN applies to the reference's fp32 island on NVIDIA by default).
 * weight: [2][Cout][KH][KW][Cin] pre-split at pack time.  Same fused epilogue minus gamma.
 * out_relu (optional, may be NULL): a second output relu(out) in the same layout -- [U] ResidualConvUnit
 * is conv2(relu(conv1(relu(x)))) + x, so the producer of x writes x (for the skip) AND relu(x) (conv1's
 * input), which takes the ReLU out of the consumer's MFMA loop (bit-identical to relu_in=1). */
int ufm_conv2d_nhwc_bf16x3(const uint16_t* in, int B, int H, int W, int Cin, const uint16_t* weight,
                           int Cout, int KH, int KW, int stride, int pad, int relu_in,
                           const float* bias, int act, const uint16_t* res1, const uint16_t* res2,
                           int shuffle, uint16_t* out, uint16_t* out_relu, const uint16_t* zero_page,
                           int passes /* 3 = bf16x3; 1 = hi planes only: a plain bf16 convolution with fp32 accumulation,
                                         the arithmetic bf16 autocast gives the reference's UNet (ufm.py:915-917) */,
                           void* stream);

/* `groups` convolutions of IDENTICAL geometry in one launch: the two DPT heads are the same graph with different weights
 * (/root/reference/uniflowmatch/models/ufm.py:553-556, 637-642), so each of their layers is one grid of twice the tiles --
 * fuller rounds on the small maps, no stream ping-pong.  B, H, W: per group.  weight: [2][groups][Cout][KH][KW][Cin]; bias:
 * [groups][Cout] ([groups][Cout / shuffle^2] in shuffle mode); in: [2][groups * B] images, or [2][B] images read by every group
 * when in_shared != 0 (the pyramid level both heads start from); res1 / res2 / out / out_relu: [2][groups * B] images, group-major.
 * Every group's result is bit-identical to its own ufm_conv2d_nhwc_bf16x3 call.  groups == 1 is that call. */
int ufm_conv2d_nhwc_bf16x3_grouped(const uint16_t* in, int groups, int in_shared, int B, int H, int W, int Cin,
                                   const uint16_t* weight, int Cout, int KH, int KW, int stride, int pad, int relu_in,
                                   const float* bias, int act, const uint16_t* res1, const uint16_t* res2, int shuffle,
                                   uint16_t* out, uint16_t* out_relu, const uint16_t* zero_page, int passes,
                                   void* splitk_ws /* may be NULL */, long long splitk_ws_bytes, void* stream);
/* Round 6: an INTERLEAVED copy of a convolution's split weights -- [groups * Cout][KH * KW * Cin / 32][hi 32 | lo 32], the values of the planar
 * [2][groups * Cout][KH * KW * Cin] tensor `planar` -- registered against that tensor's address.  Launches of ufm_conv2d_nhwc_bf16x3(_grouped) whose
 * `weight` is `planar` and that run on the row-window halo kernel (3x3 / stride 1 / pad 1 layers on the 8-phase tile) stage W from the copy:
 * whole 128-byte LDS-DMA rows.  Results are bitwise the same.  il = NULL removes the entry; the caller removes it before freeing either
 * buffer.  At most 512 entries.  Host only. */
int ufm_conv_x3_register_interleaved_weights(const void* planar, const void* il);
/* Deterministic split-K.  With a workspace (zero-filled when allocated; the kernel leaves its counters zero again) the K loop of
 * the small-map, long-K layers (<= 1600 output pixels per image and >= 48 K-tiles of 32 channels: the 19^2 / 37^2 layers of the DPT
 * heads) is cut into 2-6 consecutive ranges, one workgroup each; fp32 partial tiles meet in the workspace and the last workgroup
 * of a tile adds them in range order and runs the epilogue.  The factor depends on the layer's geometry only -- not on B or
 * groups -- so a pixel's bits do not depend on the batch it is computed in; they DO differ (in the last bits) from the same
 * layer without a workspace.  One workspace per stream.  Size: ufm_conv_x3_splitk_ws_bytes (0 = this layer is never split). */
long long ufm_conv_x3_splitk_ws_bytes(int groups, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);

/* =====================================================================================
 * Numerics mode "precise": the transformer trunk on the split format (fp32-class accuracy at bf16 MFMA rates / 3).
 * Replaces the same nn.Linear / SDPA call sites as ufm_gemm_bf16 / ufm_attention_bf16 ([U] Attention.qkv/proj,
 * Mlp.fc1/fc2, Attention.forward), evaluated so that the end-to-end flow stays within the north star's 1e-3 px of the
 * fp32 reference path (what the reference computes on a CPU, models/base.py:272-274 with autocast disabled).
 *
 * ufm_gemm_bf16x3: out[M][N] = epilogue(A[M][K] . W[N][K]^T), every product hi*hi + hi*lo + lo*hi, fp32 accumulate.
 *   A: UFM_BF16X2 [2][M][K] (ufm_layernorm / ufm_attention_bf16x3 / a previous ufm_gemm_bf16x3 write it);
 *   W: UFM_BF16X2 [2][N][K], pre-split at pack time.   v = acc + bias[n]; v = act(v) (GELU: branch-free erf, |err| <= 1.5e-7); v *= gamma[n];
 *   GELU accuracy contract of every split-format epilogue (this entry point, ufm_conv2d_nhwc_bf16x3): ABSOLUTE error <= 4.4e-7
 *   against the fp64 erf GELU over [-8, 8] (measured, tests/test_kernels_gpu.py::test_split_format_gelu_epilogue_against_fp64_gelu);
 *   relative error <= 6e-7 for x >= -1 and 5.6e-5 on [-3, -1]; NO relative bound below x = -3 (gelu < 4e-3 there and the
 *   erfc polynomial bounds the absolute error only) -- an order below the 2^-17 resolution of the split store for the fp32-scale
 *   values the trunk carries, not a relative-accuracy GELU for tiny outputs (ufm_gemm_bf16's bf16 GELU, gelu_bf16_x4, keeps
 *   relative accuracy in the negative tail instead).
 *   out_dtype UFM_BF16X2: out = split(v) as [2][M][N];   out_dtype UFM_F32: v += res[m][n] (fp32, may be NULL, may alias
 *   out: the fp32 residual stream is updated in place), out[m][n] = v.
 *   Requirements: K % 32 == 0, N % 32 == 0, 16-byte aligned pointers.  zero_page: >= 128 B of zeros (device).
 * ufm_attention_bf16x3: softmax(scale * q k^T) v per (batch, head), head_dim 64, non-causal; qkv: UFM_BF16X2
 *   [2][B*N][3*H*64] (the raw output of the qkv Linear), out: UFM_BF16X2 [2][B*N][H*64]; Q.K^T and P.V both as three
 *   bf16 MFMA passes (P is split into hi/lo in registers), softmax statistics in fp32.
 * ===================================================================================== */
int ufm_gemm_bf16x3(const uint16_t* A, const uint16_t* W, int M, int N, int K, const float* bias, int act,
                    const float* gamma, const float* res, void* out, int out_dtype, const uint16_t* zero_page,
                    void* stream);
/* Round 6: the same Linear layer on INTERLEAVED split operands -- A: [M][K / 32][hi 32 | lo 32] (ufm_layernorm with out_dtype UFM_BF16X2_IL
 * writes it), W: [N][K / 32][hi 32 | lo 32] (interleaved at pack time) -- so that every LDS-DMA row of the kernel's loop is one whole 128-byte
 * line.  Same K order, same products, same epilogues and output formats as ufm_gemm_bf16x3: bit-identical results.  N % 256 == 0, K % 32 == 0,
 * K >= 64; out_dtype additionally UFM_BF16X2_IL (the split output interleaved too: fc1 -> fc2).  Serves all four nn.Linear call sites of a block
 * (Attention.qkv / proj, Mlp.fc1 / fc2) in numerics "precise": LayerNorm, ufm_attention_bf16x3_il and the fc1 epilogue write the operands. */
int ufm_gemm_bf16x3_il(const uint16_t* A, const uint16_t* W, int M, int N, int K, const float* bias, int act, const float* gamma,
                       const float* res, void* out, int out_dtype, const uint16_t* zero_page, void* stream);
int ufm_attention_bf16x3(const uint16_t* qkv, uint16_t* out, int B, int N, int H, float scale, void* stream);
/* Round 6: the same attention with O stored INTERLEAVED (UFM_BF16X2_IL: [B*N][H*64 / 32][hi 32 | lo 32]) -- the A operand of ufm_gemm_bf16x3_il.
 * scale == 0: the Q columns of qkv arrive pre-scaled by softmax_scale * log2(e) (the QKV Linear's gamma does it before the split store), and the
 * kernel applies no per-score multiply (the convention of ufm_attention_bf16_strided / ufm_cross_attention_bf16). */
int ufm_attention_bf16x3_il(const uint16_t* qkv, uint16_t* out, int B, int N, int H, float scale, void* stream);

/* Bilinear resize, align_corners=True, NHWC fp32 ([U] FeatureFusionBlock x2 upsample,
 * DPTRegressionProcessor interpolate-to-target).  src = dst*(in-1)/(out-1).  crop_h/crop_w > 0:
 * only the top-left crop_h x crop_w of the (Ho, Wo) result is computed and stored (densely) --
 * the `[:, :, :h, :w]` slice after refinenet4 in the DPT head. */
int ufm_upsample_bilinear_nhwc(const void* in, int dtype /* UFM_F32 | UFM_BF16X2, in and out */, int B,
                               int H, int W, int C, void* out, int Ho, int Wo, int crop_h, int crop_w,
                               void* stream);

/* Head tail: per pixel, y[c] = w[c,:] . x[:] + b[c] over Cin<=64 channels, then the adaptor
 * ([U] DPTRegressionProcessor.conv2[2] + FlowAdaptor / MaskAdaptor; call sites
 * models/ufm.py:644-660).  kind[c]: 0 = affine y*a[c]+d[c] (FlowAdaptor), 1 = sigmoid
 * (MaskAdaptor: writes mask to out, logits to out_logits if non-NULL).
 * x: fp32 [P][Cin] -> out: fp32 planar [B][Cout][HW] (P = B*HW). Cout <= 8. */
int ufm_head_tail(const void* x, int in_dtype /* UFM_F32 | UFM_BF16X2 */, int P, int HW, int Cin,
                  const float* w, const float* b, int Cout,
                  const int32_t* kind_host, const float* a_host, const float* d_host, float* out,
                  float* out_logits, void* stream);

/* The whole full-resolution tail of [U] DPTRegressionProcessor in numerics "fast", fused:
 * bilinear(align_corners=True) (h, w) -> (H, W)  ->  conv 3x3 pad 1 (Cin = 128 -> Cmid = 32) + bias + ReLU  ->
 * conv 1x1 (32 -> Ct <= 8) + bias  ->  Flow/Mask adaptor (as ufm_head_tail; call sites models/ufm.py:644-660).
 * in: UFM_BF16X2 [2][B][h][w][128]; w2: UFM_BF16X2 [2][32][3][3][128]; out (and out_logits): fp32 planar
 * [B][Ct][H][W].  Bit-identical to ufm_upsample_bilinear_nhwc -> ufm_conv2d_nhwc_bf16x3(act = ReLU) ->
 * ufm_head_tail, without the two full-resolution intermediate maps ever reaching HBM. */
int ufm_dpt_tail_fused(const uint16_t* in, int B, int h, int w, int Cin, const uint16_t* w2, const float* b2,
                       int Cmid, int H, int W, const float* wt, const float* bt, int Ct,
                       const int32_t* kind_host, const float* a_host, const float* d_host, float* out,
                       float* out_logits, int64_t in_plane /* elements between the hi and lo plane of `in`; 0 = B*h*w*Cin (densely packed). Non-zero when `in` is one head's
                          slice of a stacked multi-head buffer (ufm_conv2d_nhwc_bf16x3_grouped) */,
                       void* stream);

/* Output adaptors that are not a per-channel affine / sigmoid ([U] uniception prediction_heads.adaptors; the
 * uncertainty head's optional branches, call sites models/ufm.py:648-654).  `raw` = the decoded channels as written by
 * ufm_head_tail / ufm_dpt_tail_fused with kind 0, a = 1, d = 0.
 *   Covariance2DAdaptor: raw planar [B][3][HW] = (log sigma_x, log sigma_y, atanh-like rho) ->
 *     covariance [B][3][HW] = (xx, yy, xy) with rho = 0.99 tanh(r); inverse covariance [B][3][HW]; log-determinant [B][1][HW]
 *     (-> UFMFlowFieldOutput.flow_covariance / _inv / _log_det, ufm.py:648-651).
 *   ConfidenceAdaptor (keypoint_confidence, ufm.py:653-654): type 0 = min(vmin + exp(x), vmax), 1 = (vmax-vmin) sigmoid(x) + vmin,
 *     2 = identity.
 * PARITY UNPINNED: both formulas are restated from the published adaptor semantics -- the defining uniception source is absent
 * from the reference and no fixture there holds their outputs; only the un-map / rescale applied afterwards
 * (base.py:295-319) is pinned by reference-generated goldens. */
int ufm_adaptor_covariance2d(const float* raw, int B, int HW, float* cov, float* inv_cov, float* log_det, void* stream);
int ufm_adaptor_confidence(const float* raw, int64_t n, int type, float vmin, float vmax, float* out, void* stream);

/* =====================================================================================
 * Post-processing (utils/flow_resizing.py:749-877 unmap_predicted_flow and :955-1010
 * unmap_predicted_channels, called from models/base.py:279-332): ROI crop, legacy-nearest
 * resample of the field, bilinear resample of the pixel-centre grid, per-axis coordinate
 * rescale, offsets, embed into a zero canvas, validity mask.  Regions are
 * [top,bottom,left,right] HOST ints.  flow: [B][2][h][w] -> out [B][2][H0][W0], valid uint8 [B][H0][W0].
 * ===================================================================================== */
int ufm_unmap_flow(const float* flow, int B, int h, int w, const int32_t* rep0, const int32_t* src0,
                   const int32_t* src1, int H0, int W0, float* out, uint8_t* valid, void* stream);
int ufm_unmap_channels(const float* chan, int B, int C, int h, int w, const int32_t* rep0,
                       const int32_t* src0, int H0, int W0, const float* chan_scale_host /*C or NULL*/,
                       float* out, uint8_t* valid, void* stream);

/* =====================================================================================
 * UFM-Refine classification refinement, fused (models/ufm.py:1012-1178): for every pixel,
 * bicubic (A=-0.75, zeros padding, align_corners=False) samples of the view-2 feature map at
 * flow target +- R, score = q.k / T + bias, softmax / log-softmax over P*P, residual =
 * sum attn * offset.  Never materialises the (B,H,W,P,P,C) tensor.
 *   feat: fp32 planar [2B][C][H][W] (first B = view 1); flow [B][2][H][W];
 *   residual [B][2][H][W]; log_softmax [B][H][W][P][P] (may be NULL).
 * ===================================================================================== */
int ufm_refine(const float* flow, const float* feat, int B, int C, int H, int W, int P,
               float temperature, const float* bias, float* residual, float* log_softmax,
               void* stream);

/* =====================================================================================
 * UNet fine-feature path of UFM-Refine (models/unet_encoder.py:26-71, use_unet_feature=True: models/ufm.py:816-825,
 * :915-917, :967-983).  The UNet's convolutions run on ufm_conv2d_nhwc_f32 / ufm_conv2d_nhwc_bf16x3 (3x3 pad 1 + ReLU,
 * ConvTranspose2d(k=s=2) in shuffle mode, 1x1); these are the layout moves in between, all on NHWC fp32 or UFM_BF16X2.
 *   ufm_image_to_nhwc:       the normalised network-resolution image (same in_dtype/in_layout/scale3/shift3 convention as
 *                            ufm_patchify) as NHWC with the 3 channels zero-padded to Cpad (the conv kernels' K chunk)
 *   ufm_maxpool2x2_nhwc:     nn.MaxPool2d(2, 2) (unet_encoder.py:37,57); out [B][H/2][W/2][C]
 *   ufm_resize_nearest_nhwc: F.interpolate(x, size=(Ho, Wo)) in the default legacy "nearest" mode (unet_encoder.py:66-67),
 *                            written into channels [c_off, c_off + C) of an output with ldc channels per pixel -- the
 *                            torch.cat((skip, x), dim=1) slot of unet_encoder.py:68 (Ho == H, Wo == W: a plain copy)
 *   ufm_unet_combine:        per pixel, cls = MLPFeature features planar [N][16][HW], unet = NHWC [N][HW][ldu] (16 used):
 *                            method 0 "conv": conv2(relu(conv1(cat[cls, unet])))  (w1 [32][32], w2 [16][32]);
 *                            method 1 "modulate": conv2(cls * tanh(unet))        (w2 [16][16]); out planar [N][16][HW].
 * ===================================================================================== */
int ufm_image_to_nhwc(const void* img, int in_dtype, int in_layout, int B, int H, int W, const float* scale3,
                      const float* shift3, void* out, int out_dtype, int Cpad, void* stream);
int ufm_maxpool2x2_nhwc(const void* in, int dtype, int B, int H, int W, int C, void* out, void* stream);
int ufm_resize_nearest_nhwc(const void* in, int dtype, int B, int H, int W, int C, void* out, int Ho, int Wo, int ldc,
                            int c_off, void* stream);
int ufm_unet_combine(const float* cls, const void* unet, int unet_dtype, int N, int HW, int ldu, const float* w1,
                     const float* b1, const float* w2, const float* b2, int method, float* out, void* stream);

/* =====================================================================================
 * The "moge_conv" prediction head ([U] uniception MoGeConvFeature = the convolutional head of MoGe: per-level 1x1
 * projections summed, three x2 stages {concat view-plane uv, ConvTranspose2d(k=s=2), conv3x3, residual conv blocks with
 * GroupNorm}, bilinear resize to the target, concat uv, conv3x3 -> ReLU -> conv1x1; call site models/ufm.py:266-267).
 * PARITY UNPINNED: the class is absent from the reference, restated in oracle/uniception_ref.py.  Its convolutions run on
 * ufm_conv2d_nhwc_f32 / _bf16x3 (padding_mode "replicate": relu_in bit 1); these are the kernels in between (moge.hip),
 * NHWC fp32 or UFM_BF16X2 (lo plane at B*H*W*ldc elements):
 *   ufm_group_norm_nhwc:      nn.GroupNorm(groups, C)(x) (+ ReLU if relu), x [B][HW][ldc] (C channels used), two-stage
 *                             deterministic statistics; partial_ws: ufm_group_norm_ws_floats(B, HW, groups) floats
 *   ufm_fill_uv_nhwc:         channels [c_off, c_off+2) = normalized_view_plane_uv(W, H, aspect_ratio) of the pixel, the
 *                             rest [c_off+2, ldc) = 0 (padding up to the conv kernels' 32-channel K chunk)
 *   ufm_resize_bilinear_nhwc: F.interpolate(mode="bilinear", align_corners=False) of in [B][H][W][ldi] (C channels) into
 *                             channels [c_off, c_off + C) of out [B][Ho][Wo][ldc]
 * ===================================================================================== */
int ufm_group_norm_nhwc(const void* in, int dtype, int B, int HW, int C, int ldc, int groups, const float* weight,
                        const float* bias, float eps, int relu, void* out, float* partial_ws, void* stream);
int ufm_group_norm_ws_floats(int B, int HW, int groups);
int ufm_fill_uv_nhwc(void* out, int dtype, int B, int H, int W, int ldc, int c_off, float aspect_ratio, void* stream);
int ufm_resize_bilinear_nhwc(const void* in, int dtype, int B, int H, int W, int C, int ldi, void* out, int Ho, int Wo, int ldc,
                             int c_off, void* stream);

/* Pixel shuffle for MLPFeature ([U], call site models/ufm.py:965): x [B*g*g][C*p*p] (column = (c, i, j)), fp32 or the
 * UFM_BF16X2 pair of planes (value = hi + lo) -> fp32 planar [B][C][g*p][g*p]. */
int ufm_pixel_shuffle_planar(const void* x, int in_dtype /* UFM_F32 | UFM_BF16X2 */, int B, int gh, int gw, int C, int p,
                             float* out, void* stream);

/* Elementwise helpers used by the host wiring. */
int ufm_cast_f32_to_bf16(const float* in, uint16_t* out, int64_t n, void* stream);
int ufm_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UFM_HIP_H */
