/* pseld_hip.h — C ABI of libpseld_hip.so: the MI355X (gfx950) kernels of the PSELDNets training hot path.
 *
 * The reference (Jinbo-Hu/PSELDNets, 100 % Python) has no FFI: its "operator API" is three duck-typed Python
 * seams (feature extractor, network registry, loss). This library is what the Python mirror of
 * those seams (pseldnets_amd/) binds with ctypes. Conventions for every entry point:
 *   - plain pointers and sizes only; device pointers unless stated; row-major, contiguous unless an ld* is given
 *   - `dtype`: 0 = f32 (exact-f32 MFMA / parity mode), 1 = bf16 storage with f32 accumulation
 *   - `stream` is a hipStream_t; work is enqueued, never synchronised; no allocation inside (workspaces are
 *     caller-owned, sized by the matching *_workspace function)
 *   - return 0 on success, <0 on error (-1 bad argument, -2 unsupported, -3 HIP error); text via pseld_last_error()
 * Each declaration cites the reference code (path:line under /root/reference/src) it replaces.
 */
#ifndef PSELD_HIP_H
#define PSELD_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------------------------------------------- */
const char* pseld_last_error(void);
int pseld_abi_version(void);
int pseld_device_info(char* name, int n); /* returns CU count */

/* Profiling aid: launches an empty kernel named stage_marker_kernel<tag> (0 <= tag < 16) on the stream, so that an in-order kernel trace can
 * be cut into the stages of the training step (bench.py with PSELD_STAGE_MARKERS=1, tools/pmc_stages.py). */
int pseld_stage_marker(int tag, void* stream);

/* ---- K1 feature front-end ---------------------------------------------------------------------------------
 * utils/feature.py:39-56 LogmelIV_Extractor.forward, :78-91 Logmel_Extractor.forward, :93-117 intensityvector
 * (+ torchaudio 2.2.1 Spectrogram / MelScale / AmplitudeToDB semantics).
 * wave f32[B,n_ch,L] -> feat f32[B,n_ch(+3 if with_iv),T=1+L/hop,n_mels]. window f32[n_fft]; twiddle f32[n_fft,2]
 * = (cos,sin)(-2*pi*n/n_fft); the mel filter bank is passed in compact column form: filter m has weights
 * mel_w[mel_off[m] .. +mel_cnt[m]) applied to bins mel_lo[m] .. +mel_cnt[m]. */
int pseld_logmel_iv_fwd(const float* wave, float* feat, int B, int n_ch, long L, int hop, int n_fft, int n_mels,
                        const float* window, const float* twiddle, const int* mel_lo, const int* mel_cnt,
                        const int* mel_off, const float* mel_w, int nnz, int with_iv, float amin, float iv_eps,
                        void* stream);

/* ---- MFMA GEMM ----------------------------------------------------------------------------------------------
 * nn.Linear forward and input gradient: htsat.py:123,141 (qkv, proj), model_utilities.py:166 (fc1, fc2),
 * htsat.py:309 (PatchMerging.reduction), model_utilities.py:209 (PatchEmbed.proj as GEMM), accdoa.py:230 (tscam_conv).
 * C[M,N] = A[M,K] * (trans_b ? B[K,N] : B[N,K]^T), then (in this order):
 *   + bias[N] (epi&1) ; * rowscale[m / rows_per_scale] (if rowscale) ; * gelu'(aux[m,n]) (epi&4) or * aux[m,n]
 *   (epi&32) ; + resid[m,n] (epi&2) ; then either C = v (+ C_old if epi&8) or, with epi&16 (the Mlp fc1 of
 *   model_utilities.py:166), C = gelu(v) and c2 = gelu'(v) so that neither fc2 nor the backward re-evaluates erf.
 * pro&1 applies exact-erf GELU to A on load. */
int pseld_gemm(int dtype, int trans_a, int trans_b, const void* A, const void* B, void* C, int M, int N, int K,
               int lda, int ldb, int ldc, const float* bias, const void* resid, int ldr, const float* rowscale,
               int rows_per_scale, const void* aux, int ldaux, int epi, int pro, void* c2, void* stream);

/* Input gradient of a Linear whose input is a LayerNorm output, with that LayerNorm's backward in the GEMM epilogue (bf16; C = 96 or 192 so
 * that one output tile spans the row; stages 0-1 of HTS-AT: norm1 -> attn.qkv, norm2 -> mlp.fc1, htsat.py:234,262 / model_utilities.py:159):
 *   dx[M, C] = LayerNorm'(dY[M, K] . Wt[C, K]^T; x, gamma, eps) (+ dres)
 * i.e. pseld_gemm (the input gradient through Wt = the Linear's weight [K, C] TRANSPOSED to [C, K]) + pseld_layernorm_bwd in one launch; the
 * intermediate d(LN output) is rounded to bf16 exactly where the two-launch path stores it. partial: fp32
 * [pseld_gemm_dgrad_lnbwd_parts(M, C)][2][C] row-tile sums of d(gamma) and d(beta), to be reduced by pseld_reduce_slabs(_batched) like
 * the partials of pseld_layernorm_bwd. */
int pseld_gemm_dgrad_lnbwd_supported(int dtype, long M, int C, int K);
long pseld_gemm_dgrad_lnbwd_parts(long M, int C);
int pseld_gemm_dgrad_lnbwd(int dtype, const void* dY, const void* Wt, const void* x, const float* gamma, const void* dres, void* dx,
                           float* partial, long M, int C, int K, int lddy, int ldwt, float eps, void* stream);
/* (C = 384 variants of this epilogue and of its forward counterpart - LayerNorm in the residual epilogue, pseld_gemm_resid_ln - were built in
 * round 5 on a 128 x 384 row-spanning tile, measured equal to the two launches they replace and removed in round 6: docs/EXPERIMENTS.md.) */
/* Weight gradient of the same layers: dW f32[N,K] (+)= dY[Mtok,N]^T @ (gelu_on_x ? gelu(X) : X)[Mtok,K], and (when
 * dbias != NULL) the bias gradient dbias f32[N] (+)= sum_m dY[m,n] from the same pass. Split over tokens into fp32
 * slabs in `workspace`, reduced in a fixed order (bitwise reproducible). rowscale (optional, f32[Mtok / rows_per_scale]):
 * dY row m is multiplied by rowscale[m / rows_per_scale] on load — the DropPath factor of its sample
 * (model_utilities.py:216-232), so the branch gradient s*dy never has to be materialised. */
long pseld_gemm_wgrad_workspace(int Mtok, int N, int K, int* splits_out);
int pseld_gemm_wgrad(int dtype, const void* dY, const void* X, float* dW, float* dbias, int Mtok, int N, int K,
                     int lddy, int ldx, int lddw, int gelu_on_x, int accumulate, float* workspace,
                     long workspace_bytes, const float* rowscale, int rows_per_scale, void* stream);

/* The weight gradients of several Linear layers in ONE launch (bf16): entry i is dW_i f32[N_i, K_i] = (s_i dY_i)^T X_i over Mtok_i tokens
 * (+ dbias_i f32[N_i] = column sums of s_i dY_i when dbias[i] != NULL), s_i = rowscale[i] (DropPath factor per sample of
 * rows_per_scale[i] tokens) or 1 - the layers of one stage of the encoder: htsat.py:118,140 (qkv, proj), model_utilities.py:166-170
 * (fc1, fc2), htsat.py:309 (PatchMerging.reduction). Overwrites the gradients. One weight matrix alone has 4-16 output tiles and must be
 * split ~20 ways over the tokens (one fp32 slab per workgroup, reduced afterwards); the ~25 matrices of a stage together fill the chip
 * with ONE tile per CU: no token split, no slab, no reduction. Entries the persistent kernel does not take (small or ragged shapes) are
 * left untouched and reported as set bits of *skipped_mask: the caller runs pseld_gemm_wgrad for those. count <= 32. */
long pseld_gemm_wgrad_group_workspace(int count, const int* Mtok, const int* N, const int* K);
int pseld_gemm_wgrad_group(int count, const void* const* dY, const void* const* X, float* const* dW, float* const* dbias, const int* Mtok,
                           const int* N, const int* K, const int* lddy, const int* ldx, const float* const* rowscale,
                           const int* rows_per_scale, float* workspace, long workspace_bytes, unsigned* skipped_mask, void* stream);

/* ---- fused Swin MLP block (the HBM-bound stages: C = 96 / 192, hidden = 4C) ---------------------------------------
 * htsat.py:262-264  x = x + drop_path(mlp(norm2(x)));  model_utilities.py:159-171 Mlp (fc1 -> exact-erf GELU -> fc2),
 * :216-232 DropPath (rowscale = per-sample factor mask / keep_prob, rows_per_scale = tokens per sample, a multiple of 32).
 * One kernel per direction keeps the [M, 4C] hidden activations on the CU (registers / LDS); SURVEY 8b `pseld_mlp_{fwd,bwd}`.
 *   pseld_mlp_fwd:    y = x + s * (gelu(LN(x) W1^T + b1) W2^T + b2); xh_out (optional, [M, C] in the compute dtype) receives
 *                     LN(x), the operand the two backward kernels read. w1 [4C, C], w2 [C, 4C] in the compute dtype; b1, b2,
 *                     gamma, beta fp32.
 *   pseld_mlp_bwd_dx: dxh[M, C] = gradient wrt LN(x) = ((s dy) W2 * gelu'(u)) W1 with u = xh W1^T + b1 recomputed. w2t = W2^T
 *                     [4C, C], w1t = W1^T [C, 4C]. (The residual path and the LayerNorm backward: pseld_layernorm_bwd, dres = dy.)
 *   pseld_mlp_bwd_dw: dw1 f32[4C, C], db1 f32[4C], dw2 f32[C, 4C], db2 f32[C] (+)= the four parameter gradients (u and
 *                     gelu(u) recomputed); split over tokens into fp32 slabs in `workspace`, reduced in a fixed order.
 * pseld_mlp_supported: 1 when the fused kernels take (dtype, M, C, rows_per_scale), else 0 (callers then run the layer-wise path). */
int pseld_mlp_supported(int dtype, long M, int C, int rows_per_scale);
int pseld_mlp_fwd(int dtype, const void* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                  const void* w2, const float* b2, const float* rowscale, int rows_per_scale, void* y, void* xh_out,
                  long M, int C, float eps, void* stream);
int pseld_mlp_bwd_dx(int dtype, const void* xh, const void* dy, const void* w1, const float* b1, const void* w2t,
                     const void* w1t, const float* rowscale, int rows_per_scale, void* dxh, long M, int C, void* stream);
long pseld_mlp_bwd_dw_workspace(int dtype, long M, int C, int rows_per_scale);
/* Diagnostic only: device buffer of 32 x u64 per wave; the forward kernel then records s_memtime stamps (NULL disables). */
void pseld_mlp_set_debug_buffer(void* device_buffer);
int pseld_mlp_bwd_dw(int dtype, const void* xh, const void* dy, const void* w1, const float* b1, const void* w2t,
                     const float* rowscale, int rows_per_scale, float* dw1, float* db1, float* dw2, float* db2, long M,
                     int C, int accumulate, float* workspace, long workspace_bytes, void* stream);

/* ---- fused front half of the Swin attention branch (C = 96, 4 heads, bf16): LayerNorm -> qkv Linear -> window attention -----
 * htsat.py:234 (norm1), :118-138 (WindowAttention.forward up to the head merge), :23-50,239-242,257-260 (partition / reverse / roll as an
 * address map). One persistent kernel per block; x is read once; outputs exactly what pseld_layernorm_fwd + pseld_gemm (qkv) +
 * pseld_window_attn_fwd leave for the projection GEMM and for the (unchanged) backward kernels: qkv [M, 3C], out [M, C] (heads merged),
 * xh = LN(x) [M, C] (optional), lse f32[M, heads] (optional). SURVEY 8b `pseld_swin_attn_fwd` (proj + residual stay pseld_gemm with
 * its fused epilogue). pseld_swin_attn_supported: 1 when the kernel takes (dtype, res, C, heads). */
int pseld_swin_attn_supported(int dtype, int res, int C, int heads);
int pseld_swin_attn_fwd(int dtype, const void* x, const float* gamma, const float* beta, const void* wqkv, const float* bqkv,
                        const float* bias_table, void* qkv, void* out, void* xh, float* lse, int B, int res, int C, int heads,
                        int shift, float eps, void* stream);
/* The whole attention half of a Swin block in one kernel (SURVEY 8b `pseld_swin_attn_fwd`: LN -> QKV -> window attention -> proj -> DropPath
 * + shortcut; reference htsat.py:234-260 with WindowAttention.forward :118-145): xmid = x + s * (attention(norm1(x)) Wproj^T + bproj),
 * s = rowscale[sample of the token] (DropPath mask / keep_prob, model_utilities.py:216-232) or 1 when rowscale is NULL. qkv [M,3C], out
 * [M,C] (merged heads, before proj), xh = LN(x) [M,C] and lse f32 [M,heads] are the operands of the backward kernels; all four may be NULL
 * (no-grad forward: x is read, xmid is written, nothing else). Same support set as pseld_swin_attn_supported. */
int pseld_swin_block_attn_fwd(int dtype, const void* x, const float* gamma, const float* beta, const void* wqkv, const float* bqkv,
                              const float* bias_table, const void* wproj, const float* bproj, const float* rowscale, void* qkv, void* out,
                              void* xh, float* lse, void* xmid, int B, int res, int C, int heads, int shift, float eps, void* stream);
/* ... and of its backward up to the qkv gradient (SURVEY 8b `pseld_swin_attn_bwd`): dy = d(x_mid) [M,C]; the gradient of the attention output,
 * (s * dy) Wproj (attn.proj, htsat.py:139, under the block's DropPath), is formed inside the kernel from wproj_t = Wproj^T [C,C] and
 * rowscale f32[B] (or NULL); then the backward of window attention exactly as pseld_window_attn_bwd (same workspace, same dbias_table
 * NULL = deferred accumulator convention). The projection's weight gradient (dy, out), the qkv gemms and the LayerNorm backward stay
 * separate calls. bf16, C = 96, 4 heads (pseld_swin_block_attn_bwd_supported). */
int pseld_swin_block_attn_bwd_supported(int dtype, int res, int C, int heads);
int pseld_swin_block_attn_bwd(int dtype, const void* qkv, const float* bias_table, const void* out, const float* lse, const void* dy,
                              const void* wproj_t, const float* rowscale, void* dqkv, float* dbias_table, int B, int res, int C, int heads,
                              int shift, int accumulate, float* workspace, long workspace_bytes, void* stream);

/* Diagnostic only: when a device buffer (6 x u64 per workgroup) is installed, every pseld_gemm workgroup records
 * s_memtime stamps (start, first slice staged, K loop done, end, C tile staged, stores issued); NULL disables. */
void pseld_gemm_set_debug_buffer(void* device_buffer);
/* The same for the persistent eight-phase kernel (csrc/gemm8.hip: the products with K >= 192): u64 [workgroup][wave group 0/1][tile < 16][4]
 * = s_memtime at (tile start, K loop done, epilogue done) + s_memrealtime; a diagnostic instantiation runs while a buffer is installed. */
void pseld_gemm8_set_debug_buffer(void* device_buffer);
/* Routing knobs: the A/B switches of the measurement tools and tests (list: csrc/common.h PSELD_KNOB_LIST, e.g. "GEMM8", "WGRAD8",
 * "ATTN_FWD_P"; the "PSELD_" prefix is optional). Every knob has a frozen default; the environment variable PSELD_<NAME> is read once per
 * process, when the library first asks for a knob, and these two calls change one afterwards. Returns PSELD_ERR_BAD_ARG for unknown names. */
int pseld_set_knob(const char* name, int value);
int pseld_unset_knob(const char* name);
/* Measurement / test aid: force the tile shape of the eight-phase kernel for the following pseld_gemm calls (rows 256 | 128, columns
 * 256 | 192; 0 = the launch's own choice by grid fill). All four shapes sum K in the same order and give the same bits. */
void pseld_gemm8_force_tile(int rows, int cols);
/* The same for the row-panel-stationary kernel (csrc/gemm8p.hip: K = 192 / 384, the A rows of a 64 mb-row panel in registers, the weights
 * streamed through LDS; replaces htsat.py:118,140 qkv / proj and model_utilities.py:166-170 fc1 with their K <= 384 input gradients at
 * bench-size batches): mb = 16-row blocks per wave (2 | 3 at K = 384, 3 | 4 at K = 192; 0 = own choice), stag = -1 own choice | 0 | 1
 * (second wave of every SIMD runs its epilogue one tile late). Every shape gives the bits of the eight-phase kernel. */
void pseld_gemm8p_force(int mb, int stag);
/* Diagnostic only: while a device buffer (u64 [workgroup][wave group 0/1][10]) is installed, the unstaggered 192-row launches of the plain,
 * GELU-pair and scaled-aux epilogues run a stamped instantiation: cycle sums of (A panel load, vmcnt wait, barrier, LDS-DMA issue, matrix
 * part, epilogue), the workgroup's lifetime and its tile count. NULL disables. */
void pseld_gemm8p_set_debug_buffer(void* device_buffer);
/* Fused MLP forward of a C = 192 / 384 block (csrc/mlp8f.hip; reference: model_utilities.py:159-171 Mlp.forward with the block's DropPath +
 * shortcut, htsat.py:262-264): y[M, C] = resid + s[m / rows_per_scale] (gelu(xn W1^T + b1) W2^T + b2), and the two tensors the backward pass
 * reads, h = gelu(.) and g = gelu'(.) [M, H = 4 C]. bf16 operands (xn = the LayerNorm output, W1 [H, C], W2 [C, H], row-major, 16-byte
 * aligned), fp32 biases / DropPath factors (rowscale may be NULL). One launch; the bits of pseld_gemm(fc1, GELU pair) + pseld_gemm(fc2,
 * residual). _force(mb): 16-row blocks per wave (panel = 64 mb rows; 0 = own choice), measurement aid. _set_debug_buffer: while a device
 * buffer (u64 [workgroup][wave][12]) is installed a stamped instantiation runs (cycle sums of the loop's phases). */
int pseld_mlp_panel_fwd_supported(int dtype, int M, int C, int H);
int pseld_mlp_panel_fwd(int dtype, const void* xn, const void* w1, const float* b1, const void* w2, const float* b2, const void* resid,
                        const float* rowscale, int rows_per_scale, void* y, void* h, void* g, int M, int C, int H, int ldx, int ldr,
                        int ldy, int ldh, void* stream);
void pseld_mlp_panel_force(int mb);
void pseld_mlp_panel_set_debug_buffer(void* device_buffer);
/* Measurement aid: symbol of the kernel the last pseld_gemm / pseld_gemm_wgrad call of this process launched. */
const char* pseld_gemm_last_kernel(void);
/* Measurement aid: pseld_gemm_wgrad launches a GEMM kernel and a slab reduction, so events around the call do not time one kernel.
 * After pseld_gemm_wgrad_timing(1) every call brackets its GEMM kernel alone with two library-owned HIP events (switching on resets the
 * record count, 0 switches it off and keeps the records readable); _count = calls recorded, _read(i) = the i-th call's kernel time in ms (synchronises on its end event;
 * < 0 on error), _symbol(i) = the kernel symbol it launched. bench.py ranks ALL kernel symbols of the step with it. */
int pseld_gemm_wgrad_timing(int enable);
int pseld_gemm_wgrad_timing_count(void);
float pseld_gemm_wgrad_timing_read(int i);
const char* pseld_gemm_wgrad_timing_symbol(int i);

/* Bias gradient: out f32[N] (+)= sum_m X[m,n]. */
long pseld_colsum_workspace(int M, int N);
int pseld_colsum(int dtype, const void* X, float* out, int M, int N, int ld, int accumulate, float* workspace,
                 long workspace_bytes, void* stream);

/* ---- LayerNorm (plain, and the 2x2 patch-merging gather form) -----------------------------------------------
 * htsat.py:234,261,525 (norm1/norm2/final norm), model_utilities.py:212 (PatchEmbed.norm), htsat.py:290-311
 * (PatchMerging: merge_res = side of the INPUT token grid; rows are [B*(res/2)^2, C=4*Cs]). gamma/beta fp32. */
int pseld_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, long M, int C, int merge_res, float eps, void* stream);
long pseld_layernorm_bwd_workspace(long M, int C);
int pseld_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma, const void* dres, void* dx,
                        float* dgamma, float* dbeta, long M, int C, int merge_res, float eps, int accumulate,
                        float* workspace, long workspace_bytes, void* stream);
/* accumulate & 2 in pseld_layernorm_bwd: DEFERRED parameter gradients - the call leaves its column-sum partials
 * [nb = workspace_bytes / (8 C)][2][C] in `workspace` and the caller reduces the partials of several LayerNorms with one launch: */
int pseld_reduce_slabs_batched(const void* const* src, void* const* dst, const int* n, const int* splits, const int* stride, int count,
                               int accumulate, void* stream);

/* ---- the 7 "scalar" BatchNorm2d(mel) + pad + time->frequency fold + 4x4 patch extraction ----------------------
 * models/accdoa.py:223-227 (in-place per-channel BN), htsat.py:493-511 (reshape_wav2img), im2col of
 * model_utilities.py:209 (PatchEmbed.proj). stats: sums f32[3*Cin*F] = interleaved (sum x, sum x^2)[Cin*F] then,
 * if centered, sum (x-mean)^2 [Cin*F]; the first 2*Cin*F floats are what sync-BN all-reduces
 * (configs/trainer/gpu.yaml:9). finalize: scale_shift / mean_rstd f32[Cin*F][2]; running stats updated in place. */
long pseld_bn_scalar_workspace(int B, int Cin, int T);
int pseld_bn_scalar_stats(const float* feat, float* sums, int B, int Cin, int T, int F, int centered, float* workspace,
                          long workspace_bytes, void* stream);
int pseld_bn_scalar_finalize(float* sums, float count, int centered, const float* weight, const float* bias,
                             float* running_mean, float* running_var, long long* num_batches, float* mean_rstd,
                             float* scale_shift, int Cin, int F, float momentum, float eps, int training, void* stream);
int pseld_bn_fold_patchify(int dtype, const float* feat, const float* scale_shift, void* A, int B, int Cin, int c_first,
                           int Cuse, int T, void* stream);
long pseld_bn_scalar_bwd_workspace(int B, int Cuse);
int pseld_bn_scalar_bwd(int dtype, const float* feat, const float* mean_rstd, const void* dA, float* dweight,
                        float* dbias, int B, int Cin, int c_first, int Cuse, int T, int accumulate, float* workspace,
                        long workspace_bytes, void* stream);
/* DropPath backward factor per sample (model_utilities.py:216-232): y = x * scale[i / elems_per_scale] */
int pseld_rowscale(int dtype, const void* x, const float* scale, void* y, long n, long elems_per_scale, void* stream);
/* y = a + b: gradient fan-in where two heads consume the same tokens (einv2.py:420-421) */
int pseld_add(int dtype, const void* a, const void* b, void* y, long n, void* stream);

/* ---- CrossStitch soft parameter sharing of EINV2 (model_utilities.py:35-54; einv2.py:303-305) -------------------
 * x' = w00*x + w01*y ; y' = w10*x' + w11*y (y' uses the updated x'), w f32[C,2,2], tokens [M,C]. */
int pseld_cross_stitch_fwd(int dtype, const void* x, const void* y, const float* w, void* x_out, void* y_out, long M,
                           int C, void* stream);
long pseld_cross_stitch_bwd_workspace(long M, int C);
int pseld_cross_stitch_bwd(int dtype, const void* x, const void* y, const float* w, const void* dx_out,
                           const void* dy_out, void* dx, void* dy, float* dw, long M, int C, int accumulate,
                           float* workspace, long workspace_bytes, void* stream);

/* ---- (shifted-)window attention ----------------------------------------------------------------------------------
 * htsat.py:23-50 (partition/reverse), :239-260 (roll), :118-138 (WindowAttention core), :203-222 (mask).
 * qkv [B, res*res, 3C] -> out [B, res*res, C], both in natural token order; 8x8 windows, head_dim 8..32.
 * bias_table f32[225, heads]. The forward also writes lse f32[B*res*res, heads] (log-sum-exp of the masked, biased scores of
 * every query and head; may be NULL for inference). The backward takes the saved forward output and lse — P = exp(S - lse),
 * dS = P (dP - rowsum(dO o out)), so no softmax reduction is repeated — and adds d(bias_table) (deterministic up to fp32
 * atomic order). */
int pseld_window_attn_fwd(int dtype, const void* qkv, const float* bias_table, void* out, float* lse, int B, int res, int C,
                          int heads, int shift, void* stream);
long pseld_window_attn_bwd_workspace(int heads);
/* Diagnostic only: (64 x 8 x 8 + 2 x 1024) u64 - s_memtime stamps (phases of the first windows of the first workgroups of the backward), then
   for the head_dim-24 bf16 backward the entry / exit time of every workgroup in s_memrealtime ticks (100 MHz). */
void pseld_attn_set_debug_buffer(void* device_u64_buffer);
int pseld_window_attn_bwd(int dtype, const void* qkv, const float* bias_table, const void* out, const float* lse,
                          const void* dout, void* dqkv, float* dbias_table, int B, int res, int C, int heads, int shift,
                          int accumulate, float* workspace, long workspace_bytes, void* stream);

/* Deferred d(bias_table): pseld_window_attn_bwd with dbias_table == NULL leaves the block's [heads][64][64] sums in `workspace` (which
 * the caller owns and zeroed); this turns the accumulators of n blocks into their table gradients with one launch.
 * desc = n x {accumulator offset, table-gradient offset, heads} (device longs; float offsets from acc_base / grad_base). */
int pseld_bias_table_grad_batched(const float* acc_base, float* grad_base, const long* desc, int n, int max_heads, int accumulate,
                                  void* stream);

/* ---- global multi-head self-attention of the PaSST blocks -----------------------------------------------------------
 * passt.py:62-82 (Attention.forward: qkv split, q k^T * head_dim^-0.5, softmax, @ v, head merge), head_dim 64.
 * qkv [B, N, 3E] -> out [B, N, E]; lse f32[B, heads, N] (log-sum-exp of the scaled scores) is what backward needs
 * besides qkv and out; the score matrix is never materialised. Backward is deterministic (no atomics). */
int pseld_mhsa_fwd(int dtype, const void* qkv, void* out, float* lse, int B, int N, int E, int heads, void* stream);
long pseld_mhsa_bwd_workspace(int B, int N, int heads);
int pseld_mhsa_bwd(int dtype, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
                   int N, int E, int heads, float* workspace, long workspace_bytes, void* stream);

/* ---- PaSST front / back end ------------------------------------------------------------------------------------------
 * accdoa.py:320-327 scalar BN + im2col of model_utilities.py:174-213 PatchEmbed (Conv2d k16 s10 pad 3 on the
 * [C, mel, T] image): feat f32[B,Cin,T,64] -> A [B*6*Tg, Cin*256] (k = c*256 + kf*16 + kt = the conv weight's
 * flattening), Tg = pseld_passt_grid_t(T); bn_bwd folds dA back through the overlapping patches into the BN
 * weight/bias gradients (added to them when `accumulate`). Ctot >= Cin is the channel count of `feat` itself: the SED
 * encoder of einv2.PASST (einv2.py:543) reads the first Cin = 4 of its 7 channels. assemble: passt.py:219-247, X[b] = [cls+npos0, dist+npos1, P[b] + tpos[:,tg] + fpos[:,fg]]
 * ([B, 6*Tg+2, E]); its backward emits dP and the five positional/token gradients (summed over the batch).
 * pool: passt.py:296-300 mean over the 6 frequency rows, [B, 6*Tg+2, E] -> [B, Tg, E]. tanh: accdoa.py:328. */
int pseld_passt_grid_t(int T);
int pseld_passt_patchify(int dtype, const float* feat, const float* scale_shift, void* A, int B, int Cin, int Ctot,
                         int T, void* stream);
long pseld_passt_bn_bwd_workspace(int B, int Cin, int T);
int pseld_passt_bn_bwd(int dtype, const float* feat, const float* mean_rstd, const void* dA, float* dweight,
                       float* dbias, int B, int Cin, int Ctot, int T, int accumulate, float* workspace,
                       long workspace_bytes, void* stream);
int pseld_passt_assemble_fwd(int dtype, const void* P, const float* tpos, const float* fpos, const float* cls,
                             const float* dist, const float* npos, void* X, int B, int E, int Tg, void* stream);
long pseld_passt_assemble_bwd_workspace(int E, int Tg);
int pseld_passt_assemble_bwd(int dtype, const void* dX, void* dP, float* dtpos, float* dfpos, float* dcls,
                             float* ddist, float* dnpos, int B, int E, int Tg, float* workspace, long workspace_bytes,
                             void* stream);
/* Structured patch-out (passt.py:250-258,333-338: training-time removal of frequency rows of the patch grid):
 * Y[b, j, :] = X[b, map[j], :] for j < n_dst (zeros where map[j] < 0); X [B, n_src, E], Y [B, n_dst, E], map int32 [n_dst] on the
 * device. With map = kept token rows it drops tokens; with the inverse map (-1 for dropped rows) it is the adjoint. */
int pseld_rows_select(int dtype, const void* X, const int* map, void* Y, int B, int n_src, int n_dst, int E, void* stream);
int pseld_passt_pool_fwd(int dtype, const void* X, void* Y, int B, int E, int Tg, void* stream);
int pseld_passt_pool_bwd(int dtype, const void* dY, void* dX, int B, int E, int Tg, void* stream);
int pseld_tanh_fwd(int dtype, const void* z, int ldz, float* y, long rows, int D, void* stream);
int pseld_tanh_bwd(int dtype, const float* dy, const float* y, void* dz, int ldz, long rows, int D, void* stream);
/* einv2.py:562-573 (and :166-174): the per-track Linear outputs of the EINV2 networks leave the padded GEMM buffer as
 * y f32[rows, D] (row stride ldy: track t of a [rows, 3, D] tensor has ldy = 3*D) = act(z[rows, :D]), act 0 = identity (SED logits, final_act_sed is empty) or 1 = tanh (DOA); bwd writes
 * dz[rows, ldz] with zeros in the padding columns (y may be NULL when act == 0). */
int pseld_fc_out_fwd(int dtype, const void* z, int ldz, float* y, int ldy, long rows, int D, int act, void* stream);
int pseld_fc_out_bwd(int dtype, const float* dy, const float* y, int ldy, void* dz, int ldz, long rows, int D, int act,
                     void* stream);

/* ---- convolutional encoder of the CRNN networks (CNN8 / CNN12 = the PANNs CNN14 conv stack) ------------------------------
 * accdoa.py:72-90, backbone.py:6-60, model_utilities.py:92-126 (ConvBlock), utils.py:25-52 (interpolate 'repeat').
 * Activations are NHWC rows [B*T*F, C]. cnn_input: scalar BN + NCHW->NHWC (Cp >= Cin padded channels, zeros);
 * im2col3x3 / col2im3x3: A[(b,t,f)][tap*C + c] = X[b,t+dt,f+df,c], tap = (dt+1)*3 + (df+1) (tap-major: 16-byte
 * copies, C % 8 == 0) and its adjoint; conv_weight_to_tap / conv_wgrad_from_tap: the [Cout,Cin,3,3] weight <-> the
 * tap-major [Cout,9,Cp] matrix the GEMMs use (Cp >= Cin zero-padded channels), and the fp32 gradient back; bn2d_stats: sums f32[C][2] = (sum x, sum x^2) over the rows
 * (finalise with pseld_bn_scalar_finalize(Cin = 1, F = C)); bn_relu: y = relu(x*scale + shift) and its train-mode
 * backward (dgamma, dbeta overwritten); avgpool: AvgPool2d((pt,pf)), floor; rows_pool: y[b,j] = sum_k w[j][k] *
 * x[b, i0[j]+k], k < 3 (the 'repeat' x 8 + 10-frame mean map). */
int pseld_cnn_input(int dtype, const float* feat, const float* scale_shift, void* X, int B, int Cin, int T, int Cp,
                    void* stream);
long pseld_cnn_input_bwd_workspace(int B, int Cin, int T);
int pseld_cnn_input_bwd(int dtype, const float* feat, const float* mean_rstd, const void* dX, float* dweight, float* dbias,
                        int B, int Cin, int T, int Cp, float* workspace, long workspace_bytes, void* stream);
int pseld_im2col3x3(int dtype, const void* X, void* A, int B, int T, int F, int C, void* stream);
int pseld_col2im3x3(int dtype, const void* dA, void* dX, int B, int T, int F, int C, void* stream);
int pseld_conv_weight_to_tap(int dtype, const void* W, void* Wp, int Cout, int Cin, int Cp, void* stream);
int pseld_conv_wgrad_from_tap(const float* dWp, float* dW, int Cout, int Cin, int Cp, void* stream);
/* Implicit 3x3 / pad 1 convolution on NHWC rows (model_utilities.py:92-126 ConvBlock's nn.Conv2d, bias-free): the
 * im2col matrix is never built — the GEMM loaders read X[B*T*F, C] with shifted rows and zero the border.
 * conv3x3_fwd: Y[B*T*F, N] = im2col(X) @ Wp[N, 9*C]^T (Wp from pseld_conv_weight_to_tap). The same call is the input
 * gradient with X = dY (C = Cout) and Wp = pseld_conv_weight_to_tap_t(W) ([Cp, 9*Cout], taps flipped, N = Cp).
 * conv3x3_wgrad: dWp[N, 9*C] f32 (+)= dY[B*T*F, N]^T @ im2col(X) (back to [Cout,Cin,3,3] with conv_wgrad_from_tap).
 * C, N multiples of 8; B*T*F < 2^24 rows per call. */
int pseld_conv_weight_to_tap_t(int dtype, const void* W, void* Wd, int Cout, int Cin, int Cp, void* stream);
int pseld_conv3x3_fwd(int dtype, const void* X, const void* Wp, void* Y, int B, int T, int F, int C, int N, void* stream);
long pseld_conv3x3_wgrad_workspace(int B, int T, int F, int C, int N);
int pseld_conv3x3_wgrad(int dtype, const void* dY, const void* X, float* dWp, int B, int T, int F, int C, int N,
                        int accumulate, float* workspace, long workspace_bytes, void* stream);
long pseld_bn2d_workspace(long rows, int C);
int pseld_bn2d_stats(int dtype, const void* X, float* sums, long rows, int C, float* workspace, long workspace_bytes,
                     void* stream);
int pseld_bn_relu_fwd(int dtype, const void* X, const float* scale_shift, void* Y, long rows, int C, void* stream);
int pseld_bn_relu_bwd(int dtype, const void* X, const void* Y, const void* dY, const float* mean_rstd, const float* gamma,
                      void* dX, float* dgamma, float* dbeta, long rows, int C, float* workspace, long workspace_bytes,
                      void* stream);
/* The same backward in two halves for data-parallel runs with synchronised BatchNorm (configs/trainer/gpu.yaml:9 `sync_batchnorm: true`
 * = torch.nn.SyncBatchNorm on every BatchNorm of the CRNN conv stack / the Conformer): the caller all-reduces `sums` f32[C][2] =
 * (sum g * xhat, sum g) between the two calls and passes inv_count = 1 / (rows over all ranks); dgamma / dbeta are the rank-local sums
 * (the gradient all-reduce averages them). The forward side needs no new entry point: pseld_bn2d_stats' raw sums add across ranks. */
int pseld_bn_relu_bwd_sums(int dtype, const void* X, const void* Y, const void* dY, const float* mean_rstd, float* sums, float* dgamma,
                           float* dbeta, long rows, int C, float* workspace, long workspace_bytes, void* stream);
int pseld_bn_relu_bwd_apply(int dtype, const void* X, const void* Y, const void* dY, const float* mean_rstd, const float* gamma,
                            const float* sums, float inv_count, void* dX, long rows, int C, void* stream);
/* the BatchNorm1d of the Conformer conv module (conformer/convolution.py:129): same statistics kernels, no ReLU; its
 * backward is pseld_bn_relu_bwd with Y = NULL */
int pseld_bn_affine_fwd(int dtype, const void* X, const float* scale_shift, void* Y, long rows, int C, void* stream);
int pseld_avgpool_fwd(int dtype, const void* X, void* Y, int B, int T, int F, int C, int pt, int pf, void* stream);
int pseld_avgpool_bwd(int dtype, const void* dY, void* dX, int B, int T, int F, int C, int pt, int pf, void* stream);
int pseld_rows_pool_fwd(int dtype, const void* X, const int* i0, const float* w, void* Y, int B, int n_in, int n_out, int C,
                        void* stream);
int pseld_rows_pool_bwd(int dtype, const void* dY, const int* i0, const float* w, void* dX, int B, int n_in, int n_out, int C,
                        void* stream);

/* ---- Conformer decoder glue (CRNN decoder='conformer') ---------------------------------------------------------------
 * components/conformer/modules.py:23-35 (module(x)*factor + x -> axpby), activation.py (Swish, GLU), feed_forward.py,
 * convolution.py:94-151 (pointwise -> GLU -> depthwise Conv1d(k=31,'same') -> BatchNorm1d -> Swish -> pointwise),
 * attention.py:28-147 (RelativeMultiHeadAttention with the Transformer-XL relative shift). Rows are [B*T, D]
 * channel-last; n, D multiples of 8. axpby: out = a*x + b*y. mul: y = x*m*scale (dropout: m the 0/1 keep mask, scale 1/(1-p)).
 * swish: y = u*sigmoid(u). glu: x [M, 2D] -> y [M, D] = x[:, :D] * sigmoid(x[:, D:]). dwconv: w f32 [D, K]
 * (Conv1d(groups=D).weight[D,1,K]); flip != 0 gives the input gradient; wgrad overwrites dw f32 [D, K].
 * relattn: q,k,v [B*T, D] (head h in columns h*D/heads..), pos f32 [T, D] = pos_proj(PE[:T]) (batch independent),
 * u_bias/v_bias f32 [heads*hd]; score = ((q+u) k^T + shift((q+v) pos^T)) / sqrt(D); attn f32 [B,heads,T,T] is the
 * softmax output kept for the backward; mask (same dtype as q, [B,heads,T,T], 0/1 keep, times mask_scale = 1/(1-p)) or NULL; T <= 128.
 * relattn_bwd overwrites dq, dk, dv and dpos f32 [T, D], du_bias, dv_bias f32 [D] (summed over the batch). */
int pseld_axpby(int dtype, const void* x, const void* y, void* out, float a, float b, long n, void* stream);
int pseld_mul(int dtype, const void* x, const void* m, void* y, float scale, long n, void* stream);
int pseld_swish_fwd(int dtype, const void* u, void* y, long n, void* stream);
int pseld_swish_bwd(int dtype, const void* u, const void* dy, void* du, long n, void* stream);
int pseld_glu_fwd(int dtype, const void* x, void* y, long M, int D, void* stream);
int pseld_glu_bwd(int dtype, const void* x, const void* dy, void* dx, long M, int D, void* stream);
int pseld_dwconv_fwd(int dtype, const void* x, const float* w, void* y, int B, int T, int D, int K, int flip, void* stream);
long pseld_dwconv_wgrad_workspace(int B, int T, int D, int K);
int pseld_dwconv_wgrad(int dtype, const void* x, const void* dy, float* dw, int B, int T, int D, int K, float* workspace,
                       long workspace_bytes, void* stream);
int pseld_relattn_fwd(int dtype, const void* q, const void* k, const void* v, const float* pos, const float* u_bias,
                      const float* v_bias, const void* mask, float mask_scale, void* out, float* attn, int B, int T, int D,
                      int heads, void* stream);
long pseld_relattn_bwd_workspace(int B, int T, int D);
int pseld_relattn_bwd(int dtype, const void* q, const void* k, const void* v, const float* pos, const float* u_bias,
                      const float* v_bias, const void* mask, float mask_scale, const float* attn, const void* dout, void* dq,
                      void* dk, void* dv, float* dpos, float* du_bias, float* dv_bias, int B, int T, int D, int heads,
                      float* workspace, long workspace_bytes, void* stream);

/* mono -> FOA spatialisation of the mono_adapter fine-tuning recipe (data/data.py:17-59 generate_spatial_samples; the
 * host mirror draws (azimuth, elevation) per sample): foa [N,4,L] = (w, y*w, z*w, x*w) from mono rows at mono_stride, xyz
 * f64 [N,3] = (x, y, z), products in double rounded once as numpy does. spatial_label: out[n,o,a,i] = coef[n][a] *
 * lab[n,o,0,i] (a < 4; coef f64 [N,4]) — ADPIT [N,T*6,4,C] with (1,x,y,z), ACCDOA [N,T,4,C] with (0,x,y,z).
 * spatial_doa_label: EINV2 doa label [N,T,3,3], track 0 = (activity summed over tracks and classes) * (x,y,z), rest 0. */
int pseld_spatialize_mono(const float* mono, long mono_stride, const double* xyz, float* foa, int N, long L, void* stream);
int pseld_spatial_label(const float* lab, float* out, const double* coef, int N, long outer, long inner, void* stream);
int pseld_spatial_doa_label(const float* sed, const double* xyz, float* doa, int N, long T, int tracks, int C, void* stream);
/* ---- on-device augmentations -----------------------------------------------------------------
 * src/augment/specaug.py:14-63, crop.py:10-32, freqshift.py:17-38, rotate.py:10-101, trackmix.py:15-75,
 * wavmix.py:16-116; called from models/model_module.py:47-68 and components/model_module.py:83-121. The host mirror
 * draws the random parameters with the reference's generator calls; these entry points apply them to the batch.
 * All data fp32. rect_fill: x [N,C,T,F] in place, rects int32 [N*C][R][4] = (t0,t1,f0,f1) half-open (SpecAugment time
 * and iid frequency masks, Crop). time_fill: label [N,Ty,inner] in place, spans int32 [N][R][2]. freqshift: shift int32
 * [N], > 0 = 'up' (F.pad(x,(s,0),'reflect')[..., :F]), < 0 = 'down' (F.pad(x,(0,-s),'reflect')[..., -s:]). rotate_wave:
 * y[n,0] = x[n,0], y[n,1+j] = sign[n][j] * x[n, src[n][j]] (FOA [N,4,L], L % 4 == 0). rotate_label: label viewed
 * [N, outer, A, inner], y[..,a0+j,:] = sign[n][j] * x[..,a0+src[n][j],:], other positions of the axis copied.
 * mix: y[dst[p]] = lam[p]*x[dst[p]] + (1-lam[p])*x[src[p]] for P pairs of `elems` elements (y holds a copy of x).
 * mix_adpit: the ADPIT label surgery on [N,T,6,4,C]; mode 1 = partner with one source (TrackMix, WavMix add_ov '1'),
 * mode 2 = partner with up to two (WavMix add_ov '2'). mix_tracks: sed [N,T,3,C] / doa [N,T,3,3] track stacking
 * (wavmix = 0: third track zero; 1: third track = the partner's second). Outputs hold copies of the inputs. */
int pseld_aug_rect_fill(float* x, const int* rects, int N, int C, int T, int F, int R, float value, void* stream);
int pseld_aug_time_fill(float* y, const int* spans, int N, int Ty, long inner, int R, float value, void* stream);
int pseld_aug_freqshift(const float* x, float* y, const int* shift, int N, int C, int T, int F, void* stream);
int pseld_aug_rotate_wave(const float* x, float* y, const int* src, const float* sign, int N, long L, void* stream);
int pseld_aug_rotate_label(const float* x, float* y, const int* src, const float* sign, int N, long outer, int A, long inner,
                           int a0, void* stream);
int pseld_aug_mix(const float* x, float* y, const int* dst, const int* src, const float* lam, int P, long elems, void* stream);
int pseld_aug_mix_adpit(const float* lab, float* out, const int* dst, const int* src, const float* lam, int P, int T, int C, int mode,
                        void* stream);
int pseld_aug_mix_tracks(const float* sed, const float* doa, float* sed_out, float* doa_out, const int* dst, const int* src,
                         const float* lam, int P, int T, int C, int wavmix, void* stream);

/* ---- device-side ingest (device part) --------------------------------------------------------------
 * src/data/data.py:7-15 load_audio + :75-77 np.pad for the index rows of utils/data_utilities.py:6-64 segment_index, and the
 * label synthesis of data.py:87-93,207-213. pcm: int16 interleaved frames [total_frames][C] of all clips back to back
 * (resident in HBM); seg int64 [n][5] = (first frame of the clip, begin, end, pad_before, pad_after);
 * out f32 [n][C][chunk_len] = PCM / 32768, zero in the pads. polar_labels: se u8, azi i16, ele i8 (degrees), each
 * [rows*tracks][C] -> out f32 [rows*tracks][4][C] = (se, cos(az)cos(el)se, sin(az)cos(el)se, sin(el)se). */
int pseld_pcm16_chunks(const short* pcm, const long* seg, float* out, long n, int C, int chunk_len, void* stream);
int pseld_polar_labels(const unsigned char* se, const short* azi, const signed char* ele, float* out, long rows_tracks, int C,
                       void* stream);

/* ---- inference-side decoding (device part) --------------------------------------------------------
 * utils/data_utilities.py:234-244 get_accdoa_labels, :273-300 get_multi_accdoa_labels + :302-388
 * multi_accdoa_to_dcase_format (15-degree unification of same-class tracks), components/model_module.py:302-329
 * (moving average over overlapping test chunks). decode_maccdoa: pred f32 [rows, 9C] -> events f32 [rows, C, 3, 3]
 * (xyz of up to three events per frame and class, in the reference's order) and counts int32 [rows, C].
 * decode_accdoa: pred f32 [rows, 3C] -> sed u8 [rows, C] (among the max_ov largest norms and above the threshold).
 * move_avg: preds f32 [num_chunks, chunk_frames, D] of ONE recording -> out f32 [out_frames, D]; output block
 * i (hop_frames frames) = mean of the chunks covering it, frames >= valid_frames are zero. */
int pseld_decode_maccdoa(const float* pred, float* events, int* counts, long rows, int C, float sed_threshold, float unify_deg,
                         void* stream);
int pseld_decode_accdoa(const float* pred, unsigned char* sed, long rows, int C, float sed_threshold, int max_ov, void* stream);
int pseld_move_avg(const float* preds, float* out, int num_chunks, int chunk_frames, int hop_frames, int valid_frames, int out_frames,
                   long D, void* stream);

/* ---- Transformer decoder glue (CRNN decoder='transformer': components/model_utilities.py:256-259 nn.TransformerEncoder) ------
 * relu: y = max(u, 0) and its backward. sdpa_small: nn.MultiheadAttention's softmax(q k^T / sqrt(head_dim)) (x dropout mask) v
 * for T <= 128 on the relative-attention kernels with a zero positional table; zeros = f32 buffer of >= T*D zeros;
 * attn f32 [B,heads,T,T] kept for the backward; scratch f32 [T*D + 2*D]; workspace = pseld_relattn_bwd_workspace. */
int pseld_relu_fwd(int dtype, const void* u, void* y, long n, void* stream);
int pseld_relu_bwd(int dtype, const void* u, const void* dy, void* du, long n, void* stream);
int pseld_sdpa_small_fwd(int dtype, const void* q, const void* k, const void* v, const float* zeros, const void* mask, float mask_scale,
                         void* out, float* attn, int B, int T, int D, int heads, void* stream);
int pseld_sdpa_small_bwd(int dtype, const void* q, const void* k, const void* v, const float* zeros, const void* mask, float mask_scale,
                         const float* attn, const void* dout, void* dq, void* dk, void* dv, float* scratch, int B, int T, int D,
                         int heads, float* workspace, long workspace_bytes, void* stream);

/* ---- GRU decoder cell (CRNN decoder='gru': components/model_utilities.py:249-252 nn.GRU, configs/model/default.yaml) ------
 * Gate order r | z | n. The input projections of all timesteps, the recurrent product h_{t-1} W_hh^T + b_hh of a step and every
 * weight gradient are pseld_gemm / pseld_gemm_wgrad calls; these two kernels are the element-wise part of one timestep.
 * gate_fwd: gi rows at gi + b*gi_stride (3H wide), gh [B,3H], hprev rows at hp_stride (NULL = zeros) -> h rows at h_stride and
 * gates [B,4H] = (r | z | n | gh_n) kept for the backward. gate_bwd: dh rows at dh_stride (+ carry [B,H] or NULL) ->
 * dgi rows at dgi_stride = (dr | dz | dn) pre-activation gradients, dgh [B,3H] = (dr | dz | dn*r), dhprev [B,H] = dh*z
 * (the caller adds dgh W_hh). */
int pseld_gru_gate_fwd(int dtype, const void* gi, long gi_stride, const void* gh, const void* hprev, long hp_stride, void* h,
                       long h_stride, void* gates, int B, int H, void* stream);
/* The recurrence of one layer and direction in one call (T x (B-row GEMM + gate kernel) launched from C): gi [B,T,3H] input
 * projections, w_hh [3H,H] (compute dtype), b_hh f32; seq points at this direction's H columns of the layer output
 * [B,T,ld_seq]; gates [T,B,4H]; gh scratch [B,3H]. seq_bwd: dseq / seq as above, w_hh_t [H,3H] (transposed copy) or NULL,
 * writes dgi [B,T,3H], dgh [T,B,3H], hprev_all [T,B,H] (pre-zeroed by the caller); carry / direct scratch [B,H]. */
int pseld_gru_seq_fwd(int dtype, const void* gi, const void* w_hh, const float* b_hh, void* seq, long ld_seq, void* gates, void* gh,
                      int B, int T, int H, int reverse, void* stream);
int pseld_gru_seq_bwd(int dtype, const void* dseq, const void* seq, long ld_seq, const void* gates, const void* w_hh,
                      const void* w_hh_t, void* dgi, void* dgh, void* hprev_all, void* carry, void* direct, int B, int T, int H,
                      int reverse, void* stream);
/* n independent recurrences of ONE geometry (B, T, H) advanced together, one launch per timestep for all of them: the two
 * directions of a layer (model_utilities.py:250-252, bidirectional=True), or those of all six Decoder('gru') stacks of the
 * EINV2 tail (einv2.py:52-57,472-476). Every pointer argument marked [n] is a HOST array of n device pointers with the
 * meaning of the pseld_gru_seq_* argument of the same name; reverse [n] ints. Falls back to n sequential pseld_gru_seq_*
 * calls for geometries the fused step kernels do not cover (fp32, B > 64). */
int pseld_gru_multi_fwd(int dtype, int n, const void* const* gi, const void* const* w_hh, const float* const* b_hh,
                        void* const* seq, long ld_seq, void* const* gates, void* const* gh, const int* reverse, int B, int T,
                        int H, void* stream);
int pseld_gru_multi_bwd(int dtype, int n, const void* const* dseq, const void* const* seq, long ld_seq,
                        const void* const* gates, const void* const* w_hh, const void* const* w_hh_t, void* const* dgi,
                        void* const* dgh, void* const* hprev_all, void* const* carry, void* const* direct, const int* reverse,
                        int B, int T, int H, void* stream);
int pseld_gru_gate_bwd(int dtype, const void* dh, long dh_stride, const void* carry, const void* gates, const void* hprev,
                       long hp_stride, void* dgi, long dgi_stride, void* dgh, void* dhprev, int B, int H, void* stream);

/* ---- output head ---------------------------------------------------------------------------------------------------
 * htsat.py:526-534 (token -> [C,2,32] map) + im2col of accdoa.py:230 tscam_conv((2,3), pad (0,1)):
 * tok [B,64,C] -> A [B*32, C*6] (k = c*6 + cf*3 + dt, matching the conv weight's [D, C, 2, 3] flattening).
 * accdoa.py:231-242: z [B*32, ldz] -> y f32[B, n_out, D] through the fixed interpolate/crop/mean map given in
 * compact form (row f uses taps w[f][0..2] at inputs i0[f]..i0[f]+2), then tanh (act_tanh) or identity. */
int pseld_head_im2col(int dtype, const void* tok, void* A, int B, int C, void* stream);
int pseld_head_col2im(int dtype, const void* dA, void* dtok, int B, int C, void* stream);
int pseld_head_pool_fwd(int dtype, const void* z, float* y, const int* i0, const float* w, int B, int D, int ldz,
                        int n_out, int n_in, int act_tanh, void* stream);
int pseld_head_pool_bwd(int dtype, const float* dy, const float* y, void* dz, const int* t_cnt, const int* t_f,
                        const float* t_w, int B, int D, int ldz, int n_out, int n_in, int act_tanh, int max_taps,
                        void* stream);

/* ---- losses (value + gradient in one pass; fp32) ------------------------------------------------------------------
 * loss/multi_accdoa.py:16-105 ADPIT: pred [rows=B*T, 9, C] (row stride ldp), label [rows, 6, 4, C];
 * loss/accdoa.py:15-22 MSE; loss/einv2.py:59-116 tPIT (bce + mse, beta): loss_out[0..2] = all, sed, doa. */
int pseld_adpit_loss(const float* pred, const float* label, float* dpred, float* loss_out, long rows, int C, int ldp,
                     float* workspace, long workspace_bytes, void* stream);
int pseld_mse_loss(const float* pred, const float* target, float* dpred, float* loss_out, long n, float* workspace,
                   long workspace_bytes, void* stream);
int pseld_tpit_loss(const float* sed, const float* doa, const float* sed_label, const float* doa_label, float* dsed,
                    float* ddoa, float* loss_out, long rows, int C, float beta, float* workspace, long workspace_bytes,
                    void* stream);
/* model_utilities_adapt.py:19-20,40 (adapter_scalar: learnable_scalar): out[0] (+)= <a, b> / div[0] — the gradient of the
 * learnable scale s of an Adapter from its already s-scaled fc2 gradients: <dW2, W2> / s + <db2, b2> / s. workspace >= 1 KiB. */
int pseld_dot_div(const float* a, const float* b, long n, const float* div, float* out, int accumulate, float* workspace,
                  long workspace_bytes, void* stream);
/* loss/einv2.py:118-188 AGG loss (Losses_agg_pit; configs/loss/einv2_pit_agg.yaml): the EINV2 / SEDDOA outputs scored as
 * multi-ACCDOA vectors pred[k,c,:] = sigmoid(sed[k,c]) * normalize(doa[k,:]) against sed_label[k,c] * doa_label[k,:]:
 * agg = track-permutation-invariant mean error (:166-188), accdoa = mean error of the track sums (:148-153);
 * loss_all = w_agg * agg + w_acc * accdoa ((1,0) = method mACCDOA_pit, (0,1) = ACCDOA, (alpha, 1-alpha) otherwise);
 * l1 != 0 = loss_fn 'l1', else 'mse'. loss_out[0..2] = all, agg, accdoa; dsed / ddoa = gradients of loss_all. */
long pseld_agg_pit_loss_workspace(long rows);
int pseld_agg_pit_loss(const float* sed, const float* doa, const float* sed_label, const float* doa_label, float* dsed,
                       float* ddoa, float* loss_out, long rows, int C, float w_agg, float w_acc, int l1, float* workspace,
                       long workspace_bytes, void* stream);

/* ---- optimiser: clip_grad_norm_(max_norm) + AdamW over one flat fp32 arena --------------------------------------
 * models/components/model_module.py:128-146 (AdamW, torch defaults), configs/trainer/default.yaml:26 (clip 1.0). */
int pseld_grad_norm(const float* g, long n, float* norm_out, float* workspace, long workspace_bytes, void* stream);
int pseld_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, long n, const float* grad_norm,
                     float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, void* stream);
/* The same step for a training step captured into a hipGraph: lr and the two bias corrections come from device memory,
 * hyper = {lr, 1 - beta1^step, sqrt(1 - beta2^step)} (refreshed by the host in front of each replay), so that StepLR
 * (model_module.py:143-146) and the step count do not force a re-capture. */
void pseld_adamw_bias_corrections(float beta1, float beta2, int step, float* out2 /* host */);
int pseld_adamw_step_dev(float* p, const float* g, float* m, float* v, void* shadow_bf16, long n, const float* grad_norm,
                         float max_norm, float grad_scale, const float* hyper, float beta1, float beta2, float eps,
                         float weight_decay, void* stream);
int pseld_cast_f32_to_bf16(const float* x, void* y, long n, void* stream);
/* Transposed bf16 copies of the arena's 2-D weights, for the input-gradient GEMMs (dX = dY W as a k-contiguous product):
 * desc = n_desc x {element offset, rows, cols, first 32x32 tile} (device longs, n_desc <= 512), dst[off + c*rows + r] = src[off + r*cols + c]. */
int pseld_transpose_batch_bf16(const void* src, void* dst, const long* desc, int n_desc, long total_tiles, void* stream);

#ifdef __cplusplus
}
#endif
#endif
