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
 *   + bias[N] (epi&1) ; * rowscale[m / rows_per_scale] (if rowscale) ; * gelu'(aux[m,n]) (epi&4) ;
 *   + resid[m,n] (epi&2) ; + C_old (epi&8).
 * pro&1 applies exact-erf GELU to A on load (fc2 consumes the stored pre-activation). */
int pseld_gemm(int dtype, int trans_a, int trans_b, const void* A, const void* B, void* C, int M, int N, int K,
               int lda, int ldb, int ldc, const float* bias, const void* resid, int ldr, const float* rowscale,
               int rows_per_scale, const void* aux, int ldaux, int epi, int pro, void* stream);

/* Weight gradient of the same layers: dW f32[N,K] (+)= dY[Mtok,N]^T @ (gelu_on_x ? gelu(X) : X)[Mtok,K];
 * split over tokens into fp32 slabs in `workspace`, reduced in a fixed order (bitwise reproducible). */
long pseld_gemm_wgrad_workspace(int Mtok, int N, int K, int* splits_out);
int pseld_gemm_wgrad(int dtype, const void* dY, const void* X, float* dW, int Mtok, int N, int K, int lddy, int ldx,
                     int lddw, int gelu_on_x, int accumulate, float* workspace, long workspace_bytes, void* stream);

/* Bias gradient: out f32[N] (+)= sum_m X[m,n]. */
long pseld_colsum_workspace(int M, int N);
int pseld_colsum(int dtype, const void* X, float* out, int M, int N, int ld, int accumulate, float* workspace,
                 long workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
