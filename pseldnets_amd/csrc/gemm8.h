// Persistent 256 x 256 x 64 eight-phase bf16 GEMM (gemm8.hip): the descriptor pseld_gemm / pseld_gemm_wgrad hand over.
#pragma once
#include "common.h"

struct Gemm8Desc {
    const void* A;          // [M, lda] bf16, k-contiguous rows
    const void* B;          // [N, ldb] bf16, k-contiguous rows (C = A B^T)
    void* C;                // [M, ldc] bf16
    void* C2;               // GELU-dual: second output gelu'(v), [M, ldc]
    const float* bias;      // [N] or null
    const void* resid;      // [M, ldr] bf16 or null
    const void* aux;        // [M, ldaux] bf16 multiplier or null
    const float* rowscale;  // [ceil(M / rows_per_scale)] or null
    int M, N, K;
    int lda, ldb, ldc, ldr, ldaux;
    int rows_per_scale;
    int gelu_dual;
};

// 1 when the eight-phase kernel takes the product (shape / alignment / size limits), else 0
int pseld_gemm8_supported(const Gemm8Desc& d);
int pseld_gemm8_launch(const Gemm8Desc& d, hipStream_t stream);
const char* pseld_gemm8_last_symbol();      // instantiation of the last launch, as rocprofv3 prints it

// Row-panel-stationary kernel for K = 192 / 384 (gemm8p.hip): the A rows stay in registers, the weights stream through LDS. Same bits as
// the eight-phase kernel.
int pseld_gemm8p_supported(const Gemm8Desc& d);
int pseld_gemm8p_wanted(const Gemm8Desc& d);       // the routing knobs' say (epilogue kinds, K)
int pseld_gemm8p_launch(const Gemm8Desc& d, hipStream_t stream);
const char* pseld_gemm8p_last_symbol();

// weight gradient on the same loop (gemm8w.hip): plan returns the split count (0 = shape not taken), tile width and tokens per split
int pseld_gemm8w_plan(int Mtok, int N, int K, int lddy, int ldx, int rows_per_scale, int has_rowscale, int max_splits, int* bn_out, int* kchunk_out);
int pseld_gemm8w_launch(const void* dY, const void* X, float* slabs, float* colsum, long slab_stride, long colsum_stride, int Mtok, int N,
                        int K, int lddy, int ldx, int bn, int kchunk, int splits, const float* rowscale, int rows_per_scale, hipStream_t stream);
