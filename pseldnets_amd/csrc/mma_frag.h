// MFMA 32x32 operand-fragment helpers shared by the attention kernels (attn.hip: Swin windows; mhsa.hip: global).
// bf16 uses v_mfma_f32_32x32x16_bf16 with ds_read_b64_tr_b16 transposed reads; f32 (parity mode) uses
// v_mfma_f32_32x32x2_f32 with the same logical fragment shapes, so every kernel body is written once.
#pragma once
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;
typedef __attribute__((ext_vector_type(8))) short short8v;

template <typename T> struct AMma;
template <> struct AMma<bf16_t> {
    using Frag = bf16x8;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ Frag keep_if(const Frag& f, bool live) {
        f32x4 v = __builtin_bit_cast(f32x4, f);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = live ? v[i] : 0.f;
        return __builtin_bit_cast(Frag, v);
    }
    static __device__ __forceinline__ Frag from_acc(const f32x16& a, int s) {
        Frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (bf16_t)a[8 * s + j];
        return f;
    }
    // 8 consecutive elements of one LDS row
    static __device__ __forceinline__ Frag ld_row(const char* p) { return *(const bf16x8*)p; }
    // element j = M[row0 + 8*(j>>2) + 4*h2 + (j&3)][col0 + (lane&31)]  (the k-order of an accumulator operand)
    static __device__ __forceinline__ Frag ld_cols(const char* base, int strideB, int row0, int col0, int lane) {
        const int i = lane & 15, q = i >> 2, p = i & 3, gsel = (lane >> 4) & 1, h2 = lane >> 5;
        const char* addr = base + (row0 + 4 * h2 + q) * strideB + (col0 + 16 * gsel + 4 * p) * 2;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr + 8 * strideB));
        short8v s;
        s[0] = lo[0]; s[1] = lo[1]; s[2] = lo[2]; s[3] = lo[3];
        s[4] = hi[0]; s[5] = hi[1]; s[6] = hi[2]; s[7] = hi[3];
        return __builtin_bit_cast(bf16x8, s);
    }
    // element j = M[row0 + 8*h2 + j][col0 + (lane&31)]  (the standard k-order of an operand fragment)
    static __device__ __forceinline__ Frag ld_cols_std(const char* base, int strideB, int row0, int col0, int lane) {
        const int i = lane & 15, q = i >> 2, p = i & 3, gsel = (lane >> 4) & 1, h2 = lane >> 5;
        const char* addr = base + (row0 + 8 * h2 + q) * strideB + (col0 + 16 * gsel + 4 * p) * 2;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr + 4 * strideB));
        short8v s;
        s[0] = lo[0]; s[1] = lo[1]; s[2] = lo[2]; s[3] = lo[3];
        s[4] = hi[0]; s[5] = hi[1]; s[6] = hi[2]; s[7] = hi[3];
        return __builtin_bit_cast(bf16x8, s);
    }
    // ld_cols_std on a [32 rows][32 bf16] image of 64-byte rows whose 8-byte pieces are XOR-swizzled by (row >> 1) & 7
    // (img_piece below): the 8-byte image WRITES of 16 consecutive rows then fall in 16 different bank pairs instead of 2, and
    // the transposed reads stay conflict-free (a row's pieces are permuted inside the row)
    static __device__ __forceinline__ Frag ld_img(const char* base, int row0, int lane) {
        const int i = lane & 15, q = i >> 2, p = i & 3, gsel = (lane >> 4) & 1, h2 = lane >> 5;
        const int row = row0 + 8 * h2 + q, c = 4 * gsel + p;
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base + row * 64 + ((c ^ ((row >> 1) & 7)) << 3)));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base + (row + 4) * 64 + ((c ^ (((row + 4) >> 1) & 7)) << 3)));
        short8v s;
        s[0] = lo[0]; s[1] = lo[1]; s[2] = lo[2]; s[3] = lo[3];
        s[4] = hi[0]; s[5] = hi[1]; s[6] = hi[2]; s[7] = hi[3];
        return __builtin_bit_cast(bf16x8, s);
    }
    // byte offset of the 4-key piece `piece` (0..7) of image row `row`
    static __device__ __forceinline__ int img_piece(int row, int piece) { return row * 64 + ((piece ^ ((row >> 1) & 7)) << 3); }
};
template <typename T> __device__ __forceinline__ void store4(char* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<bf16_t>(char* p, float a, float b, float c, float d) {
    bf16x4 v; v[0] = (bf16_t)a; v[1] = (bf16_t)b; v[2] = (bf16_t)c; v[3] = (bf16_t)d;
    *(bf16x4*)p = v;
}
template <> __device__ __forceinline__ void store4<float>(char* p, float a, float b, float c, float d) {
    f32x4 v = {a, b, c, d};
    *(f32x4*)p = v;
}
struct AFragF32 { float v[8]; };
template <> struct AMma<float> {
    using Frag = AFragF32;
    static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], c, 0, 0, 0);
    }
    static __device__ __forceinline__ Frag keep_if(Frag f, bool live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f.v[i] = live ? f.v[i] : 0.f;
        return f;
    }
    static __device__ __forceinline__ Frag from_acc(const f32x16& a, int s) {
        Frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = a[8 * s + j];
        return f;
    }
    static __device__ __forceinline__ Frag ld_row(const char* p) {
        const f32x4 a = ((const f32x4*)p)[0], b = ((const f32x4*)p)[1];
        Frag f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
        return f;
    }
    static __device__ __forceinline__ Frag ld_cols(const char* base, int strideB, int row0, int col0, int lane) {
        const int r = lane & 31, h2 = lane >> 5;
        Frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            f.v[j] = *(const float*)(base + (row0 + 8 * (j >> 2) + 4 * h2 + (j & 3)) * strideB + (col0 + r) * 4);
        return f;
    }
    static __device__ __forceinline__ Frag ld_cols_std(const char* base, int strideB, int row0, int col0, int lane) {
        const int r = lane & 31, h2 = lane >> 5;
        Frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = *(const float*)(base + (row0 + 8 * h2 + j) * strideB + (col0 + r) * 4);
        return f;
    }
    // the f32 images (144-byte rows of 32 floats) are not swizzled: same entry points as the bf16 struct
    static __device__ __forceinline__ Frag ld_img(const char* base, int row0, int lane) { return ld_cols_std(base, 144, row0, 0, lane); }
    static __device__ __forceinline__ int img_piece(int row, int piece) { return row * 144 + piece * 16; }
};

// row held by accumulator register e of a 32x32 tile for lane half h2
__device__ __forceinline__ int acc_row(int e, int h2) { return (e & 3) + 8 * (e >> 2) + 4 * h2; }

}  // namespace
