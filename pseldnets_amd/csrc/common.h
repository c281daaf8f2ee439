// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the SELD hot path.
// Wave size is 64 everywhere; nothing here compiles for any other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define PSELD_WAVE 64

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ---- status codes returned by every extern "C" entry point -----------------------------------
enum {
    PSELD_OK = 0,
    PSELD_ERR_BAD_ARG = -1,
    PSELD_ERR_UNSUPPORTED = -2,
    PSELD_ERR_HIP = -3,
};
enum { PSELD_F32 = 0, PSELD_BF16 = 1 };

void pseld_set_error(const char* fmt, ...);

#define PSELD_CHECK_ARG(cond, ...)                                                               \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            pseld_set_error(__VA_ARGS__);                                                        \
            return PSELD_ERR_BAD_ARG;                                                            \
        }                                                                                        \
    } while (0)

#define PSELD_LAUNCH_CHECK(name)                                                                 \
    do {                                                                                         \
        hipError_t e__ = hipGetLastError();                                                      \
        if (e__ != hipSuccess) {                                                                 \
            pseld_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));              \
            return PSELD_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)

static inline int pseld_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- routing knobs ---------------------------------------------------------------------------------
// The A/B switches of the measurement tools and tests. Every knob has a FROZEN default in the code that asks for it (the routing the
// committed numbers were measured with). The environment (PSELD_<NAME>) is read ONCE per process, at the first query of any knob
// (runtime.hip), and pseld_set_knob() / pseld_unset_knob() change one in-process (tests, tools). No launch path calls getenv: which
// kernel a layer gets is fixed when the library is loaded, and a query is an array read.
#define PSELD_KNOB_LIST(X)                                                                                                            \
    X(ATTN_HG) X(ATTN_FWD_P) X(ATTN_BWD_WGS) X(ATTN_BWD_V2) X(ALLOW_WRONG_RESULTS) X(GRU_BWD_UNITS) X(GEMM_XCD)       \
    X(GEMM_BIG) X(WGRAD_TILE) X(GEMM8) X(GEMM8_MINK) X(GEMM_DMA) X(GEMM_FWD_RING) X(GEMM_RING3) X(WGRAD_FILL) X(WGRAD_MINTOK)         \
    X(WGRAD_RING) X(WGRAD8_MINN) X(WGRAD8) X(GEMM8_BN) X(GEMM8_BM) X(LN_EXACT) X(GEMM8W_BN) X(MLP_VARIANT) X(GEMM8P) X(GEMM8P_MINM) X(GEMM8P_MODES) X(GEMM8P_K) X(GEMM8P_STAG)
enum PseldKnob {
#define PSELD_KNOB_ENUM(n) KNOB_##n,
    PSELD_KNOB_LIST(PSELD_KNOB_ENUM)
#undef PSELD_KNOB_ENUM
    KNOB_COUNT
};
int pseld_knob(PseldKnob k, int dflt);       // the value the environment / pseld_set_knob gave the knob, dflt when it is unset
bool pseld_knob_is_set(PseldKnob k);

// ---- element conversion -------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 8-element vector of T (16 B for bf16, 32 B for f32): the unit every kernel moves.
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    f32x4 lo, hi;
    __device__ __forceinline__ float get(int i) const { return i < 4 ? lo[i] : hi[i - 4]; }
    __device__ __forceinline__ void set(int i, float v) { if (i < 4) lo[i] = v; else hi[i - 4] = v; }
};
template <> struct Vec8<bf16_t> {
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (bf16_t)f; }
};

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&out)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&out)[8]) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { out[i] = a[i]; out[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&out)[8]) {
    bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&in)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&in)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[i]; b[i] = in[4 + i]; }
    *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&in)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)in[i];
    *(bf16x8*)p = a;
}

// two consecutive elements
template <typename T> __device__ __forceinline__ void load2(const T* p, float (&out)[2]);
template <> __device__ __forceinline__ void load2<float>(const float* p, float (&out)[2]) { const f32x2 a = *(const f32x2*)p; out[0] = a[0]; out[1] = a[1]; }
template <> __device__ __forceinline__ void load2<bf16_t>(const bf16_t* p, float (&out)[2]) { const bf16x2 a = *(const bf16x2*)p; out[0] = (float)a[0]; out[1] = (float)a[1]; }
template <typename T> __device__ __forceinline__ void store2(T* p, float a, float b);
template <> __device__ __forceinline__ void store2<float>(float* p, float a, float b) { *(f32x2*)p = f32x2{a, b}; }
template <> __device__ __forceinline__ void store2<bf16_t>(bf16_t* p, float a, float b) { bf16x2 v; v[0] = (bf16_t)a; v[1] = (bf16_t)b; *(bf16x2*)p = v; }

// ---- wave-level reductions (64 lanes) -----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over an aligned group of G lanes (G power of two <= 64)
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact-erf GELU and its derivative (reference: nn.GELU default, model_utilities.py:145-166).
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. below fp32 round-off of the product x*cdf for
// the activations seen here): 1 v_rcp_f32 (the raw 1-ulp instruction: __frcp_rn expands to the correctly rounded
// division sequence, 8 more instructions) + 1 v_exp_f32 + 6 FMA instead of the ~35-instruction libm erff, which
// matters because GELU runs in the epilogue of every fc1 GEMM.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
// gelu(x) and gelu'(x) together (they share the polynomial and exp(-x^2/2) = the A&S exponential)
__device__ __forceinline__ void gelu_both(float x, float& y, float& dy) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);                 // = exp(-x^2 / 2)
    const float erfv = copysignf(1.0f - p * t * e, x);
    const float cdf = 0.5f * (1.0f + erfv);
    y = x * cdf;
    dy = cdf + x * 0.3989422804014327f * e;
}
// the same for two values at once: the polynomial runs on v_pk_fma_f32 / v_pk_mul_f32 (two fp32 lanes per instruction)
__device__ __forceinline__ void gelu_both2(f32x2 x, f32x2& y, f32x2& dy) {
    f32x2 ax;
    ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
    ax = ax * 0.70710678118654752f;
    const f32x2 den = ax * 0.3275911f + 1.0f;
    f32x2 t;
    t[0] = __builtin_amdgcn_rcpf(den[0]); t[1] = __builtin_amdgcn_rcpf(den[1]);
    f32x2 p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    const f32x2 a2 = ax * ax * (-1.4426950408889634f);           // exp(-ax^2) = exp2(-ax^2 * log2 e)
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(a2[0]); e[1] = __builtin_amdgcn_exp2f(a2[1]);
    f32x2 er = 1.0f - p * t * e;
    er[0] = copysignf(er[0], x[0]); er[1] = copysignf(er[1], x[1]);
    const f32x2 cdf = er * 0.5f + 0.5f;
    y = x * cdf;
    dy = x * 0.3989422804014327f * e + cdf;
}
// gelu'(x) alone, one exponential (the A&S erf and the Gaussian density share exp(-x^2 / 2))
__device__ __forceinline__ float gelu_grad_shared(float x) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);                 // = exp(-x^2 / 2)
    const float cdf = 0.5f + copysignf(0.5f - 0.5f * p * t * e, x);
    return fmaf(x * 0.3989422804014327f, e, cdf);
}
// ---- GELU and its derivative from an LDS table with linear interpolation (the fused MLP kernels in bf16 mode; f32 parity mode keeps the
// A&S erf). 512 entries of 16 bytes over [-8, 8): {Phi(x0), Phi(x1) - Phi(x0), gelu'(x0), gelu'(x1) - gelu'(x0)}, x1 = x0 + 1/32, Phi the
// normal CDF: gelu(x) = x Phi(x). Both tabulated functions are CONSTANT outside the table (0 / 0 below -8, 1 / 1 above +8 to 1e-14), so the
// clamped index is all the range handling there is - round 5: the table used to hold gelu itself, which needs a compare + select for
// x >= 8 per element; two vector instructions less in the forward and dx kernels' inner loops. |error| <= 1.5e-5 |x| for gelu (h^2 / 16
// max|Phi''| = 1.5e-5, every chord shifted by half its midpoint gap: centred, no systematic bias against the erf evaluation) and 6e-5 for
// gelu', a sixtieth of the bf16 spacing at 1 the results are rounded to. One ds_read_b128 + 2-3 FMA-class + index arithmetic per element
// against 1 v_rcp + 1 v_exp + ~12 FMA-class instructions (tools/experiments/gelu_table.hip: 59 against 85 cycles per wave for the pair).
constexpr int GELU_TAB_N = 512;
constexpr int GELU_TAB_BYTES = GELU_TAB_N * 16;
__device__ __forceinline__ void gelu_tab_fill(f32x4* tab, int tid, int nthreads) {
    auto phi_d = [](float x, float& phi, float& d) {          // Phi(x) and gelu'(x) = Phi + x pdf: gelu_both's A&S evaluation, the CDF kept
        const float ax = fabsf(x) * 0.70710678118654752f;
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
        float p = fmaf(1.061405429f, t, -1.453152027f);
        p = fmaf(p, t, 1.421413741f);
        p = fmaf(p, t, -0.284496736f);
        p = fmaf(p, t, 0.254829592f);
        const float e = __expf(-ax * ax);
        phi = 0.5f * (1.0f + copysignf(1.0f - p * t * e, x));
        d = phi + x * 0.3989422804014327f * e;
    };
    for (int i = tid; i < GELU_TAB_N; i += nthreads) {
        const float x0 = (float)(i - GELU_TAB_N / 2) * (1.0f / 32.0f), x1 = x0 + (1.0f / 32.0f);
        float p0, d0, p1, d1, pm, dm;
        phi_d(x0, p0, d0);
        phi_d(x1, p1, d1);
        phi_d(x0 + (1.0f / 64.0f), pm, dm);
        // the chord of a convex (concave) piece lies above (below) the function, by at most its gap at the midpoint: each interval's chord is
        // lowered by half that gap, so the error is centred
        const float ep = 0.5f * (0.5f * (p0 + p1) - pm), ed = 0.5f * (0.5f * (d0 + d1) - dm);
        tab[i] = f32x4{p0 - ep, p1 - p0, d0 - ed, d1 - d0};
    }
}
__device__ __forceinline__ f32x4 gelu_tab_entry(const f32x4* tab, float x, float& fr) {
    const float u = __builtin_amdgcn_fmed3f(fmaf(x, 32.0f, (float)(GELU_TAB_N / 2)), 0.0f, (float)GELU_TAB_N - 0.01f);
    fr = __builtin_amdgcn_fractf(u);
    return tab[(int)u];
}
__device__ __forceinline__ float gelu_tab_f(const f32x4* tab, float x) {
    float fr;
    const f32x4 e = gelu_tab_entry(tab, x, fr);
    return x * fmaf(e[1], fr, e[0]);
}
__device__ __forceinline__ float gelu_tab_grad(const f32x4* tab, float x) {
    float fr;
    const f32x4 e = gelu_tab_entry(tab, x, fr);
    return fmaf(e[3], fr, e[2]);
}
__device__ __forceinline__ void gelu_tab_both(const f32x4* tab, float x, float& y, float& dy) {
    float fr;
    const f32x4 e = gelu_tab_entry(tab, x, fr);
    y = x * fmaf(e[1], fr, e[0]);
    dy = fmaf(e[3], fr, e[2]);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
