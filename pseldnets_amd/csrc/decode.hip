// Inference-side decoding of the network outputs on the device (gfx950); tiny HBM-bound kernels.
//
// Replaces (reference, /root/reference/src): utils/data_utilities.py:273-300 get_multi_accdoa_labels (track activity =
// |xyz| > threshold) fused with :302-388 multi_accdoa_to_dcase_format (same-class tracks closer than 15 degrees are
// unified by averaging — including the reference's quirk for the "only tracks 0 and 2 are close" case), :234-244
// get_accdoa_labels (top-3 norms above the threshold), and the moving-average stitching of overlapping test chunks,
// models/components/model_module.py:302-329. The DCASE dictionaries / CSV files are assembled on the host from the
// compact event arrays these kernels write.
#include "common.h"

namespace {

__device__ __forceinline__ float ang_dist_deg(float x1, float y1, float z1, float x2, float y2, float z2) {
    // data_utilities.py:212-231 (fp32, as numpy computes it on fp32 scalars)
    const float n1 = sqrtf(x1 * x1 + y1 * y1 + z1 * z1 + 1e-10f), n2 = sqrtf(x2 * x2 + y2 * y2 + z2 * z2 + 1e-10f);
    float d = (x1 / n1) * (x2 / n2) + (y1 / n1) * (y2 / n2) + (z1 / n1) * (z2 / n2);
    d = fminf(fmaxf(d, -1.f), 1.f);
    return acosf(d) * 180.f / 3.14159265358979323846f;
}

// pred [rows, 9C] (track-major: columns (3*track + axis)*C + c). events [rows, C, 3, 3] xyz, counts [rows, C]
__global__ void decode_maccdoa_kernel(const float* __restrict__ pred, float* __restrict__ events, int* __restrict__ counts, int C,
                                      float thr, float unify, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % C);
    const long row = id / C;
    const float* p = pred + row * 9 * C + c;
    float e[3][3];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = p[(3 * k) * C], y = p[(3 * k + 1) * C], z = p[(3 * k + 2) * C];
        if (sqrtf(x * x + y * y + z * z) > thr) { e[n][0] = x; e[n][1] = y; e[n][2] = z; ++n; }
    }
    float o[3][3];
    int m = 0;
    auto put = [&](const float* v) { o[m][0] = v[0]; o[m][1] = v[1]; o[m][2] = v[2]; ++m; };
    auto avg2 = [&](int a, int b) { o[m][0] = (e[a][0] + e[b][0]) / 2; o[m][1] = (e[a][1] + e[b][1]) / 2; o[m][2] = (e[a][2] + e[b][2]) / 2; ++m; };
    if (n == 1) put(e[0]);
    else if (n == 2) {
        if (ang_dist_deg(e[0][0], e[0][1], e[0][2], e[1][0], e[1][1], e[1][2]) < unify) avg2(0, 1);
        else { put(e[0]); put(e[1]); }
    } else if (n == 3) {
        const int s01 = ang_dist_deg(e[0][0], e[0][1], e[0][2], e[1][0], e[1][1], e[1][2]) < unify;
        const int s12 = ang_dist_deg(e[1][0], e[1][1], e[1][2], e[2][0], e[2][1], e[2][2]) < unify;
        const int s02 = ang_dist_deg(e[0][0], e[0][1], e[0][2], e[2][0], e[2][1], e[2][2]) < unify;
        const int s = s01 + s12 + s02;
        if (s == 0) { put(e[0]); put(e[1]); put(e[2]); }
        else if (s == 1) {
            if (s01) { avg2(0, 1); put(e[2]); }
            else if (s12) { put(e[0]); avg2(1, 2); }
            else { put(e[0]); avg2(0, 2); }                    // data_utilities.py:371-377: event 0 and the (0,2) average, as the reference
        } else {
            o[0][0] = (e[0][0] + e[1][0] + e[2][0]) / 3; o[0][1] = (e[0][1] + e[1][1] + e[2][1]) / 3; o[0][2] = (e[0][2] + e[1][2] + e[2][2]) / 3;
            m = 1;
        }
    }
    counts[id] = m;
    float* out = events + id * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) out[i * 3 + j] = i < m ? o[i][j] : 0.f;
}

// pred [rows, 3C]: sed[row, c] = (norm_c is among the max_ov largest of the row) && norm_c > thr. One workgroup per row.
__global__ __launch_bounds__(256) void decode_accdoa_kernel(const float* __restrict__ pred, unsigned char* __restrict__ sed, int C, float thr,
                                                            int max_ov) {
    extern __shared__ float norms[];
    const long row = blockIdx.x;
    const float* p = pred + row * 3 * C;
    for (int c = threadIdx.x; c < C; c += 256) { const float x = p[c], y = p[C + c], z = p[2 * C + c]; norms[c] = sqrtf(x * x + y * y + z * z); }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = norms[c];
        int rank = 0;
        for (int j = 0; j < C; ++j) rank += (norms[j] > v) || (norms[j] == v && j < c);
        sed[row * C + c] = (rank < max_ov && v > thr) ? 1 : 0;
    }
}

// out[f, :] = mean over the chunks j that cover output block i = f / hop: preds[j, (i - j) * hop + f % hop, :]
__global__ void move_avg_kernel(const float* __restrict__ preds, float* __restrict__ out, int num_chunks, int chunk_frames, int hop, int reach,
                                int valid_frames, long D, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const long d = id % D;
    const int f = (int)(id / D);
    float v = 0.f;
    if (f < valid_frames) {
        const int i = f / hop, off = f - i * hop;
        const int lo = max(0, i - reach + 1), hi = min(i + 1, num_chunks);
        float s = 0.f;
        for (int j = lo; j < hi; ++j) s += preds[((long)j * chunk_frames + (i - j) * hop + off) * D + d];
        v = s / (float)(hi - lo);
    }
    out[id] = v;
}

}  // namespace

extern "C" int pseld_decode_maccdoa(const float* pred, float* events, int* counts, long rows, int C, float sed_threshold, float unify_deg,
                                    void* stream) {
    PSELD_CHECK_ARG(pred && events && counts && rows > 0 && C > 0, "decode_maccdoa: bad argument");
    const long total = rows * C;
    hipLaunchKernelGGL(decode_maccdoa_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, pred, events, counts, C, sed_threshold,
                       unify_deg, total);
    PSELD_LAUNCH_CHECK("decode_maccdoa");
    return PSELD_OK;
}
extern "C" int pseld_decode_accdoa(const float* pred, unsigned char* sed, long rows, int C, float sed_threshold, int max_ov, void* stream) {
    PSELD_CHECK_ARG(pred && sed && rows > 0 && C > 0 && C <= 8192 && max_ov > 0, "decode_accdoa: bad argument");
    hipLaunchKernelGGL(decode_accdoa_kernel, dim3((unsigned)rows), dim3(256), C * sizeof(float), (hipStream_t)stream, pred, sed, C, sed_threshold, max_ov);
    PSELD_LAUNCH_CHECK("decode_accdoa");
    return PSELD_OK;
}
extern "C" int pseld_move_avg(const float* preds, float* out, int num_chunks, int chunk_frames, int hop_frames, int valid_frames, int out_frames,
                              long D, void* stream) {
    PSELD_CHECK_ARG(preds && out && num_chunks > 0 && chunk_frames > 0 && hop_frames > 0 && chunk_frames % hop_frames == 0 && out_frames > 0 && D > 0,
                    "move_avg: bad argument");
    const long total = (long)out_frames * D;
    hipLaunchKernelGGL(move_avg_kernel, dim3(pseld_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, preds, out, num_chunks, chunk_frames, hop_frames,
                       chunk_frames / hop_frames, valid_frames, D, total);
    PSELD_LAUNCH_CHECK("move_avg");
    return PSELD_OK;
}
