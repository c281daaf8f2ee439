// libpseld_comm.so: gradient all-reduce of the data-parallel loop on RCCL (include/pseld_comm.h). gfx950 / xGMI only.
//
// Two algorithms behind one entry point. ALGO_RCCL hands the bucket to ncclAllReduce. ALGO_DIRECT is written for what xGMI is: a full
// mesh of point-to-point links (7 per GPU on an 8-GPU node, ~153 GB/s each). A ring all-reduce of S bytes pushes 2 (W-1)/W S through
// one link per direction (1.58 ms for the 138 MB fp32 gradient arena at W = 8); sending chunk p straight to rank p and the reduced
// chunk straight back uses all W-1 links at once: S / W per link per phase (~0.23 ms). RCCL's grouped ncclSend / ncclRecv
// give exactly that traffic pattern; the only arithmetic is this file's fixed-order sum kernel.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../../include/pseld_comm.h"

namespace {
thread_local char g_err[512] = "";
void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
#define COMM_CHECK_ARG(cond, ...) do { if (!(cond)) { set_err(__VA_ARGS__); return -1; } } while (0)
#define COMM_NCCL(call, what) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { set_err("%s: %s", what, ncclGetErrorString(r__)); return -3; } } while (0)

struct Comm { ncclComm_t nccl; int rank, world; bool aborted; };

// An error between the two phases of the DIRECT algorithm (or inside one of its ncclGroups) leaves the peers, which have entered the phase,
// waiting for this rank for ever: close the open group, abort the communicator (the peers' pending operations then fail instead of hanging)
// and report. Every later call on the communicator fails at once; the caller is expected to exit non-zero so that the launcher tears the
// job down.
static int comm_fatal(Comm* c, bool in_group, const char* what, const char* why) {
    set_err("%s: %s (communicator aborted)", what, why);
    if (in_group) (void)ncclGroupEnd();
    if (!c->aborted) { c->aborted = true; (void)ncclCommAbort(c->nccl); }
    return -3;
}
#define COMM_NCCL_FATAL(c, in_group, call, what) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) return comm_fatal(c, in_group, what, ncclGetErrorString(r__)); } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

// own[i] = sum over ranks r = 0..W-1 (in that order) of the copy rank r holds of this chunk: rank `me`'s copy is `own` itself, rank r's
// (r != me) arrived at scratch + slot(r) * stride, slot(r) = r < me ? r : r - 1. n4 = chunk length in 16-byte pieces.
template <typename V>
__global__ __launch_bounds__(256) void sum_in_rank_order(V* __restrict__ own, const V* __restrict__ scratch, long n, long stride, int me, int world) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const V mine = own[i];
        bool first = true;
        f4 acc4 = {0.f, 0.f, 0.f, 0.f};
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r) {
            const V v = r == me ? mine : scratch[(long)(r < me ? r : r - 1) * stride + i];
            if constexpr (sizeof(V) == 16 && __is_same(V, f4)) {
                acc4 = first ? v : acc4 + v;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc8[k] = first ? (float)v[k] : acc8[k] + (float)v[k];
            }
            first = false;
        }
        if constexpr (__is_same(V, f4)) own[i] = acc4;
        else {
            V o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (__bf16)acc8[k];      // bf16 payload: summed in fp32, rounded once
            own[i] = o;
        }
    }
}
}  // namespace

extern "C" const char* pseld_comm_last_error(void) { return g_err; }

extern "C" int pseld_comm_unique_id(void* id_out) {
    COMM_CHECK_ARG(id_out, "comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) <= PSELD_COMM_ID_BYTES, "ncclUniqueId does not fit the id buffer");
    ncclUniqueId id;
    COMM_NCCL(ncclGetUniqueId(&id), "ncclGetUniqueId");
    memset(id_out, 0, PSELD_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof id);
    return 0;
}

extern "C" int pseld_comm_init(const void* idb, int rank, int world, void** comm_out) {
    COMM_CHECK_ARG(idb && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d of %d)", rank, world);
    ncclUniqueId id;
    memcpy(&id, idb, sizeof id);
    Comm* c = new Comm{nullptr, rank, world, false};
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { set_err("ncclCommInitRank(rank %d of %d): %s", rank, world, ncclGetErrorString(r)); delete c; return -3; }
    *comm_out = c;
    return 0;
}
extern "C" int pseld_comm_rank(void* comm) { return comm ? ((Comm*)comm)->rank : -1; }
extern "C" int pseld_comm_world(void* comm) { return comm ? ((Comm*)comm)->world : -1; }

extern "C" long pseld_comm_direct_plan(long count, int world, long* off, long* len) {
    if (count < 0 || world < 1) return -1;
    // chunk stride: ceil(count / world) rounded up to 8 elements (16-byte pieces for f32 and bf16 alike)
    long stride = (count + world - 1) / world;
    stride = (stride + 7) / 8 * 8;
    for (int p = 0; p < world; ++p) {
        const long o = (long)p * stride;
        if (off) off[p] = o < count ? o : count;
        if (len) len[p] = o >= count ? 0 : (count - o < stride ? count - o : stride);
    }
    return stride;
}

extern "C" long pseld_comm_scratch_bytes(void* comm, long count, int elem_bytes, int algo) {
    if (!comm || algo != PSELD_COMM_ALGO_DIRECT) return 0;
    const Comm* c = (const Comm*)comm;
    if (c->world == 1) return 0;
    return (long)(c->world - 1) * pseld_comm_direct_plan(count, c->world, nullptr, nullptr) * elem_bytes;
}

extern "C" int pseld_comm_allreduce_bucket(void* comm, void* buf, long count, int dtype, int algo, void* scratch, long scratch_bytes, void* stream) {
    COMM_CHECK_ARG(comm && buf && count >= 0 && (dtype == 0 || dtype == 1), "comm_allreduce_bucket: bad arguments");
    Comm* c = (Comm*)comm;
    hipStream_t s = (hipStream_t)stream;
    if (c->world == 1 || count == 0) return 0;
    COMM_CHECK_ARG(!c->aborted, "comm_allreduce_bucket: the communicator was aborted by an earlier error");
    const ncclDataType_t dt = dtype == 0 ? ncclFloat32 : ncclBfloat16;
    const int es = dtype == 0 ? 4 : 2;
    if (algo == PSELD_COMM_ALGO_RCCL) {
        COMM_NCCL(ncclAllReduce(buf, buf, (size_t)count, dt, ncclSum, c->nccl, s), "ncclAllReduce");
        return 0;
    }
    COMM_CHECK_ARG(algo == PSELD_COMM_ALGO_DIRECT, "comm_allreduce_bucket: unknown algorithm %d", algo);
    COMM_CHECK_ARG(c->world <= 64, "comm_allreduce_bucket: direct algorithm is built for one node (world %d)", c->world);
    COMM_CHECK_ARG(((unsigned long)buf & 15) == 0 && ((unsigned long)scratch & 15) == 0, "comm_allreduce_bucket: buffers must be 16-byte aligned");
    long off[64], len[64];
    const long stride = pseld_comm_direct_plan(count, c->world, off, len);
    // a bucket whose length is not a multiple of 8 elements would leave one chunk with a ragged 16-byte tail: such a bucket (every rank
    // sees the same count and takes the same branch) goes through ncclAllReduce instead
    if (count % 8 != 0) {
        COMM_NCCL(ncclAllReduce(buf, buf, (size_t)count, dt, ncclSum, c->nccl, s), "ncclAllReduce");
        return 0;
    }
    COMM_CHECK_ARG(scratch && scratch_bytes >= (long)(c->world - 1) * stride * es, "comm_allreduce_bucket: scratch %ld bytes < %ld", scratch_bytes,
                   (long)(c->world - 1) * stride * es);
    char* b = (char*)buf;
    char* sc = (char*)scratch;
    const int me = c->rank;
    // phase 1 (reduce-scatter): my copy of chunk p -> rank p; rank r's copy of chunk `me` -> scratch slot of r
    COMM_NCCL_FATAL(c, false, ncclGroupStart(), "ncclGroupStart");
    for (int p = 0; p < c->world; ++p) {
        if (p == me) continue;
        if (len[p] > 0) COMM_NCCL_FATAL(c, true, ncclSend(b + off[p] * es, (size_t)len[p], dt, p, c->nccl, s), "ncclSend(reduce-scatter)");
        if (len[me] > 0) COMM_NCCL_FATAL(c, true, ncclRecv(sc + (long)(p < me ? p : p - 1) * stride * es, (size_t)len[me], dt, p, c->nccl, s), "ncclRecv(reduce-scatter)");
    }
    COMM_NCCL_FATAL(c, false, ncclGroupEnd(), "ncclGroupEnd(reduce-scatter)");
    if (len[me] > 0) {
        // (count % 8 == 0 and the stride is a multiple of 8: every chunk is whole 16-byte pieces)
        const long n16 = len[me] * es / 16;
        const int blocks = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
        if (dtype == 0) hipLaunchKernelGGL(sum_in_rank_order<f4>, dim3(blocks), dim3(256), 0, s, (f4*)(b + off[me] * es), (const f4*)sc, n16, stride * es / 16, me, c->world);
        else hipLaunchKernelGGL(sum_in_rank_order<bf16x8v>, dim3(blocks), dim3(256), 0, s, (bf16x8v*)(b + off[me] * es), (const bf16x8v*)sc, n16, stride * es / 16, me, c->world);
        // (the peers are already on their way into phase 2: a rank that stops here must not leave them waiting)
        if (hipGetLastError() != hipSuccess) return comm_fatal(c, false, "comm_allreduce_bucket", "sum kernel launch failed");
    }
    // phase 2 (all-gather): my reduced chunk -> every peer; theirs arrive in place
    COMM_NCCL_FATAL(c, false, ncclGroupStart(), "ncclGroupStart");
    for (int p = 0; p < c->world; ++p) {
        if (p == me) continue;
        if (len[me] > 0) COMM_NCCL_FATAL(c, true, ncclSend(b + off[me] * es, (size_t)len[me], dt, p, c->nccl, s), "ncclSend(all-gather)");
        if (len[p] > 0) COMM_NCCL_FATAL(c, true, ncclRecv(b + off[p] * es, (size_t)len[p], dt, p, c->nccl, s), "ncclRecv(all-gather)");
    }
    COMM_NCCL_FATAL(c, false, ncclGroupEnd(), "ncclGroupEnd(all-gather)");
    return 0;
}

// test aid: the fixed-order sum kernel of ALGO_DIRECT on its own (one GPU, no communicator): own[count] (+)= the world - 1 peer copies in scratch
extern "C" int pseld_comm_sum_in_rank_order(void* own, const void* scratch, long count, long stride, int me, int world, int dtype, void* stream) {
    COMM_CHECK_ARG(own && (scratch || world == 1) && count >= 0 && count % 8 == 0 && stride % 8 == 0 && me >= 0 && me < world && (dtype == 0 || dtype == 1),
                   "comm_sum_in_rank_order: bad arguments");
    const int es = dtype == 0 ? 4 : 2;
    const long n16 = count * es / 16;
    if (n16 == 0) return 0;
    const int blocks = (int)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL(sum_in_rank_order<f4>, dim3(blocks), dim3(256), 0, s, (f4*)own, (const f4*)scratch, n16, stride * es / 16, me, world);
    else hipLaunchKernelGGL(sum_in_rank_order<bf16x8v>, dim3(blocks), dim3(256), 0, s, (bf16x8v*)own, (const bf16x8v*)scratch, n16, stride * es / 16, me, world);
    if (hipGetLastError() != hipSuccess) { set_err("comm_sum_in_rank_order: launch failed"); return -3; }
    return 0;
}

extern "C" int pseld_comm_finalize(void* comm) {
    if (!comm) return 0;
    Comm* c = (Comm*)comm;
    ncclResult_t r = c->aborted ? ncclSuccess : ncclCommDestroy(c->nccl);      // (ncclCommAbort has already released an aborted one)
    delete c;
    if (r != ncclSuccess) { set_err("ncclCommDestroy: %s", ncclGetErrorString(r)); return -3; }
    return 0;
}
