/* C ABI of libpseld_comm.so: the thin RCCL layer of the data-parallel training loop (SURVEY.md 8b "comm helpers around RCCL",
 * 8e "Collective 1"). Replaces, for the gradient all-reduce, what the reference gets from Lightning's DDP strategy
 * (/root/reference/configs/trainer/gpu.yaml:4-10 `strategy: ddp`; torch DDP's bucketed NCCL all-reduce under
 * src/models/components/model_module.py's LightningModule). One communicator per process (one process per GPU); every call
 * enqueues on the HIP stream it is given and returns - nothing here synchronises with the host.
 *
 * Conventions as include/pseld_hip.h: plain pointers and sizes, int status (0 = ok, < 0 = error, text in pseld_comm_last_error()).
 * The library links RCCL (librccl.so.1: inside a PyTorch process the copy PyTorch has already loaded is the one that resolves). */
#ifndef PSELD_COMM_H
#define PSELD_COMM_H
#ifdef __cplusplus
extern "C" {
#endif

#define PSELD_COMM_ID_BYTES 128
enum { PSELD_COMM_ALGO_RCCL = 0,        /* ncclAllReduce: whatever ring / tree RCCL picks */
       PSELD_COMM_ALGO_DIRECT = 1 };    /* reduce-scatter + all-gather as grouped point-to-point transfers to ALL peers at once */

const char* pseld_comm_last_error(void);
/* Rank 0 creates the rendezvous id (ncclGetUniqueId); the caller ships its PSELD_COMM_ID_BYTES bytes to the other ranks (the
 * Python side broadcasts them over the torch.distributed group that already exists for the rendezvous). */
int pseld_comm_unique_id(void* id_out);
/* ncclCommInitRank on the CURRENT HIP device. *comm_out is an opaque handle. world == 1 is allowed (every collective is then the identity). */
int pseld_comm_init(const void* id, int rank, int world, void** comm_out);
int pseld_comm_rank(void* comm);
int pseld_comm_world(void* comm);
/* Bytes of device scratch the DIRECT algorithm needs for a bucket of `count` elements of `elem_bytes` (the peers' pieces of this rank's
 * chunk land there before they are summed): (world - 1) * chunk. 0 for ALGO_RCCL. */
long pseld_comm_scratch_bytes(void* comm, long count, int elem_bytes, int algo);
/* In-place SUM all-reduce of buf[count] (dtype: 0 = f32, 1 = bf16) over the communicator, enqueued on `stream`.
 * ALGO_DIRECT (xGMI is point-to-point, 7 links per GPU: a ring moves 2 (W-1)/W S over ONE link per direction, this moves S / W per link per
 * phase - SURVEY 8e): the bucket is cut into W chunks (multiples of 4 elements); phase 1, one ncclGroup: chunk p goes to rank p, the
 * peers' copies of chunk `rank` arrive in scratch; a kernel sums them onto the own copy IN RANK ORDER 0..W-1 (every rank's chunk is summed in
 * the same order whatever the arrival order: the result is the same bits on every rank and run to run); phase 2, one ncclGroup: the
 * reduced chunk goes to every peer, theirs arrive in place. */
int pseld_comm_allreduce_bucket(void* comm, void* buf, long count, int dtype, int algo, void* scratch, long scratch_bytes, void* stream);
/* The transfer plan of ALGO_DIRECT as plain arithmetic (testable without RCCL): for a bucket of `count` elements on `world` ranks,
 * chunk p is elements [off[p], off[p] + len[p]); off / len have `world` entries. Returns the chunk stride (elements). */
long pseld_comm_direct_plan(long count, int world, long* off, long* len);
/* Test aid: ALGO_DIRECT's fixed-order sum kernel on its own (one GPU, no communicator): own[count] = sum over ranks r = 0..world-1, in that
 * order, of rank r's copy - rank `me`'s is own itself, rank r's sits at scratch + (r < me ? r : r - 1) * stride (elements). */
int pseld_comm_sum_in_rank_order(void* own, const void* scratch, long count, long stride, int me, int world, int dtype, void* stream);
int pseld_comm_finalize(void* comm);

#ifdef __cplusplus
}
#endif
#endif
