"""Thin RCCL layer of the data-parallel loop (include/pseld_comm.h, csrc/comm/comm.hip): what replaces Lightning's DDP strategy
(/root/reference/configs/trainer/gpu.yaml:4-10) for the gradient all-reduce when `FusedTrainer(comm='rccl' | 'rccl_direct')` asks
for it. One communicator per process, created from the torch.distributed group that already exists for the rendezvous (its only job
here: shipping rank 0's 128-byte RCCL id); buckets are reduced in place on a dedicated HIP stream with event hand-off to and from
the compute stream. 'rccl' = ncclAllReduce, 'rccl_direct' = grouped point-to-point reduce-scatter + all-gather to all peers at once
(xGMI is a full mesh of point-to-point links: S / W bytes per link per phase instead of a ring's 2 (W-1)/W S over one link)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpseld_comm.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "pseld_comm.h")
ALGO = {'rccl': 0, 'rccl_direct': 1}
ID_BYTES = 128
_lib = None


class CommError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CommError(f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        from ._lib import parse_header
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header(HEADER_PATH).items():
            fn = getattr(_lib, name, None)
            if fn is None:
                raise CommError(f"{LIB_PATH} lacks `{name}` declared in include/pseld_comm.h: rebuild it")
            fn.restype, fn.argtypes = restype, argtypes
    return _lib


def declared_symbols():
    from ._lib import parse_header
    return sorted(parse_header(HEADER_PATH))


def _check(rc, what):
    if rc != 0:
        raise CommError(f"{what} failed with status {rc}: {lib().pseld_comm_last_error().decode('utf-8', 'replace')}")


def direct_plan(count, world):
    """(stride, [(offset, length), ...]) of the DIRECT algorithm's chunks (pseld_comm_direct_plan: plain arithmetic, no RCCL call)."""
    off, ln = (ctypes.c_long * world)(), (ctypes.c_long * world)()
    stride = lib().pseld_comm_direct_plan(count, world, off, ln)
    return stride, list(zip(off, ln))


class _EventWork:
    """What the trainer waits on: the compute stream waits for the bucket's completion event (no host synchronisation)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class _NoWork:
    def wait(self):
        pass


class RcclComm:
    def __init__(self, group, device, algo='rccl_direct'):
        import torch.distributed as dist
        if algo not in ALGO:
            raise ValueError(f"comm algorithm {algo!r}: one of {sorted(ALGO)}")
        self.algo, self.device = ALGO[algo], torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        L = lib()
        idbuf = ctypes.create_string_buffer(ID_BYTES)
        if self.rank == 0:
            _check(L.pseld_comm_unique_id(idbuf), "pseld_comm_unique_id")
        box = [bytes(idbuf.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if hasattr(dist, 'get_global_rank') else 0, group=group)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(L.pseld_comm_init(box[0], self.rank, self.world, ctypes.byref(handle)), "pseld_comm_init")
        self.handle = handle
        # The communication stream exists only where there is something to communicate (world size 1 returns before any stream hand-off).
        # (Measured, round 6: whether this stream exists or not makes no difference to the one-rank step - 19.54 / 19.63 ms with it, 19.51 /
        # 19.63 without, profiles/r06_comm_world1.txt; what a second communicator costs there is looked for elsewhere)
        self.stream = torch.cuda.Stream(device=self.device) if (self.world > 1 or os.environ.get('PSELD_COMM_IDLE_STREAM') == '1') else None
        self._scratch = None

    def reserve(self, numel, elem_size):
        """Size the DIRECT algorithm's scratch once, for the largest bucket, before the first bucket is issued. The buffer is allocated
        WITH THE COMMUNICATION STREAM CURRENT, so the caching allocator ties it to that stream: when it is replaced or freed, its memory
        is handed out again only in that stream's order - never to compute-stream tensors while a receive or the rank-order sum may still
        be running on it."""
        if self.world == 1:
            return
        need = lib().pseld_comm_scratch_bytes(self.handle, int(numel), int(elem_size), self.algo)
        if need > 0 and (self._scratch is None or self._scratch.numel() < need):
            with torch.cuda.stream(self.stream):
                self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)

    def allreduce_(self, t):
        """In-place sum of the contiguous f32 / bf16 tensor t over the ranks: issued on the communication stream behind everything the
        compute stream has enqueued so far; returns an object whose wait() makes the compute stream wait for the result."""
        if not t.is_contiguous() or t.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("allreduce_: contiguous f32 / bf16 tensors only")
        if self.world == 1:
            return _NoWork()                                # the identity: no stream hand-off, no events
        L = lib()
        es = t.element_size()
        need = L.pseld_comm_scratch_bytes(self.handle, t.numel(), es, self.algo)
        self.reserve(t.numel(), es)                         # (a no-op once the largest bucket has been seen or reserved)
        ready = torch.cuda.Event()
        ready.record()                                      # compute stream: the bucket's gradients are final
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            _check(L.pseld_comm_allreduce_bucket(self.handle, t.data_ptr(), t.numel(), 0 if t.dtype == torch.float32 else 1, self.algo,
                                                 self._scratch.data_ptr() if need > 0 else None, self._scratch.numel() if need > 0 else 0,
                                                 self.stream.cuda_stream),
                   "pseld_comm_allreduce_bucket")
            done = torch.cuda.Event()
            done.record(self.stream)
        t.record_stream(self.stream)
        return _EventWork(done)

    def close(self):
        if self.handle is not None:
            torch.cuda.synchronize(self.device)
            _check(lib().pseld_comm_finalize(self.handle), "pseld_comm_finalize")
            self.handle = None
