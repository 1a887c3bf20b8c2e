"""Communicators for the data-parallel EM step (reference: evo/utils/parallel.py + the inline
mpi4py calls of bsc.py:230-231,257,274 / sssc.py:671-691,763,773-780 / _models.py:540-547).

The reference shards the N datapoints over MPI ranks (``np.array_split`` order,
parallel.py:102) and sums the M-step accumulators with ``comm.Allreduce``.  Here one process
drives one GPU and the packed accumulator is summed with ONE in-place RCCL all-reduce over
xGMI, issued inside libevo_amd on the stream that produced it (``RcclComm``).  The model
classes only need the small duck-typed surface below, which mpi4py's ``MPI.COMM_WORLD`` also
satisfies, so a reference user can keep passing their communicator:

    comm.rank, comm.size
    comm.allreduce(python_scalar_or_ndarray) -> summed value      (pickle path in mpi4py)
    comm.bcast(obj, root=0)
    comm.Barrier()

``SerialComm``      one rank (default).
(A gloo transport with the same surface lives in tests/_torch_comm.py: the CPU multi-process tests use it;
nothing under evo_amd/ imports PyTorch.)
``RcclComm``        RCCL through libevo_amd: the device accumulator is reduced in place by
                    ``evoamd_stats``; host scalars go through a tiny device bounce buffer.
"""
import os
import sys
import time

import numpy as np


def pprint(obj="", comm=None, end="\n"):
    """Rank-0 print (parallel.py:23-42)."""
    if comm is not None and comm.rank != 0:
        return
    sys.stdout.write((obj if isinstance(obj, str) else repr(obj)) + end)
    sys.stdout.flush()


def shard_bounds(N, size):
    """Start offsets of np.array_split(range(N), size): the first N % size shards get one extra
    row (parallel.py:102-112)."""
    base, extra = divmod(int(N), int(size))
    counts = np.full(size, base, dtype=np.int64)
    counts[:extra] += 1
    starts = np.concatenate(([0], np.cumsum(counts)))
    return starts


def shard(array, rank, size):
    """This rank's contiguous block of ``array`` along axis 0 (what Scatterv delivers,
    parallel.py:117-151)."""
    b = shard_bounds(array.shape[0], size)
    return array[b[rank]:b[rank + 1]]


_DTYPES = (np.float64, np.float32, np.int64, np.int32, np.int16, np.uint8, np.bool_, np.uint16, np.uint32, np.uint64)


def _sum_array(comm, a):
    """Element-wise sum over ranks of a float64 array through whatever the communicator offers."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if hasattr(comm, "allreduce_array"):
        return np.asarray(comm.allreduce_array(a)).reshape(a.shape)
    return np.asarray(comm.allreduce(a)).reshape(a.shape)  # mpi4py: pickle path, op = SUM


def scatter_to_processes(to_scatter, comm=None):
    """Split an array that lives on rank 0 into np.array_split blocks along axis 0 and hand rank r
    block r (parallel.py:117-151: get_chunk_dimensions + Scatterv).  Non-root ranks may pass None.
    The communicators here expose sums only, so shape, dtype and payload travel as zero-padded
    contributions of rank 0 (one-off data ingest, not a hot path).  Values must be exactly
    representable in float64 (every dtype the reference's typemap lists except 64-bit integers
    beyond 2^53)."""
    comm = comm or SerialComm()
    if comm.size == 1:
        return np.array(to_scatter)
    root = comm.rank == 0
    meta = np.zeros(10)
    if root:
        a = np.asarray(to_scatter)
        assert a.ndim <= 8
        meta[0] = [np.dtype(d) for d in _DTYPES].index(a.dtype)
        meta[1] = a.ndim
        meta[2:2 + a.ndim] = a.shape
    meta = _sum_array(comm, meta)
    dtype = np.dtype(_DTYPES[int(meta[0])])
    shape = tuple(int(v) for v in meta[2:2 + int(meta[1])])
    payload = np.asarray(to_scatter, dtype=np.float64) if root else np.zeros(shape)
    full = _sum_array(comm, payload)
    return np.ascontiguousarray(shard(full, comm.rank, comm.size)).astype(dtype)


def gather_from_processes(chunk, comm=None):
    """Concatenate the per-rank blocks along axis 0 in rank order (parallel.py:154-173: Gatherv).
    Every rank receives the gathered array (the reference fills it on rank 0 only)."""
    comm = comm or SerialComm()
    chunk = np.asarray(chunk)
    if comm.size == 1:
        return chunk.copy()
    counts = np.zeros(comm.size)
    counts[comm.rank] = chunk.shape[0]
    counts = _sum_array(comm, counts).astype(np.int64)
    starts = np.concatenate(([0], np.cumsum(counts)))
    full = np.zeros((int(starts[-1]),) + chunk.shape[1:])
    full[starts[comm.rank]:starts[comm.rank + 1]] = chunk
    return _sum_array(comm, full).astype(chunk.dtype)


class SerialComm:
    rank = 0
    size = 1
    device_reduces = False  # the packed accumulator needs no reduction

    def allreduce(self, value, op=None):
        return value

    def bcast(self, value, root=0):
        return value

    def Barrier(self):
        return None

    def allreduce_array(self, a):
        return a


class RcclComm:
    """RCCL communicator attached to an Engine's context (one process per GPU).

    ``device_reduces`` tells the models that ``Engine.stats()`` already returns globally summed
    accumulators.  Scalars and small host arrays use ``evoamd_comm_allreduce_host``."""
    device_reduces = True

    def __init__(self, engine, rank, size, unique_id):
        self.engine = engine
        self.rank = int(rank)
        self.size = int(size)
        engine.comm_init(unique_id, rank, size)

    def allreduce_array(self, a, op="sum"):
        a = np.asarray(a, dtype=np.float64)
        return self.engine.comm_allreduce(a.ravel(), op).reshape(a.shape)

    def allreduce(self, value, op=None):
        if isinstance(value, np.ndarray):
            return self.allreduce_array(value)
        out = self.allreduce_array(np.array([value], dtype=np.float64))[0]
        return type(value)(out) if isinstance(value, (int, np.integer)) else float(out)

    def allreduce_max(self, value):
        return float(self.allreduce_array(np.array([value], dtype=np.float64), "max")[0])

    def bcast(self, value, root=0):
        """Numeric scalars / float arrays only (everything Theta holds): the root contributes the
        value, the others zeros, and the sum is the broadcast."""
        a = np.asarray(value, dtype=np.float64)
        send = a if self.rank == root else np.zeros_like(a)
        out = self.allreduce_array(send)
        return out if isinstance(value, np.ndarray) else type(value)(out)

    def Barrier(self):
        self.allreduce_array(np.zeros(1))

    def close(self):
        self.engine.comm_destroy()


def _parent_start_time():
    """Start time (clock ticks since boot) of the parent process: with the parent pid it names ONE launcher
    process, so a recycled (MASTER_PORT, ppid) pair of an earlier launch cannot match a stale file."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            return f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def rendezvous_path(tag=None):
    if tag is None:
        tag = "%s_%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.getppid(),
                               os.environ.get("EVO_AMD_LAUNCH_NONCE") or _parent_start_time(),
                               os.environ.get("TORCHELASTIC_RUN_ID", "none"))
    return os.path.join(os.environ.get("EVO_AMD_RDZV_DIR", "/tmp"), "evo_amd_rccl_%s.id" % tag)


def rendezvous_unique_id(rank, size, make_id, tag=None, timeout_s=300.0):
    """Share rank 0's RCCL unique id with the other local ranks through a file.

    One node, one process per GPU (the bench contract): the launcher (torchrun, or bench.py's own
    launch_ranks) gives every rank the same MASTER_PORT and the same parent process, which together name
    the rendezvous file under /tmp (plus a per-launch nonce / the parent's start time, so a file left by an
    earlier launch can never match).  Rank 0 writes the id atomically (temp file + rename); the others
    poll; init_rccl_from_env removes the file once every rank is in the communicator."""
    if size == 1:
        return make_id()
    path = rendezvous_path(tag)
    if rank == 0:
        uid = make_id()
        tmp = path + ".tmp.%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == 128:
                return uid
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError("no RCCL unique id at %s after %.0f s" % (path, timeout_s))
        time.sleep(0.05)


def init_rccl_from_env(engine):
    """RANK / WORLD_SIZE (torchrun) -> RcclComm on ``engine``; SerialComm when WORLD_SIZE is 1."""
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if size == 1:
        return SerialComm()
    uid = rendezvous_unique_id(rank, size, engine.comm_unique_id)
    comm = RcclComm(engine, rank, size, uid)
    comm.Barrier()  # every rank has read the id file by now
    if rank == 0:
        try:
            os.unlink(rendezvous_path())
        except OSError:
            pass
    return comm
