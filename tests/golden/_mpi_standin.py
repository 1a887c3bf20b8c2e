"""Single-rank stand-in for ``mpi4py`` used ONLY by make_golden.py in the build container.

The reference (/root/reference) imports ``mpi4py`` at module top level and the image has
no MPI.  This object provides the handful of communicator calls the reference's hot path
makes, for one rank, and records every reduction so that per-rank accumulators (which are
function locals in the reference) can be written into the golden fixtures.
It is our own code; nothing here is taken from the reference.
"""
import sys
import time
import types

import numpy as np


class RecordingComm:
    rank = 0
    size = 1

    def __init__(self):
        self.log = []  # (kind, payload copy)
        self.recording = False

    # -- helpers -----------------------------------------------------------------
    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def _rec(self, kind, payload):
        if self.recording:
            self.log.append((kind, np.array(payload, copy=True)))

    @staticmethod
    def _buf(spec):
        return spec[0] if isinstance(spec, (list, tuple)) else spec

    # -- pickle-path collectives ---------------------------------------------------
    def allreduce(self, value, op=None):
        self._rec("allreduce", value)
        return value

    def bcast(self, value, root=0):
        return value

    # -- buffer-path collectives ---------------------------------------------------
    def Allreduce(self, send, recv, op=None):
        src = self._buf(send)
        dst = self._buf(recv)
        self._rec("Allreduce", src)
        dst[...] = src

    def Bcast(self, buf, root=0):
        return None

    def Barrier(self):
        return None

    def Scatterv(self, send, recv, root=0):
        recv[...] = self._buf(send).reshape(recv.shape)

    def Gatherv(self, send, recv, root=0):
        self._buf(recv)[...] = send


def install():
    """Put a fake ``mpi4py`` package into sys.modules (idempotent). Returns the communicator."""
    if "mpi4py" in sys.modules and hasattr(sys.modules["mpi4py"], "_evo_amd_standin"):
        return sys.modules["mpi4py"].MPI.COMM_WORLD
    pkg = types.ModuleType("mpi4py")
    mpi = types.ModuleType("mpi4py.MPI")
    comm = RecordingComm()
    mpi.COMM_WORLD = comm
    for name in ("DOUBLE", "FLOAT", "BOOL", "SHORT", "INT", "LONG", "UNSIGNED_SHORT",
                 "UNSIGNED_INT", "UNSIGNED_LONG", "SUM"):
        setattr(mpi, name, name)
    mpi.Wtime = time.time
    mpi.Intracomm = RecordingComm
    pkg.MPI = mpi
    pkg._evo_amd_standin = True
    sys.modules["mpi4py"] = pkg
    sys.modules["mpi4py.MPI"] = mpi
    return comm
