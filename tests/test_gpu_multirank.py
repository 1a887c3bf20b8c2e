"""N > 1 on the GPU box: two processes share the one MI355X (own contexts), every rank runs the HIP
E-step and statistics on its np.array_split shard, the packed accumulators are summed over gloo and every
rank updates Theta; the result must be the reference's single-rank step (tests/golden).  (RCCL itself
needs one GPU per rank: its 1-rank path is covered in test_gpu_models.py, 2-8 GPUs are the driver's.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("fixture", ["ebsc_mid", "es3c_mid"])
def test_two_ranks_one_gpu(fixture):
    port = str(free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r), "2", port, fixture],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out)
        assert "rank %d ok" % r in out
