"""N > 1 on CPU: two processes, gloo backend, the product's data-parallel host path
(shard like np.array_split -> packed accumulator -> all-reduce -> redundant Theta update)."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("fixture", ["ebsc_mid", "es3c_mid"])
def test_two_rank_mstep(fixture):
    port = str(free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(r), "2", port, fixture],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out)
        assert "rank %d ok" % r in out
