"""SURVEY.md section 8(e) row 2 on the device: the candidate matrix of a HOST-CALLBACK function sharded by column blocks over the ranks
of a process group (t4a_gpu_tci2_set_pi_shard, csrc/pishard.hpp; tensorci2.rs:1859-1893), rank-revealing LU replicated.  Two processes
share GPU 0 and exchange their blocks over gloo (the 8-GPU node runs the same code with one process per GPU): index sets, bond errors and
tensor-train values must be BITWISE those of the unsharded run, on every rank, and each rank's callback must have seen about half of the
points."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, tmp):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "pishard_worker.py"), str(r), str(world), str(port), str(tmp)])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(os.path.join(str(tmp), f"pishard_{world}_{r}.npz")) for r in range(world)]


def test_column_block_shard_selects_the_pivots_of_the_unsharded_run(tmp_path):
    (ref,) = _run(1, tmp_path)
    assert int(ref["link_dims"].max()) >= 4 and int(ref["gathers"][0]) == 0
    ranks = _run(2, tmp_path)
    for r, got in enumerate(ranks):
        for k in ref.files:
            if k in ("calls", "gathers"):
                continue
            assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), (r, k)   # bitwise: same matrices, same LU
        assert int(got["gathers"][0]) > 0
    # the work really was split: every callback-evaluated matrix cost one gather, and the two ranks sent the same number of bytes
    assert int(ranks[0]["gathers"][0]) == int(ranks[1]["gathers"][0])
    assert int(ranks[0]["gathers"][1]) == int(ranks[1]["gathers"][1])
