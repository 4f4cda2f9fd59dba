"""GPU test of the multi-GPU plumbing bench.py uses at --gpus N > 1, on ONE GPU: pipelined optimize (the last
fill_site_tensors stays in flight), asynchronous core export ordered after that fill, and an RCCL all-gather
(world_size 1 process group) waiting for the export on the device."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_pipelined_export_and_rccl_all_gather():
    import torch
    import torch.distributed as dist
    import t4a_amd
    if t4a_amd.device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, chi = 14, 16
        spec = t4a_amd.quantics_trig_exp(n)  # cos(10x) exp(-x): low rank, so the capped interpolant is accurate
        tci = t4a_amd.TensorCI2([2] * n)
        tci.set_function(spec)
        tci.add_global_pivots([[0] * n])
        tci.set_keep_site_tensors(True)
        opt = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=2, ncheck_history=10 ** 6, nsearch=0,
                                  max_nglobal_pivot=0)
        cap = chi * 2 * chi
        send = [torch.zeros(n * cap, dtype=torch.float64, device="cuda") for _ in range(2)]
        recv = [torch.zeros(n * cap, dtype=torch.float64, device="cuda") for _ in range(2)]
        stream = torch.cuda.current_stream()
        for sweep in range(4):
            k = sweep % 2
            tci.optimize(opt, final_sweep1site=False)          # returns with the last fill still in flight
            tci.export_site_tensors_async(send[k].data_ptr(), cap, stream.cuda_stream)
            work = dist.all_gather_into_tensor(recv[k], send[k], async_op=True)
            work.wait()
        torch.cuda.synchronize()
        got = recv[1].cpu().numpy()                             # sweep 3 used buffer pair 1
        for s in range(n):
            core = tci.site_tensor(s)
            assert core.size > 0
            flat = core.reshape(-1, order="F")
            assert np.array_equal(got[s * cap:s * cap + flat.size], flat), f"core {s} differs after the gather"
        # the gathered cores form the interpolant of the last sweep
        pts = np.random.default_rng(0).integers(0, 2, size=(50, n))
        from oracle_binding import fn_eval
        assert np.abs(tci.evaluate(pts) - fn_eval(spec, pts)).max() < 1e-6
    finally:
        dist.destroy_process_group()
