"""BASELINE configs[4] on ONE GPU through the product path of the farm: parallel.PaddedPatchFarm + DevicePatchExporter (world size 1
over RCCL) — 64 patches, chi = 128, eight at a time through t4a_gpu_tci2_optimize_group, cores exported device-to-device and
gathered with one all_gather_into_tensor.  Prints the wall time per farm run and per patch."""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before the library: see INTEGRATION.md section 4)
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
import t4a_amd  # noqa: E402
from t4a_amd import parallel  # noqa: E402

n_patches = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
n, chi = bench.N_SITES, 128
opt = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)


def make_patch(p):
    t = t4a_amd.TensorCI2([2] * n)
    t.set_function(bench.patch_spec(p, 64))
    t.add_global_pivots([[0] * n])
    t.set_max_sample_value(1.0)
    return t


for rnd in range(rounds):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    farm = parallel.PaddedPatchFarm(dist, torch, n_patches, n, chi * 2 * chi, "cuda")
    farm.run(parallel.DevicePatchExporter(t4a_amd, torch, make_patch, opt, group=8))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"round {rnd}: {n_patches} patches in {dt * 1e3:.1f} ms = {dt / n_patches * 1e3:.2f} ms per patch "
          f"(link dims of patch 0: max {max(farm.core_dims(0, s)[2] for s in range(n - 1))})", flush=True)
dist.destroy_process_group()
