"""Worker of tests/test_gpu_pishard.py: one rank of a `world`-process gloo group, all on GPU 0 (fresh process, started before anything
touches the GPU).  argv: rank world port out_dir"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))


def user_fn(idx):
    # a genuinely "user" function (no built-in functor): rank ~6 on 8 sites of dimension 3
    x = sum(int(v) * 3.0 ** (-k - 1) for k, v in enumerate(idx))
    return float(np.cos(9.0 * x) * np.exp(-x) + 0.25 * np.sin(23.0 * x) / (1.0 + idx[0] + idx[7]))


def batched(idx):
    idx = np.asarray(idx, dtype=np.float64)
    w = 3.0 ** (-np.arange(1, idx.shape[1] + 1))
    x = idx @ w
    return np.cos(9.0 * x) * np.exp(-x) + 0.25 * np.sin(23.0 * x) / (1.0 + idx[:, 0] + idx[:, 7])


user_fn.batched = batched


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    import t4a_amd
    from t4a_amd import parallel
    gather = None
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        gather = parallel.PiShardGather(dist, torch)
    t4a_amd.set_device(0)
    n = 8
    tci = t4a_amd.TensorCI2([3] * n)
    tci.set_function(user_fn)
    if world > 1:
        tci.set_pi_shard(rank, world, gather)
    o = t4a_amd.TCI2Options(tolerance=1e-9, max_bond_dim=12, max_iter=6, nsearch=0, max_nglobal_pivot=0, seed=1)
    tci.crossinterpolate2([[0] * n], o)
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 3, size=(200, n))
    res = {"link_dims": np.asarray(tci.link_dims()), "values": tci.evaluate(pts), "sum": np.asarray([tci.sum()]),
           "bond_errors": np.asarray(tci.bond_errors()), "calls": np.asarray([tci.n_callback_calls]),
           "gathers": np.asarray([tci.pi_shard_stats()["gathers"], tci.pi_shard_stats()["bytes_sent"]])}
    for s in range(n):
        res[f"i{s}"] = np.asarray(tci.i_set(s)).reshape(-1)
        res[f"j{s}"] = np.asarray(tci.j_set(s)).reshape(-1)
    np.savez(os.path.join(out_dir, f"pishard_{world}_{rank}.npz"), **res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
