"""Multi-GPU orchestration of the parts of the TCI2 path that shard: one process per GPU,
torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).

 * patch farm (BASELINE.json configs[4]; reference: partitionedtt::adaptiveinterpolate runs one independent
   crossinterpolate2 per patch, crates/tensor4all-partitionedtt/src/adaptive_interpolation.rs:151-330):
   patches are dealt round-robin to ranks, every rank interpolates its patches, then the patch cores are
   all-gathered (the only exchange step).  FIFO patch order is preserved when re-assembling.
 * site-sharded fill_site_tensors (configs[3]; tensorci2.rs:1065-1186: sites are independent given the final
   I/J sets): rank r fills the sites s with s % world == r, then cores are all-gathered.
 * column-block shard of the candidate matrix for expensive (host-callback) functions (tensorci2.rs:1859-1893: entries of one
   candidate matrix are independent): rank r evaluates ceil(N / world) columns through its callback, one all-gather per matrix,
   the rank-revealing LU replicated (PiShardGather + TensorCI2.set_pi_shard; csrc/pishard.hpp).

The local compute is injected (`run_patch`, `fill_sites`) so that the same orchestration runs with the device
handle in production and with the CPU oracle in tests/test_cpu_parallel.py.
"""
import numpy as np


def leave_legacy_stream(torch):
    """Every device-side exchange of this module hands torch's CURRENT stream to the library as the consumer / producer of its
    events.  In a process that never selected a stream that is stream 0, the legacy default stream, whose implicit ordering against
    the streams a HIP graph launch runs on broke the captured fill_site_tensors graph (GPU memory faults, bisected in round 5:
    profiles/r05_fill_graph_fault_bisect.txt).  Make a non-blocking side stream current instead — once per thread; a caller that has
    selected its own stream keeps it.  (The library itself falls back to direct issue when it is handed stream 0.)"""
    import os
    if os.environ.get("T4A_SS_LEGACY_STREAM"):  # (diagnosis: stay on stream 0; the library then issues its fills directly)
        return
    if torch.cuda.is_available() and torch.cuda.current_stream().cuda_stream == 0:
        #  the side stream is non-blocking: it is ordered behind everything the caller has enqueued on stream 0 so far
        # (tensor initialisation, the caller's inputs).  Side effect, on purpose and permanent: the thread's current stream changes;
        # tensors the caller allocated on stream 0 and hands in later should be passed with record_stream() or after a synchronize.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(side)


def patches_of_rank(n_patches, rank, world):
    """Static round-robin farm: patch p runs on rank p % world (FIFO order inside a rank)."""
    return [p for p in range(n_patches) if p % world == rank]


def pack_cores(cores):
    """cores: list of (l, s, r) float64 arrays (Fortran order = simplett Tensor3 layout) -> (header, flat)."""
    header = np.array([d for c in cores for d in c.shape], dtype=np.int64)
    flat = np.concatenate([np.asarray(c, dtype=np.float64).ravel(order="F") for c in cores]) if cores else np.zeros(0)
    return header, flat


def unpack_cores(header, flat):
    cores, off = [], 0
    for k in range(0, len(header), 3):
        l, s, r = (int(x) for x in header[k:k + 3])
        n = l * s * r
        cores.append(np.array(flat[off:off + n]).reshape((l, s, r), order="F"))
        off += n
    return cores


def all_gather_variable(dist, torch, arr, device, dtype):
    """all-gather of per-rank 1-D arrays of different lengths (sizes first, then padded payload)."""
    world = dist.get_world_size()
    n = torch.tensor([len(arr)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=dtype, device=device)
    if len(arr):
        send[:len(arr)] = torch.as_tensor(arr, dtype=dtype, device=device)
    recv = [torch.zeros(cap, dtype=dtype, device=device) for _ in range(world)]
    dist.all_gather(recv, send)
    return [recv[r][:sizes[r]].cpu().numpy() for r in range(world)]


def run_patch_farm(dist, torch, n_patches, run_patch, device="cpu"):
    """run_patch(p) -> list of core arrays of patch p.  Returns the cores of ALL patches, in patch order, on
    every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = patches_of_rank(n_patches, rank, world)
    headers, flats, counts = [], [], []
    for p in mine:
        h, f = pack_cores(run_patch(p))
        headers.append(h)
        flats.append(f)
        counts.append(len(h) // 3)
    hdr = np.concatenate([np.array([len(mine)], dtype=np.int64), np.array(counts, dtype=np.int64)] + headers) \
        if mine else np.array([0], dtype=np.int64)
    payload = np.concatenate(flats) if flats else np.zeros(0)
    all_hdr = all_gather_variable(dist, torch, hdr, device, torch.int64)
    all_pay = all_gather_variable(dist, torch, payload, device, torch.float64)
    result = [None] * n_patches
    for r in range(world):
        h, f = all_hdr[r], all_pay[r]
        npat = int(h[0])
        cnts = [int(x) for x in h[1:1 + npat]]
        hoff, foff = 1 + npat, 0
        for k, p in enumerate(patches_of_rank(n_patches, r, world)):
            hh = h[hoff:hoff + 3 * cnts[k]]
            size = int(sum(int(hh[i]) * int(hh[i + 1]) * int(hh[i + 2]) for i in range(0, len(hh), 3)))
            result[p] = unpack_cores(hh, f[foff:foff + size])
            hoff += 3 * cnts[k]
            foff += size
    return result


class PaddedPatchFarm:
    """Device-resident patch farm (BASELINE.json configs[4]; reference: adaptive_interpolation.rs:171-330 runs one independent
    crossinterpolate2 per patch and keeps the FIFO patch order when it re-assembles, :303-330).

    Patches are dealt round-robin to the ranks (patch p on rank p % world, FIFO inside a rank).  Every rank writes the cores of
    its patches into ONE padded device tensor [per_rank, n_sites, cap] (cap = the largest core, e.g. chi * d * chi) plus a small
    int64 tensor with the (l, s, r) of every core; ONE all_gather_into_tensor moves the payload and one the shapes — nothing is
    staged through host memory (the host-packed run_patch_farm above moves ~0.5 GB through PCIe for 64 patches x chi 128).
    The gathered payload stays on the device; `core(p, s)` / `cores(p)` copy single cores out for whoever wants them on the host.

    `export_patches(patches, send)`: the local compute, injected.  `patches` = this rank's patch numbers in FIFO order, `send` =
    a float64 tensor view [len(patches), n_sites, cap] on `device`; it interpolates the patches, writes their cores (column-major
    (l, s, r), zero padded) and returns the shapes as a list (per patch) of lists (per site) of (l, s, r).
    DevicePatchExporter (below) does that with t4a_gpu_tci2_optimize_group — up to eight patches at a time, one XCD each — and
    device-to-device exports; tests/test_cpu_parallel.py runs the same orchestration on gloo with the CPU oracle."""

    def __init__(self, dist, torch, n_patches, n_sites, cap, device):
        self.dist, self.torch = dist, torch
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.n_patches, self.n_sites, self.cap = n_patches, n_sites, cap
        self.per_rank = (n_patches + self.world - 1) // self.world
        self.mine = patches_of_rank(n_patches, self.rank, self.world)
        if str(device).startswith("cuda"):
            leave_legacy_stream(torch)
        self.send = torch.zeros(self.per_rank * n_sites * cap, dtype=torch.float64, device=device)
        self.recv = torch.zeros(self.world * self.per_rank * n_sites * cap, dtype=torch.float64, device=device)
        self.send_dims = torch.zeros(self.per_rank * n_sites * 3, dtype=torch.int64, device=device)
        self.recv_dims = torch.zeros(self.world * self.per_rank * n_sites * 3, dtype=torch.int64, device=device)
        self.dims = None

    def run(self, export_patches, timed=False):
        """timed=True (bench.py --mode patch-farm): the device is synchronised between the local compute and the gather, and
        last_export_s / last_gather_s hold the two wall times (the gather is then fully exposed: nothing overlaps it)."""
        import time
        torch = self.torch
        view = self.send.view(self.per_rank, self.n_sites, self.cap)[:len(self.mine)]
        t0 = time.perf_counter()
        shapes = export_patches(self.mine, view) if self.mine else []
        if timed and str(self.send.device).startswith("cuda"):
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        flat = [int(v) for patch in shapes for d in patch for v in d]
        flat += [0] * (self.per_rank * self.n_sites * 3 - len(flat))
        self.send_dims.copy_(torch.tensor(flat, dtype=torch.int64), non_blocking=True)
        if self.world > 1:
            works = [self.dist.all_gather_into_tensor(self.recv, self.send, async_op=True),
                     self.dist.all_gather_into_tensor(self.recv_dims, self.send_dims, async_op=True)]
            for w in works:
                w.wait()
        else:
            self.recv.copy_(self.send)
            self.recv_dims.copy_(self.send_dims)
        self.dims = self.recv_dims.cpu().numpy().reshape(self.world, self.per_rank, self.n_sites, 3)
        if timed and str(self.send.device).startswith("cuda"):
            torch.cuda.synchronize()
        self.last_export_s, self.last_gather_s = t1 - t0, time.perf_counter() - t1
        self.gather_bytes = int(self.recv.numel() * 8 + self.recv_dims.numel() * 8)
        return self

    def _slot(self, p):
        return p % self.world, p // self.world

    def core_dims(self, p, s):
        r, k = self._slot(p)
        return tuple(int(v) for v in self.dims[r, k, s])

    def core_device(self, p, s):
        """Core s of patch p as a device tensor view (l * d * r doubles, column-major (l, s, r))."""
        r, k = self._slot(p)
        l, d, rr = self.core_dims(p, s)
        return self.recv.view(self.world, self.per_rank, self.n_sites, self.cap)[r, k, s, :l * d * rr]

    def core(self, p, s):
        l, d, rr = self.core_dims(p, s)
        return self.core_device(p, s).cpu().numpy().reshape((l, d, rr), order="F")

    def cores(self, p):
        return [self.core(p, s) for s in range(self.n_sites)]


class DevicePatchExporter:
    """export_patches callback of PaddedPatchFarm for device handles: `make_patch(p)` returns a fresh TensorCI2 handle with its
    function and initial pivots set; the patches of this rank are optimised `group` at a time (<= 8: one XCD each,
    t4a_gpu_tci2_optimize_group), their site tensors filled and exported device-to-device.  `keep(p, tci)` is called with every
    finished handle (tests keep a sample for the comparison with the oracle)."""

    def __init__(self, t4a_amd, torch, make_patch, options, group=8, keep=None):
        self.t4a, self.torch, self.make_patch, self.options, self.group, self.keep = t4a_amd, torch, make_patch, options, max(1, min(group, 8)), keep

    def __call__(self, patches, send):
        shapes = []
        leave_legacy_stream(self.torch)
        stream = self.torch.cuda.current_stream().cuda_stream
        cap = send.shape[2]
        for k0 in range(0, len(patches), self.group):
            chunk = patches[k0:k0 + self.group]
            tcis = [self.make_patch(p) for p in chunk]
            for t in tcis:
                # pipelined mode: the fill_site_tensors of the LAST iteration stays valid (and may still be in flight) when
                # optimize returns — the export below is ordered behind it on the device.  Without it every patch paid a
                # second, synchronous fill of the same index sets here (eight in a row: a fifth of a group's time).
                t.set_keep_site_tensors(True)
            self.t4a.optimize_group(tcis, self.options, final_sweep1site=False)
            for j, (p, t) in enumerate(zip(chunk, tcis)):
                t.export_site_tensors_async(send[k0 + j].data_ptr(), cap, stream)
                ld, loc = t.link_dims(), t.local_dims
                shapes.append([_site_dims(ld, loc, s) for s in range(len(loc))])
                if self.keep is not None:
                    self.keep(p, t)
            self.torch.cuda.current_stream().synchronize()  # (the handles of this chunk are released next: their cores must have left)
        return shapes


def sharded_fill(dist, torch, n_sites, fill_my_sites, get_core, set_core, device="cpu"):
    """Site-sharded fill_site_tensors + core all-gather.  fill_my_sites(rank, world) fills the local sites;
    get_core(s) -> array, set_core(s, array) installs a core received from another rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    fill_my_sites(rank, world)
    mine = [s for s in range(n_sites) if s % world == rank]
    h, f = pack_cores([get_core(s) for s in mine])
    all_hdr = all_gather_variable(dist, torch, h, device, torch.int64)
    all_pay = all_gather_variable(dist, torch, f, device, torch.float64)
    for r in range(world):
        if r == rank:
            continue
        cores = unpack_cores(all_hdr[r], all_pay[r])
        for s, c in zip([s for s in range(n_sites) if s % world == r], cores):
            set_core(s, c)


class ShardedCoreExchange:
    """Device-resident core exchange of the site-sharded fill (BASELINE.json configs[3]): every rank fills the sites
    s % world == rank, packs them into a padded buffer [per_rank][cap] (cap = largest core), ONE all_gather_into_tensor per
    half-sweep moves them, every rank unpacks the other ranks' sites.

    Ordering.  `exchange()` is called right after the local fill of a half-sweep has been issued and does everything for THAT
    half-sweep without blocking the host: export (ordered after the fill on the device), collective (asynchronous), import
    (ordered after the collective on the device, on the handle's import stream).  Export and import therefore see the same
    (replicated) index sets — the importer sizes the remote cores from its own bond dimensions, which is only right while no
    bond update has run in between.  The data itself may still be travelling while the next half-sweep's bond updates run
    (they only touch the index sets); the first reader of a core waits for it.  Two buffer pairs alternate, so a buffer is
    reused two half-sweeps after its collective was issued.

    Shape check.  Every rank also gathers the (l, s, r) of the cores it exported; when a buffer pair comes up for reuse (or at
    `finish()`), the gathered shapes are compared with the ones the importer assumed and a mismatch raises — the ranks were
    not replicas of each other (different options, a missed half-sweep), and the imported cores would have been
    reinterpreted silently.

    `adapter` hides where the cores live:
      export_shard(send_tensor) -> [(l, s, r)] of the local sites   local cores -> send_tensor viewed as [per_rank, cap]
      import_shard(recv_tensor, per_rank) -> {site: (l, s, r)}      remote cores <- recv_tensor viewed as [world, per_rank, cap]
    DeviceShardAdapter (below) does both with device-to-device copies through the C ABI; tests/test_cpu_parallel.py runs the
    same orchestration on gloo with a numpy adapter around the CPU oracle."""

    def __init__(self, dist, torch, n_sites, cap, adapter, device):
        self.dist, self.torch, self.adapter = dist, torch, adapter
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.n_sites = n_sites
        self.per_rank = (n_sites + self.world - 1) // self.world
        self.cap = cap
        self.is_cuda = str(device).startswith("cuda")
        if self.is_cuda:
            leave_legacy_stream(torch)
        self.send = [torch.zeros(self.per_rank * cap, dtype=torch.float64, device=device) for _ in range(2)]
        self.recv = [torch.zeros(self.world * self.per_rank * cap, dtype=torch.float64, device=device) for _ in range(2)]
        self.send_dims = [torch.zeros(self.per_rank * 3, dtype=torch.int64, device=device) for _ in range(2)]
        self.recv_dims = [torch.zeros(self.world * self.per_rank * 3, dtype=torch.int64, device=device) for _ in range(2)]
        self.assumed = [None, None]         # {site: (l, s, r)} the importer used for exchange k
        self.done = [None, None]            # device event / work handles: everything of exchange k has been enqueued / finished
        self.outstanding = [False, False]
        self.count = 0

    def exchange(self):
        """Called after the local fill of a half-sweep: export, all-gather and import of THIS half-sweep, nothing waited for
        on the host except the (long finished) exchange that used the same buffer pair two half-sweeps ago."""
        torch = self.torch
        k = self.count % 2
        self.count += 1
        self.finish(k)
        dims = self.adapter.export_shard(self.send[k])
        flat = [int(v) for d in dims for v in d] + [0] * (3 * self.per_rank - 3 * len(dims))
        self.send_dims[k].copy_(torch.tensor(flat, dtype=torch.int64), non_blocking=True)
        works = []
        if self.world > 1:
            works.append(self.dist.all_gather_into_tensor(self.recv[k], self.send[k], async_op=True))
            works.append(self.dist.all_gather_into_tensor(self.recv_dims[k], self.send_dims[k], async_op=True))
            for w in works:
                w.wait()  # RCCL: stream-level dependency only; gloo: completion
        else:
            self.recv[k].copy_(self.send[k])
            self.recv_dims[k].copy_(self.send_dims[k])
        self.assumed[k] = self.adapter.import_shard(self.recv[k], self.per_rank)
        if self.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.done[k] = ev
        self.outstanding[k] = True
        return k

    def finish(self, k=None):
        """Completes exchange k (default: every outstanding one, oldest first) on the host and checks the gathered shapes."""
        for kk in ([k] if k is not None else [self.count % 2, (self.count + 1) % 2]):
            if not self.outstanding[kk]:
                continue
            if self.done[kk] is not None:
                self.done[kk].synchronize()
                self.done[kk] = None
            got = self.recv_dims[kk].cpu().numpy().reshape(self.world, self.per_rank, 3)
            for s, want in (self.assumed[kk] or {}).items():
                have = tuple(int(v) for v in got[s % self.world, s // self.world])
                if have != tuple(int(v) for v in want):
                    raise RuntimeError(f"site-sharded core exchange: rank {s % self.world} exported site {s} as {have}, rank "
                                       f"{self.rank} imported it as {tuple(want)} — the ranks do not hold the same index sets")
            self.assumed[kk] = None
            self.outstanding[kk] = False


def _site_dims(link_dims, local_dims, s):
    n = len(local_dims)
    l = 1 if s == 0 else max(int(link_dims[s - 1]), 1)
    r = 1 if s == n - 1 else max(int(link_dims[s]), 1)
    return (l, int(local_dims[s]), r)


class DeviceShardAdapter:
    """Cores stay in HBM: export / import are device-to-device copies ordered by events (t4a_gpu_tci2_export_site_shard_async,
    t4a_gpu_tci2_import_site_shard_async); the all-gather runs on torch's current stream.  Shapes come from the handle's
    (replicated) bond dimensions at the time of the call."""

    def __init__(self, tci, torch, cap):
        self.tci, self.torch, self.cap = tci, torch, cap

    def _dims(self):
        ld, loc = self.tci.link_dims(), self.tci.local_dims
        return [_site_dims(ld, loc, s) for s in range(len(loc))]

    def export_shard(self, send):
        rank, world = self.tci.site_shard()
        self.tci.export_site_shard_async(send.data_ptr(), self.cap, self.torch.cuda.current_stream().cuda_stream)
        dims = self._dims()
        return [dims[s] for s in range(rank, len(dims), world)]

    def import_shard(self, recv, per_rank):
        rank, world = self.tci.site_shard()
        self.tci.import_site_shard_async(recv.data_ptr(), self.cap, per_rank, self.torch.cuda.current_stream().cuda_stream)
        dims = self._dims()
        return {s: dims[s] for s in range(len(dims)) if s % world != rank}


class NumpyShardAdapter:
    """Cores as numpy arrays in a dict {site: array(l, s, r)} (CPU tests over gloo).  dims_of(site) -> (l, s, r) of a remote
    core, known on every rank from the replicated index sets."""

    def __init__(self, torch, store, n_sites, rank, world, cap, dims_of):
        self.torch, self.store, self.n_sites, self.rank, self.world, self.cap, self.dims_of = torch, store, n_sites, rank, world, cap, dims_of

    def export_shard(self, send):
        buf = send.view(-1, self.cap)
        dims = []
        for k, s in enumerate(range(self.rank, self.n_sites, self.world)):
            flat = np.asarray(self.store[s], dtype=np.float64).ravel(order="F")
            buf[k, :flat.size] = self.torch.from_numpy(flat.copy())
            dims.append(tuple(int(v) for v in np.asarray(self.store[s]).shape))
        return dims

    def import_shard(self, recv, per_rank):
        buf = recv.view(self.world, per_rank, self.cap).numpy()
        used = {}
        for s in range(self.n_sites):
            r = s % self.world
            if r == self.rank:
                continue
            l, d, rr = self.dims_of(s)
            self.store[s] = np.array(buf[r, s // self.world, :l * d * rr]).reshape((l, d, rr), order="F")
            used[s] = (int(l), int(d), int(rr))
        return used


class PiShardGather:
    """The all-gather of TensorCI2.set_pi_shard as a callable on host buffers: send (count doubles) -> world * count doubles,
    rank-major.  gloo: CPU tensors; nccl (= RCCL): staged through device tensors of the current device (the values of a host
    callback are host values; 8 M N / world bytes per rank and matrix).  Buffers are cached per size."""

    def __init__(self, dist, torch, device="cpu"):
        self.dist, self.torch, self.device = dist, torch, device
        self.world = dist.get_world_size()
        self._buf = {}
        self.calls = 0
        self.bytes = 0

    def __call__(self, send):
        torch = self.torch
        n = int(send.size)
        if n not in self._buf:
            self._buf[n] = (torch.zeros(n, dtype=torch.float64, device=self.device),
                            torch.zeros(self.world * n, dtype=torch.float64, device=self.device))
        import time
        t0 = time.perf_counter()
        s, r = self._buf[n]
        s.copy_(torch.from_numpy(np.ascontiguousarray(send)))
        self.dist.all_gather_into_tensor(r, s)
        self.calls += 1
        self.bytes += 8 * n
        out = r.cpu().numpy()  # (staged through host + device copies per matrix: fine for an expensive f)
        self.seconds = getattr(self, "seconds", 0.0) + (time.perf_counter() - t0)
        return out
