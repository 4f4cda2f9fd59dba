// pishard.hpp — SURVEY.md section 8(e) row 2: the candidate matrix of one bond sharded by COLUMN BLOCKS over the ranks of a process
// group, for functions that are expensive to evaluate (the host batch callback: every real Rust closure).  Entries of one candidate
// matrix are independent (tensor4all-tensorci/src/tensorci2.rs:1859-1893): rank r evaluates the columns [r cb, (r + 1) cb) with
// cb = ceil(N / world) through ITS callback — points in the reference's order restricted to the block, row index outer, column index
// inner (tensorci2.rs:1862-1869) —, one all-gather of the M x cb blocks (8 M cb bytes per rank) makes the whole matrix known to every
// rank, and the rank-revealing LU runs replicated: it is deterministic, so every rank selects the same pivots and no broadcast is needed.
// Host logic only (the values of a callback are host values): used by Tci2::eval_matrix and, through t4a_gpu_pi_shard_eval, by the
// CPU-only world-2 test (tests/test_cpu_parallel.py).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/t4a_gpu.h"
#include "common.hpp"

namespace t4a {

struct PiShard {
    size_t rank = 0, world = 1;
    t4a_gpu_allgather_fn gather = nullptr;
    void* gather_ctx = nullptr;
    size_t n_gathers = 0, bytes_sent = 0; // statistics
    bool active() const { return world > 1 && gather != nullptr; }
};

// rows: `na` multi-index halves of `wa` digits placed at site a0, columns: `nb` halves of `wb` digits at site b0 (wa + wb = n_sites).
// out: na x nb ROW-major (what the unsharded callback path produces before its transpose).
inline void pi_shard_evaluate(PiShard& ps, t4a_gpu_batch_eval_fn cb, void* cb_ctx, size_t n_sites, const uint32_t* a_digits, size_t wa, size_t a0,
                              size_t na, const uint32_t* b_digits, size_t wb, size_t b0, size_t nb, double* out)
{
    const size_t W = ps.world;
    const size_t cbk = (nb + W - 1) / W; // columns per rank (the last blocks may be short or empty)
    const size_t c0 = ps.rank * cbk < nb ? ps.rank * cbk : nb;
    const size_t c1 = c0 + cbk < nb ? c0 + cbk : nb;
    const size_t mine = c1 - c0;
    // One extra slot per rank carries a status: a rank whose callback failed STILL takes part in the all-gather (the others have
    // entered the collective already: staying away is a hang with gloo / RCCL, not an error), and every rank throws afterwards.
    const size_t blk = na * cbk + 1;
    std::vector<double> send(blk, 0.0), recv(W * blk);
    std::string my_error;
    if (mine > 0) {
        const size_t npts = na * mine;
        std::vector<uint32_t> idx(npts * n_sites);
        for (size_t ia = 0; ia < na; ++ia)
            for (size_t ib = c0; ib < c1; ++ib) {
                uint32_t* dst = idx.data() + (ia * mine + (ib - c0)) * n_sites;
                std::memcpy(dst + a0, a_digits + ia * wa, wa * sizeof(uint32_t));
                std::memcpy(dst + b0, b_digits + ib * wb, wb * sizeof(uint32_t));
            }
        std::vector<double> vals(npts);
        const int64_t got = cb(cb_ctx, idx.data(), n_sites, npts, vals.data());
        if (got < 0 || (size_t)got != npts) {
            my_error = "batch callback returned " + std::to_string(got) + " values for " + std::to_string(npts) + " requested entries (column block " +
                       std::to_string(ps.rank) + " of " + std::to_string(W) + ")";
            send[na * cbk] = 1.0;
        } else {
            for (size_t ia = 0; ia < na; ++ia) std::memcpy(send.data() + ia * cbk, vals.data() + ia * mine, mine * sizeof(double));
        }
    }
    const int32_t st = ps.gather(ps.gather_ctx, send.data(), send.size(), recv.data());
    if (st != 0) throw Error(T4A_GPU_CALLBACK_ERROR, "all-gather callback of the column-block shard failed with status " + std::to_string(st));
    ps.n_gathers += 1;
    ps.bytes_sent += send.size() * sizeof(double);
    if (!my_error.empty()) throw Error(T4A_GPU_CALLBACK_ERROR, my_error);
    for (size_t r = 0; r < W; ++r)
        if (recv[r * blk + na * cbk] != 0.0)
            throw Error(T4A_GPU_CALLBACK_ERROR, "batch callback failed on rank " + std::to_string(r) + " of the column-block shard (reported through the all-gather)");
    for (size_t r = 0; r < W; ++r) {
        const size_t r0 = r * cbk < nb ? r * cbk : nb;
        const size_t r1 = r0 + cbk < nb ? r0 + cbk : nb;
        for (size_t ia = 0; ia < na; ++ia)
            if (r1 > r0) std::memcpy(out + ia * nb + r0, recv.data() + r * blk + ia * cbk, (r1 - r0) * sizeof(double));
    }
}

} // namespace t4a
