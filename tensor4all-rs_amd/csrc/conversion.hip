// conversion.hip — TensorCI2::from_tensor_train on the device
// (crates/tensor4all-tensorci/src/conversion.rs:66-433): alternating one-site LUCI sweeps over a copy of the
// tensor train collect the nested I/J sets; the factor that is not kept is carried into the neighbour by a GEMM.
#include "tci2.hpp"

#include <algorithm>
#include <cmath>

namespace t4a {

namespace {

struct ConvState {
    std::vector<DevCore> cores; // working copy of the train (conversion.rs takes `tt` by value)
};

void move_core(DevCore& dst, DevCore&& src) { dst = std::move(src); }

IndexSet select(const IndexSet& set, const std::vector<int>& perm, int rank) // conversion.rs:401-415
{
    IndexSet out;
    out.width = set.width;
    for (int k = 0; k < rank; ++k) {
        if ((size_t)perm[k] >= set.count)
            throw Error(T4A_GPU_INTERNAL_ERROR, "conversion selected index " + std::to_string(perm[k]) +
                                                    " from set of length " + std::to_string(set.count));
        if (set.width == 0)
            ++out.count;
        else
            out.push(set.at((size_t)perm[k]));
    }
    return out;
}

bool same_sets(const std::vector<IndexSet>& a, const std::vector<IndexSet>& b)
{
    if (a.size() != b.size()) return false;
    for (size_t p = 0; p < a.size(); ++p)
        if (a[p].width != b[p].width || a[p].count != b[p].count || a[p].d != b[p].d) return false;
    return true;
}

// conversion.rs:123-205 (sweep1site_get_indices) with sweep_pair :207-271 inlined
void sweep1site_get_indices(Engine& eng, std::vector<DevCore>& tt, bool forward, std::vector<IndexSet>* spectators,
                            const FromTensorTrainOptions& opt, std::vector<IndexSet>& index_set,
                            std::vector<double>& pivot_errors, DevBuf<double>& d_m1, DevBuf<double>& d_m2)
{
    const size_t n = tt.size();
    hipStream_t st = eng.stream();
    index_set.clear();
    {
        IndexSet root;
        root.width = 0;
        root.count = 1; // vec![vec![]]
        index_set.push_back(root);
    }
    size_t rank = 1;
    for (size_t i = 1; i < n; ++i) rank = std::max(rank, tt[i].l);
    pivot_errors.assign(rank + 1, 0.0);
    for (size_t step = 0; step + 1 < n; ++step) {
        const size_t site = forward ? step : n - step - 1;
        const size_t next_site = forward ? site + 1 : site - 1;
        DevCore& cur = tt[site];
        DevCore& nxt = tt[next_site];
        const int cl = (int)cur.l, cs = (int)cur.s, cr = (int)cur.r;
        const int nl = (int)nxt.l, ns = (int)nxt.s, nr = (int)nxt.r;
        // group_indices(current, forward, false): forward -> left matrix, backward -> right matrix
        d_m1.reserve(std::max<size_t>(cur.size(), 1));
        core_reshape_launch(cur.buf.get(), cl, cs, cr, forward ? 0 : 2, d_m1.get(), st);
        RrLUOptions o;
        o.max_bond_dim = opt.max_bond_dim == 0 ? std::numeric_limits<size_t>::max() : opt.max_bond_dim;
        o.rel_tol = opt.tolerance;
        o.abs_tol = 0.0;
        o.left_orthogonal = forward;
        const int M = forward ? cl * cs : cl, N = forward ? cr : cs * cr;
        LuciResult f = eng.luci(d_m1.get(), M, N, o, true, false);
        const int r = f.rank;
        if (r == 0)
            throw Error(T4A_GPU_INVALID_ARGUMENT, "TensorCI2 conversion: a bond matrix of the tensor train is zero");
        const IndexSet& base = index_set.back();
        IndexSet cand;
        if (forward) { // kronecker_append :361-377
            cand.width = base.width + 1;
            std::vector<uint32_t> tmp(cand.width);
            for (size_t b = 0; b < base.count; ++b)
                for (int loc = 0; loc < cs; ++loc) {
                    if (base.width) std::copy_n(base.at(b), base.width, tmp.data());
                    tmp[base.width] = (uint32_t)loc;
                    cand.push(tmp.data());
                }
            index_set.push_back(select(cand, f.row_perm, r));
            if (spectators) (*spectators)[site] = select((*spectators)[site], f.col_perm, r);
            // next <- right factor (r x cr) * right matrix of next (nl x ns*nr), nl == cr
            d_m1.reserve(std::max<size_t>(nxt.size(), 1));
            core_reshape_launch(nxt.buf.get(), nl, ns, nr, 2, d_m1.get(), st);
            d_m2.reserve(std::max<size_t>((size_t)r * ns * nr, 1));
            GemmDesc g;
            g.m = r;
            g.n = ns * nr;
            g.k = cr;
            g.A = eng.right();
            g.lda = r;
            g.strideA = 0;
            g.transA = 0;
            g.B = d_m1.get();
            g.ldb = nl;
            g.strideB = 0;
            g.transB = 0;
            g.C = d_m2.get();
            g.ldc = r;
            g.strideC = 0;
            g.alpha = 1.0;
            g.beta = 0.0;
            g.batch = 1;
            gemm_launch(g, st);
            DevCore nc, nn;
            nc.l = cl;
            nc.s = cs;
            nc.r = r;
            nc.buf.reserve(std::max<size_t>(nc.size(), 1));
            core_reshape_launch(eng.left(), cl, cs, r, 1, nc.buf.get(), st);
            nn.l = r;
            nn.s = ns;
            nn.r = nr;
            nn.buf.reserve(std::max<size_t>(nn.size(), 1));
            core_reshape_launch(d_m2.get(), r, ns, nr, 3, nn.buf.get(), st);
            eng.sync();
            move_core(tt[site], std::move(nc));
            move_core(tt[next_site], std::move(nn));
        } else { // kronecker_prepend :379-399
            cand.width = base.width + 1;
            std::vector<uint32_t> tmp(cand.width);
            for (int loc = 0; loc < cs; ++loc)
                for (size_t b = 0; b < base.count; ++b) {
                    tmp[0] = (uint32_t)loc;
                    if (base.width) std::copy_n(base.at(b), base.width, tmp.data() + 1);
                    cand.push(tmp.data());
                }
            index_set.push_back(select(cand, f.col_perm, r));
            if (spectators) (*spectators)[site] = select((*spectators)[site], f.row_perm, r);
            // next <- left matrix of next (nl*ns x nr) * left factor (cl x r), nr == cl
            d_m1.reserve(std::max<size_t>(nxt.size(), 1));
            core_reshape_launch(nxt.buf.get(), nl, ns, nr, 0, d_m1.get(), st);
            d_m2.reserve(std::max<size_t>((size_t)nl * ns * r, 1));
            GemmDesc g;
            g.m = nl * ns;
            g.n = r;
            g.k = cl;
            g.A = d_m1.get();
            g.lda = nl * ns;
            g.strideA = 0;
            g.transA = 0;
            g.B = eng.left();
            g.ldb = cl;
            g.strideB = 0;
            g.transB = 0;
            g.C = d_m2.get();
            g.ldc = nl * ns;
            g.strideC = 0;
            g.alpha = 1.0;
            g.beta = 0.0;
            g.batch = 1;
            gemm_launch(g, st);
            DevCore nc, nn;
            nc.l = r;
            nc.s = cs;
            nc.r = cr;
            nc.buf.reserve(std::max<size_t>(nc.size(), 1));
            core_reshape_launch(eng.right(), r, cs, cr, 3, nc.buf.get(), st);
            nn.l = nl;
            nn.s = ns;
            nn.r = r;
            nn.buf.reserve(std::max<size_t>(nn.size(), 1));
            core_reshape_launch(d_m2.get(), nl, ns, r, 1, nn.buf.get(), st);
            eng.sync();
            move_core(tt[site], std::move(nc));
            move_core(tt[next_site], std::move(nn));
        }
        // merge_pivot_errors :417-424
        if (pivot_errors.size() < f.pivot_errors.size()) pivot_errors.resize(f.pivot_errors.size(), 0.0);
        for (size_t k = 0; k < f.pivot_errors.size(); ++k) pivot_errors[k] = std::fmax(pivot_errors[k], f.pivot_errors[k]);
    }
    if (!forward) std::reverse(index_set.begin(), index_set.end());
}

} // namespace

void Tci2::assign_from_tensor_train(const TensorTrain& src, const FromTensorTrainOptions& opt)
{
    // validate_options :100-121
    if (!std::isfinite(opt.tolerance) || opt.tolerance < 0.0)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "TensorCI2 conversion tolerance must be finite and nonnegative");
    if (opt.max_iter < 2) throw Error(T4A_GPU_INVALID_ARGUMENT, "TensorCI2 conversion max_iter must be at least 2");
    if (src.len() < 2)
        throw Error(T4A_GPU_INVALID_ARGUMENT, "TensorCI2 conversion requires at least 2 tensor-train sites");
    if (src.len() != n_) throw Error(T4A_GPU_INVALID_ARGUMENT, "TensorCI2 conversion: site count mismatch");
    for (size_t s = 0; s < n_; ++s)
        if (src.cores[s].s != local_dims[s])
            throw Error(T4A_GPU_INVALID_ARGUMENT, "TensorCI2 conversion: local dimension mismatch");
    fill_wait();
    hipStream_t st = eng.stream();
    std::vector<DevCore> tt(n_);
    for (size_t s = 0; s < n_; ++s) {
        tt[s].l = src.cores[s].l;
        tt[s].s = src.cores[s].s;
        tt[s].r = src.cores[s].r;
        tt[s].buf.reserve(std::max<size_t>(tt[s].size(), 1));
        if (tt[s].size())
            T4A_HIP(hipMemcpyAsync(tt[s].buf.get(), src.cores[s].buf.get(), tt[s].size() * sizeof(double),
                                   hipMemcpyDeviceToDevice, st));
    }
    DevBuf<double> d_m1, d_m2;
    std::vector<IndexSet> iset, jset, tmp;
    std::vector<double> errs, perr;
    sweep1site_get_indices(eng, tt, true, nullptr, opt, iset, errs, d_m1, d_m2);
    sweep1site_get_indices(eng, tt, false, nullptr, opt, jset, perr, d_m1, d_m2);
    for (size_t iter = 3; iter <= opt.max_iter; ++iter) {
        if (iter % 2 == 1) {
            std::vector<IndexSet> filtered = jset;
            sweep1site_get_indices(eng, tt, true, &filtered, opt, tmp, errs, d_m1, d_m2);
            jset = filtered;
            perr = errs;
            if (same_sets(tmp, iset)) break;
            iset = tmp;
        } else {
            std::vector<IndexSet> filtered = iset;
            sweep1site_get_indices(eng, tt, false, &filtered, opt, tmp, errs, d_m1, d_m2);
            iset = filtered;
            perr = errs;
            if (same_sets(tmp, jset)) break;
            jset = tmp;
        }
    }
    // from_parts_for_conversion (tensorci2.rs:406-447)
    i_set = iset;
    j_set = jset;
    mark_sets_changed();
    for (size_t p = 0; p < n_; ++p) {
        i_set[p].width = p;
        j_set[p].width = n_ - p - 1;
    }
    pivot_errors = perr;
    bond_errors.assign(n_ - 1, 0.0);
    clear_history();
    // max_site_tensor_abs :426-433
    T4A_HIP(hipMemsetAsync(d_maxbits_.get(), 0, sizeof(unsigned long long), st));
    for (size_t s = 0; s < n_; ++s) absmax_launch(tt[s].buf.get(), tt[s].size(), d_maxbits_.get(), st);
    unsigned long long bits = 0;
    T4A_HIP(hipMemcpyAsync(&bits, d_maxbits_.get(), sizeof(bits), hipMemcpyDeviceToHost, st));
    eng.sync();
    double mx;
    std::memcpy(&mx, &bits, sizeof(mx));
    max_sample_value = mx;
    for (size_t s = 0; s < n_; ++s) cores[s] = std::move(tt[s]);
    T4A_HIP(hipGetLastError());
}

} // namespace t4a
