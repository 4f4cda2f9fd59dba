// kernels_rrlu_w1.hip — launchers of the one-wave rrLU kernel (kernels_rrlu_w1_body.hpp): matrices of at most 64 x 64.
// Solo launch: wave 0 of workgroup 0 factorises, the other workgroups (bond chain) evaluate the next bond's candidate matrix;
// group launch: workgroup x factorises slot x.  Plans carry wg = 2, RPT = 1, CPT = register columns (8, 16, 32, 64).
#include "kernels_rrlu_w1_body.hpp"

namespace t4a {

namespace {

template <int NC, bool ROWMAJOR, bool FACTORS>
__global__ void __launch_bounds__(XT) rrlu_w1_kernel(RrluXcdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[]; // W1Lds<NC, FACTORS>::bytes + 16
    if (blockIdx.x != 0) {
        if (p.spec.out && p.dims) {
            const int m_spec = p.dims[2] != 0 ? 0 : (p.dims_swap ? p.dims[1] : p.dims[0]);
            if (m_spec > 0 && m_spec <= p.M)
                xcd_spec_work(reinterpret_cast<const XcdSpecArgs*>(kernarg_base() + offsetof(RrluXcdArgs, spec)), m_spec,
                              reinterpret_cast<int*>(lds + W1Lds<NC, FACTORS>::bytes));
        }
        return;
    }
    if (threadIdx.x >= 64) return;
    (void)rrlu_w1_body<NC, ROWMAJOR, FACTORS>(p, lds);
}

template <int NC, bool ROWMAJOR>
__global__ void __launch_bounds__(64) rrlu_w1_group_kernel(RrluXcdGroupArgs g)
{
    __shared__ __attribute__((aligned(16))) char lds[W1Lds<NC, false>::bytes];
    (void)g;
    const RrluXcdArgs& p = *reinterpret_cast<const RrluXcdArgs*>(kernarg_base() + (size_t)blockIdx.x * sizeof(RrluXcdArgs));
    if (p.xcc < 0) return;
    (void)rrlu_w1_body<NC, ROWMAJOR, false>(p, lds);
}

template <int NC, bool ROWMAJOR, bool FACTORS> void w1_launch_solo(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream)
{
    constexpr size_t lds_bytes = W1Lds<NC, FACTORS>::bytes + 16; // (+ the tile word of the speculating workgroups)
    if constexpr (lds_bytes > 64 * 1024) {
        static std::once_flag attr_once;
        std::call_once(attr_once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rrlu_w1_kernel<NC, ROWMAJOR, FACTORS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
    }
    hipLaunchKernelGGL((rrlu_w1_kernel<NC, ROWMAJOR, FACTORS>), dim3(plan.grid), dim3(plan.grid > 1 ? XT : 64), lds_bytes, stream, a);
}

template <int NC> void w1_launch_nc(const RrluXcdPlan& plan, bool row_major, const RrluXcdArgs* solo, const RrluXcdGroupArgs* group, hipStream_t stream)
{
    if (solo) {
        const bool factors = solo->Aout != nullptr;
        if (row_major) {
            if (factors) w1_launch_solo<NC, true, true>(plan, *solo, stream);
            else w1_launch_solo<NC, true, false>(plan, *solo, stream);
        } else {
            if (factors) w1_launch_solo<NC, false, true>(plan, *solo, stream);
            else w1_launch_solo<NC, false, false>(plan, *solo, stream);
        }
    } else {
        if (row_major) hipLaunchKernelGGL((rrlu_w1_group_kernel<NC, true>), dim3(8), dim3(64), 0, stream, *group);
        else hipLaunchKernelGGL((rrlu_w1_group_kernel<NC, false>), dim3(8), dim3(64), 0, stream, *group);
    }
}

void w1_dispatch(const RrluXcdPlan& plan, bool row_major, const RrluXcdArgs* solo, const RrluXcdGroupArgs* group, hipStream_t stream)
{
    switch (plan.CPT) {
    case 8: w1_launch_nc<8>(plan, row_major, solo, group, stream); break;
    case 16: w1_launch_nc<16>(plan, row_major, solo, group, stream); break;
    case 32: w1_launch_nc<32>(plan, row_major, solo, group, stream); break;
    default: w1_launch_nc<64>(plan, row_major, solo, group, stream); break;
    }
}

} // namespace

// One-wave plan for an M x N matrix (upper bounds in a bond chain), or false when it does not fit (more than 64 rows or columns).
bool rrlu_w1_make_plan(int M, int N, RrluXcdPlan* out, int spec_blocks)
{
    static const bool off = std::getenv("T4A_NO_W1") != nullptr;
    // measured (tools/probe_wg.py): one wave issues an instruction every ~10 cycles whatever it is, so the per-column work of the
    // update (5 instructions) overtakes the one-workgroup kernel's barrier between 16 and 32 columns
    static const int max_n = std::getenv("T4A_W1_MAXN") ? std::atoi(std::getenv("T4A_W1_MAXN")) : 16;
    if (off || M < 1 || N < 1 || M > 64 || N > 64 || N > max_n) return false;
    RrluXcdPlan plan;
    plan.W = 1;
    plan.RPT = 1;
    plan.CPT = N <= 8 ? 8 : N <= 16 ? 16 : N <= 32 ? 32 : 64;
    plan.grid = 1 + (spec_blocks > 0 ? spec_blocks : 0);
    plan.lds_bytes = 0;
    plan.wg = 2;
    *out = plan;
    return true;
}

void rrlu_w1_launch(const RrluXcdPlan& plan, const RrluXcdArgs& a, hipStream_t stream) { w1_dispatch(plan, a.tie_row_major != 0, &a, nullptr, stream); }

void rrlu_w1_group_launch(const RrluXcdPlan& plan, const RrluXcdGroupArgs& a, bool tie_row_major, hipStream_t stream)
{
    w1_dispatch(plan, tie_row_major, nullptr, &a, stream);
}

} // namespace t4a
