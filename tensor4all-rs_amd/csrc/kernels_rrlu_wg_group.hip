// kernels_rrlu_wg_group.hip — the group-launch instantiations of the one-workgroup rrLU kernel (one factorisation per workgroup,
// eight per launch: rrlu_wg_group_launch) as their own translation unit, so that they compile beside the solo instantiations.
#define T4A_WG_GROUP_TU 1
#include "kernels_rrlu_wg.hip"
