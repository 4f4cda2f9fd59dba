// kernels_rrlu_xcd_group.hip — the group-launch instantiations of the single-XCD rrLU kernel (eight factorisations, one per XCD,
// in one launch: rrlu_xcd_group_launch) as their own translation unit, so that they compile beside the solo instantiations.
#define T4A_XCD_GROUP_TU 1
#include "kernels_rrlu_xcd.hip"
