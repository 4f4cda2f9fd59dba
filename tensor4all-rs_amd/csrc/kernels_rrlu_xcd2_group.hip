// kernels_rrlu_xcd2_group.hip — the group-launch instantiations of the second-generation single-XCD rrLU kernel (eight
// factorisations, one per XCD, in one launch: rrlu_xcd2_group_launch) as their own translation unit.
#define T4A_XCD_GROUP_TU 1
#include "kernels_rrlu_xcd2.hip"
