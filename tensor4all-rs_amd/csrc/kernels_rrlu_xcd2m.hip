// kernels_rrlu_xcd2m.hip — the second-generation single-XCD rrLU kernel (kernels_rrlu_xcd2.hip: same body, same protocol, same
// bit-exact contract — matrixlu.rs:480-519, :735-819) for matrices beyond one XCD's register file: up to 1 536 rows (RPT = 24 row slots
// per lane) and the columns over the agents of K = 1 ... 3 neighbouring XCDs (KX template parameter: write-through mailbox stores,
// 4 K key loads per polling lane).  BASELINE.json configs[3] (d = 40, chi = 512): the saturated bonds with the history extras are
// ~1 450 x 1 450 — 2.1 M values, two XCDs' worth of registers — and ran on the chip-wide round-1 kernel at 6.8 us per pivot step.
#define T4A_XCD2_MULTI_TU 1
#include "kernels_rrlu_xcd2.hip"
