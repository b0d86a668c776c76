// k_syncmer_sel.hip -- the two-pass syncmer plan: k_syncmer_sel<W = k - s> (selection), k_sel_scan, k_syncmer_emit (hash what was selected),
// + k_syncmer_fast<W, true> over the listed reads, and their dispatch.
#define BSK_IMPL_SYNSEL
#include "kernels_syncmer_sel.hpp"
