// k_syncmer_fix.hip -- instantiations of k_syncmer_fast<W = k - s, true>, the exact 64-bit machine over the reads the packed syncmer kernels
// listed (kernels_syncmer_pk.hpp), and their dispatch: its own translation unit (the slowest of the three to compile).
#define BSK_IMPL_SYNFIX
#include "kernels_syncmer_pk.hpp"
