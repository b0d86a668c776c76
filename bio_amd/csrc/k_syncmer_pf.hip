// k_syncmer_pf.hip -- instantiations of k_syncmer_pf<W = k - s>, the packed syncmer machine with the emit fused into every unit
// (kernels_syncmer_pf.hpp), and their dispatch; the listed reads' exact machine is k_syncmer_fix.hip's.
#define BSK_IMPL_SYNPF
#include "kernels_syncmer_pf.hpp"
