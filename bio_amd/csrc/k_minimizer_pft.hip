// k_minimizer_pft.hip -- instantiations of k_minimizer_pft<W>, minimizers of long sequences as dense tiles (kernels_minimizer_pf.hpp), and their dispatch.
#define BSK_IMPL_MINPFT
#include "kernels_minimizer_pf.hpp"
