// k_minimizer_pkd.hip -- instantiations of k_minimizer_pkd<W> (w = 2..13) and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_PKD
#include "kernels_pkd.hpp"
