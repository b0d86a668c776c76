// k_minimizer_dense.hip -- instantiations of k_minimizer_dense<W> (w = 2..16) and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_DENSE
#include "kernels_fast.hpp"
