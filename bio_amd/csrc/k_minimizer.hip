// k_minimizer.hip -- instantiations of k_minimizer_fast<W> (w = 2..32) and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_MINIMIZER
#include "kernels_fast.hpp"
