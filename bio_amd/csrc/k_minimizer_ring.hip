// k_minimizer_ring.hip -- instantiations of k_minimizer_ring<W> (w = 2..13) and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_RING
#include "kernels_ring.hpp"
