// k_minimizer_seg.hip -- instantiations of k_minimizer_seg<W> (w = 2..32) and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_SEG
#include "kernels_seg.hpp"
