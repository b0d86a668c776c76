// k_minimizer_pk.hip -- instantiations of k_minimizer_pk<W> (w = 2..16) and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_PK
#include "kernels_pk.hpp"
