// k_syncmer.hip -- instantiations of k_syncmer_fast<W = k - s> and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_SYNCMER
#include "kernels_syncmer.hpp"
