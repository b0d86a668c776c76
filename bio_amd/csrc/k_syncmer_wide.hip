// k_syncmer_wide.hip -- k_syncmer_fast<W = k - s> for k - s = 25..32 and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_SYNCMER_WIDE
#include "kernels_syncmer.hpp"
