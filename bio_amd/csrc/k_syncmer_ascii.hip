// k_syncmer_ascii.hip -- instantiations of k_syncmer_fast<W, false, true>, the syncmer side launch of a mixed batch fed from ASCII
// (kernels_syncmer.hpp), and their dispatch.
#define BSK_IMPL_SYNCMER_ASCII
#include "kernels_syncmer.hpp"
