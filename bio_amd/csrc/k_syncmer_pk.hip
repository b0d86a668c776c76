// k_syncmer_pk.hip -- instantiations of k_syncmer_pk<W = k - s> (+ k_syncmer_fast<W, true>, its fix pass) and their dispatch.
#define BSK_IMPL_SYNPK
#include "kernels_syncmer_pk.hpp"
