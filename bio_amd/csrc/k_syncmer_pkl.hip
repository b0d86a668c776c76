// k_syncmer_pkl.hip -- instantiations of k_syncmer_pkl<W = k - s>, the packed syncmer machine's long plan (kernels_syncmer_pk.hpp), and
// their dispatch: its own translation unit so that the two plans compile side by side.
#define BSK_IMPL_SYNPKL
#include "kernels_syncmer_pk.hpp"
