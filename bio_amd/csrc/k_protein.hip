// k_protein.hip -- instantiations of the register-wyhash protein kernels and their dispatch (fast_dispatch.hpp).
#define BSK_IMPL_PROTEIN
#include "kernels_protein.hpp"
