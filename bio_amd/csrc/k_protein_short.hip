// k_protein_short.hip -- the register-wyhash protein kernels for k = 4..8 residues (kernels_protein.hpp) and their dispatch: a translation unit
// of its own beside k_protein.hip (k = 9..16).
#define BSK_IMPL_PROTEIN_SHORT
#include "kernels_protein.hpp"
