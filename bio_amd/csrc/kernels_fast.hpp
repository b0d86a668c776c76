// kernels_fast.hpp -- specialised kernels (compile-time window) -- placeholder until written
#pragma once
#include "kernels_generic.hpp"
namespace bsk {
static inline bool fast_minimizer_supported(int) { return false; }
static inline int fast_minimizer_blocks_per_cu(int) { return 1; }
static inline void fast_minimizer_launch(int, int, hipStream_t, const KArgs &) {}
}  // namespace bsk
