// fast_dispatch.hpp -- host-side entry points of the specialised kernel families.  Each family is instantiated in its own
// translation unit (k_minimizer.hip, k_syncmer.hip, k_protein.hip) so that the library builds in parallel; biosketch.hip only
// sees these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_generic.hpp"  // KArgs

#define BSK_FAST_CAP 32   // tuples staged per read by k_minimizer_fast (slab = 64 * CAP tuples per unit)
#define BSK_SYN_CAP 28    // same for k_syncmer_fast (19.9 KB of LDS per wavefront: still 8 per CU)
#define BSK_ASCII_PAD 1024  // slack behind residue / ASCII buffers: the 16-byte staging loads read past a chunk's end

namespace bsk {
bool fast_minimizer_supported(int w);
int fast_minimizer_blocks_per_cu(int w);
void fast_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a);
bool dense_minimizer_supported(int w);  // per-read slabs + mid-read flushes: windows that select more than 32 positions per read
int dense_minimizer_blocks_per_cu(int w);
void dense_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a);
void dense_minimizer_ascii_launch(int w, int grid, hipStream_t stream, const KArgs &a);  // the ASCII side launch of a mixed batch (KArgs::subset, ascii, aoff, out_base)
int dense_minimizer_ascii_blocks_per_cu(int w);

bool pk_minimizer_supported(int w);  // packed 32-bit window machine, w <= 16 (kernels_pk.hpp)
int pk_minimizer_blocks_per_cu(int w);
u32 pk_minimizer_short_bases();
void pk_minimizer_launch(int w, bool long_reads, int grid, hipStream_t stream, const KArgs &a);
void pk_minimizer_list_launch(int w, int grid, hipStream_t stream, const KArgs &a);  // the list pass alone: k_minimizer_dense<W, true>

bool pkd_minimizer_supported(int w);  // the packed machine over per-read slabs and mid-read flushes: reads of any length below 32 768 bases (kernels_pkd.hpp)
int pkd_minimizer_blocks_per_cu(int w);
void pkd_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a);  // (+ the list pass with KArgs::list_slab per listed read)

bool ring_minimizer_supported(int w);  // packed window machine, unit rows through a ring of 16 staged rows: three waves per SIMD (kernels_ring.hpp)
int ring_minimizer_blocks_per_cu(int w);
u32 ring_minimizer_short_bases();
void ring_minimizer_launch(int w, bool long_reads, int grid, hipStream_t stream, const KArgs &a);

bool seg_minimizer_supported(int w);  // per-read slabs + a flush of everything staged every few blocks: three wavefronts per SIMD
int seg_minimizer_blocks_per_cu(int w);
void seg_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a);

void wpr_minimizer_launch(int grid, hipStream_t stream, const KArgs &a);  // one read per wavefront, w = 11 (the A/B experiment, kernels_wpr.hpp)
int wpr_minimizer_blocks_per_cu();

bool fast_syncmer_supported(int k, int s);
int fast_syncmer_blocks_per_cu(int w);
void fast_syncmer_launch(int w, int grid, hipStream_t stream, const KArgs &a);
void fast_syncmer_ascii_launch(int w, int grid, hipStream_t stream, const KArgs &a);  // the ASCII side launch of a mixed batch
bool fast_syncmer_wide_supported(int w);  // k - s = 25..32 (k_syncmer_wide.hip), reached through the three functions above
int fast_syncmer_wide_blocks_per_cu(int w);
void fast_syncmer_wide_launch(int w, int grid, hipStream_t stream, const KArgs &a);

// packed window machine (kernels_syncmer_pk.hpp).  lng = false: k_syncmer_pk, three waves per SIMD, reads up to 224 bases in 23-row
// columns; lng = true: k_syncmer_pkl, two waves per SIMD, reads up to 480 bases in 58-row columns, k - s up to 24
bool pk_syncmer_supported(int w, bool lng);
u32 pk_syncmer_max_bases(bool lng);
u32 pk_syncmer_pair_rows(bool lng);
int pk_syncmer_blocks_per_cu(int w, bool lng);
void pk_syncmer_launch(int w, bool lng, int grid, int fix_grid, hipStream_t stream, const KArgs &a);

// the same machine with the emit fused into every unit (kernels_syncmer_pf.hpp, round 6): the s-mer side alone + from-scratch hashes of
// the selected k-mers at the end of the unit; no staging columns; k <= 64; the exact machine over its listed reads is k_syncmer_fix.hip's
// lng: k_syncmer_pfl -- 32 words of a read in registers (reads of up to 480 bases), k - s up to 24, two waves per SIMD
bool pf_syncmer_supported(int w, bool lng);
u32 pf_syncmer_max_bases(bool lng);
u32 pf_syncmer_mask_rows(bool lng);
u32 pf_syncmer_unit_tuples(bool lng);  // what a unit's emit phase takes
int pf_syncmer_blocks_per_cu(int w, bool lng);
void pf_syncmer_launch(int w, bool lng, int grid, int fix_grid, hipStream_t stream, const KArgs &a);

// minimizers of long sequences as DENSE tiles (kernels_minimizer_pf.hpp, round 6): the tile kernel writes final tuples, no stitch pass; w = 4..13, k <= 64
bool pft_minimizer_supported(int w);
u32 pft_minimizer_max_tile_bases();
u32 pft_minimizer_mask_rows();
u32 pft_minimizer_unit_tuples();
size_t pft_minimizer_scratch_words(u32 nunits);
int pft_minimizer_blocks_per_cu(int w);
void pft_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a);

// the two-pass plan (kernels_syncmer_sel.hpp): selection by the packed s-mer machine, then the selected k-mers hashed from scratch
bool sel_syncmer_supported(int w);
u32 sel_syncmer_max_bases();
int sel_syncmer_blocks_per_cu(int w);
void sel_syncmer_launch(int w, int grid, int fix_grid, int cus, u32 nw, hipStream_t stream, const KArgs &a);  // nw: packed words of a read pass 2 stages (longest read + k)

bool fast_prot_supported(int w, int k);
int fast_prot_blocks_per_cu(int w, int k);
void fast_prot_launch(int w, int k, int grid, hipStream_t stream, const KArgs &a);
void fast_prot_dna_launch(int w, int k, int grid, hipStream_t stream, const KArgs &a);  // 2-bit DNA batch, translation fused in

bool fast_prot_hash_supported(int k);
int fast_prot_hash_blocks_per_cu(int k);
void fast_prot_hash_launch(int k, int grid, hipStream_t stream, const KArgs &a);
int fast_prot_hash_dna_blocks_per_cu(int k);
void fast_prot_hash_dna_launch(int k, int grid, hipStream_t stream, const KArgs &a);  // 2-bit DNA batch, translation fused in
}  // namespace bsk
