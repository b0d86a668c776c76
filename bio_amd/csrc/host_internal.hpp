// host_internal.hpp -- what the host translation units of libbiosketch.so share: biosketch.hip (ABI, contexts, batches, results), planner.hip,
// launch.hip, tiles.hip, classes.hip.  Nothing here crosses the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "biosketch.h"
#include "host_types.hpp"
#include "kernels_generic.hpp"
#include "fast_dispatch.hpp"
#include "planner_table.hpp"
#include "kernels_fast.hpp"      // (templates only: launch.hip is the one translation unit that instantiates them)
#include "kernels_more.hpp"
#include "kernels_simhash.hpp"
#include "kernels_translate.hpp"

using namespace bsk;

// internal return codes: they never cross the C ABI (public_rc, biosketch.hip)
#define BSK_RESIZE (-1001)         // internal: a timed re-run outgrew the regions the result was sized with (run_planned_resizing sizes again, once)
#define BSK_REPLAN_CLASS (-1002)   // internal: a PART of a class plan overflowed on the launch the caller sees (run_classed sizes the parts again)
#define BSK_REPLAN_UNFUSED (-1000)  // internal: the fused DNA -> protein plan gave up, run the two-step path
static constexpr u64 kMaxPrefetchWords = 32;  // >= SynPkLdsL::NW (static_assert beside pk_syncmer_max_bases, kernels_syncmer_pk.hpp)
static constexpr u32 kSynTileMin = (u32)PlannerTable::syn_tile_min_bases;

template <class K>
static int blocks_per_cu(K kernel) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 64, 0) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}

// Which kernel runs a (batch, params) pair, on how many workgroups, and how the tuple arrays are organised.
enum Which { K_MIN_GEN_P, K_MIN_GEN_A, K_NT_P, K_NT_A, K_MIN_FAST, K_NT_FAST, K_SYN_P, K_SYN_A, K_KMER_P, K_KMER_A, K_SIM_P, K_SIM_A,
             K_PROT_HASH, K_PROT_MIN, K_SYN_FAST, K_PROT_MIN_FAST, K_PROT_HASH_FAST, K_SIM_FAST, K_MIN_DENSE, K_MIN_SEG, K_MIN_WPR, K_MIN_PK, K_SYN_PK, K_MIN_RING, K_SYN_SEL, K_MIN_PKD, K_MIN_DENSE_A, K_SYN_FAST_A };
struct Plan {
    Which which = K_MIN_GEN_P;
    int grid = 1;
    int fast_w = 0;
    bool syn_fused = false; // K_SYN_PK: k_syncmer_pf (the emit fused into every unit: no staging columns, kernels_syncmer_pf.hpp)
    bool syn_long = false;  // K_SYN_PK: k_syncmer_pkl (longer columns, more words in registers, two waves per SIMD)
    bool slab = false;     // true: unit u owns tuples [u*slab_unit, (u+1)*slab_unit) (+ overflow region); no look-back
    u64 slab_unit = 0;     // tuples per unit slab
    u64 slab_total = 0;    // nunits * slab_unit
    u32 nunits = 0;
    u32 ring_w = 0;
    size_t ring_entries = 0;
    u64 slab_read = 0;     // per-sequence slabs (protein fast path)
    int fast_k = 0;
    bool compact = false;  // stream kernels, fixed-length batch: runs without padding (k_nthash_fast<MODE, true>)
    u32 bin_gran = 0;      // != 0: the kernel runs over the batch's length-binned descriptors (ensure_binned), classes of this many bases
    bool fused_dna = false;  // protein minimizer of a 2-bit DNA batch: the kernel translates where it fetches its residues
    // mixed batch: the fast 2-bit kernel over all reads + the general ASCII kernel over the reads with a non-ACGT letter
    bool mixed = false;
    Which side_which = K_MIN_GEN_A;
    u32 side_nunits = 0, side_ring_w = 0;
    u64 side_slab = 0;     // K_MIN_DENSE_A: tuples of a read's slab in the side launch's region
    int side_grid = 1;
};

static_assert(sizeof(Plan) <= sizeof(((bsk_result *)nullptr)->plan_blob) && std::is_trivially_copyable<Plan>::value, "bsk_result::plan_blob holds a Plan");

// class plans (classes.hip)
struct ClassPart {
    bsk_batch *sub = nullptr;   // borrowed view: desc = the class's descriptors (context pool), words = the parent's
    bsk_result *res = nullptr;  // refs / status of its own; hash / pos = the parent's tail once the parent exists
    u32 *list = nullptr;        // the class's reads (batch positions), ascending (context pool)
    u64 n = 0, bases = 0, off = 0, extent = 0;
    u32 lo = 0, hi = 0;
    bool tiled = false;  // longer than the kind's tile threshold: the part runs over tiles (sketch_tiled), its result is wide and copied into the tail
    bool fresh = false;  // ... and was just run by the sizing call (the parent's first launch does not run it again)
    bool async = false;  // ... without a synchronisation of its own (sketch_tiled, tile_async): its overflow flags wait in the side context's d_ticket[24]
};
struct ClassSet {
    std::vector<ClassPart> parts;
    bsk_batch *view = nullptr;  // the bulk class's view of the batch
    u64 tail = 0;
    u64 n = 0, n_bases = 0;     // what it was cut from
    u32 maxlen = 0, blo = 0, bhi = 0;
    const u64 *desc = nullptr;
    const u32 *words = nullptr;
    float build_ms = 0.0f;
    bool masked = false;   // no view array: the bulk's kernel masks by length itself (KArgs::cls_lo / cls_hi / cls_pretend)
    u32 pretend = 0;
};

struct ClassSig {
    int which = -1, octave = 0;
    bool syn_long = false, syn_fused = false;
    bool operator==(const ClassSig &o) const { return which == o.which && octave == o.octave && syn_long == o.syn_long && syn_fused == o.syn_fused; }
};
struct ClassCut {
    u32 lo, hi;   // the class takes the lengths [lo, hi]
    u32 shortest; // the shortest read it holds
    u64 n, bases;
    ClassSig sig;
};
// (internal: none of these is an export of libbiosketch.so)
#pragma GCC visibility push(hidden)
// ---- biosketch.hip ----
void spare_register(bsk_ctx *ctx, bool add);
void spare_flush(bsk_ctx *ctx);
void spare_give(bsk_ctx *ctx, void *p);
hipError_t spare_take(bsk_ctx *ctx, void **out, size_t bytes);
int grid_for(bsk_ctx *ctx, u64 items, int block);
u64 pad_words(u32 maxlen);
u32 env_u32(const char *name, u32 dflt);
u32 min_tile_min(const bsk_ctx *ctx);
bsk_ctx *side_ctx(bsk_ctx *ctx);
int bin_with_batch(bsk_ctx *ctx, bsk_batch *b);
bool kind_has_pos(int kind);
int result_prepare(bsk_ctx *ctx, bsk_result **res, u64 n, int kind, u64 cap, u64 tail = 0);
int make_circular(bsk_ctx *ctx, const bsk_batch *b, int k, bsk_batch **out);
// ---- planner.hip ----
int validate(const bsk_params *p, int alphabet);
int ensure_scratch(bsk_ctx *ctx, size_t nunits, size_t ring_entries);
int build_subset(bsk_ctx *ctx, bsk_batch *b);
void plan_record(bsk_result *res, const bsk_batch *b, const bsk_params *p, int circ_ext, const Plan &pl);
bool plan_recall(const bsk_result *res, const bsk_batch *b, const bsk_params *p, int circ_ext, Plan &pl);
bool slab_budget_ok(const bsk_batch *b, u64 slab_read);
u32 bin_gran_for(const bsk_ctx *ctx, const bsk_batch *b, int step);
int ensure_binned(bsk_ctx *ctx, const bsk_batch *b, u32 lo, u32 gran, u32 mlo, u32 mhi, u32 mpretend, bool fine = false);
int make_plan(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, Plan &pl);
int make_plan_enc(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, Plan &pl, bool use_ascii);
void plan_name(const Plan &pl, const bsk_params *p, bool tiled, int cus, bsk_result *res);
u64 syn_pk_fixcap(u64 n, int grid);
u64 estimate_cap(const bsk_batch *b, const bsk_params *p, int circ_ext);
u64 estimate_cap_n(const bsk_params *p, u64 bases, u64 nreads);
// ---- launch.hip ----
int launch(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, bsk_result *res, int circ_ext, const Plan &pl, hipEvent_t ev0, hipEvent_t ev1);
int run_planned(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms);
int run_planned_resizing(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms);
// workgroups per CU of the kernels only launch.hip instantiates (the planner asks by name)
enum OccId { OCC_MIN_GEN_P, OCC_MIN_GEN_A, OCC_NT_FAST0, OCC_NT_FAST1, OCC_NT_FAST2, OCC_NT_FAST3, OCC_NT_FAST4, OCC_NT_FAST0C, OCC_NT_FAST1C, OCC_NT_FAST2C, OCC_NT_P, OCC_NT_A, OCC_SYN_P, OCC_SYN_A,
             OCC_KMER_P, OCC_KMER_A, OCC_SIMF_5S, OCC_SIMF_6S, OCC_SIMF_5M, OCC_SIMF_6M, OCC_SIMF_5, OCC_SIMF_6, OCC_SIM_P, OCC_SIM_A, OCC_PROT_HASH, OCC_PROT_MIN };
int occ(OccId id);
// ---- tiles.hip ----
bool kind_tiles(const bsk_params *p);
u32 tile_min_for(const bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p);
int sketch_tiled(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p_in, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms);
// ---- classes.hip ----
void class_set_free(ClassSet *cs);
int launch_parts(bsk_ctx *ctx, ClassSet *cs, const bsk_params *p, bsk_result *res, bool tiled);
int adopt_parts(bsk_ctx *ctx, ClassSet *cs, bsk_result *res);
bool class_decide(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, std::vector<ClassCut> &cuts, int &bulk);
int run_classed(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, int circ_ext, bsk_result **result, int warmup, int iters, float *kernel_ms, bool *applied);
int class_build(bsk_ctx *ctx, const bsk_batch *b, const bsk_params *p, const std::vector<ClassCut> &cuts, int bulk, ClassSet *cs);
#pragma GCC visibility pop
