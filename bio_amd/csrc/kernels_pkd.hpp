// kernels_pkd.hpp -- k_minimizer_pkd<W>: the packed window machine of k_minimizer_pk (kernels_pk.hpp: one 32-bit word  key | slot  per
// window element, three v_min_u32 per step, selections as bits) over the staging and the output of k_minimizer_dense (kernels_fast.hpp:
// every read owns a slab of `slab_read` tuples, the lane's staging rows are a ring that is flushed to the slab every NB blocks in whole
// 128-byte lines, flush_groups / flush_last).  For reads that select more tuples than a pair of k_minimizer_pk's columns holds -- from
// about 160 bases at w = 11 to 32 767 -- where the choice used to be the unit-row kernel (k_minimizer_ring: its lanes drift apart with
// the read length, 991 / 770 / 723 Gbases/s at 250 / 300 / 350 bases) or the exact 64-bit machine (k_minimizer_dense, ~700 at any length).
// NextMinimizer, sketches/sketch.go:205-309 (closed form: the leftmost argmin of every window, emitted when it changes).
//
// Exactness as in k_minimizer_pk: a read in which two equal 27-bit keys met in a min operation goes to the list of reads for the exact
// machine (list_append, k_minimizer_dense<W, true>), and so does a read that outgrows its slab (a homopolymer selects every position;
// the slab holds 2.6 / (w + 1) of the windows + 16) -- the flush skips such a lane and reports it (`lost`) instead of failing the call.
#pragma once
#include "kernels_pk.hpp"

namespace bsk {

template <int W>
struct PkdCfg {
    static constexpr int NB = DenseCfg<W>::NB, GL = DenseCfg<W>::GL, G = DenseCfg<W>::G, CAP = DenseCfg<W>::CAP;
    // Blocks per flush round: TWICE what the ring is sized for (CAP = a left-over of up to G - 1 rows + NB W new ones).  A flush is a fifth
    // of the kernel at one per NB blocks (knock-out at 400 bases, w = 11: 783 -> 999 Gbases/s without the regular rounds) and most of it is
    // fixed cost; a lane selects a sixth of its windows on average, so 2 NB W windows overflow the ring only where a read selects more
    // than half of them -- the flush sees that (the write pointer moved 2 NB W < CAP + 1 rows at most, so the distance is unambiguous),
    // drops the lane's staged rows and the read goes to the exact machine's list with the tied ones (`lost`).
#ifndef PKD_ROUND
#define PKD_ROUND 2
#endif
    // (w >= 8 only: narrower windows select a third or more of their positions, and a round of twice the windows would run a few reads
    // per thousand over their rings)
    static constexpr int NBF = (W >= 8 ? PKD_ROUND : 1) * NB;
    static_assert(NBF * W < CAP + 1, "the write pointer's distance per round must be unambiguous");
    typedef FLds<CAP, true> LY;
};

template <int W>
__global__ __launch_bounds__(64, 2) void k_minimizer_pkd(KArgs a) {
    typedef PkdCfg<W> C;
    typedef typename C::LY LY;
    constexpr int CAP = C::CAP, NBF = C::NBF, GL = C::GL, G = C::G;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    {  // the two hash tables: written once, nothing of the flush is laid over them
        PkTabs tabs;
        tabs.init(a.k, lane, (u32)LY::TAB, (u32)LY::TAB2);
        tabs.write(ldsq);
    }
    __syncthreads();
    const u64 slab_read = a.slab_read;
    constexpr u32 RB = (u32)(LY::ROW * 8);
    const u32 lseg = a.fixcap / a.list_grid;  // this workgroup's segment of the list of reads for the exact machine (list_append)
    u32 lcur = 0;
    // What a unit reads first -- its descriptors, input flags and the first four words of every read -- is requested while the PREVIOUS
    // unit hashes and waited for at that unit's first flushes, before their stores: loads and stores share one in-order vmcnt, so a
    // load at the start of a unit would wait for every store of the last flush to reach memory, three dependent loads deep.
    u64 d_cur = 0, d_n1 = 0;
    u32x4 pw_cur = {0, 0, 0, 0}, pw_n1 = {0, 0, 0, 0};
    u32 rfl_cur = 0, rfl_n1 = 0;
    bool have = false;
    const u64 rmax = a.n - 1;
    const u32 tku = a.tk ? a.tk : 4u;
    for (u32 unit = next_ticket(a.ticket, lane) * tku, uend = unit + tku; unit < a.nunits; ++unit, ({
             if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * tku;
                 uend = unit + tku;
             }
         })) {
        const u64 r = (u64)unit * 64 + lane;
        const bool nxt = unit + 1 != uend && unit + 1 < a.nunits;  // the next unit is this wave's too
        if (!have) {  // first unit of a ticket: nothing was requested ahead (indices beyond the batch are clamped to its last read)
            d_cur = a.desc[r < rmax ? r : rmax];
            rfl_cur = a.rflags ? a.rflags[r < rmax ? r : rmax] : 0u;
            pw_cur = *reinterpret_cast<const u32x4 *>(a.words + (d_cur >> 24));
        }
        d_n1 = a.desc[r + 64 < rmax ? r + 64 : rmax];
        rfl_n1 = a.rflags ? a.rflags[r + 64 < rmax ? r + 64 : rmax] : 0u;
        int nflush = 0;
        const u64 d = d_cur;
        const u64 off = d >> 24;
        u64 L = 0, ro = r;
        const u32 rfl = rfl_cur;
        if (r < a.n) {
            L = desc_len(a, d);
            ro = out_index(a, r, d);  // (length-binned batches: the read's own place in its chunk)
        }
        const bool ok = r < a.n && L >= (u64)a.circ_ext && (L - (u64)a.circ_ext) + 1 >= (u64)a.k + (u64)W;
        const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
        const u32 nk_max = wave_max_u32(nk);
        const u32 nk_min = ~wave_max_u32(~(ok ? nk : 0xffffffffu));  // (over the lanes with a read: the blocks all of them fill take no per-lane window test)
        const u64 ubase = (u64)unit * 64 * slab_read;
        u32 done = 0, tmin_lane = 0xffffffffu, lost = 0;
        if (nk_max) {
            PkMin<W, true, LY, true> pm;
            pm.w = a.words + off;
            pm.lds = ldsq;
            pm.k = a.k;
            pm.lane = lane;
            pm.nk = nk;
            pm.pw = pw_cur;
            pm.begin((u32)lane * 8u, ok ? (int)RB : 0, 0u);  // (a lane without a read stays on row 0 of its own column)
            u32 head = 0, left = 0, wprev = 0;
            // what the lane staged since the last flush, whole groups of G to the read's slab; `last`: everything
            auto flush = [&](bool last) {
                // the next block's words (requested a block of hashing ago) are waited for HERE, before the flush's stores are issued: vmcnt
                // is one in-order counter, and a wait behind the stores waits for every one of them to reach memory (k_minimizer_dense)
                asm volatile("" : "+v"(pm.in_lo), "+v"(pm.in_hi), "+v"(pm.out_lo), "+v"(pm.out_hi), "+v"(d_n1), "+v"(rfl_n1));
                if (nflush == 0) pw_n1 = *reinterpret_cast<const u32x4 *>(a.words + (d_n1 >> 24));  // (the next unit's descriptor is in: its words go out)
                else asm volatile("" : "+v"(pw_n1));
                ++nflush;
                const u32 wrow = (pm.slot - (u32)lane * 8u) / RB;
                const u32 moved = wrow >= wprev ? wrow - wprev : wrow + (u32)(CAP + 1) - wprev;  // rows staged since the last flush
                const bool over = left + moved > (u32)CAP;  // the ring ran over the lane's own rows: what it holds is gone (see PkdCfg)
                lost |= over ? 1u : 0u;
                head = over ? wrow : head;
                const u32 cnt = over ? 0u : left + moved;
#ifdef PKD_NOFLUSH  // dev knock-out (timing only): the regular rounds do not move anything
                if (last) flush_last<LY, true, GL, CAP + 1>(lds, lane, cnt < (u32)G ? cnt : (u32)(G - 1), done, slab_read, ubase, a, head, &lost);
#else
                if (last) flush_last<LY, true, GL, CAP + 1>(lds, lane, cnt, done, slab_read, ubase, a, head, &lost);
                else flush_groups<LY, true, GL, CAP + 1>(lds, lane, cnt, done, slab_read, ubase, a, head, &lost);
#endif
                const u32 nfl = last ? cnt : (cnt & ~(u32)(G - 1));
                head += nfl;
                head = head >= (u32)(CAP + 1) ? head - (u32)(CAP + 1) : head;
                head = head >= (u32)(CAP + 1) ? head - (u32)(CAP + 1) : head;
                done += nfl;
                left = cnt - nfl;
                wprev = wrow;
            };
            pm.template block<true, false, 0>(0, ok ? 1u : 0u, nk_max > (u32)W);
            u32 i0 = W;
            int par = 1, inround = 0;
            while (i0 < nk_max) {
                const bool more = i0 + (u32)W < nk_max, full = i0 + (u32)W <= nk_min;
                if (par) {
                    if (full) pm.template block<false, false, 1>(i0, 1u, more);
                    else pm.template block<false, true, 1>(i0, 1u, more);
                } else {
                    if (full) pm.template block<false, false, 0>(i0, 1u, more);
                    else pm.template block<false, true, 0>(i0, 1u, more);
                }
                if (++inround == NBF || !more) {  // (the last block's round leaves room for the W slots the drain emits)
                    inround = 0;
                    flush(false);
                }
                i0 += (u32)W;
                par ^= 1;
            }
            if (par) pm.template drain<0>(i0 - (u32)W);  // the last block's own slots
            else pm.template drain<1>(i0 - (u32)W);
            flush(true);
            tmin_lane = pm.tmin;
        }
        if (!nk_max) pw_n1 = *reinterpret_cast<const u32x4 *>(a.words + (d_n1 >> 24));  // (a unit of short reads only: nothing was flushed)
        d_cur = d_n1;
        rfl_cur = rfl_n1;
        pw_cur = pw_n1;
        have = nxt;
        const u64 redo = __builtin_amdgcn_ballot_w64(ok && (tmin_lane < 32u || lost));
        if (redo) list_append(a, a.rlist, lseg, lcur, redo, lane, r);
        if (r < a.n) {
            if (!((redo >> lane) & 1)) a.refs[ro] = ((ubase + (u64)lane * slab_read) << 24) | done;  // (listed reads: the list pass writes theirs)
            u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
            if (ok) sbyte |= (u8)rfl;
            a.status[ro] = sbyte;
        }
    }
    list_close(a.rlist, lseg, lcur, lane);
}

#ifdef BSK_IMPL_PKD
#ifndef BSK_PKD_WS
#define BSK_PKD_WS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#endif
bool pkd_minimizer_supported(int w) { return w >= 2 && w <= 13; }
int pkd_minimizer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
    switch (w) {
#define X(WW) \
    case WW: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_pkd<WW>, 64, 0); break;
        BSK_PKD_WS(X)
#undef X
        default: break;
    }
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
// (the list pass, k_minimizer_dense<W, true>, is instantiated in k_minimizer_pk.hip: pk_minimizer_list_launch)
void pkd_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
    switch (w) {
#define X(WW) \
    case WW: hipLaunchKernelGGL((k_minimizer_pkd<WW>), dim3(grid), dim3(64), 0, stream, a); break;
        BSK_PKD_WS(X)
#undef X
        default: break;
    }
    KArgs al = a;
    al.slab_read = a.list_slab;  // a listed read's slab: one tuple per window
    pk_minimizer_list_launch(w, grid, stream, al);
}
#endif  // BSK_IMPL_PKD

}  // namespace bsk
