// kernels_syncmer_sel.hpp -- the two-pass syncmer plan (round 5): SELECT, then hash only what was selected.
//
// NextSyncmer (sketches/sketch.go:312-477) emits the canonical hash of a k-mer for ~1.5 / (k - s + 1) of the windows -- 7.1 of the 101
// windows of a 150-base read at k = 31, s = 11 -- but a one-pass kernel (k_syncmer_pk) rolls the k-mer hash over EVERY position,
// canonicalises it and stages it, because which position is selected is only known 2(k - s) - 1 steps later.  The k-mer side is a third
// of its instructions and all of its staging (timing-only build without it: 1 482 against 966 Gbases/s, profiles/r05).  Here:
//
//   pass 1  k_syncmer_sel<W>   the packed s-mer window machine of k_syncmer_pk alone (SynPk<W, LY, SEL = true>): one selection WORD per
//                              block of W windows and lane leaves to HBM (mask rows [unit][block][lane]: whole 256-byte lines), plus
//                              per read its count and its offset in the unit, per unit its total; reads with a key tie go to the list
//                              of reads for the exact machine as before (k_syncmer_fast<W, true>)
//   scan    k_sel_scan         exclusive scan of the unit totals -> where every unit's tuples start (dense: no slabs, no gaps)
//   pass 2  k_syncmer_emit     one LANE PER TUPLE: owner read by a search over the unit's 64 offsets, the position by select-nth-set-bit
//                              over the read's mask words, then the canonical ntHash of those k bases FROM SCRATCH -- four bases per
//                              table row (256 rows of (fwd, rev) contributions in LDS) -- and coalesced 512 / 256-byte stores.
//
// ntHash is XOR-linear in its bases: fwd = XOR_j rol(seed[b_j], k-1-j), rev = XOR_j rol(seed[comp b_j], j), so four bases are ONE row
// F4 = rol(sF[b0],3)^rol(sF[b1],2)^rol(sF[b2],1)^sF[b3], R4 = sR[b0]^rol(sR[b1],1)^rol(sR[b2],2)^rol(sR[b3],3):
// fwd = rol(fwd, 4) ^ F4, rev ^= rol(R4, 4 g) for group g.  Results are bit-identical to the rolling form (same 64-bit arithmetic).
#pragma once
#include "kernels_syncmer_pk.hpp"

namespace bsk {

// LDS plan of pass 1: the s-mer table, the [W][64] column of parked suffix minima (first-window test), the LDS-DMA buffers
struct SynSelLds {
    static constexpr int PR = 1, ROW = 33;  // (unused by the SEL machine; SynPk names them)
    static constexpr int NW = PKNW;
    static constexpr bool DMA = true;
    static constexpr int TABK = 0, TABS = 0;  // one table
    static constexpr int PARK = 320;           // u32 [24][64]
    static constexpr int SH = PARK, SP = PARK;
    static constexpr int WBUF = PARK + 24 * 256;
    static constexpr int DBUF = WBUF + NW * 64 * 4;
    static constexpr int TOTAL = DBUF + 512;   // 11 072 B: fourteen waves per CU by LDS
};

#define BSK_SEL_LISTED 0x80000000u  // sel_cnt[slot]: the read went to the list for the exact machine (pass 2 leaves it alone)

#ifndef SYNSEL_LB
#define SYNSEL_LB 3
#endif
template <int W>
__global__ __launch_bounds__(64, SYNSEL_LB) void k_syncmer_sel(KArgs a) {
    typedef SynSelLds LY;
    constexpr int NQ = LY::NW / 4;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    {  // the s-mer table: lanes 0..19 write their rows once (nothing overwrites them: no copy-out in this kernel)
        SynPkTabs tabs;
        tabs.init(a.s, a.s, lane);
        if (lane < 20) *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::TABS + lane * 16) = tabs.row;
        wave_sync_lds();
    }
    u64 d_cur = 0;
    bool have = false;
    const u32 lseg = a.fixcap / a.list_grid;
    u32 lcur = 0;
    const u32 wbuf = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(ldsq + LY::WBUF));
    const u32 dbuf = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(ldsq + LY::DBUF));
    for (u32 unit = next_ticket(a.ticket, lane) * 8u, uend = unit + 8u; unit < a.nunits; ++unit, ({
             if (unit == uend) {
                 unit = next_ticket(a.ticket, lane) * 8u;
                 uend = unit + 8u;
             }
         })) {
        const u64 r = (u64)unit * 64 + lane;
        const bool nxt = unit + 1 != uend && unit + 1 < a.nunits;
        const u64 rmax = a.n - 1;
        typename SynVec<LY::NW>::type wr;
        u64 d_n1;
        u32 rfl;
        if (!have) {  // first unit of a ticket: nothing was requested ahead
            d_cur = a.desc[r < rmax ? r : rmax];
            synpk_dma_words<NQ>(a.words + (d_cur >> 24), wbuf);
            synpk_dma_desc(a.desc + (r + 64 < rmax ? r + 64 : rmax), dbuf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        {
            const LDSQ u32x4 *wb = reinterpret_cast<const LDSQ u32x4 *>(ldsq + LY::WBUF) + lane;
            const LDSQ u32 *db = reinterpret_cast<const LDSQ u32 *>(ldsq + LY::DBUF) + lane;
            u32x4 wq[NQ];
#pragma unroll
            for (int j = 0; j < NQ; ++j) wq[j] = wb[64 * j];
            u32 dl = db[0], dh = db[64];
            static_assert(NQ == 4, "k_syncmer_sel");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]), "+v"(dl), "+v"(dh)::"memory");
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                wr[4 * j] = wq[j].x;
                wr[4 * j + 1] = wq[j].y;
                wr[4 * j + 2] = wq[j].z;
                wr[4 * j + 3] = wq[j].w;
            }
            d_n1 = ((u64)dh << 32) | dl;
        }
        synpk_dma_words<NQ>(a.words + (d_n1 >> 24), wbuf);
        synpk_dma_desc(a.desc + (r + 128 < rmax ? r + 128 : rmax), dbuf);
        rfl = pk_load_u8(a.rflags ? a.rflags + (r < rmax ? r : rmax) : reinterpret_cast<const u8 *>(a.desc));
        const u64 d = d_cur;
        const u64 L = desc_len(a, d);
        const u64 ro = out_index(a, r, d);
        const long long Lorig = (long long)L - a.circ_ext;
        const bool ok = r < a.n && Lorig >= 0 && Lorig >= 2LL * a.k - a.s - 1 && L >= (u64)a.k;  // sketch.go:149
        const u32 nwin = ok ? (u32)(L - 2 * (u64)a.k + a.s + 2) : 0u;                             // end + 1
        const u32 ns = ok ? (u32)(L - a.s + 1) : 0u;
        const u32 ns_max = wave_max_u32(ns);
        const u32 nwin_min = ~wave_max_u32(ok ? ~nwin : 0u);
        u32 cnt = 0, tmin_lane = 0xffffffffu;
        if (ns_max) {
            SynPk<W, LY, 1> sp;
            sp.lds = ldsq;
            sp.k = a.k;
            sp.s = a.s;
            sp.lane = lane;
            sp.end_plus1 = nwin;
            sp.wr = wr;
            sp.gmask = a.sel_mask + (u64)unit * a.sel_nb * 64u + (u32)lane;
            sp.run(ns_max, nwin_min, 0u, 0, 0u);
            cnt = sp.nsel;
            tmin_lane = sp.tmin;
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rfl)::"memory");  // the next unit's words and the descriptors after them are in LDS
        d_cur = d_n1;
        have = nxt;
        // reads the exact machine must run: two equal 27-bit keys met in one of their min operations (kernels_syncmer_pk.hpp)
        const u64 redo = __builtin_amdgcn_ballot_w64(ok && tmin_lane < 32u);
        if (redo) {
            list_append(a, reinterpret_cast<u32 *>(a.fixlist), lseg, lcur, redo, lane, r);
            if ((redo >> lane) & 1) cnt = 0;
        }
        const u32 incl = wave_incl_scan_u32(cnt, lane);
        const u32 excl = incl - cnt;
        if (r < a.n) {
            const bool listed = (redo >> lane) & 1;
            a.sel_cnt[r] = (listed ? BSK_SEL_LISTED : 0u) | (excl << 8) | cnt;  // (a read of <= 224 bases selects < 256 positions, a unit < 2^23; a listed read counts 0)
            if (!listed) {
                u8 sbyte = ok ? BSK_ST_OK : BSK_ST_SHORT;
                if (ok && a.rflags) sbyte |= (u8)rfl;
                a.status[ro] = sbyte;
            }
        }
        if (lane == 63) a.sel_utot[unit] = incl;
    }
    list_close(reinterpret_cast<u32 *>(a.fixlist), lseg, lcur, lane);
}

// exclusive scan of the unit totals (u32) -> unit bases (u64); a ticket is 1 024 units, decoupled look-back between tickets;
// total[0] receives the sum.  (2 10^6 units for 1.25 10^8 reads: ~20 us)
__global__ __launch_bounds__(64) void k_sel_scan(const u32 *utot, u32 nunits, u32 nblocks, u32 *ticket, u64 *lookback, u64 *ubase, u64 *total) {
    const int lane = lane_id();
    for (;;) {
        const u32 blk = next_ticket(ticket, lane);
        if (blk >= nblocks) break;
        u32 v[16];
        u64 sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {  // lane owns 16 consecutive units: [blk * 1024 + lane * 16, +16)
            const u32 u = blk * 1024u + (u32)lane * 16u + (u32)j;
            v[j] = u < nunits ? utot[u] : 0u;
            sum += v[j];
        }
        u64 incl = sum;  // wave inclusive scan of the lanes' sums
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(((lane - dlt) & 63) * 4, (int)(u32)incl);
            const u32 hi = (u32)__builtin_amdgcn_ds_bpermute(((lane - dlt) & 63) * 4, (int)(u32)(incl >> 32));
            if (lane >= dlt) incl += ((u64)hi << 32) | lo;
        }
        const u64 T = wave_bcast_u64(incl, 63);
        const u64 base = lookback_exclusive(lookback, blk, T, lane);
        u64 at = base + incl - sum;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 u = blk * 1024u + (u32)lane * 16u + (u32)j;
            if (u < nunits) ubase[u] = at;
            at += v[j];
        }
        if (blk == nblocks - 1 && lane == 63) total[0] = base + T;
    }
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------------------------
// FIVE bases per table row (1 024 rows of (fwd, rev) contributions, 16 KB of LDS per workgroup of eight waves): k = 31 is one single base
// and six rows.  fwd is Horner's rule, fwd = rol(fwd, 5) ^ F5; for rev = XOR_g rol(R5_g, 5 g) the accumulator is kept rotated the other
// way, acc = ror(acc ^ R5_g, 5), so that every step rotates by a CONSTANT and one rotation by 5 G (+ the single bases) at the end puts it
// right.  The first k mod 5 bases are taken singly (any split of the k bases into consecutive pieces gives the same XOR).
// LDS: the table (static) + per wave (dynamic, sized by the host from the batch): u32 excl[64], the unit's selection words [nb][64] and
// the unit's packed words [64][nw | 1] -- everything a unit needs is loaded up front, so that no load sits behind the tuple stores of the
// loop (gfx9 counts loads and stores in ONE in-order vmcnt: a load per iteration waited for the previous iteration's stores, and the
// kernel was bound by that latency -- 10.1 ms for 8.9 10^8 tuples at sixteen waves per CU).
struct SynEmitLds {
    static constexpr int T5 = 0;              // u32x4 [1024]
    static constexpr int T1 = 16384;          // u32x4 [4]: one base
    static constexpr int TOTAL = T1 + 64;
    static constexpr int DYN_MAX = 48 * 1024; // per workgroup: with the table 64 KB, two workgroups per CU
};
#define BSK_EMIT_TCAP 1024u  // tuples of a unit expanded per round (a unit of 150-base reads at k = 31, s = 11 has ~450)
__host__ __device__ inline u32 emit_wave_bytes(u32 nb, u32 nw) { return nb * 256u + 64u * (nw | 1u) * 4u + BSK_EMIT_TCAP * 2u; }

__device__ __forceinline__ void emit_rol64(u32 &lo, u32 &hi, u32 rot) {  // rot wave-uniform, 0..63
    if (rot & 32u) {
        const u32 t = lo;
        lo = hi;
        hi = t;
    }
    const u32 rs = rot & 31u;
    if (rs) {
        const u32 nl = __builtin_amdgcn_alignbit(lo, hi, 32u - rs), nh = __builtin_amdgcn_alignbit(hi, lo, 32u - rs);
        lo = nl;
        hi = nh;
    }
}

// G = k / 5 rows per k-mer (a template parameter: the rows' table reads are then issued together and waited for once)
template <int G>
__global__ __launch_bounds__(512) void k_syncmer_emit(KArgs a, u32 nw) {
    typedef SynEmitLds LY;
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 1024; e += (int)blockDim.x) {  // row e: codes c0 | c1 << 2 | ... | c4 << 8, c0 the FIRST base
        u64 f = 0, r = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const unsigned c = ((unsigned)e >> (2 * j)) & 3u;
            f ^= rol64(seed_fwd_code(c), (unsigned)(4 - j));
            r ^= rol64(seed_rev_code(c), (unsigned)j);
        }
        reinterpret_cast<u32x4 *>(lds + LY::T5)[e] = (u32x4){(u32)f, (u32)(f >> 32), (u32)r, (u32)(r >> 32)};
    }
    if (tid < 4) {
        const u64 f1 = seed_fwd_code((unsigned)tid), r1 = seed_rev_code((unsigned)tid);
        reinterpret_cast<u32x4 *>(lds + LY::T1)[tid] = (u32x4){(u32)f1, (u32)(f1 >> 32), (u32)r1, (u32)(r1 >> 32)};
    }
    __syncthreads();
    const u32 nb = a.sel_nb, ws = nw | 1u;  // (odd row stride: the owners' rows spread over the banks)
    u32 *const s_mask = reinterpret_cast<u32 *>(dyn + (size_t)wv * emit_wave_bytes(nb, nw));
    u32 *const s_words = s_mask + nb * 64u;
    unsigned short *const s_tp = reinterpret_cast<unsigned short *>(s_words + 64u * ws);  // tuple -> (owner lane << 8) | idx
    const u32 W = (u32)(a.k - a.s);
    const u32 krem = (u32)a.k - 5u * (u32)G;  // 0..4
    u32 *const ticket = a.ticket + 4;  // (pass 1 used [0]; the host zeroes both)
    for (;;) {
        u32 t0 = 0;
        if (lane == 0) t0 = atomicAdd(ticket, 1u);
        const u32 unit0 = (u32)__builtin_amdgcn_readfirstlane((int)t0) * 4u;  // a ticket is four units
        if (unit0 >= a.nunits) break;
        for (u32 unit = unit0; unit < unit0 + 4u && unit < a.nunits; ++unit) {
            const u64 r = (u64)unit * 64 + lane;
            const u64 d = r < a.n ? a.desc[r] : a.desc[a.n - 1];
            const u32 sc = r < a.n ? a.sel_cnt[r] : 0u;
            const bool listed = (sc & BSK_SEL_LISTED) != 0;
            const u64 ubase = a.sel_ubase[unit];
            const u32 T = a.sel_utot[unit];
            const u32 cnt = sc & 0xffu, excl = (sc & ~BSK_SEL_LISTED) >> 8;
            if (r < a.n && !listed) a.refs[out_index(a, r, d)] = ((ubase + excl) << 24) | cnt;
            if (ubase + T > a.ovf_base) {  // the dense region is too small for this batch: the host sizes it again (total[0] has the need)
                if (lane == 0) atomicOr(&a.ticket[1], 1u);
                continue;
            }
            if (T == 0) continue;
            // the unit's selection words (coalesced rows) and packed words (every lane its own read's, nw of them) into LDS
            const u32 *const mrow = a.sel_mask + (u64)unit * nb * 64u;
            for (u32 mm = 0; mm < nb; ++mm) s_mask[mm * 64u + (u32)lane] = mrow[mm * 64u + (u32)lane];
            {
                const u32 *const wsrc = a.words + (d >> 24);
                u32 *const wdst = s_words + (u32)lane * ws;
                for (u32 j = 0; j < nw; ++j) wdst[j] = wsrc[j];
            }
            for (u32 c0 = 0; c0 < T; c0 += BSK_EMIT_TCAP) {  // (one round unless the unit selected more than 1 024 positions)
                // every read's selection words expanded into the unit's tuple list: tuple excl + j is (this lane, its j-th selected window)
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                {
                    u32 at = excl, left = cnt;  // this lane's next tuple; how many it still has (rows behind the unit's last block were never written)
                    for (u32 mm = 0; mm < nb; ++mm) {
                        u32 w = left ? s_mask[mm * 64u + (u32)lane] : 0u;
                        const u32 ibase = mm * W + 1u - W;  // bit O of word mm: idx = (mm - 1) W + 1 + O  (kernels_syncmer_pk.hpp, SEL)
                        while (__builtin_amdgcn_ballot_w64(w != 0u)) {
                            if (w) {
                                const u32 O = (u32)__builtin_ctz(w);
                                w &= w - 1u;
                                if (at - c0 < BSK_EMIT_TCAP) s_tp[at - c0] = (unsigned short)(((u32)lane << 8) | (ibase + O));
                                ++at;
                                if (--left == 0) w = 0;
                            }
                        }
                        if (__builtin_amdgcn_ballot_w64(left != 0u) == 0) break;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const u32 tend = T - c0 < BSK_EMIT_TCAP ? T - c0 : BSK_EMIT_TCAP;
                for (u32 tb = 0; tb < tend; tb += 64) {
                    const u32 tl = tb + (u32)lane;
                    const bool live = tl < tend;
                    const u32 tp = live ? (u32)s_tp[tl] : 0u;
                    const u32 o = tp >> 8, idx = tp & 0xffu;
                    // the k bases from idx on: five words of the owner's row, funnel-shifted so that base idx sits at bit 0 of h0
                    const u32 *const wp = s_words + o * ws + (idx >> 4);
                    const u32 q0 = wp[0], q1 = wp[1], q2 = wp[2], q3 = wp[3], q4 = wp[4];
                    const u32 sh0 = (idx & 15u) * 2u;
                    const u32 h0 = __builtin_amdgcn_alignbit(q1, q0, sh0), h1 = __builtin_amdgcn_alignbit(q2, q1, sh0), h2 = __builtin_amdgcn_alignbit(q3, q2, sh0),
                              h3 = __builtin_amdgcn_alignbit(q4, q3, sh0);  // 64 bases
                    // the rows of five: bases from idx + krem on; group q takes bits [10 q, 10 q + 10) of g0..g3
                    const u32 sh1 = krem * 2u;  // (< 10 bits)
                    const u32 g0 = __builtin_amdgcn_alignbit(h1, h0, sh1), g1 = __builtin_amdgcn_alignbit(h2, h1, sh1), g2 = __builtin_amdgcn_alignbit(h3, h2, sh1), g3 = h3 >> sh1;
                    u32 code[12];
                    code[0] = g0, code[1] = g0 >> 10, code[2] = g0 >> 20, code[3] = __builtin_amdgcn_alignbit(g1, g0, 30), code[4] = g1 >> 8, code[5] = g1 >> 18;
                    code[6] = __builtin_amdgcn_alignbit(g2, g1, 28), code[7] = g2 >> 6, code[8] = g2 >> 16, code[9] = __builtin_amdgcn_alignbit(g3, g2, 26), code[10] = g3 >> 4,
                    code[11] = g3 >> 14;
                    u32x4 x[G > 0 ? G : 1];
#pragma unroll
                    for (int q = 0; q < G; ++q) x[q] = *reinterpret_cast<const u32x4 *>(lds + LY::T5 + ((code[q] & 0x3ffu) << 4));
                    u32 fl = 0, fh = 0, rl = 0, rh = 0;
                    for (u32 j = 0; j < krem; ++j) {  // the first k mod 5 bases, singly
                        const u32 c1 = (h0 >> (2u * j)) & 3u;
                        const u32x4 y = *reinterpret_cast<const u32x4 *>(lds + LY::T1 + c1 * 16u);
                        const u32 nfl = __builtin_amdgcn_alignbit(fl, fh, 31) ^ y.x, nfh = __builtin_amdgcn_alignbit(fh, fl, 31) ^ y.y;  // rol(f, 1)
                        fl = nfl;
                        fh = nfh;
                        u32 xl = y.z, xh = y.w;
                        emit_rol64(xl, xh, j);
                        rl ^= xl;
                        rh ^= xh;
                    }
                    u32 al = 0, ah = 0;  // the rev accumulator, rotated: acc = ror(acc ^ R5, 5)
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        const u32 nfl = __builtin_amdgcn_alignbit(fl, fh, 27) ^ x[q].x, nfh = __builtin_amdgcn_alignbit(fh, fl, 27) ^ x[q].y;  // rol(f, 5)
                        fl = nfl;
                        fh = nfh;
                        const u32 tl2 = al ^ x[q].z, th2 = ah ^ x[q].w;
                        al = __builtin_amdgcn_alignbit(th2, tl2, 5);  // ror 5
                        ah = __builtin_amdgcn_alignbit(tl2, th2, 5);
                    }
                    // rev of the rows = rol(acc, 5 G) (the accumulator was rotated right by 5 once per row), shifted by the krem single bases
                    emit_rol64(al, ah, (5u * (u32)G + krem) & 63u);
                    rl ^= al;
                    rh ^= ah;
                    const bool rev = rh < fh || (rh == fh && rl < fl);  // nthash returns rev only when strictly smaller
                    if (live) {
                        a.hash[ubase + c0 + tl] = rev ? (((u64)rh << 32) | rl) : (((u64)fh << 32) | fl);
                        a.pos[ubase + c0 + tl] = idx | (rev ? BSK_POS_STRAND_BIT : 0u);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

#ifdef BSK_IMPL_SYNSEL
#ifndef BSK_SYNSEL_WS
#define BSK_SYNSEL_WS(X) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)
#endif
bool sel_syncmer_supported(int w) {
#define X(WW) \
    if (w == WW) return true;
    BSK_SYNSEL_WS(X)
#undef X
    return false;
}
u32 sel_syncmer_max_bases() { return 16u * (u32)(SynSelLds::NW - 2); }
int sel_syncmer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
#define X(WW) \
    if (w == WW) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_syncmer_sel<WW>, 64, 0);
    BSK_SYNSEL_WS(X)
#undef X
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
// pass 1, the scan of the unit totals, pass 2, and the exact machine over the listed reads
void sel_syncmer_launch(int w, int grid, int fix_grid, int cus, u32 nw, hipStream_t stream, const KArgs &a) {
#define X(WW) \
    if (w == WW) hipLaunchKernelGGL((k_syncmer_sel<WW>), dim3(grid), dim3(64), 0, stream, a);
    BSK_SYNSEL_WS(X)
#undef X
    const u32 nblocks = (a.nunits + 1023u) / 1024u;
    hipLaunchKernelGGL(k_sel_scan, dim3(std::min<u32>(nblocks, (u32)cus * 4u)), dim3(64), 0, stream, a.sel_utot, a.nunits, nblocks, a.ticket + 5, a.sel_lookback, a.sel_ubase,
                       a.total);
    {  // pass 2: as many waves per workgroup as 48 KB of dynamic LDS hold (eight for 150-base reads at k - s = 20), two workgroups per CU
        const u32 per_wave = emit_wave_bytes(a.sel_nb, nw);
        const int waves = (int)std::max<u32>(1u, std::min<u32>(8u, (u32)SynEmitLds::DYN_MAX / per_wave));
        const dim3 g2((unsigned)std::max(1, std::min<int>((int)((a.nunits + 4u * (u32)waves - 1u) / (4u * (u32)waves)), cus * 2))), b2((unsigned)(64 * waves));
        const u32 dynb = per_wave * (u32)waves;
        switch (a.k / 5) {
#define Y(GG) \
    case GG: hipLaunchKernelGGL((k_syncmer_emit<GG>), g2, b2, dynb, stream, a, nw); break;
            Y(1) Y(2) Y(3) Y(4) Y(5) Y(6) Y(7) Y(8) Y(9) Y(10) Y(11) Y(12)
#undef Y
            default: break;
        }
    }
#define X(WW) \
    if (w == WW) hipLaunchKernelGGL((k_syncmer_fast<WW, true>), dim3(fix_grid), dim3(64), 0, stream, a);
    BSK_SYNSEL_WS(X)
#undef X
}
#endif  // BSK_IMPL_SYNSEL

}  // namespace bsk
