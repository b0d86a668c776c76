// kernels_minimizer_pf.hpp -- k_minimizer_pft<W>: minimizers of LONG sequences without a stitch pass (round 6; opt-in: BSK_TILE_DENSE=1).
//
// A long sequence runs as overlapping tiles, one per lane (kernels_tile.hpp).  The shipped path writes every tile's tuples -- the overlap's
// included, positions tile-local -- into slabs, and k_tile_stitch reads them back to drop, shift and pack them.  Here the tile kernel writes
// the FINAL tuples:
//
//   hash phase   PkMin<W, ..., SELM>: the packed window machine of k_minimizer_pk, selection only -- one word of selection bits per block and
//                lane into LDS rows (nothing is staged);
//   repair       a tile in which two equal 27-bit keys met (the packed machine cannot vouch for it: kernels_pk.hpp; 1e-4 of the tiles on
//                random sequence, low-complexity stretches on real genomes) is done again RIGHT HERE by all 64 lanes (pft_repair_tile):
//                a dense stream has no room for a list pass afterwards;
//   ownership    bits of positions outside the tile's own [lo, hi) are dropped (TileTab::keep); the rows move into registers and the lane's
//                count is final: the unit's total is published (one granule per unit);
//   -- the unit waits here, hashed and counted, while the wave hashes its NEXT unit --
//   place        the tuples of all units before it: the totals of the units before it in its chunk of 64 + a decoupled look-back over the
//                chunks (a unit waits for LOWER units only: the wave holds tickets it has not begun);
//   emit         as k_syncmer_pf: the lanes' packed words back to LDS, the selection rows expanded into a list (in passes of TCAP tuples),
//                one LANE PER TUPLE hashes the k bases from scratch (pf_hash_kmer), picks the strand, adds the tile's offset in its
//                sequence and stores 64 consecutive final tuples per round: a sequence's tuples end up contiguous and in position order.
//
// Tickets: one unit per ticket, from eight heads (device_common.hpp).  Measured (NOTEBOOK 6.4): exact, and NOT faster than slabs + stitch --
// 6.2 ms against 5.4-5.9 for 2 10^9 bases: with the tickets out of the way the kernel is VALU-bound, and hashing the selected 17.7 % of all
// positions a second time costs what the stitch pass costs in HBM time.  Tiles of at most 160 bases (12 packed words), k <= 64, w = 4..13;
// LDS 13 440 B, 168 VGPRs: twelve waves per CU.
#pragma once
#include "kernels_syncmer_pf.hpp"

namespace bsk {

struct MinPfLds {
    static constexpr int PR = 1, ROW = 33;  // (PkMin::begin names them; the selection machine stages nothing)
    static constexpr int NW = PKNW;
    static constexpr int TCAP = 1280;
    static constexpr int TAB = 0;             // 20 x uint4: the k-mer update table (PkTabs)
    static constexpr int TAB2 = 320;          // 16 x uint4: two-base warm-up table
    static constexpr int KTA = 576;           // u32x4 [64] (pf_hash_kmer)
    static constexpr int KT1 = KTA + 1024;    // u32x4 [4]
    static constexpr int MROWS = 16;          // selection words: one row per block of W k-mers
    static constexpr int MASK = KT1 + 64;     // u32 [MROWS][64]
    static constexpr int EST = 12;            // words of a tile kept for the emit phase (160 bases + look-ahead)
    static constexpr int EBUF = MASK;         // u32 [64][EST], over the dead selection rows
    static constexpr int KTB = EBUF + 64 * EST * 4;  // u32x4 [64], written at the start of every emit phase
    static constexpr int OWN = MASK + MROWS * 256;   // u32 [64] (+ 256 spare): per tile its offset in its sequence
    static constexpr int FLAT = OWN + 512;    // u16 [TCAP]: (lane << 9) | position in the tile
    static constexpr int WBUF = FLAT + TCAP * 2;
    static constexpr int DBUF = WBUF + NW * 64 * 4;
    static constexpr int TOTAL = DBUF + 512;
    static_assert(KTB + 1024 <= MASK + MROWS * 256 && WBUF % 16 == 0 && EBUF % 16 == 0 && (EST * 4) % 16 == 0 && TOTAL <= 13648, "MinPfLds: twelve waves per CU");
};
__host__ __device__ constexpr u32 min_pft_max_tile_bases() { return 16u * (u32)(MinPfLds::EST - 2); }

// ---- the exact machine for the rare tile: all 64 lanes on ONE tile, its selection rows rewritten ------------------------------------------
// A tile in which two equal 27-bit keys met (tmin < 32) is done again exactly before anything depends on its count: the tile's packed words
// go into LDS scratch (the tuple list's place, not yet in use), every lane hashes the k-mers at positions lane, lane + 64, ... from scratch
// (64-bit canonical ntHash), every lane then takes the windows lane, lane + 64, ...: leftmost argmin of the w hashes (strictly smaller wins:
// the leftmost of equal hashes stays, sketch.go:263-295), selected when it differs from the window before -- the closed form of
// NextMinimizer (sketch.go:205-309) -- and the positions are OR-ed into the tile's selection rows (cleared first).  ~1 000 instructions per
// tile: a unit's place in the dense stream is not held up.  Returns BSK_ST_FIRST_WINDOW_TIE or 0 (oracle tie_flag: some h[a] of the first w
// hashes equals the minimum of the hashes behind it).
template <int W>
__device__ __noinline__ u32 pft_repair_tile(char *lds, const u32 *gwords, u32 L, int k, int bl, int lane) {
    typedef MinPfLds LY;
    LDSQ char *const ldsq = (LDSQ char *)lds;
    LDSQ u32 *const sw = reinterpret_cast<LDSQ u32 *>(ldsq + LY::FLAT);            // [16] the tile's words
    LDSQ u64 *const sh = reinterpret_cast<LDSQ u64 *>(ldsq + LY::FLAT + 64);       // [<= 160] canonical hashes
    LDSQ u16 *const sp = reinterpret_cast<LDSQ u16 *>(ldsq + LY::FLAT + 64 + 160 * 8);  // [<= 160] argmin of every window
    static_assert(64 + 160 * 8 + 160 * 2 <= LY::TCAP * 2, "pft_repair_tile: scratch inside the tuple list");
    const u32 nk = L - (u32)k + 1u, nwin = nk - (u32)W + 1u, nb = (nk + (u32)W - 1u) / (u32)W;
    if (lane < 16) sw[lane] = lane < LY::EST ? gwords[lane] : 0u;
    for (u32 mm = (u32)lane; mm < nb; mm += 64u) *reinterpret_cast<LDSQ u32 *>(ldsq + LY::MASK + mm * 256u + (u32)bl * 4u) = 0u;
    wave_sync_lds();
    for (u32 p = (u32)lane; p < nk; p += 64u) {
        u64 fh = 0, rh = 0;
        for (int t = 0; t < k; ++t) {
            const u32 q = p + (u32)t, c = (sw[q >> 4] >> (2u * (q & 15u))) & 3u;
            fh = rol64(fh, 1) ^ seed_fwd_code(c);
            rh ^= rol64(seed_rev_code(c), (unsigned)t);
        }
        sh[p] = rh < fh ? rh : fh;
    }
    wave_sync_lds();
    for (u32 j = (u32)lane; j < nwin; j += 64u) {
        u64 best = sh[j];
        u32 bp = j;
#pragma unroll
        for (int t = 1; t < W; ++t) {
            const u64 c = sh[j + (u32)t];
            if (c < best) {
                best = c;
                bp = j + (u32)t;
            }
        }
        sp[j] = (u16)bp;
    }
    wave_sync_lds();
    for (u32 j = (u32)lane; j < nwin; j += 64u) {
        const u32 bp = sp[j];
        if (j == 0 || sp[j - 1] != bp) atomicOr(reinterpret_cast<u32 *>(lds + LY::MASK + (bp / (u32)W) * 256u + (u32)bl * 4u), 1u << (bp % (u32)W));
    }
    u32 flag = 0;
    if (W > 1) {  // (every lane the same scan: w <= 13 reads)
        u64 m = sh[W - 1];
        for (int a = W - 2; a >= 0; --a) {
            const u64 c = sh[a];
            if (c < m) m = c;
            else if (c == m) flag = BSK_ST_FIRST_WINDOW_TIE;
        }
    }
    wave_sync_lds();
    return flag;
}

template <int W>
__global__ __launch_bounds__(64, 3) void k_minimizer_pft(KArgs a) {
    typedef MinPfLds LY;
    constexpr int NQ = LY::NW / 4;
    static_assert(NQ == 4, "k_minimizer_pft");
    __shared__ __attribute__((aligned(16))) char lds[LY::TOTAL];
    LDSQ char *const ldsq = (LDSQ char *)lds;
    const int lane = lane_id();
    u32x4 ktb_row;
    {
        PkTabs tabs;
        tabs.init(a.k, lane, (u32)LY::TAB, (u32)LY::TAB2);
        tabs.write(ldsq);
        const unsigned c0 = (unsigned)lane & 3u, c1 = ((unsigned)lane >> 2) & 3u, c2 = ((unsigned)lane >> 4) & 3u;  // c0 the FIRST base
        const u64 f3 = rol64(seed_fwd_code(c0), 2) ^ rol64(seed_fwd_code(c1), 1) ^ seed_fwd_code(c2);
        const u64 r3 = seed_rev_code(c0) ^ rol64(seed_rev_code(c1), 1) ^ rol64(seed_rev_code(c2), 2);
        const u64 fa = rol64(f3, 3), rb = rol64(r3, 3);
        *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::KTA + lane * 16) = (u32x4){(u32)fa, (u32)(fa >> 32), (u32)r3, (u32)(r3 >> 32)};
        ktb_row = (u32x4){(u32)f3, (u32)(f3 >> 32), (u32)rb, (u32)(rb >> 32)};
        if (lane < 4) {
            const u64 f1 = seed_fwd_code((unsigned)lane), r1 = seed_rev_code((unsigned)lane);
            *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::KT1 + lane * 16) = (u32x4){(u32)f1, (u32)(f1 >> 32), (u32)r1, (u32)(r1 >> 32)};
        }
        wave_sync_lds();
    }
    u64 d_cur = 0;
    bool have = false;
    const u32 wbuf = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(ldsq + LY::WBUF));
    const u32 dbuf = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(size_t)(ldsq + LY::DBUF));
    // ONE unit per ticket (a unit's place in the stream depends on every unit before it: tickets of several units would make the first unit of a
    // ticket wait for the last of the ticket before), and the wave holds FOUR units: the one whose tuples it writes (hashed and counted an
    // iteration ago), the one it hashes, the one whose words are on their way and the one whose descriptors are.  Hash first, write after:
    // a unit's place is known when every unit before it has COUNTED, and those had a whole hash phase more to do so -- the wait is (mostly) over
    // before it begins.  A unit waits for units before it only, and counts are published before any wait: whoever holds the lowest unit not yet
    // counted is not waiting for anything.
    const u64 rmax = a.n - 1;
    auto rd = [&](u32 u) {  // the read a lane holds in unit u (beyond the batch: clamped, never used)
        const u64 q = (u64)u * 64 + (u64)lane;
        return q < rmax ? q : rmax;
    };
    const u32 nchunks = (a.nunits + 63u) >> 6;
    u64 *const lbc = a.lookback, *const utot = lbc + nchunks;
    // Tickets from eight heads (device_common.hpp): one unit per ticket, 240 000 units of 2 10^9 bases -- one head word WAS 2.7 of the kernel's
    // 6.2 ms.  A wave holds units of ONE head only, in rising order (it goes to the next head with nothing in hand): it waits for lower units
    // only, and whoever holds (or will next pull) the lowest unit not yet counted is not waiting.
    // the pending unit
    u32 p_rows[LY::MROWS], p_w[LY::EST];
    u32 p_exclo = 0, p_TO = 0, p_cnt = 0, p_shift = 0, p_nb = 0, p_unit = 0, p_stat = 0;
    bool p_valid = false;
#pragma unroll
    for (int mm = 0; mm < LY::MROWS; ++mm) p_rows[mm] = 0;
#pragma unroll
    for (int j = 0; j < LY::EST; ++j) p_w[j] = 0;
    u32 *const heads = reinterpret_cast<u32 *>(utot + ((a.nunits + 15u) & ~15u));
    const u32 xcc = xcc_id();
  for (u32 hh = 0; hh < 8u; ++hh) {
    const u32 hx = (xcc + hh) & 7u;
    u32 *const head = heads + hx * 32u;
    auto pull = [&]() { return next_ticket(head, lane) * 8u + hx; };
    u32 unit = pull();
    if (unit >= a.nunits) continue;
    u32 unit_n1 = pull(), unit_n2 = pull();
    have = false;
    p_valid = false;
    for (;;) {
        const bool live = unit < a.nunits;
        u32 rows[LY::MROWS];
        u32x16 wr;
        u32 cnt = 0, exclo = 0, TO = 0, shift = 0, nb = 0, stat = 0;
        if (live) {
            const u64 r = (u64)unit * 64 + lane;
            u64 d_n1;
            if (!have) {  // the wave's first unit: nothing was requested ahead
                d_cur = a.desc[rd(unit)];
                synpk_dma_words<NQ>(a.words + (d_cur >> 24), wbuf);
                synpk_dma_desc(a.desc + rd(unit_n1), dbuf);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                have = true;
            }
            {
                u32x4 wq[NQ];
                const LDSQ u32x4 *wb = reinterpret_cast<const LDSQ u32x4 *>(ldsq + LY::WBUF) + lane;
                const LDSQ u32 *db = reinterpret_cast<const LDSQ u32 *>(ldsq + LY::DBUF) + lane;
#pragma unroll
                for (int j = 0; j < NQ; ++j) wq[j] = wb[64 * j];
                u32 dl = db[0], dh = db[64];
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
            // the tile's own positions and its place in its sequence (in flight through the hash phase)
            const u64 keep_w = __builtin_nontemporal_load(&a.tkeep[rd(unit)]), shf_w = __builtin_nontemporal_load(&a.tshift[rd(unit)]);
            synpk_dma_words<NQ>(a.words + (d_n1 >> 24), wbuf);
            synpk_dma_desc(a.desc + rd(unit_n2), dbuf);
            const u64 d = d_cur;
            const u64 L = d & (u64)a.len_mask;
            const bool ok = r < a.n && L + 1 >= (u64)a.k + (u64)W;  // sketch.go:92 (tiles: the sequence's own length rule was applied when it was cut)
            const u32 nk = ok ? (u32)(L - a.k + 1) : 0u;
            const u32 nk_max = wave_max_u32(nk);
            const u32 nk_min = ~wave_max_u32(~(ok ? nk : 0xffffffffu));
            u32 tmin_lane = 0xffffffffu;
            if (nk_max) {
                PkMin<W, false, LY, false, true> pm;
                pm.w = a.words + (d >> 24);
                pm.lds = ldsq;
                pm.k = a.k;
                pm.lane = lane;
                pm.nk = nk;
                pm.wr = wr;
                pm.run(nk_max, ok, nk_min, 0u, 0, 0u);
                tmin_lane = pm.tmin;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next unit's words and descriptors are in LDS
            d_cur = d_n1;
            const u32 lo = (u32)keep_w, hi = (u32)(keep_w >> 32);
            shift = (u32)shf_w;
            // tiles the packed machine cannot vouch for (a key tie): the exact machine rewrites their selection rows, here and now
            u32 tieflag = 0;
            {
                u64 bad = __builtin_amdgcn_ballot_w64(ok && tmin_lane < 32u);
                while (bad) {
                    const int bl = __builtin_ctzll(bad);
                    bad &= bad - 1;
                    const u32 dlo = (u32)__builtin_amdgcn_readlane((int)(u32)d, bl), dhi = (u32)__builtin_amdgcn_readlane((int)(u32)(d >> 32), bl);
                    const u64 bd = ((u64)dhi << 32) | dlo;
                    const u32 f = pft_repair_tile<W>(lds, a.words + (bd >> 24), (u32)(bd & (u64)a.len_mask), a.k, bl, lane);
                    if (lane == bl) tieflag = f;
                }
            }
            stat = (r < a.n ? 0x100u : 0u) | (ok ? BSK_ST_OK : BSK_ST_SHORT) | tieflag;
            // ownership: only positions in [lo, hi) are this tile's; the lane's count of them.  The rows move into registers here: their place
            // in LDS becomes the pending unit's words (EBUF), and they stay there until this unit's own tuples are written, an iteration on.
            nb = nk_max ? (nk_max + (u32)W - 1u) / (u32)W : 0u;
#pragma unroll
            for (int mm = 0; mm < LY::MROWS; ++mm) {
                u32 wv = 0;
                if ((u32)mm < nb && ok) {
                    wv = *reinterpret_cast<const LDSQ u32 *>(ldsq + LY::MASK + mm * 256 + lane * 4);
                    const int p0 = mm * W;  // position of bit 0
                    const int dl = (int)lo - p0, dh = (int)hi - p0;  // bits [dl, dh) stay
                    const u32 below = dl <= 0 ? 0u : dl >= 32 ? 0xffffffffu : ((1u << dl) - 1u);
                    const u32 upto = dh <= 0 ? 0u : dh >= 32 ? 0xffffffffu : ((1u << dh) - 1u);
                    wv &= upto & ~below;
                }
                rows[mm] = wv;
                cnt += (u32)__builtin_popcount(wv);
            }
            const u32 inclo = wave_incl_scan_u32(cnt, lane);
            exclo = inclo - cnt;
            TO = wave_bcast_u32(inclo, 63);
            if (lane == 0) lb_store(&utot[unit], (u64)TO | LB_AGG);  // counted.  (No fence: an agent-scope release writes the L2 back -- per unit, that WAS the kernel's time; the granule is the payload)
        }
        if (p_valid) {
            // The pending unit's place in the dense stream = the tuples of all units before it.  A unit-by-unit look-back is a serial chain
            // (92 ms for 2 10^9 bases); instead CHUNKS of 64 consecutive units: the unit reads the totals of the units before it IN ITS CHUNK
            // (one granule per lane); the chunk's last unit takes the chunk's place by a decoupled look-back over the CHUNKS (a chain 64 times
            // shorter) and the others walk the same granules without leaving a mark.
            u64 base;
            {
                const u32 c = p_unit >> 6, ui = p_unit & 63u, c0u = c << 6;
                const bool mine = (u32)lane < ui;
                u64 t = mine ? lb_load(&utot[c0u + (u32)lane]) : LB_AGG;
                while (__builtin_amdgcn_ballot_w64(!(t >> 62))) {
                    __builtin_amdgcn_s_sleep(1);
                    if (!(t >> 62)) t = lb_load(&utot[c0u + (u32)lane]);
                }
                const u32 pre = wave_bcast_u32(wave_incl_scan_u32(mine ? (u32)(t & 0xffffffffULL) : 0u, lane), 63);
                u64 cb = 0;
                if (ui == 63u || p_unit == a.nunits - 1u) cb = lookback_exclusive(lbc, c, (u64)pre + p_TO, lane);
                else if (c) cb = lookback_peek(lbc, c, lane);
                base = cb + pre;
            }
            const bool fits = base + p_TO <= a.cap;
            if (!fits && lane == 0) atomicOr(&a.ticket[1], 1u);
            u64 *const gh = a.hash + base;
            u32 *const gp = a.pos + base;
            if (p_TO && fits) {
                LDSQ unsigned short *const flat = reinterpret_cast<LDSQ unsigned short *>(ldsq + LY::FLAT);
                wave_sync_lds();  // (the selection rows of the unit just hashed have been read)
                *reinterpret_cast<LDSQ u32 *>(ldsq + LY::OWN + lane * 4) = p_shift;  // the tile's offset in its sequence
                {
                    LDSQ u32x4 *eb = reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::EBUF + lane * (LY::EST * 4));
#pragma unroll
                    for (int j = 0; j < LY::EST / 4; ++j) eb[j] = (u32x4){p_w[4 * j], p_w[4 * j + 1], p_w[4 * j + 2], p_w[4 * j + 3]};
                    *reinterpret_cast<LDSQ u32x4 *>(ldsq + LY::KTB + lane * 16) = ktb_row;
                }
                for (u32 c0 = 0; c0 < p_TO; c0 += (u32)LY::TCAP) {  // (one pass unless the unit selects more than the list holds: kilobases of short-period repeats)
                    const u32 tend = p_TO - c0 < (u32)LY::TCAP ? p_TO - c0 : (u32)LY::TCAP;
                    wave_sync_lds();  // (the pass before has read its list)
                    {   // expand: tuple exclo + j of the unit is (this lane, its j-th owned selection)
                        u32 at = p_exclo;
                        const u32 tag = (u32)lane << 9;
#pragma unroll
                        for (int mm = 0; mm < LY::MROWS; ++mm) {
                            u32 wv = (u32)mm < p_nb ? p_rows[mm] : 0u;
                            const u32 val0 = tag + (u32)(mm * W);
                            while (__builtin_amdgcn_ballot_w64(wv != 0u)) {
                                if (wv) {
                                    const u32 O = (u32)__builtin_ctz(wv);
                                    wv &= wv - 1u;
                                    if (at - c0 < (u32)LY::TCAP) flat[at - c0] = (unsigned short)(val0 + O);
                                    ++at;
                                }
                            }
                        }
                    }
                    wave_sync_lds();
                    auto request = [&](u32 tb, u32 &tp, u32 &q0, u32 &q1, u32 &q2, u32 &psh) {
                        const u32 tl = tb + (u32)lane;
                        tp = (u32)flat[tl < tend ? tl : tend - 1u];
                        const LDSQ u32 *const wp = reinterpret_cast<const LDSQ u32 *>(ldsq + LY::EBUF) + (tp >> 9) * (u32)LY::EST + ((tp & 0x1ffu) >> 4);
                        q0 = wp[0];
                        q1 = wp[1];
                        q2 = wp[2];
                        psh = *reinterpret_cast<const LDSQ u32 *>(ldsq + LY::OWN + (tp >> 9) * 4u);
                    };
                    u32 tp, q0, q1, q2, psh;
                    request(0u, tp, q0, q1, q2, psh);
                    for (u32 tb = 0; tb < tend; tb += 64) {
                        const u32 tl = tb + (u32)lane;
                        const bool lv = tl < tend;
                        const u32 idx = tp & 0x1ffu, o = tp >> 9, c0w = q0, c1w = q1, c2w = q2, sh = psh;
                        if (tb + 64u < tend) request(tb + 64u, tp, q0, q1, q2, psh);
                        const PfHash h = pf_hash_kmer<LY>(ldsq, o, idx, (u32)a.k, c0w, c1w, c2w);
                        const bool rev = h.rh < h.fh || (h.rh == h.fh && h.rl < h.fl);  // nthash returns rev only when strictly smaller
                        if (lv) {
                            __builtin_nontemporal_store(rev ? (((u64)h.rh << 32) | h.rl) : (((u64)h.fh << 32) | h.fl), &gh[c0 + tl]);
                            __builtin_nontemporal_store((idx + sh) | (rev ? BSK_POS_STRAND_BIT : 0u), &gp[c0 + tl]);
                        }
                    }
                }
                wave_sync_lds();  // (the next unit's selection rows lie where EBUF is)
            }
            if (p_stat & 0x100u) {
                const u64 pr = (u64)p_unit * 64 + lane;
                a.refs[pr] = ((base + p_exclo) << 24) | p_cnt;
                a.status[pr] = (u8)(p_stat & 0xffu);
            }
        }
        if (!live) break;
#pragma unroll
        for (int mm = 0; mm < LY::MROWS; ++mm) p_rows[mm] = rows[mm];
#pragma unroll
        for (int j = 0; j < LY::EST; ++j) p_w[j] = wr[j];
        p_exclo = exclo, p_TO = TO, p_cnt = cnt, p_shift = shift, p_nb = nb, p_unit = unit, p_stat = stat;
        p_valid = true;
        unit = unit_n1, unit_n1 = unit_n2, unit_n2 = pull();
    }
  }
}

#ifndef BSK_MINPFT_WS
#define BSK_MINPFT_WS(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#endif
#ifdef BSK_IMPL_MINPFT
bool pft_minimizer_supported(int w) {
#define X(WW) \
    if (w == WW) return true;
    BSK_MINPFT_WS(X)
#undef X
    return false;
}
u32 pft_minimizer_max_tile_bases() { return min_pft_max_tile_bases(); }
u32 pft_minimizer_mask_rows() { return (u32)MinPfLds::MROWS; }
u32 pft_minimizer_unit_tuples() { return (u32)MinPfLds::TCAP; }
size_t pft_minimizer_scratch_words(u32 nunits) { return (size_t)((nunits + 63u) >> 6) + (size_t)((nunits + 15u) & ~15u) + 8 * 16 + 16; }
int pft_minimizer_blocks_per_cu(int w) {
    int nb = 0;
    hipError_t e = hipErrorInvalidValue;
#define X(WW) \
    if (w == WW) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_minimizer_pft<WW>, 64, 0);
    BSK_MINPFT_WS(X)
#undef X
    if (e != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 1;
    }
    return nb;
}
void pft_minimizer_launch(int w, int grid, hipStream_t stream, const KArgs &a) {
#define X(WW) \
    if (w == WW) hipLaunchKernelGGL((k_minimizer_pft<WW>), dim3(grid), dim3(64), 0, stream, a);
    BSK_MINPFT_WS(X)
#undef X
}
#endif  // BSK_IMPL_MINPFT

}  // namespace bsk
