// pipeline.cpp -- the end-to-end path: file (or host memory) -> pinned chunks -> H2D + pack -> sketch kernel -> tuples on the host,
// with the stages of different chunks overlapping.
//
// This is host code that uses nothing but the public C ABI -- what the reference-side host (Go: a producer goroutine over
// fastx.Reader as ChunkChan is, seqio/fastx/reader.go:562-608, and one worker goroutine per stream) would write itself:
//   * ONE producer thread reads chunks (bsk_fastx_read_chunk) and copies them into pinned buffers from a free list;
//   * n_streams worker threads, each with its own context (= HIP stream), re-fill one batch object (bsk_batch_refill_ascii:
//     no allocation per chunk), run bsk_sketch into one re-used result and fetch the tuples into pinned memory
//     (bsk_result_fetch).  While one worker waits for its copy or kernel the others' streams run: H2D, kernels and D2H of
//     different chunks overlap without any cross-stream choreography.
// The statistics say where the time went; `seconds` is the wall time from the first read to the last tuple on the host.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "host_types.hpp"

namespace {
std::mutex g_h2d_mutex[16], g_d2h_mutex[16];  // per device: one copy per direction at a time (run_pipeline)
// Measured (scripts/perf_e2e.py, 3.2e7 reads from host memory, 3 / 5 / 8 streams x chunks of 2^18..2^20 records): 12.0 / 12.6 / 10.4 Gbases/s
// with the locks, 13.5 / 12.1 / 10.6 without -- the workers are bound by their own host-side work (the copy into pinned memory, the
// pass over the fetched tuples), not by interleaved copies.  Off unless BSK_PIPE_COPY_LOCKS is set.
const bool g_copy_locks = getenv("BSK_PIPE_COPY_LOCKS") != nullptr;


using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

std::atomic<uint64_t> g_pin_ns{0};  // time spent pinning / unpinning host memory (all threads), for the statistics

// Pinned buffers outlive a pipeline call: pinning costs ~0.35 ms per MB (0.2 s for the buffers of one three-stream run), so a
// finished run parks its buffers in a process-wide pool (per device, best fit, bounded) and the next run -- the next file of a
// long-lived host -- takes them from there.  bsk_pipeline_trim() gives the memory back.
struct PinPool {
    struct Blk {
        void *p;
        size_t cap;
        int dev;
    };
    std::mutex m;
    std::vector<Blk> blocks;
    size_t bytes = 0;
    static constexpr size_t LIMIT = (size_t)8 << 30;
    void *take(int dev, size_t want, size_t *cap) {
        std::lock_guard<std::mutex> l(m);
        int best = -1;
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].dev == dev && blocks[i].cap >= want && (best < 0 || blocks[i].cap < blocks[(size_t)best].cap)) best = (int)i;
        if (best < 0 || blocks[(size_t)best].cap > 2 * want + (1u << 20)) return nullptr;  // do not burn a large block on a small request
        Blk b = blocks[(size_t)best];
        blocks.erase(blocks.begin() + best);
        bytes -= b.cap;
        *cap = b.cap;
        return b.p;
    }
    bool give(int dev, void *p, size_t cap) {
        std::lock_guard<std::mutex> l(m);
        if (bytes + cap > LIMIT) return false;
        blocks.push_back({p, cap, dev});
        bytes += cap;
        return true;
    }
    void trim() {
        std::vector<Blk> old;
        {
            std::lock_guard<std::mutex> l(m);
            old.swap(blocks);
            bytes = 0;
        }
        for (auto &b : old) (void)hipHostFree(b.p);
    }
};
PinPool &pin_pool() {
    static PinPool *pool = new PinPool;  // never destroyed: the HIP runtime may be gone by the time static destructors run
    return *pool;
}

struct PinBuf {  // grow-only pinned host buffer
    void *p = nullptr;
    size_t cap = 0;
    int dev = 0;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;  // a copy would hand the same block to the pool twice
    PinBuf &operator=(const PinBuf &) = delete;
    void release() {
        if (p && !pin_pool().give(dev, p, cap)) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    bool ensure(size_t bytes) {
        if (cap >= bytes) return true;
        const auto t0 = clk::now();
        release();
        (void)hipGetDevice(&dev);
        const size_t want = bytes + bytes / 4 + 4096;
        bool ok = true;
        p = pin_pool().take(dev, want, &cap);
        if (!p) {
            ok = hipHostMalloc(&p, want) == hipSuccess;
            cap = ok ? want : 0;
            if (!ok) p = nullptr;
        }
        g_pin_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count();
        return ok;
    }
    ~PinBuf() { release(); }
};

// ---- 2-bit packing on the host (memory source): a chunk of pure ACGT reads crosses the link as 38 + 8 bytes per 150-base read instead
// of 150 + 8, and the device-side pack kernel is not needed.  A = 0, C = 1, G = 2, T = 3 (upper or lower case), 16 bases per u32, base i
// of a word at bits [2i, 2i + 2), every read on a word boundary: the layout of k_pack (biosketch.hip).  Any other byte: the chunk goes the
// ASCII way (the device keeps the ASCII of such reads for the side launch).
static inline int code_of(uint8_t c) {
    switch (c & 0xDF) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}
static bool pack_read_scalar(const uint8_t *src, uint64_t len, uint32_t *dst) {
    const uint64_t nw = (len + 15) / 16;
    for (uint64_t w = 0; w < nw; ++w) {
        uint32_t v = 0;
        const uint64_t i0 = w * 16, m = std::min<uint64_t>(16, len - i0);
        for (uint64_t j = 0; j < m; ++j) {
            const int c = code_of(src[i0 + j]);
            if (c < 0) return false;
            v |= (uint32_t)c << (2 * j);
        }
        dst[w] = v;
    }
    return true;
}
#if defined(__x86_64__)
#include <immintrin.h>
// 32 bases per step; `safe` = bytes that may be read from src (the caller guarantees len <= safe; the vector loop reads whole 32-byte
// blocks only while they lie inside it).  Writes ceil(len / 16) words, possibly one more zero word behind them (the next read's first).
__attribute__((target("avx2"))) static bool pack_read_avx2(const uint8_t *src, uint64_t len, uint64_t safe, uint32_t *dst) {
    const __m256i up = _mm256_set1_epi8((char)0xDF), three = _mm256_set1_epi8(3);
    const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
    const __m256i idx = _mm256_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
    const __m256i m1 = _mm256_set1_epi16(0x0401), m2 = _mm256_set1_epi32(0x00100001);
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    uint64_t i = 0;
    for (; i < len && i + 32 <= safe; i += 32) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i u = _mm256_and_si256(v, up);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)), _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
        const uint64_t rem = len - i;
        __m256i inside = _mm256_set1_epi8((char)0xFF);
        if (rem < 32) inside = _mm256_cmpgt_epi8(_mm256_set1_epi8((char)rem), idx);  // lanes < rem
        if ((uint32_t)_mm256_movemask_epi8(_mm256_or_si256(ok, _mm256_andnot_si256(inside, _mm256_set1_epi8((char)0xFF)))) != 0xFFFFFFFFu) return false;
        __m256i c = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), three);
        c = _mm256_and_si256(c, inside);
        const __m256i q = _mm256_madd_epi16(_mm256_maddubs_epi16(c, m1), m2);  // every 32-bit lane: one packed byte (four bases)
        const __m256i g = _mm256_shuffle_epi8(q, pick);
        dst[i / 16] = (uint32_t)_mm256_extract_epi32(g, 0);
        if (rem > 16) dst[i / 16 + 1] = (uint32_t)_mm256_extract_epi32(g, 4);
    }
    if (i < len) return pack_read_scalar(src + i, len - i, dst + i / 16);  // (the last bytes of the source: no room for a whole block)
    return true;
}
static bool have_avx2() {
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}
#endif
// reads [a, a + m) of (bytes, offsets) -> words / descriptors; false: a byte outside ACGTacgt (nothing usable was written)
static bool pack_chunk(const uint8_t *bytes, const uint64_t *offsets, uint64_t a, uint64_t m, uint64_t total_bytes, uint32_t *words, uint64_t *desc, uint64_t *n_words) {
    uint64_t w = 0;
    for (uint64_t r = 0; r < m; ++r) {
        const uint64_t b0 = offsets[a + r], len = offsets[a + r + 1] - b0;
        if (len >= (1ull << 24)) return false;
        bool ok;
#if defined(__x86_64__)
        if (have_avx2()) ok = pack_read_avx2(bytes + b0, len, total_bytes - b0, words + w);
        else
#endif
            ok = pack_read_scalar(bytes + b0, len, words + w);
        if (!ok) return false;
        desc[r] = (w << 24) | len;
        w += (len + 15) / 16;
    }
    *n_words = w;
    return true;
}

struct Chunk {
    PinBuf bytes, offs;
    uint64_t n = 0, nbytes = 0;
    uint64_t src_at = 0;  // memory source: first record of the chunk (the worker copies it into the pinned buffers itself)
    std::vector<bsk_fastx_piece *> parts;  // block-parallel file source: the parsed pieces of the chunk (copied by the worker)
    struct Slot *slot = nullptr;           // where the chunk came from (several files can be in flight)
    int alphabet = BSK_ALPHA_DNA;
    bool packed = false;   // bytes holds 2-bit packed words (16 bases per u32, every read on a word boundary), offs the descriptors
    uint64_t n_words = 0;
};

struct Queue {  // chunks handed from the producer to the workers, and back
    std::mutex m;
    std::condition_variable cv;
    std::deque<Chunk *> q;
    bool closed = false;
    void push(Chunk *c) {
        {
            std::lock_guard<std::mutex> l(m);
            q.push_back(c);
        }
        cv.notify_one();
    }
    Chunk *pop() {  // nullptr: closed and empty
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return nullptr;
        Chunk *c = q.front();
        q.pop_front();
        return c;
    }
    void close() {
        {
            std::lock_guard<std::mutex> l(m);
            closed = true;
        }
        cv.notify_all();
    }
};

struct Source {  // next chunk into c; returns 0 at the end, <0 on error
    virtual ~Source() {}
    virtual int next(Chunk *c, uint64_t max_records) = 0;
    virtual int materialize(Chunk *) { return BSK_OK; }  // worker side: whatever the producer left to do (runs in parallel)
    virtual void discard(Chunk *) {}                      // a chunk that was handed out but never materialised (a failed run)
    int alphabet = BSK_ALPHA_DNA;
    std::string err;
};

struct Slot {  // one source of a run: closed as soon as it is exhausted and its last chunk is on the device
    Source *src = nullptr;
    std::atomic<int> inflight{0};
    std::atomic<bool> finished{false}, closed{false};
};

struct FastxSource : Source {
    bsk_fastx *f = nullptr;
    int want_alpha = -1;
    int next(Chunk *c, uint64_t max_records) override {
        uint64_t n = 0;
        const uint8_t *sb = nullptr;
        const uint64_t *so = nullptr;
        const int rc = bsk_fastx_read_chunk(f, max_records, 0, &n, &sb, &so, nullptr, nullptr, nullptr);
        if (rc != BSK_OK) {
            err = bsk_fastx_error(f);
            return -rc;
        }
        if (n == 0) return 0;
        if (want_alpha < 0) {
            int isq = 0, a = -1;
            bsk_fastx_info(f, &isq, &a);
            alphabet = a < 0 ? BSK_ALPHA_UNLIMIT : a;  // the guessed alphabet (a nucleotide flavour or protein); "Unlimit" (-1) keeps its own pairing (seq.go:381-383: reversed, not complemented)
        } else {
            alphabet = want_alpha;
        }
        if (!c->bytes.ensure(so[n] + 1) || !c->offs.ensure((n + 1) * 8)) return -BSK_ERR_NOMEM;
        memcpy(c->bytes.p, sb, so[n]);
        memcpy(c->offs.p, so, (n + 1) * 8);
        c->n = n;
        c->nbytes = so[n];
        return 1;
    }
};

// plain files: n parser threads work ahead (bsk_fastx_par_*); the producer only strings pieces together, the workers copy them
struct ParFastxSource : Source {
    bsk_fastx_par *f = nullptr;
    int want_alpha = -1;
    int next(Chunk *c, uint64_t max_records) override {
        c->parts.clear();
        c->n = c->nbytes = 0;
        while (max_records == 0 || c->n < max_records) {
            bsk_fastx_piece *pc = nullptr;
            const int rc = bsk_fastx_par_next(f, &pc);
            if (rc != BSK_OK) {
                err = bsk_fastx_par_error(f);
                for (auto *q : c->parts) bsk_fastx_piece_release(f, q);
                c->parts.clear();
                return -rc;
            }
            if (!pc) break;
            uint64_t n = 0;
            const uint64_t *so = nullptr;
            bsk_fastx_piece_data(pc, &n, nullptr, &so);
            c->parts.push_back(pc);
            c->n += n;
            c->nbytes += so[n];
        }
        if (c->parts.empty()) return 0;
        if (want_alpha < 0) {
            int a = -1;
            bsk_fastx_par_info(f, nullptr, &a, nullptr);
            alphabet = a < 0 ? BSK_ALPHA_UNLIMIT : a;
        } else {
            alphabet = want_alpha;
        }
        return 1;
    }
    int materialize(Chunk *c) override {
        int rc = BSK_OK;
        if (!c->bytes.ensure(c->nbytes + 1) || !c->offs.ensure((c->n + 1) * 8)) rc = BSK_ERR_NOMEM;
        uint64_t *o = (uint64_t *)c->offs.p;
        uint64_t at = 0, nb = 0;
        for (auto *pc : c->parts) {
            uint64_t n = 0;
            const uint8_t *sb = nullptr;
            const uint64_t *so = nullptr;
            bsk_fastx_piece_data(pc, &n, &sb, &so);
            if (rc == BSK_OK) {
                memcpy((uint8_t *)c->bytes.p + nb, sb, so[n]);
                for (uint64_t i = 0; i < n; ++i) o[at + i] = nb + so[i];
                at += n;
                nb += so[n];
            }
            bsk_fastx_piece_release(f, pc);
        }
        c->parts.clear();
        if (rc == BSK_OK) o[at] = nb;
        return rc;
    }
    void discard(Chunk *c) override {
        for (auto *pc : c->parts) bsk_fastx_piece_release(f, pc);
        c->parts.clear();
    }
};

struct MemorySource : Source {
    const uint8_t *bytes = nullptr;
    const uint64_t *offsets = nullptr;
    uint64_t n = 0, at = 0;
    int repeat = 1, pass = 0;
    int next(Chunk *c, uint64_t max_records) override {
        if (at >= n) {
            if (++pass >= repeat) return 0;
            at = 0;
        }
        // the bytes are already in host memory: the producer only hands out ranges, the staging copy into pinned memory is the
        // workers' (one memcpy thread could not keep several streams busy: 6.7 GB/s against the 3 x 20 GB/s they can move)
        const uint64_t m = std::min<uint64_t>(max_records ? max_records : n, n - at);
        c->src_at = at;
        c->n = m;
        c->nbytes = offsets[at + m] - offsets[at];
        at += m;
        return 1;
    }
    int materialize(Chunk *c) override {
        const uint64_t a = c->src_at, m = c->n, b0 = offsets[a], nb = c->nbytes;
        c->packed = false;
        static const bool no_pack = getenv("BSK_PIPE_NO_HOST_PACK") != nullptr;  // dev: every chunk the ASCII way
        if (c->alphabet == BSK_ALPHA_DNA && !no_pack && m) {  // DNA: 2-bit words straight into the pinned buffer (a quarter of the bytes); other alphabets keep their own pairing rules with the batch
            const uint64_t max_words = nb / 16 + m + 2;
            if (!c->bytes.ensure(max_words * 4 + 64) || !c->offs.ensure((m + 1) * 8)) return BSK_ERR_NOMEM;
            if (pack_chunk(bytes, offsets, a, m, offsets[n], (uint32_t *)c->bytes.p, (uint64_t *)c->offs.p, &c->n_words)) {
                c->packed = true;
                return BSK_OK;
            }
        }
        if (!c->bytes.ensure(nb + 1) || !c->offs.ensure((m + 1) * 8)) return BSK_ERR_NOMEM;
        memcpy(c->bytes.p, bytes + b0, nb);
        uint64_t *o = (uint64_t *)c->offs.p;
        for (uint64_t i = 0; i <= m; ++i) o[i] = offsets[a + i] - b0;
        return BSK_OK;
    }
};

// The sources of a run are opened by the producer threads themselves, in order (`open(i)`: nullptr + *rc on failure) -- one file after
// the other with one producer, several files at once with several.
struct SourceSet {
    virtual ~SourceSet() {}
    virtual int count() const = 0;
    virtual Source *open(int i, int *rc, std::string *errtext) = 0;
    virtual void close(Source *s) = 0;
};
struct OneSource : SourceSet {
    Source *s;
    explicit OneSource(Source *s_) : s(s_) {}
    int count() const override { return 1; }
    Source *open(int, int *, std::string *) override { return s; }
    void close(Source *) override {}
};

// devices[n_dev]: the GPUs of the run.  Every device gets n_streams workers (a context = a HIP stream each); all workers take chunks from the one
// queue the producers fill, so a node's GPUs share one input the way the reference's workers share ChunkChan (seqio/fastx/reader.go:562-608) --
// reads are independent, nothing is exchanged between devices, and the order-independent digest of the statistics is the whole job's.
int run_pipeline(const int *devices, int n_dev, SourceSet &set, int n_producers, const bsk_params *p, int n_streams, uint64_t chunk_records, int fetch, bsk_pipeline_stats *st) {
    if (!p || !st || !devices || n_dev < 1 || n_dev > 64 || n_streams < 1 || n_streams > 16 || n_producers < 1 || n_producers > 64) return BSK_ERR_ARG;
    memset(st, 0, sizeof *st);
    for (int d = 0; d < n_dev; ++d)
        if (hipSetDevice(devices[d]) != hipSuccess) return BSK_ERR_NO_DEVICE;
    const int device = devices[0];  // (the producers' pinned buffers)
    const int n_workers = n_dev * n_streams;
    const int nchunks = 2 * n_workers + n_producers;  // double buffering per stream + the one every producer is filling
    std::vector<Chunk> chunks(nchunks);
    Queue free_q, full_q;
    for (auto &c : chunks) free_q.push(&c);
    std::atomic<int> error{0};
    std::mutex stm;
    std::string errtext;
    auto fail = [&](int code, const std::string &text) {
        int z = 0;
        if (error.compare_exchange_strong(z, code)) {
            std::lock_guard<std::mutex> l(stm);
            errtext = text;
        }
        full_q.close();
        free_q.close();
    };
    const auto t_start = clk::now();
    const uint64_t pin0 = g_pin_ns.load();
    double reader_s = 0, reader_wait_s = 0;
    std::atomic<int> next_source{0}, producers_left{n_producers};
    std::unique_ptr<Slot[]> slots(new Slot[(size_t)std::max(1, set.count())]);
    auto close_once = [&](Slot *sl) {
        if (!sl->closed.exchange(true)) set.close(sl->src);
    };
    std::vector<std::thread> producers;
    for (int pi = 0; pi < n_producers; ++pi) {
        producers.emplace_back([&] {
            (void)hipSetDevice(device);
            double rs = 0, ws = 0;
            for (;;) {
                const int si = next_source++;
                if (si >= set.count() || error.load()) break;
                int orc = BSK_OK;
                std::string otext;
                Source *src = set.open(si, &orc, &otext);
                if (!src) {
                    fail(orc, otext);
                    break;
                }
                Slot *sl = &slots[si];
                sl->src = src;
                for (;;) {
                    const auto w0 = clk::now();
                    Chunk *c = free_q.pop();
                    ws += secs(w0, clk::now());
                    if (!c || error.load()) break;
                    const auto r0 = clk::now();
                    const int rc = src->next(c, chunk_records);
                    rs += secs(r0, clk::now());
                    if (rc <= 0) {
                        if (rc < 0) fail(-rc, src->err);
                        else free_q.push(c);  // the source is exhausted: the buffer goes back
                        break;
                    }
                    c->slot = sl;
                    c->alphabet = src->alphabet;
                    sl->inflight++;
                    full_q.push(c);
                }
                if (error.load()) break;
                sl->finished = true;
                if (sl->inflight.load() == 0) close_once(sl);
            }
            {
                std::lock_guard<std::mutex> l(stm);
                reader_s += rs;
                reader_wait_s += ws;
            }
            if (--producers_left == 0) full_q.close();
        });
    }
    std::vector<std::thread> workers;
    for (int w = 0; w < n_workers; ++w) {
        workers.emplace_back([&, w] {
            const int device = devices[w % n_dev];  // (shadows the producers' device: the copy locks below are per device)
            bsk_ctx *ctx = nullptr;
            if (bsk_ctx_create(device, &ctx) != BSK_OK) {
                fail(BSK_ERR_NO_DEVICE, "bsk_ctx_create");
                return;
            }
            bsk_batch *batch = nullptr;
            bsk_result *res = nullptr;
            PinBuf o_off, o_st, o_hash, o_pos;
            bsk_pipeline_stats loc;
            memset(&loc, 0, sizeof loc);
            for (;;) {
                Chunk *c = full_q.pop();
                if (!c || error.load()) break;
                auto t0 = clk::now();
                Slot *sl = c->slot;
                int rc = sl->src->materialize(c);
                if (rc == BSK_OK) {
                    // ONE host-to-device copy in flight per device, and one device-to-host (below): the link moves 48 + 48 GB/s with one
                    // large-copy stream per direction and 17 + 31 .. 24 + 43 when n streams interleave 39-MB and 46 + 23-MB copies both
                    // ways (scripts/ubench/pcie.py).  The kernels and the host-side work of other chunks still overlap the copies.
                    std::unique_lock<std::mutex> lk(g_h2d_mutex[device & 15], std::defer_lock);
                    if (g_copy_locks) lk.lock();
                    if (c->packed) rc = bsk_batch_refill_packed(ctx, &batch, (const uint32_t *)c->bytes.p, c->n_words, (const uint64_t *)c->offs.p, c->n);
                    else rc = bsk_batch_refill_ascii(ctx, &batch, (const uint8_t *)c->bytes.p, (const uint64_t *)c->offs.p, c->n, c->alphabet);
                }
                c->slot = nullptr;
                if (--sl->inflight == 0 && sl->finished.load()) close_once(sl);
                const uint64_t n = c->n, nb = c->nbytes;
                free_q.push(c);  // the bytes are on the device: the producer may refill this buffer
                auto t1 = clk::now();
                loc.h2d_pack_seconds += secs(t0, t1);
                if (rc == BSK_OK) rc = bsk_sketch(ctx, batch, p, &res);
                auto t2 = clk::now();
                loc.kernel_seconds += secs(t1, t2);
                uint64_t nr = 0, nt = 0;
                int hp = 0;
                if (rc == BSK_OK) rc = bsk_result_info(res, &nr, &nt, &hp);
                if (rc == BSK_OK && fetch) {
                    if (!o_off.ensure((nr + 1) * 8) || !o_st.ensure(nr + 1) || !o_hash.ensure((nt + 1) * 8) || (hp && !o_pos.ensure((nt + 1) * 4))) rc = BSK_ERR_NOMEM;
                    // kinds with explicit positions leave through the narrow fetch (u32 offsets scanned on the device, u16 positions: 10 bytes
                    // per tuple + 5 per read over the link instead of 12 + 17); reads of 32 768 bases or more fall back to the wide one
                    static const bool wide_only = getenv("BSK_PIPE_WIDE_FETCH") != nullptr;
                    bool narrow = hp && !wide_only && nt < (1ull << 32);
                    if (rc == BSK_OK && narrow) {
                        std::unique_lock<std::mutex> lk(g_d2h_mutex[device & 15], std::defer_lock);
                        if (g_copy_locks) lk.lock();
                        rc = bsk_result_fetch_narrow(ctx, res, 0, nr, (uint32_t *)o_off.p, (uint8_t *)o_st.p, (uint64_t *)o_hash.p, (uint16_t *)o_pos.p, nt + 1, nullptr);
                        if (rc == BSK_ERR_UNSUPPORTED) {
                            narrow = false;
                            rc = BSK_OK;
                        }
                    }
                    if (rc == BSK_OK && !narrow) {
                        std::unique_lock<std::mutex> lk(g_d2h_mutex[device & 15], std::defer_lock);
                        if (g_copy_locks) lk.lock();
                        rc = bsk_result_fetch(ctx, res, 0, nr, (uint64_t *)o_off.p, (uint8_t *)o_st.p, (uint64_t *)o_hash.p, hp ? (uint32_t *)o_pos.p : nullptr, nt + 1);
                    }
                    static const bool nodigest = getenv("BSK_PIPE_NO_DIGEST") != nullptr;  // dev: the run without the consumer stand-in (checksum stays 0)
                    if (rc == BSK_OK && !nodigest && narrow) {
                        const uint64_t *h = (const uint64_t *)o_hash.p;
                        const uint16_t *ps = (const uint16_t *)o_pos.p;
                        const uint64_t T = ((const uint32_t *)o_off.p)[nr];
                        uint64_t sum = 0;
                        for (uint64_t j = 0; j < T; ++j) sum += h[j] * (2 * (uint64_t)(ps[j] & BSK_POS16_MASK) + 1);
                        loc.checksum += sum;
                    } else
                    if (rc == BSK_OK && !nodigest) {  // the caller's consumer would start here; the statistics keep an order-independent digest
                        const uint64_t *h = (const uint64_t *)o_hash.p, *oo = (const uint64_t *)o_off.p;
                        const uint32_t *ps = (const uint32_t *)o_pos.p;
                        uint64_t sum = 0;
                        if (hp) {  // explicit positions: one flat pass (the term does not depend on the read)
                            const uint64_t T = oo[nr];
                            for (uint64_t j = 0; j < T; ++j) sum += h[j] * (2 * (uint64_t)(ps[j] & BSK_POS_MASK) + 1);
                        } else {
                            for (uint64_t r = 0; r < nr; ++r)
                                for (uint64_t j = oo[r]; j < oo[r + 1]; ++j) sum += h[j] * (2 * (j - oo[r]) + 1);
                        }
                        loc.checksum += sum;
                    }
                } else if (rc == BSK_OK) {
                    uint64_t ck = 0, ntt = 0;
                    rc = bsk_result_digest(ctx, res, &ck, &ntt, nullptr);
                    loc.checksum += ck;
                }
                loc.fetch_seconds += secs(t2, clk::now());
                if (rc != BSK_OK) {
                    fail(rc, bsk_last_error(ctx));
                    break;
                }
                loc.records += n;
                loc.bases += nb;
                loc.tuples += nt;
                loc.chunks += 1;
            }
            bsk_result_release(res);
            bsk_batch_destroy(batch);
            bsk_ctx_destroy(ctx);
            std::lock_guard<std::mutex> l(stm);
            st->records += loc.records;
            st->bases += loc.bases;
            st->tuples += loc.tuples;
            st->chunks += loc.chunks;
            st->checksum += loc.checksum;
            st->h2d_pack_seconds += loc.h2d_pack_seconds;
            st->kernel_seconds += loc.kernel_seconds;
            st->fetch_seconds += loc.fetch_seconds;
        });
    }
    for (auto &t : producers) t.join();
    for (auto &t : workers) t.join();
    for (auto &c : chunks)
        if (c.slot && !c.slot->closed.load()) c.slot->src->discard(&c);
    for (int i = 0; i < set.count(); ++i)
        if (slots[i].src) close_once(&slots[i]);
    st->seconds = secs(t_start, clk::now());
    st->reader_seconds = reader_s;
    st->reader_wait_seconds = reader_wait_s;
    st->n_streams = n_workers;
    st->pin_seconds = (double)(g_pin_ns.load() - pin0) * 1e-9;
    return error.load();
}

}  // namespace

extern "C" int bsk_pipeline_fastx_multi(const int *devices, int n_devices, const char *path, int alphabet, const bsk_params *p, int n_streams,
                                        uint64_t chunk_records, int fetch_tuples, bsk_pipeline_stats *stats) {
    if (!path || !devices || n_devices < 1) return BSK_ERR_ARG;
    // a plain file: block-parallel parsing (the serial record reader delivers ~0.8 Gbases/s, a third of what ONE stream sketches)
    if (!getenv("BSK_FASTX_SERIAL")) {
        ParFastxSource ps;
        ps.want_alpha = alphabet;
        const char *tv = getenv("BSK_FASTX_THREADS");
        int nt = tv && atoi(tv) > 0 ? atoi(tv) : (int)std::thread::hardware_concurrency() - n_streams * n_devices - 1;
        nt = std::max(1, std::min(nt, 12));
        const int orc = bsk_fastx_par_open(path, nt, 0, &ps.f);
        if (orc == BSK_OK) {
            OneSource one(&ps);
            const int rc = run_pipeline(devices, n_devices, one, 1, p, n_streams, chunk_records, fetch_tuples, stats);
            uint64_t rep = 0;
            bsk_fastx_par_info(ps.f, nullptr, nullptr, &rep);
            if (stats) stats->reader_threads = nt, stats->reparsed_pieces = rep;
            bsk_fastx_par_close(ps.f);
            return rc;
        }
        if (orc != BSK_ERR_UNSUPPORTED) return orc;  // gzip / stdin: the serial reader below
    }
    FastxSource src;
    src.want_alpha = alphabet;
    int rc = bsk_fastx_open(path, &src.f);
    if (rc != BSK_OK) return rc;
    OneSource one(&src);
    rc = run_pipeline(devices, n_devices, one, 1, p, n_streams, chunk_records, fetch_tuples, stats);
    bsk_fastx_close(src.f);
    return rc;
}
extern "C" int bsk_pipeline_fastx(int device, const char *path, int alphabet, const bsk_params *p, int n_streams, uint64_t chunk_records,
                                  int fetch_tuples, bsk_pipeline_stats *stats) {
    return bsk_pipeline_fastx_multi(&device, 1, path, alphabet, p, n_streams, chunk_records, fetch_tuples, stats);
}

extern "C" int bsk_pipeline_memory_multi(const int *devices, int n_devices, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet,
                                         const bsk_params *p, int n_streams, uint64_t chunk_records, int repeat, int fetch_tuples, bsk_pipeline_stats *stats) {
    if (!bytes || !offsets || !n || repeat < 1 || !devices || n_devices < 1) return BSK_ERR_ARG;
    MemorySource src;
    src.bytes = bytes;
    src.offsets = offsets;
    src.n = n;
    src.repeat = repeat;
    src.alphabet = alphabet;
    OneSource one(&src);
    return run_pipeline(devices, n_devices, one, 1, p, n_streams, chunk_records, fetch_tuples, stats);
}
extern "C" int bsk_pipeline_memory(int device, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet, const bsk_params *p,
                                   int n_streams, uint64_t chunk_records, int repeat, int fetch_tuples, bsk_pipeline_stats *stats) {
    return bsk_pipeline_memory_multi(&device, 1, bytes, offsets, n, alphabet, p, n_streams, chunk_records, repeat, fetch_tuples, stats);
}

// several files, n_readers of them read at once (each by its own producer thread: the block-parallel reader for a plain file, the serial
// one for a gzip file -- which is how gzip input scales: one zlib stream per file, several files)
namespace {
struct FileSet : SourceSet {
    std::vector<std::string> paths;
    int want_alpha = -1, threads_per_file = 1;
    std::mutex m;
    uint64_t reparsed = 0;
    int par_files = 0;
    int count() const override { return (int)paths.size(); }
    Source *open(int i, int *rc, std::string *errtext) override {
        if (!getenv("BSK_FASTX_SERIAL")) {
            auto *ps = new ParFastxSource();
            ps->want_alpha = want_alpha;
            const int orc = bsk_fastx_par_open(paths[i].c_str(), threads_per_file, 0, &ps->f);
            if (orc == BSK_OK) {
                std::lock_guard<std::mutex> l(m);
                par_files++;
                return ps;
            }
            delete ps;
            if (orc != BSK_ERR_UNSUPPORTED) {
                *rc = orc;
                *errtext = "cannot read " + paths[i];
                return nullptr;
            }
        }
        auto *fs = new FastxSource();
        fs->want_alpha = want_alpha;
        const int orc = bsk_fastx_open(paths[i].c_str(), &fs->f);
        if (orc != BSK_OK) {
            delete fs;
            *rc = orc;
            *errtext = "cannot open " + paths[i];
            return nullptr;
        }
        return fs;
    }
    void close(Source *s) override {
        if (auto *ps = dynamic_cast<ParFastxSource *>(s)) {
            uint64_t rep = 0;
            bsk_fastx_par_info(ps->f, nullptr, nullptr, &rep);
            {
                std::lock_guard<std::mutex> l(m);
                reparsed += rep;
            }
            bsk_fastx_par_close(ps->f);
        } else if (auto *fs = dynamic_cast<FastxSource *>(s)) {
            bsk_fastx_close(fs->f);
        }
        delete s;
    }
};
}  // namespace

extern "C" int bsk_pipeline_fastx_files(int device, const char *const *paths, int n_paths, int alphabet, const bsk_params *p, int n_streams, int n_readers,
                                        uint64_t chunk_records, int fetch_tuples, bsk_pipeline_stats *stats) {
    if (!paths || n_paths < 1 || n_readers < 0) return BSK_ERR_ARG;
    FileSet set;
    for (int i = 0; i < n_paths; ++i) {
        if (!paths[i]) return BSK_ERR_ARG;
        set.paths.emplace_back(paths[i]);
    }
    set.want_alpha = alphabet;
    if (n_readers == 0) n_readers = std::min(n_paths, 8);
    n_readers = std::min(n_readers, std::min(n_paths, 64));
    const char *tv = getenv("BSK_FASTX_THREADS");
    const int budget = tv && atoi(tv) > 0 ? atoi(tv) : std::max(1, (int)std::thread::hardware_concurrency() - n_streams - n_readers);
    set.threads_per_file = std::max(1, std::min(budget, 12) / n_readers);
    const int rc = run_pipeline(&device, 1, set, n_readers, p, n_streams, chunk_records, fetch_tuples, stats);
    if (stats) {
        stats->reader_threads = set.par_files ? set.threads_per_file : 0;
        stats->reparsed_pieces = set.reparsed;
    }
    return rc;
}

extern "C" void bsk_pipeline_trim(void) { pin_pool().trim(); }
