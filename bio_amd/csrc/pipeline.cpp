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
#include <map>
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
// (w0: the chunk's words before this piece -- descriptors count from the chunk's first word)
static bool pack_chunk(const uint8_t *bytes, const uint64_t *offsets, uint64_t a, uint64_t m, uint64_t total_bytes, uint32_t *words, uint64_t *desc, uint64_t *n_words,
                       uint64_t w0 = 0) {
    uint64_t w = w0;
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
    uint64_t seq = 0, first_record = 0;  // place in the run's delivery order; index of the chunk's first record in its source
    int source_index = 0;
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
        c->packed = false;  // chunks come from ONE free queue shared by every source of the run: a chunk last filled by a host-packing
        c->n_words = 0;     // source (plain file, memory) must not keep its flag when the serial reader fills it with ASCII
        c->parts.clear();
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
        c->packed = false;
        static const bool no_pack = getenv("BSK_PIPE_NO_HOST_PACK") != nullptr;  // dev: every chunk the ASCII way
        if (c->alphabet == BSK_ALPHA_DNA && !no_pack && c->n) {  // pure ACGT chunks cross the link as 2-bit words (as MemorySource)
            const uint64_t max_words = c->nbytes / 16 + c->n + 2;
            if (c->bytes.ensure(max_words * 4 + 64) && c->offs.ensure((c->n + 1) * 8)) {
                uint64_t at = 0, w = 0;
                bool ok = true;
                for (auto *pc : c->parts) {
                    uint64_t n = 0;
                    const uint8_t *sb = nullptr;
                    const uint64_t *so = nullptr;
                    bsk_fastx_piece_data(pc, &n, &sb, &so);
                    if (!pack_chunk(sb, so, 0, n, so[n], (uint32_t *)c->bytes.p, (uint64_t *)c->offs.p + at, &w, w)) {
                        ok = false;
                        break;
                    }
                    at += n;
                }
                if (ok) {
                    for (auto *pc : c->parts) bsk_fastx_piece_release(f, pc);
                    c->parts.clear();
                    c->n_words = w;
                    c->packed = true;
                    return BSK_OK;
                }
            }
        }
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

// ---- the pipeline object behind bsk_pipeline_open_* / _next / _release / _close (include/biosketch.h) -----------------------------
// devices[n_dev]: the GPUs of the run.  Every device gets n_streams workers (a context = a HIP stream each); all workers take chunks from the one
// queue the producers fill, so a node's GPUs share one input the way the reference's workers share ChunkChan (seqio/fastx/reader.go:562-608) --
// reads are independent, nothing is exchanged between devices.  What the reference's ChunkChan hands its consumer -- every chunk, in input
// order -- is what bsk_pipeline_next hands out: a worker finishes its chunk into an OUTPUT BUFFER (pinned; a bounded pool), the buffers are
// delivered in the order the chunks were queued, and the consumer gives each one back with bsk_pipeline_release.
//   A worker takes its output buffer BEFORE it takes a chunk: chunks leave the queue in sequence order, so the lowest sequence number not
//   yet delivered is always held by a worker that already owns a buffer -- the bounded pool cannot dead-lock the ordered delivery.
template <class T>
struct PQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T *> q;
    bool closed = false;
    void push(T *c) {
        {
            std::lock_guard<std::mutex> l(m);
            q.push_back(c);
        }
        cv.notify_one();
    }
    T *pop() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return nullptr;
        T *c = q.front();
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

struct OutBuf {
    PinBuf off, st, hash, pos;
    bsk_chunk pub;
    uint64_t seq = 0;
};

}  // namespace

struct bsk_pipeline {
    std::vector<int> devices;
    int n_streams = 1, n_producers = 1, sink = BSK_SINK_TUPLES, sets_scale = 1, host_checksum = 0;
    uint64_t chunk_records = 0;
    bsk_params params{};
    std::unique_ptr<SourceSet> set;
    std::unique_ptr<Source> own_src;  // the single source of a memory / one-file run (the set refers to it)
    // machinery
    std::vector<Chunk> chunks;
    Queue free_q, full_q;
    std::vector<std::unique_ptr<OutBuf>> outs;
    PQueue<OutBuf> free_out;
    std::mutex om;  // ordered delivery
    std::condition_variable ocv;
    std::map<uint64_t, OutBuf *> ready;
    uint64_t next_seq = 0;
    int workers_alive = 0;
    std::mutex seqm;  // sequence numbers are given in queue order
    uint64_t seq_counter = 0;
    std::atomic<int> error{0};
    std::mutex stm;
    std::string errtext;
    std::unique_ptr<Slot[]> slots;
    std::atomic<int> next_source{0}, producers_left{0};
    std::vector<std::thread> producers, workers;
    clk::time_point t_start, t_last;
    uint64_t pin0 = 0;
    double reader_s = 0, reader_wait_s = 0;
    bsk_pipeline_stats acc{};
    bool joined = false;
    // the block-parallel reader's figures, filled in when its file closes
    int reader_threads = 0;
    uint64_t reparsed = 0;

    void fail(int code, const std::string &text) {
        int z = 0;
        if (error.compare_exchange_strong(z, code)) {
            std::lock_guard<std::mutex> l(stm);
            errtext = text;
        }
        full_q.close();
        free_q.close();
        free_out.close();
        {
            std::lock_guard<std::mutex> l(om);
        }
        ocv.notify_all();
    }
    void close_once(Slot *sl) {
        if (!sl->closed.exchange(true)) set->close(sl->src);
    }
    void producer_main();
    void worker_main(int w);
    int start();
    void join_all();
};

namespace {
inline uint64_t host_checksum_tuples(const bsk_chunk &c) {
    uint64_t sum = 0;
    const uint64_t *h = c.hash;
    if (c.offsets32) {
        const uint64_t T = c.offsets32[c.n_records];
        if (c.pos16)
            for (uint64_t j = 0; j < T; ++j) sum += h[j] * (2 * (uint64_t)(c.pos16[j] & BSK_POS16_MASK) + 1);
        else
            for (uint64_t r = 0; r < c.n_records; ++r)
                for (uint64_t j = c.offsets32[r]; j < c.offsets32[r + 1]; ++j) sum += h[j] * (2 * (j - c.offsets32[r]) + 1);
    } else {
        const uint64_t *oo = c.offsets64;
        if (c.pos32) {  // explicit positions: one flat pass (the term does not depend on the read)
            const uint64_t T = oo[c.n_records];
            for (uint64_t j = 0; j < T; ++j) sum += h[j] * (2 * (uint64_t)(c.pos32[j] & BSK_POS_MASK) + 1);
        } else {
            for (uint64_t r = 0; r < c.n_records; ++r)
                for (uint64_t j = oo[r]; j < oo[r + 1]; ++j) sum += h[j] * (2 * (j - oo[r]) + 1);
        }
    }
    return sum;
}
}  // namespace

void bsk_pipeline::producer_main() {
    (void)hipSetDevice(devices[0]);
    double rs = 0, ws = 0;
    for (;;) {
        const int si = next_source++;
        if (si >= set->count() || error.load()) break;
        int orc = BSK_OK;
        std::string otext;
        Source *src = set->open(si, &orc, &otext);
        if (!src) {
            fail(orc, otext);
            break;
        }
        Slot *sl = &slots[si];
        sl->src = src;
        uint64_t first_record = 0;
        for (;;) {
            const auto w0 = clk::now();
            Chunk *c = free_q.pop();
            ws += secs(w0, clk::now());
            if (!c || error.load()) break;
            const auto r0 = clk::now();
            c->packed = false;  // the free queue is shared by every source of the run: whatever the last source left in the chunk is not this one's
            c->n_words = 0;
            const int rc = src->next(c, chunk_records);
            rs += secs(r0, clk::now());
            if (rc <= 0) {
                if (rc < 0) fail(-rc, src->err);
                else free_q.push(c);  // the source is exhausted: the buffer goes back
                break;
            }
            c->slot = sl;
            c->alphabet = src->alphabet;
            c->source_index = si;
            c->first_record = first_record;
            first_record += c->n;
            sl->inflight++;
            {
                std::lock_guard<std::mutex> l(seqm);  // number and queue in one step: the workers take chunks in sequence order
                c->seq = seq_counter++;
                full_q.push(c);
            }
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
}

void bsk_pipeline::worker_main(int w) {
    const int n_dev = (int)devices.size();
    const int device = devices[(size_t)(w % n_dev)];
    bsk_ctx *ctx = nullptr;
    bsk_batch *batch = nullptr;
    bsk_result *res = nullptr;
    bsk_sets *sets = nullptr;
    bsk_pipeline_stats loc;
    memset(&loc, 0, sizeof loc);
    const bsk_params *p = &params;
    if (bsk_ctx_create(device, &ctx) != BSK_OK) {
        fail(BSK_ERR_NO_DEVICE, "bsk_ctx_create");
    } else {
        for (;;) {
            OutBuf *ob = free_out.pop();
            if (!ob || error.load()) break;
            Chunk *c = full_q.pop();
            if (!c || error.load()) {
                if (c) free_q.push(c);
                free_out.push(ob);
                break;
            }
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
            bsk_chunk &pub = ob->pub;
            memset(&pub, 0, sizeof pub);
            ob->seq = c->seq;
            pub.sequence = c->seq;
            pub.source_index = c->source_index;
            pub.first_record = c->first_record;
            pub.n_records = n;
            pub.n_bases = nb;
            pub.device = device;
            pub.sink = sink;
            pub.opaque = ob;
            free_q.push(c);  // the bytes are on the device: the producer may refill this buffer
            auto t1 = clk::now();
            loc.h2d_pack_seconds += secs(t0, t1);
            if (rc == BSK_OK) rc = bsk_sketch(ctx, batch, p, &res);
            auto t2 = clk::now();
            loc.kernel_seconds += secs(t1, t2);
            uint64_t nr = 0, nt = 0;
            int hp = 0;
            if (rc == BSK_OK) rc = bsk_result_info(res, &nr, &nt, &hp);
            pub.n_tuples = nt;
            pub.has_pos = hp;
            if (rc == BSK_OK && sink == BSK_SINK_TUPLES) {
                if (!ob->off.ensure((nr + 1) * 8) || !ob->st.ensure(nr + 1) || !ob->hash.ensure((nt + 1) * 8) || (hp && !ob->pos.ensure((nt + 1) * 4))) rc = BSK_ERR_NOMEM;
                // kinds with explicit positions leave through the narrow fetch (u32 offsets scanned on the device, u16 positions: 10 bytes
                // per tuple + 5 per read over the link instead of 12 + 17); reads of 32 768 bases or more fall back to the wide one
                static const bool wide_only = getenv("BSK_PIPE_WIDE_FETCH") != nullptr;
                bool narrow = hp && !wide_only && nt < (1ull << 32);
                if (rc == BSK_OK && narrow) {
                    std::unique_lock<std::mutex> lk(g_d2h_mutex[device & 15], std::defer_lock);
                    if (g_copy_locks) lk.lock();
                    rc = bsk_result_fetch_narrow(ctx, res, 0, nr, (uint32_t *)ob->off.p, (uint8_t *)ob->st.p, (uint64_t *)ob->hash.p, (uint16_t *)ob->pos.p, nt + 1, nullptr);
                    if (rc == BSK_ERR_UNSUPPORTED) {
                        narrow = false;
                        rc = BSK_OK;
                    }
                }
                if (rc == BSK_OK && !narrow) {
                    std::unique_lock<std::mutex> lk(g_d2h_mutex[device & 15], std::defer_lock);
                    if (g_copy_locks) lk.lock();
                    rc = bsk_result_fetch(ctx, res, 0, nr, (uint64_t *)ob->off.p, (uint8_t *)ob->st.p, (uint64_t *)ob->hash.p, hp ? (uint32_t *)ob->pos.p : nullptr, nt + 1);
                }
                pub.status = (const uint8_t *)ob->st.p;
                pub.hash = (const uint64_t *)ob->hash.p;
                if (narrow) {
                    pub.offsets32 = (const uint32_t *)ob->off.p;
                    pub.pos16 = (const uint16_t *)ob->pos.p;
                    pub.link_bytes = nr * 5 + nt * 10;
                } else {
                    pub.offsets64 = (const uint64_t *)ob->off.p;
                    pub.pos32 = hp ? (const uint32_t *)ob->pos.p : nullptr;
                    pub.link_bytes = nr * 9 + nt * (hp ? 12 : 8);
                }
                pub.n_values = nt;
                static const bool nodigest = getenv("BSK_PIPE_NO_DIGEST") != nullptr;  // dev: the run without the consumer stand-in (checksum stays 0)
                if (rc == BSK_OK && host_checksum && !nodigest) pub.checksum = host_checksum_tuples(pub);
            } else if (rc == BSK_OK && sink == BSK_SINK_SETS) {
                // the on-device reduction the reference's consumers do on the host (collect, sort, de-duplicate, FracMinHash filter:
                // iterator.go:181-185): only the distinct values that pass cross the link
                rc = bsk_result_sets_reuse(ctx, res, BSK_SETS_PER_SEQUENCE, sets_scale, &sets);
                uint64_t ns = 0, nv = 0;
                if (rc == BSK_OK) rc = bsk_sets_info(sets, &ns, &nv);
                if (rc == BSK_OK && (!ob->off.ensure((ns + 1) * 4) || !ob->st.ensure(nr + 1) || !ob->hash.ensure((nv + 1) * 8))) rc = BSK_ERR_NOMEM;
                if (rc == BSK_OK) {
                    std::unique_lock<std::mutex> lk(g_d2h_mutex[device & 15], std::defer_lock);
                    if (g_copy_locks) lk.lock();
                    rc = bsk_sets_fetch_narrow(ctx, sets, (uint32_t *)ob->off.p, (uint64_t *)ob->hash.p, nv + 1);
                    if (rc == BSK_OK) rc = bsk_result_fetch_status(ctx, res, 0, nr, (uint8_t *)ob->st.p);
                }
                pub.offsets32 = (const uint32_t *)ob->off.p;
                pub.status = (const uint8_t *)ob->st.p;
                pub.hash = (const uint64_t *)ob->hash.p;
                pub.n_values = nv;
                pub.link_bytes = nr * 5 + nv * 8;
                if (rc == BSK_OK && host_checksum) {
                    uint64_t sum = 0;
                    for (uint64_t j = 0; j < nv; ++j) sum += pub.hash[j];
                    pub.checksum = sum;
                }
            } else if (rc == BSK_OK) {  // BSK_SINK_COUNTS: nothing but counts and the device-side digest leave the device
                uint64_t ck = 0, ntt = 0;
                rc = bsk_result_digest(ctx, res, &ck, &ntt, nullptr);
                pub.checksum = ck;
            }
            loc.fetch_seconds += secs(t2, clk::now());
            if (rc != BSK_OK) {
                fail(rc, bsk_last_error(ctx));
                free_out.push(ob);
                break;
            }
            loc.records += n;
            loc.bases += nb;
            loc.tuples += nt;
            loc.chunks += 1;
            loc.checksum += pub.checksum;
            {
                std::lock_guard<std::mutex> l(om);
                ready[ob->seq] = ob;
            }
            ocv.notify_all();
        }
    }
    if (sets) bsk_sets_release(sets);
    bsk_result_release(res);
    bsk_batch_destroy(batch);
    if (ctx) bsk_ctx_destroy(ctx);
    {
        std::lock_guard<std::mutex> l(stm);
        acc.records += loc.records;
        acc.bases += loc.bases;
        acc.tuples += loc.tuples;
        acc.chunks += loc.chunks;
        acc.checksum += loc.checksum;
        acc.h2d_pack_seconds += loc.h2d_pack_seconds;
        acc.kernel_seconds += loc.kernel_seconds;
        acc.fetch_seconds += loc.fetch_seconds;
    }
    {
        std::lock_guard<std::mutex> l(om);
        --workers_alive;
    }
    ocv.notify_all();
}

int bsk_pipeline::start() {
    const int n_dev = (int)devices.size();
    for (int d = 0; d < n_dev; ++d)
        if (hipSetDevice(devices[(size_t)d]) != hipSuccess) return BSK_ERR_NO_DEVICE;
    const int n_workers = n_dev * n_streams;
    const int nchunks = 2 * n_workers + n_producers;  // double buffering per stream + the one every producer is filling
    chunks = std::vector<Chunk>((size_t)nchunks);
    for (auto &c : chunks) free_q.push(&c);
    const int nouts = 2 * n_workers + 2;  // one in work per stream, one waiting for its turn or held by the consumer, two to spare
    for (int i = 0; i < nouts; ++i) {
        outs.emplace_back(new OutBuf());
        free_out.push(outs.back().get());
    }
    slots.reset(new Slot[(size_t)std::max(1, set->count())]);
    t_start = t_last = clk::now();
    pin0 = g_pin_ns.load();
    producers_left = n_producers;
    workers_alive = n_workers;
    for (int pi = 0; pi < n_producers; ++pi) producers.emplace_back([this] { producer_main(); });
    for (int w = 0; w < n_workers; ++w) workers.emplace_back([this, w] { worker_main(w); });
    return BSK_OK;
}

void bsk_pipeline::join_all() {
    if (joined) return;
    joined = true;
    for (auto &t : producers) t.join();
    for (auto &t : workers) t.join();
    for (auto &c : chunks)
        if (c.slot && !c.slot->closed.load()) c.slot->src->discard(&c);
    for (int i = 0; i < set->count(); ++i)
        if (slots[i].src) close_once(&slots[i]);
}

extern "C" int bsk_pipeline_next(bsk_pipeline *pl, const bsk_chunk **chunk) {
    if (!pl || !chunk) return BSK_ERR_ARG;
    *chunk = nullptr;
    std::unique_lock<std::mutex> l(pl->om);
    pl->ocv.wait(l, [&] { return pl->ready.count(pl->next_seq) || pl->workers_alive == 0 || pl->error.load(); });
    if (pl->error.load()) return pl->error.load();
    auto it = pl->ready.find(pl->next_seq);
    if (it == pl->ready.end()) return BSK_OK;  // every worker has left and nothing is waiting: the end
    OutBuf *ob = it->second;
    pl->ready.erase(it);
    pl->next_seq++;
    pl->t_last = clk::now();
    *chunk = &ob->pub;
    return BSK_OK;
}

extern "C" int bsk_pipeline_release(bsk_pipeline *pl, const bsk_chunk *chunk) {
    if (!pl || !chunk || !chunk->opaque) return BSK_ERR_ARG;
    pl->free_out.push(static_cast<OutBuf *>(chunk->opaque));
    return BSK_OK;
}

extern "C" const char *bsk_pipeline_error(const bsk_pipeline *pl) {
    if (!pl) return "null pipeline";
    std::lock_guard<std::mutex> l(const_cast<bsk_pipeline *>(pl)->stm);
    return pl->errtext.c_str();
}

extern "C" int bsk_pipeline_close(bsk_pipeline *pl, bsk_pipeline_stats *st) {
    if (!pl) return BSK_ERR_ARG;
    bool drained;
    {
        std::lock_guard<std::mutex> l(pl->om);
        drained = pl->workers_alive == 0 && pl->ready.empty();
    }
    int rc = pl->error.load();
    if (!drained && !rc) {  // closed before the end: stop the threads (not an error of the run)
        int z = 0;
        pl->error.compare_exchange_strong(z, -1);  // BEFORE the queues close: a producer that finds its queue closed must see the run as stopped, not its source as exhausted
        pl->full_q.close();
        pl->free_q.close();
        pl->free_out.close();
    }
    pl->join_all();
    if (st) {
        *st = pl->acc;
        st->seconds = secs(pl->t_start, drained ? pl->t_last : clk::now());
        st->reader_seconds = pl->reader_s;
        st->reader_wait_seconds = pl->reader_wait_s;
        st->n_streams = (int32_t)(pl->devices.size() * (size_t)pl->n_streams);
        st->pin_seconds = (double)(g_pin_ns.load() - pl->pin0) * 1e-9;
        st->reader_threads = pl->reader_threads;
        st->reparsed_pieces = pl->reparsed;
    }
    pl->set.reset();
    pl->own_src.reset();
    delete pl;
    return rc;
}

extern "C" int bsk_pipeline_cancel(bsk_pipeline *pl) {
    if (!pl) return BSK_ERR_ARG;
    {
        std::lock_guard<std::mutex> l(pl->om);
        if (pl->workers_alive == 0 && pl->ready.empty()) return BSK_OK;  // the run has ended and everything was delivered: nothing to stop
    }
    int z = 0;
    pl->error.compare_exchange_strong(z, -1);
    pl->full_q.close();
    pl->free_q.close();
    pl->free_out.close();
    {
        std::lock_guard<std::mutex> l(pl->om);
    }
    pl->ocv.notify_all();  // a consumer blocked in bsk_pipeline_next returns -1
    return BSK_OK;
}

extern "C" int bsk_pipeline_run(bsk_pipeline *pl, bsk_chunk_fn on_chunk, void *user, bsk_pipeline_stats *stats) {
    if (!pl) return BSK_ERR_ARG;
    int rc = BSK_OK, crc = 0;
    for (;;) {
        const bsk_chunk *c = nullptr;
        rc = bsk_pipeline_next(pl, &c);
        if (rc != BSK_OK || !c) break;
        if (on_chunk) crc = on_chunk(user, c);
        bsk_pipeline_release(pl, c);
        if (crc) break;  // the consumer stops the run
    }
    const int close_rc = bsk_pipeline_close(pl, stats);
    if (crc) return BSK_ERR_STOPPED;  // the consumer's choice, not an error of the run (nor of its arguments)
    return rc != BSK_OK ? rc : close_rc;
}

namespace {
int check_config(const bsk_pipeline_config *cfg, const bsk_params *p) {
    if (!cfg || !p || !cfg->devices || cfg->n_devices < 1 || cfg->n_devices > 64 || cfg->n_streams < 1 || cfg->n_streams > 16) return BSK_ERR_ARG;
    if (cfg->sink != BSK_SINK_COUNTS && cfg->sink != BSK_SINK_TUPLES && cfg->sink != BSK_SINK_SETS) return BSK_ERR_ARG;
    if (cfg->sink == BSK_SINK_SETS && cfg->sets_scale < 0) return BSK_ERR_ARG;
    return BSK_OK;
}
bsk_pipeline *new_pipeline(const bsk_pipeline_config *cfg, const bsk_params *p) {
    auto *pl = new bsk_pipeline();
    pl->devices.assign(cfg->devices, cfg->devices + cfg->n_devices);
    pl->n_streams = cfg->n_streams;
    pl->chunk_records = cfg->chunk_records;
    pl->sink = cfg->sink;
    pl->sets_scale = cfg->sets_scale > 0 ? cfg->sets_scale : 1;
    pl->host_checksum = cfg->host_checksum;
    pl->params = *p;
    return pl;
}
int start_or_drop(bsk_pipeline *pl, bsk_pipeline **out) {
    const int rc = pl->start();
    if (rc != BSK_OK) {
        pl->set.reset();
        pl->own_src.reset();
        delete pl;
        return rc;
    }
    *out = pl;
    return BSK_OK;
}

// the statistics-only entry points of rounds 2-4, now one consumer loop over the pipeline object
int run_pipeline(bsk_pipeline *pl, bsk_pipeline_stats *st) { return bsk_pipeline_run(pl, nullptr, nullptr, st); }

// several files, n_readers of them read at once (each by its own producer thread: the block-parallel reader for a plain file, the serial
// one for a gzip file -- which is how gzip input scales: one zlib stream per file, several files)
struct FileSet : SourceSet {
    std::vector<std::string> paths;
    int want_alpha = -1, threads_per_file = 1;
    std::mutex m;
    uint64_t reparsed = 0;
    int par_files = 0;
    bsk_pipeline *owner = nullptr;
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
            if (orc != BSK_ERR_UNSUPPORTED) {  // (gzip / stdin: the serial reader below)
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
                if (owner) owner->reparsed = reparsed, owner->reader_threads = threads_per_file;
            }
            bsk_fastx_par_close(ps->f);
        } else if (auto *fs = dynamic_cast<FastxSource *>(s)) {
            bsk_fastx_close(fs->f);
        }
        delete s;
    }
};
}  // namespace

extern "C" int bsk_pipeline_open_fastx(const bsk_pipeline_config *cfg, const char *const *paths, int n_paths, const bsk_params *p, bsk_pipeline **out) {
    if (!out) return BSK_ERR_ARG;
    *out = nullptr;
    int rc = check_config(cfg, p);
    if (rc != BSK_OK || !paths || n_paths < 1 || cfg->n_readers < 0) return BSK_ERR_ARG;
    auto *set = new FileSet();
    for (int i = 0; i < n_paths; ++i) {
        if (!paths[i]) {
            delete set;
            return BSK_ERR_ARG;
        }
        set->paths.emplace_back(paths[i]);
    }
    set->want_alpha = cfg->alphabet;
    int n_readers = cfg->n_readers ? cfg->n_readers : std::min(n_paths, 8);
    n_readers = std::min(n_readers, std::min(n_paths, 64));
    const int n_workers = cfg->n_streams * cfg->n_devices;
    const char *tv = getenv("BSK_FASTX_THREADS");
    const int budget = tv && atoi(tv) > 0 ? atoi(tv) : std::max(1, (int)std::thread::hardware_concurrency() - n_workers - n_readers);
    set->threads_per_file = std::max(1, std::min(budget, 12) / n_readers);
    bsk_pipeline *pl = new_pipeline(cfg, p);
    set->owner = pl;
    pl->set.reset(set);
    pl->n_producers = n_readers;
    return start_or_drop(pl, out);
}

extern "C" int bsk_pipeline_open_memory(const bsk_pipeline_config *cfg, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int repeat,
                                        const bsk_params *p, bsk_pipeline **out) {
    if (!out) return BSK_ERR_ARG;
    *out = nullptr;
    if (check_config(cfg, p) != BSK_OK || !bytes || !offsets || !n || repeat < 1) return BSK_ERR_ARG;
    auto *src = new MemorySource();
    src->bytes = bytes;
    src->offsets = offsets;
    src->n = n;
    src->repeat = repeat;
    src->alphabet = cfg->alphabet < 0 ? BSK_ALPHA_DNA : cfg->alphabet;
    bsk_pipeline *pl = new_pipeline(cfg, p);
    pl->own_src.reset(src);
    pl->set.reset(new OneSource(src));
    pl->n_producers = 1;
    return start_or_drop(pl, out);
}

// ---- the statistics-only entry points (rounds 2-4): one consumer loop over the pipeline object, the digest folded by the workers ----
static int stats_run_files(const int *devices, int n_devices, const char *const *paths, int n_paths, int alphabet, const bsk_params *p, int n_streams, int n_readers,
                           uint64_t chunk_records, int fetch_tuples, bsk_pipeline_stats *stats) {
    if (!stats) return BSK_ERR_ARG;
    memset(stats, 0, sizeof *stats);
    bsk_pipeline_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.devices = devices;
    cfg.n_devices = n_devices;
    cfg.n_streams = n_streams;
    cfg.chunk_records = chunk_records;
    cfg.sink = fetch_tuples == 2 ? BSK_SINK_SETS : fetch_tuples ? BSK_SINK_TUPLES : BSK_SINK_COUNTS;
    cfg.sets_scale = 1;
    cfg.alphabet = alphabet;
    cfg.host_checksum = 1;
    cfg.n_readers = n_readers;
    bsk_pipeline *pl = nullptr;
    const int rc = bsk_pipeline_open_fastx(&cfg, paths, n_paths, p, &pl);
    if (rc != BSK_OK) return rc;
    return run_pipeline(pl, stats);
}

extern "C" int bsk_pipeline_fastx_multi(const int *devices, int n_devices, const char *path, int alphabet, const bsk_params *p, int n_streams,
                                        uint64_t chunk_records, int fetch_tuples, bsk_pipeline_stats *stats) {
    if (!path || !devices || n_devices < 1) return BSK_ERR_ARG;
    return stats_run_files(devices, n_devices, &path, 1, alphabet, p, n_streams, 1, chunk_records, fetch_tuples, stats);
}
extern "C" int bsk_pipeline_fastx(int device, const char *path, int alphabet, const bsk_params *p, int n_streams, uint64_t chunk_records,
                                  int fetch_tuples, bsk_pipeline_stats *stats) {
    return bsk_pipeline_fastx_multi(&device, 1, path, alphabet, p, n_streams, chunk_records, fetch_tuples, stats);
}

extern "C" int bsk_pipeline_memory_multi(const int *devices, int n_devices, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet,
                                         const bsk_params *p, int n_streams, uint64_t chunk_records, int repeat, int fetch_tuples, bsk_pipeline_stats *stats) {
    if (!bytes || !offsets || !n || repeat < 1 || !devices || n_devices < 1 || !stats) return BSK_ERR_ARG;
    memset(stats, 0, sizeof *stats);
    bsk_pipeline_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.devices = devices;
    cfg.n_devices = n_devices;
    cfg.n_streams = n_streams;
    cfg.chunk_records = chunk_records;
    cfg.sink = fetch_tuples == 2 ? BSK_SINK_SETS : fetch_tuples ? BSK_SINK_TUPLES : BSK_SINK_COUNTS;
    cfg.sets_scale = 1;
    cfg.alphabet = alphabet;
    cfg.host_checksum = 1;
    bsk_pipeline *pl = nullptr;
    const int rc = bsk_pipeline_open_memory(&cfg, bytes, offsets, n, repeat, p, &pl);
    if (rc != BSK_OK) return rc;
    return run_pipeline(pl, stats);
}
extern "C" int bsk_pipeline_memory(int device, const uint8_t *bytes, const uint64_t *offsets, uint64_t n, int alphabet, const bsk_params *p,
                                   int n_streams, uint64_t chunk_records, int repeat, int fetch_tuples, bsk_pipeline_stats *stats) {
    return bsk_pipeline_memory_multi(&device, 1, bytes, offsets, n, alphabet, p, n_streams, chunk_records, repeat, fetch_tuples, stats);
}

extern "C" int bsk_pipeline_fastx_files(int device, const char *const *paths, int n_paths, int alphabet, const bsk_params *p, int n_streams, int n_readers,
                                        uint64_t chunk_records, int fetch_tuples, bsk_pipeline_stats *stats) {
    if (!paths || n_paths < 1 || n_readers < 0) return BSK_ERR_ARG;
    return stats_run_files(&device, 1, paths, n_paths, alphabet, p, n_streams, n_readers, chunk_records, fetch_tuples, stats);
}

extern "C" void bsk_pipeline_trim(void) { pin_pool().trim(); }
