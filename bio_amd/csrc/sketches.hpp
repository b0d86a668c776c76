// sketches.hpp -- C++17 host-side mirror of the reference `sketches` package over the
// C ABI (include/biosketch.h).  Header only; links against libbiosketch.so.
//
// Same names, argument order and error behaviour as shenwei356/bio sketches/:
//   NewHashIterator / NextHash            iterator.go:615,658
//   NewKmerIterator / NextKmer            iterator.go:668,708
//   NewSimHashIterator / NextSimHash      iterator.go:113,191
//   NewMinimizerSketch / NextMinimizer    sketch.go:85,205
//   NewSyncmerSketch / NextSyncmer        sketch.go:142,312
//   NewProteinIterator / Next             iterator-protein.go:46,76
//   NewProteinMinimizerSketch / Next      sketch-protein.go:62,106
//   Index()                               iterator.go:776, sketch.go:488, ...
// Go returns (obj, err); here constructors return a std::unique_ptr and write the
// sentinel code to *err (bsk_err values 1..11 are the reference's sentinels).
// The device works on batches: Engine::batch() + Batch::run() is the fast path and
// Result::sketch(i) / iterator(i) hand out the same cursor types.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "biosketch.h"

namespace sketches {

constexpr int ErrInvalidK = BSK_ERR_INVALID_K, ErrEmptySeq = BSK_ERR_EMPTY_SEQ, ErrShortSeq = BSK_ERR_SHORT_SEQ,
              ErrIllegalBase = BSK_ERR_ILLEGAL_BASE, ErrKTooLarge = BSK_ERR_K_TOO_LARGE, ErrInvalidM = BSK_ERR_INVALID_M,
              ErrInvalidScale = BSK_ERR_INVALID_SCALE, ErrInvalidS = BSK_ERR_INVALID_S, ErrInvalidW = BSK_ERR_INVALID_W;

struct Seq {  // seq.Seq (seq/seq.go:29-34): only the alphabet identity and the bytes cross the boundary
    bool protein = false;
    std::string Seq_;
};

class Cursor {  // common part of Iterator / Sketch / ProteinIterator / ProteinMinimizerSketch
   public:
    bool next(uint64_t &code) {
        if (i_ >= codes_.size()) return false;
        code = codes_[i_];
        idx_ = pos_.empty() ? (long)i_ : (long)(pos_[i_] & BSK_POS_MASK);
        if (per_strand_) idx_ %= (long)per_strand_;
        strand_ = pos_.empty() ? 0 : (int)(pos_[i_] >> 31);
        ++i_;
        return true;
    }
    long Index() const { return idx_; }
    int Strand() const { return strand_; }    // extension: 1 iff the reverse-strand hash was canonical
    uint8_t Status() const { return status_; }  // extension: BSK_ST_* flags of the read
    std::vector<uint64_t> codes_;
    std::vector<uint32_t> pos_;
    uint8_t status_ = 0;
    size_t i_ = 0, per_strand_ = 0;
    long idx_ = -1;
    int strand_ = 0;
};

struct Iterator : Cursor {
    bool NextHash(uint64_t &c) { return next(c); }
    bool NextSimHash(uint64_t &c) { return next(c); }
    bool NextKmer(uint64_t &c, int *err) {  // iterator.go:708: (code, ok, err)
        if (err) *err = 0;
        if (next(c)) return true;
        if (err && (status_ & BSK_ST_CODE_MASK) == BSK_ST_ILLEGAL) *err = ErrIllegalBase;
        return false;
    }
};
struct Sketch : Cursor {
    bool NextMinimizer(uint64_t &c) { return next(c); }
    bool NextSyncmer(uint64_t &c) { return next(c); }
    bool Next(uint64_t &c) { return next(c); }
};
struct ProteinIterator : Cursor {
    bool Next(uint64_t &c) { return next(c); }
};
struct ProteinMinimizerSketch : Cursor {
    bool Next(uint64_t &c) { return next(c); }
};

class Engine;

class Result {
   public:
    std::vector<uint64_t> offsets, hash;
    std::vector<uint32_t> pos;
    std::vector<uint8_t> status;
    template <class C>
    std::unique_ptr<C> cursor(size_t i, int *err) const {
        if ((status[i] & BSK_ST_CODE_MASK) == BSK_ST_SHORT) {
            if (err) *err = ErrShortSeq;
            return nullptr;
        }
        if (err) *err = 0;
        auto c = std::make_unique<C>();
        c->codes_.assign(hash.begin() + offsets[i], hash.begin() + offsets[i + 1]);
        if (!pos.empty()) c->pos_.assign(pos.begin() + offsets[i], pos.begin() + offsets[i + 1]);
        c->status_ = status[i];
        return c;
    }
};

class Engine {
   public:
    explicit Engine(int device = 0) {
        int rc = bsk_ctx_create(device, &ctx_);
        if (rc != BSK_OK) throw std::runtime_error(std::string("bsk_ctx_create: ") + bsk_err_name(rc));
    }
    ~Engine() { bsk_ctx_destroy(ctx_); }
    Engine(const Engine &) = delete;
    // one launch over a batch of sequences; returns the sentinel / engine error code (0 = ok)
    int run(const std::vector<std::string_view> &seqs, bool protein, const bsk_params &p, Result &out) {
        std::vector<uint64_t> offs(seqs.size() + 1, 0);
        std::string bytes;
        for (size_t i = 0; i < seqs.size(); ++i) {
            bytes.append(seqs[i]);
            offs[i + 1] = bytes.size();
        }
        bsk_batch *b = nullptr;
        int rc = bsk_batch_from_ascii(ctx_, (const uint8_t *)bytes.data(), offs.data(), seqs.size(),
                                      protein ? BSK_ALPHA_PROTEIN : BSK_ALPHA_DNA, &b);
        if (rc != BSK_OK) return rc;
        bsk_result *r = nullptr;
        rc = bsk_sketch(ctx_, b, &p, &r);
        if (rc == BSK_OK) {
            uint64_t n = 0, t = 0;
            int hp = 0;
            bsk_result_info(r, &n, &t, &hp);
            out.offsets.assign(n + 1, 0);
            out.status.assign(n + 1, 0);
            out.hash.assign(t + 1, 0);
            out.pos.assign(hp ? t + 1 : 0, 0);
            rc = bsk_result_fetch(ctx_, r, 0, n, out.offsets.data(), out.status.data(), out.hash.data(),
                                  hp ? out.pos.data() : nullptr, t + 1);
            out.hash.resize(t);
            if (hp) out.pos.resize(t);
        }
        if (r) bsk_result_release(r);
        bsk_batch_destroy(b);
        return rc;
    }
    const char *last_error() const { return bsk_last_error(ctx_); }

   private:
    bsk_ctx *ctx_ = nullptr;
};

inline Engine &default_engine() {
    static Engine e(0);
    return e;
}

namespace detail {
template <class C>
std::unique_ptr<C> single(const Seq &s, const bsk_params &p, int *err, Engine *eng) {
    Result r;
    int rc = (eng ? *eng : default_engine()).run({std::string_view(s.Seq_)}, s.protein, p, r);
    if (rc != BSK_OK) {
        if (err) *err = rc;
        return nullptr;
    }
    return r.cursor<C>(0, err);
}
inline bsk_params params(int kind, int k) {
    bsk_params p{};
    p.kind = kind;
    p.k = k;
    p.canonical = 1;
    p.codon_table = 1;
    p.frame = 1;
    return p;
}
}  // namespace detail

inline std::unique_ptr<Iterator> NewHashIterator(const Seq &s, int k, bool canonical, bool circular, int *err, Engine *e = nullptr) {
    auto p = detail::params(BSK_NTHASH, k);
    p.canonical = canonical;
    p.circular = circular;
    return detail::single<Iterator>(s, p, err, e);
}
inline std::unique_ptr<Iterator> NewKmerIterator(const Seq &s, int k, bool canonical, bool circular, int *err, Engine *e = nullptr) {
    auto p = detail::params(BSK_KMER, k);
    p.canonical = canonical;
    p.circular = circular;
    auto it = detail::single<Iterator>(s, p, err, e);
    if (it && !canonical) it->per_strand_ = it->codes_.size() / 2;  // Index() restarts on the second strand (iterator.go:720)
    return it;
}
inline std::unique_ptr<Iterator> NewSimHashIterator(const Seq &s, int k, int m, int scale, bool canonical, bool circular, int *err,
                                                    Engine *e = nullptr) {
    auto p = detail::params(BSK_SIMHASH, k);
    p.m = m;
    p.scale = scale;
    p.canonical = canonical;
    p.circular = circular;
    return detail::single<Iterator>(s, p, err, e);
}
inline std::unique_ptr<Sketch> NewMinimizerSketch(const Seq &S, int k, int w, bool circular, int *err, Engine *e = nullptr) {
    auto p = detail::params(BSK_MINIMIZER, k);
    p.w = w;
    p.circular = circular;
    return detail::single<Sketch>(S, p, err, e);
}
inline std::unique_ptr<Sketch> NewSyncmerSketch(const Seq &S, int k, int s, bool circular, int *err, Engine *e = nullptr) {
    auto p = detail::params(BSK_SYNCMER, k);
    p.s = s;
    p.circular = circular;
    return detail::single<Sketch>(S, p, err, e);
}
inline std::unique_ptr<ProteinIterator> NewProteinIterator(const Seq &s, int k, int codonTable, int frame, int *err, Engine *e = nullptr) {
    auto p = detail::params(BSK_PROT_HASH, k);
    p.codon_table = codonTable;
    p.frame = frame;
    return detail::single<ProteinIterator>(s, p, err, e);
}
inline std::unique_ptr<ProteinMinimizerSketch> NewProteinMinimizerSketch(const Seq &S, int k, int codonTable, int frame, int w, int *err,
                                                                         Engine *e = nullptr) {
    if (k >= 1 && S.Seq_.size() < (size_t)k * 3) {  // upstream's order: k, then this length check (sketch-protein.go:66), only then w (:69)
        if (err) *err = BSK_ERR_SHORT_SEQ;
        return nullptr;
    }
    auto p = detail::params(BSK_PROT_MINIMIZER, k);
    p.w = w;
    p.codon_table = codonTable;
    p.frame = frame;
    return detail::single<ProteinMinimizerSketch>(S, p, err, e);
}

}  // namespace sketches
