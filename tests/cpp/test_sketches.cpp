// C++ mirror of the reference's sketches/sketch_test.go + iterator_test.go, driven through
// bio_amd/csrc/sketches.hpp -> C ABI -> GPU.  Built by __graft_entry__.build(), run by tests/test_gpu_cpp_mirror.py.
#include <cstdio>
#include <string>
#include <vector>

#include "sketches.hpp"
using namespace sketches;

static int fails = 0;
#define CHECK(c)                                                    \
    do {                                                            \
        if (!(c)) {                                                 \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            ++fails;                                                \
        }                                                           \
    } while (0)

int main() {
    int err = 0;
    {  // TestMinimizer sketch_test.go:33-76
        Seq s{false, "GGCAAGTTCGTCA"};
        auto sk = NewMinimizerSketch(s, 5, 3, false, &err);
        CHECK(sk && err == 0);
        std::vector<uint64_t> codes;
        std::vector<long> idx;
        uint64_t c;
        while (sk && sk->NextMinimizer(c)) {
            codes.push_back(c);
            idx.push_back(sk->Index());
        }
        const std::vector<uint64_t> want = {973456138564179607ULL, 2645801399420473919ULL, 1099502864234245338ULL,
                                            6763474888237448943ULL, 2737971715116251183ULL};
        CHECK(codes == want);
        CHECK((idx == std::vector<long>{0, 1, 4, 7, 8}));
    }
    {  // TestSyncmer sketch_test.go:78-117 (the reference asserts nothing; the commented-out expectation is 5 codes)
        Seq s{false, "GGCAAGTTCGTCATCGATC"};
        auto sk = NewSyncmerSketch(s, 5, 2, false, &err);
        CHECK(sk && err == 0);
        int n = 0;
        uint64_t c;
        while (sk && sk->NextSyncmer(c)) ++n;
        CHECK(n == 5);
    }
    const std::string s100 =
        "AAGTTTGAATCATTCAACTATCTAGTTTTCAGAGAACAATGTTCTCTAAAGAATAGAAAAGAGTCATTGTGCGGTGATGATGGCGGGAAGGATCCACCTG";
    {  // TestKmerIterator / TestHashIterator iterator_test.go:31-103
        Seq s{false, s100};
        auto it = NewKmerIterator(s, 10, true, false, &err);
        CHECK(it && err == 0);
        size_t n = 0;
        uint64_t c;
        int e2 = 0;
        while (it && it->NextKmer(c, &e2)) ++n;
        CHECK(e2 == 0 && n == s100.size() - 10 + 1);
        auto ih = NewHashIterator(s, 10, true, false, &err);
        CHECK(ih && err == 0);
        n = 0;
        while (ih && ih->NextHash(c)) ++n;
        CHECK(n == s100.size() - 10 + 1);
    }
    for (const char *q : {"GAACAATGTTCTCTAAAATTG", "GcACAATGTTCTCTAAAATTG"}) {  // TestSimHashIterator iterator_test.go:105-145
        Seq s{false, q};
        auto it = NewSimHashIterator(s, 21, 5, 5, true, false, &err);
        CHECK(it && err == 0);
        size_t n = 0;
        uint64_t c;
        while (it && it->NextSimHash(c)) ++n;
        CHECK(n == 1);
    }
    {  // error surface: constructors return the reference's sentinels
        Seq s{false, "ACGTACG"};
        CHECK(!NewMinimizerSketch(s, 5, 4, false, &err) && err == ErrShortSeq);
        CHECK(!NewMinimizerSketch(s, 0, 4, false, &err) && err == ErrInvalidK);
        CHECK(!NewMinimizerSketch(s, 5, 0, false, &err) && err == ErrInvalidW);
        CHECK(!NewSyncmerSketch(s, 5, 6, false, &err) && err == ErrInvalidS);
        CHECK(!NewKmerIterator(s, 33, true, false, &err) && err == ErrKTooLarge);
    }
    {  // protein (amino-acid input)
        Seq s{true, "ACDEFGHIKLMNPQRSTVWYACDEFGHIKLMNPQRSTVWY"};
        auto it = NewProteinIterator(s, 9, 1, 1, &err);
        CHECK(it && err == 0);
        size_t n = 0;
        uint64_t c;
        while (it && it->Next(c)) ++n;
        CHECK(n == 40 - 9 + 1);
        auto sk = NewProteinMinimizerSketch(s, 9, 1, 1, 5, &err);
        CHECK(sk && err == 0);
    }
    {  // TestProteinIterator iterator-protein_test.go:29-63 / TestProteinMinimizer sketch-protein_test.go:29-58: DNA input
        const std::string dna = "AAGTTTGAATCATTCAACTATCTAGTTTTCAGAGAACAATGTTCTCTAAAGAATAGAAAAGAGTCATTGTGCGGTGATGATGGCGGGAAGGATCCACCTG";
        Seq s{false, dna};
        const int k = 10;
        auto it = NewProteinIterator(s, k, 1, 1, &err);
        CHECK(it && err == 0);
        size_t n = 0;
        uint64_t c;
        while (it && it->Next(c)) ++n;
        CHECK(n == dna.size() / 3 - k + 1);  // iterator-protein_test.go:59
        auto sk = NewProteinMinimizerSketch(s, k, 1, 1, 3, &err);
        CHECK(sk && err == 0);
        size_t m = 0;
        while (sk && sk->Next(c)) ++m;
        CHECK(m >= 1 && m <= n);
        Seq tiny{false, dna.substr(0, 3 * k - 1)};
        CHECK(!NewProteinIterator(tiny, k, 1, 1, &err) && err == ErrShortSeq);  // iterator-protein.go:50
    }
    {  // plain C ABI, file to sketch sets: FASTQ -> bsk_fastx -> device batch -> minimizers -> sorted distinct hashes
        const char *path = "/tmp/bsk_cpp_test.fq";
        FILE *f = std::fopen(path, "w");
        CHECK(f != nullptr);
        const std::string r1 = "AAGTTTGAATCATTCAACTATCTAGTTTTCAGAGAACAATGTTCTCTAAAGAATAGAAAAGAGTCATTGTGCGGTGATGATGGCGGGAAGGATCCACCTG";
        if (f) {
            std::fprintf(f, "@r1 first\n%s\n+\n%s\n@r2\n%s\n+\n%s\n@r3\nACGT\n+\n@@@@\n", r1.c_str(), std::string(r1.size(), 'I').c_str(),
                         r1.c_str(), std::string(r1.size(), '@').c_str());
            std::fclose(f);
        }
        bsk_ctx *ctx = nullptr;
        CHECK(bsk_ctx_create(0, &ctx) == BSK_OK);
        bsk_fastx *fx = nullptr;
        CHECK(bsk_fastx_open(path, &fx) == BSK_OK);
        bsk_batch *b = nullptr;
        uint64_t nrec = 0;
        CHECK(bsk_batch_from_fastx(ctx, fx, 0, 0, -1, &b, &nrec) == BSK_OK && nrec == 3);
        int isq = 0, alpha = -1;
        CHECK(bsk_fastx_info(fx, &isq, &alpha) == BSK_OK && isq == 1 && alpha == BSK_ALPHA_DNA_PLAIN);
        bsk_params p{};
        p.kind = BSK_MINIMIZER;
        p.k = 21;
        p.w = 11;
        p.canonical = 1;
        bsk_result *res = nullptr;
        CHECK(bsk_sketch(ctx, b, &p, &res) == BSK_OK);
        bsk_sets *sets = nullptr;
        CHECK(bsk_result_sets(ctx, res, BSK_SETS_PER_SEQUENCE, 1, &sets) == BSK_OK);
        uint64_t ns = 0, nv = 0;
        CHECK(bsk_sets_info(sets, &ns, &nv) == BSK_OK && ns == 3 && nv > 0);
        std::vector<uint64_t> offs(ns + 1), vals(nv + 1);
        CHECK(bsk_sets_fetch(ctx, sets, 0, ns, offs.data(), vals.data(), nv + 1) == BSK_OK);
        CHECK(offs[1] - offs[0] == offs[2] - offs[1] && offs[3] == offs[2]);  // r1 == r2, r3 is too short
        bool same = true, sorted = true;
        for (uint64_t i = 0; i < offs[1]; ++i) {
            same = same && vals[i] == vals[offs[1] + i];
            sorted = sorted && (i == 0 || vals[i - 1] < vals[i]);
        }
        CHECK(same && sorted);
        bsk_sets_release(sets);
        bsk_result_release(res);
        bsk_batch_destroy(b);
        uint64_t again = 1;
        CHECK(bsk_batch_from_fastx(ctx, fx, 0, 0, -1, &b, &again) == BSK_OK && again == 0);  // end of file
        bsk_fastx_close(fx);
        bsk_ctx_destroy(ctx);
        std::remove(path);
    }
    {  // the exact C-ABI call sequences of the Go shim (bindings/go/sketches/engine.go), one block per shim function
        bsk_ctx *ctx = nullptr;
        CHECK(bsk_ctx_create(0, &ctx) == BSK_OK);  // NewEngine
        const std::string dna = "AAGTTTGAATCATTCAACTATCTAGTTTTCAGAGAACAATGTTCTCTAAAGAATAGAAAAGAGTCATTGTGCGGTGATGATGGCGGGAAGGATCCACCTG";
        const std::string shortr = "ACGTACGTAC";
        // Engine.NewBatchFromSeqs: one malloc'd byte buffer + offsets[n+1], bsk_batch_from_ascii, buffers freed after the call
        std::string bytes = dna + shortr + dna;
        std::vector<uint64_t> offs = {0, dna.size(), dna.size() + shortr.size(), 2 * dna.size() + shortr.size()};
        bsk_batch *b = nullptr;
        CHECK(bsk_batch_from_ascii(ctx, (const uint8_t *)bytes.data(), offs.data(), 3, BSK_ALPHA_DNA, &b) == BSK_OK);
        bytes.assign(bytes.size(), 'x');  // the shim frees its C buffer right after: the library must not look at it again
        // Batch.run (every *Iterators / *Sketches method): bsk_sketch, bsk_result_info, bsk_result_fetch(cap = n_tuples + 1), release
        auto run = [&](bsk_batch *bb, bsk_params p, std::vector<uint64_t> &o, std::vector<uint8_t> &st, std::vector<uint64_t> &h,
                       std::vector<uint32_t> &pos) {
            bsk_result *r = nullptr;
            if (bsk_sketch(ctx, bb, &p, &r) != BSK_OK) return false;
            uint64_t nr = 0, nt = 0;
            int hp = 0;
            bsk_result_info(r, &nr, &nt, &hp);
            o.assign(nr + 1, 0);
            st.assign(nr + 1, 0);
            h.assign(nt + 1, 0);
            pos.assign(hp ? nt + 1 : 0, 0);
            const int rc = bsk_result_fetch(ctx, r, 0, nr, o.data(), st.data(), h.data(), hp ? pos.data() : nullptr, nt + 1);
            bsk_result_release(r);
            return rc == BSK_OK && o[nr] == nt;
        };
        std::vector<uint64_t> o, h, o2, h2;
        std::vector<uint8_t> st, st2;
        std::vector<uint32_t> pos, pos2;
        bsk_params pm{};
        pm.kind = BSK_MINIMIZER;
        pm.k = 21;
        pm.w = 11;
        pm.canonical = 1;
        CHECK(run(b, pm, o, st, h, pos));  // Batch.MinimizerSketches
        CHECK((st[1] & BSK_ST_CODE_MASK) == BSK_ST_SHORT && o[1] == o[2]);  // Result.slice -> ErrShortSeq for record 1
        CHECK(o[1] - o[0] == o[3] - o[2] && o[1] > 0 && pos.size() == h.size());
        // Batch.ProteinMinimizerSketches on a DNA batch == Batch.Translate + the same call on the translated batch
        bsk_params pp{};
        pp.kind = BSK_PROT_MINIMIZER;
        pp.k = 9;
        pp.w = 3;
        pp.codon_table = 1;
        pp.frame = 1;
        CHECK(run(b, pp, o, st, h, pos));
        bsk_batch *t = nullptr;
        CHECK(bsk_batch_translate(ctx, b, 1, 1, &t) == BSK_OK);  // Batch.Translate (the result is a protein batch with a finalizer)
        uint64_t tn = 0, tb = 0;
        CHECK(bsk_batch_info(t, &tn, &tb, nullptr, nullptr) == BSK_OK && tn == 3 && tb == 2 * (dna.size() / 3) + shortr.size() / 3);
        CHECK(run(t, pp, o2, st2, h2, pos2));
        CHECK(h == h2 && pos == pos2 && o[1] == o2[1] && o[3] == o2[3]);
        pp.kind = BSK_PROT_HASH;  // Batch.ProteinIterators
        CHECK(run(b, pp, o, st, h, pos) && pos.empty() && o[1] == dna.size() / 3 - 9 + 1);
        // JoinEngines / GatherCounts: bsk_comm_init_all over the engines' contexts, bsk_gather_counts_all after the work
        bsk_ctx *ctxs[1] = {ctx};
        const uint64_t mine[5] = {3, dna.size() * 2 + shortr.size(), o[3], 0, 0};
        uint64_t all[5] = {0};
        CHECK(bsk_comm_init_all(ctxs, 1) == BSK_OK);
        CHECK(bsk_gather_counts_all(ctxs, 1, mine, 5, all) == BSK_OK && all[0] == 3 && all[2] == o[3]);
        bsk_batch_destroy(t);
        bsk_batch_destroy(b);
        bsk_ctx_destroy(ctx);  // also drops the communicator
    }
    std::printf(fails ? "FAILED %d checks\n" : "all C++ mirror checks passed\n", fails);
    return fails ? 1 : 0;
}
