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
        CHECK(bsk_fastx_info(fx, &isq, &alpha) == BSK_OK && isq == 1 && alpha == BSK_ALPHA_DNA);
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
    std::printf(fails ? "FAILED %d checks\n" : "all C++ mirror checks passed\n", fails);
    return fails ? 1 : 0;
}
