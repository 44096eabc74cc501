// format_host -- CPU harness of sambamba_amd/csrc/format_core.hpp (the row emitter of K6, `__host__ __device__`): the same statements
// the device runs, checked against the C library.
//   * dec4 for every x < 10000, n_digits32 / n_digits4 / n_digits64 at and around every power of ten;
//   * RowSink: random rows `name \t pos \t cov \t a \t c \t g \t t \t del \t refskip [\t sample] [\t y|n] \n` written at every start
//     alignment into a guarded buffer, compared with snprintf's text; the bytes around the row must stay untouched.
// Prints "ok <rows>" or the first difference; exit status 0 / 1.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../sambamba_amd/csrc/format_core.hpp"

using namespace sbx::fmt;

static int fail(const char* what, uint64_t v) { printf("FAIL %s at %" PRIu64 "\n", what, v); return 1; }

int main(int argc, char** argv) {
    const uint64_t n_rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 200000;
    for (uint32_t x = 0; x < 10000; ++x) {
        char want[8];
        snprintf(want, sizeof want, "%04u", x);
        const uint32_t got = dec4(x);
        if (memcmp(&got, want, 4) != 0) return fail("dec4", x);
    }
    {
        uint64_t p = 1;
        for (int k = 0; k < 20; ++k, p *= 10) {
            for (uint64_t v : {p - 1, p, p + 1, p * 9, p * 9 + 8}) {
                char tmp[32];
                const uint32_t want = (uint32_t)snprintf(tmp, sizeof tmp, "%" PRIu64, v);
                if (n_digits64(v) != want) return fail("n_digits64", v);
                if (v <= 0xFFFFFFFFull && n_digits32((uint32_t)v) != want) return fail("n_digits32", v);
                if (v < 10000 && n_digits4((uint32_t)v) != want) return fail("n_digits4", v);
            }
        }
        if (n_digits32(0) != 1 || n_digits32(0xFFFFFFFFu) != 10 || n_digits64(~0ull) != 20) return fail("n_digits extremes", 0);
    }
    std::mt19937_64 rng(0x5A4D0006);
    auto pick = [&](int cls) -> uint64_t {      // a number of a given magnitude class
        switch (cls) {
            case 0: return rng() % 10;
            case 1: return rng() % 10000;
            case 2: return rng() % 100000000ull;
            case 3: return rng() & 0xFFFFFFFFull;
            default: return rng() >> (rng() % 40);
        }
    };
    const char* names[] = {"1", "chr1", "chrUn_KI270442v1", "a_rather_long_contig_name_of_forty_one_ch"};
    const char* samples[] = {"", "S", "sample_07", "NA12878.illumina.hiseq"};
    std::vector<uint8_t> buf(1024);
    for (uint64_t r = 0; r < n_rows; ++r) {
        const std::string name = names[rng() % 4], sample = samples[rng() % 4];
        const bool annotate = rng() & 1, with_sample = rng() & 1;
        const int cls = (int)(rng() % 5);
        const uint32_t pos = (uint32_t)pick(cls == 4 ? 3 : cls);
        uint64_t v[6];
        for (auto& x : v) x = cls == 4 ? (uint32_t)pick(3) : (uint32_t)pick((int)(rng() % (cls + 1)));
        const uint64_t total = cls == 4 ? pick(4) : v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + (uint32_t)pick(0);
        char want[512];
        int wn = snprintf(want, sizeof want, "%s\t%u\t%" PRIu64 "\t%u\t%u\t%u\t%u\t%u\t%u", name.c_str(), pos, total, (uint32_t)v[0], (uint32_t)v[1],
                          (uint32_t)v[2], (uint32_t)v[3], (uint32_t)v[4], (uint32_t)v[5]);
        if (with_sample) wn += snprintf(want + wn, sizeof want - wn, "\t%s", sample.c_str());
        if (annotate) wn += snprintf(want + wn, sizeof want - wn, "\t%c", (r & 1) ? 'y' : 'n');
        wn += snprintf(want + wn, sizeof want - wn, "\n");
        const uint32_t at = 64 + (uint32_t)(r % 16);
        std::fill(buf.begin(), buf.end(), (uint8_t)0xEE);
        RowSink o;
        o.init(buf.data() + at);
        o.str(name.c_str(), (uint32_t)name.size());
        o.sep_num32('\t', pos, n_digits32(pos));
        const bool small = total < 10000;
        if (small) o.sep_num32('\t', (uint32_t)total, n_digits4((uint32_t)total)); else o.sep_num64('\t', total);
        for (auto x : v) o.sep_num32('\t', (uint32_t)x, small && x < 10000 ? n_digits4((uint32_t)x) : n_digits32((uint32_t)x));
        if (with_sample) { o.put((uint64_t)'\t', 1); o.str(sample.c_str(), (uint32_t)sample.size()); }
        if (annotate) o.put((uint64_t)'\t' | (uint64_t)((r & 1) ? 'y' : 'n') << 8 | (uint64_t)'\n' << 16, 3);
        else o.put((uint64_t)'\n', 1);
        const uint8_t* end = o.finish();
        if ((int)(end - (buf.data() + at)) != wn || memcmp(buf.data() + at, want, (size_t)wn) != 0) {
            printf("FAIL row %" PRIu64 ": want %.*s got %.*s\n", r, wn, want, (int)(end - (buf.data() + at)), (const char*)buf.data() + at);
            return 1;
        }
        for (uint32_t i = 0; i < at; ++i) if (buf[i] != 0xEE) return fail("bytes in front of the row", r);
        for (size_t i = at + (size_t)wn; i < buf.size(); ++i) if (buf[i] != 0xEE) return fail("bytes behind the row", r);
    }
    printf("ok %" PRIu64 "\n", n_rows);
    return 0;
}
