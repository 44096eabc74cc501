// tests/cpp/inflate2_host.cpp -- runs the lane program of K1a `huffman_decode2` (sambamba_amd/csrc/inflate2_core.hpp) on the CPU,
// one lane at a time, over every BGZF block of a file, applies the literal translation and the LZ77 resolve in plain C++ and
// compares the result with zlib's inflate of the same block (the library the reference calls, block.d:158-185).
// Test infrastructure (tests/test_inflate2_cpu.py), not part of the product.
//   usage: inflate2_host FILE [lane]      exit 0 = every block identical;  prints "<blocks> <fast> <general> <bad>"
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sambamba_amd/csrc/inflate2_core.hpp"

using namespace sbx::inf2;

static std::vector<uint8_t> read_file(const char* path) {
    std::vector<uint8_t> v;
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n);
    if (n && fread(v.data(), 1, n, f) != (size_t)n) { perror("read"); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE [lane]\n", argv[0]); return 2; }
    const uint32_t lane = argc > 2 ? (uint32_t)atoi(argv[2]) : 0u;
    std::vector<uint8_t> file = read_file(argv[1]);
    std::vector<uint8_t> lds(kWaveLds + kLenTabBytes + kDistTabBytes, 0xA5);
    uint16_t* len_tab = (uint16_t*)(lds.data() + kWaveLds);
    uint32_t* dist_tab = (uint32_t*)(lds.data() + kWaveLds + kLenTabBytes);
    for (uint32_t i = 0; i < 32; ++i) rfc_tables_entry(i, &len_tab[i], &dist_tab[i]);
    size_t pos = 0;
    long n_blocks = 0, n_fast = 0, n_general = 0, n_bad = 0;
    while (pos + 18 <= file.size()) {
        const uint8_t* h = file.data() + pos;
        if (h[0] != 31 || h[1] != 139) { fprintf(stderr, "not a BGZF block at %zu\n", pos); return 2; }
        const uint32_t xlen = h[10] | h[11] << 8;
        const uint32_t bsize = (h[16] | h[17] << 8) + 1u;      // (BGZF: the BC subfield is the first one in every file we read here)
        const uint32_t hdr = 12 + xlen, clen = bsize - hdr - 8;
        const uint32_t isize = h[bsize - 4] | h[bsize - 3] << 8 | h[bsize - 2] << 16 | (uint32_t)h[bsize - 1] << 24;
        // expected: zlib raw inflate
        std::vector<uint8_t> want(isize + 16);
        {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            inflateInit2(&zs, -15);
            zs.next_in = (Bytef*)(h + hdr);
            zs.avail_in = clen;
            zs.next_out = want.data();
            zs.avail_out = isize + 16;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.total_out != isize) { fprintf(stderr, "zlib failed at %zu\n", pos); return 2; }
            inflateEnd(&zs);
        }
        // the lane: payload copied to every alignment in turn
        const uint32_t lead = (uint32_t)(n_blocks & 3);
        std::vector<uint8_t> in(clen + 128 + 8, 0x5A);
        uint8_t* in_al = in.data() + ((8 - ((uintptr_t)in.data() & 7)) & 7) + lead;
        memcpy(in_al, h + hdr, clen);
        std::vector<uint8_t> lit(isize + 128 + 64, 0xEE);
        std::vector<uint32_t> ent(isize / 3 + isize / 255 + 64, 0xEEEEEEEEu);
        std::vector<uint8_t> scratch(kScratchBytes + 64, 0xCC);
        uint8_t* lit_al = lit.data() + ((64 - ((uintptr_t)lit.data() & 63)) & 63);
        uint8_t* scr_al = scratch.data() + ((16 - ((uintptr_t)scratch.data() & 15)) & 15);
        LaneIo io;
        io.in = in_al; io.in_bits = clen * 8u; io.osize = isize; io.lit = lit_al; io.ent = ent.data(); io.scratch = scr_al; io.live = true;
        Lane L;
        const LaneResult R = L.run(io, lds.data(), lane, len_tab, dist_tab);
        ++n_blocks;
        if (R.status != 0) {
            ++n_general;
        } else {
            ++n_fast;
            const uint32_t* info = (const uint32_t*)(scr_al + kScratchInfo);
            const uint32_t n_seg = info[0], n_lit = info[1];
            bool ok = n_lit == R.n_lit;
            // translation
            for (uint32_t s = 0; s < n_seg && ok; ++s) {
                const uint32_t a = info[2 + s], b = s + 1 < n_seg ? info[2 + s + 1] : n_lit;
                const uint8_t* tab = scr_al + kScratchTabs + 256 * s;
                for (uint32_t i = a; i < b; ++i) lit_al[i] = tab[lit_al[i]];
            }
            // resolve
            std::vector<uint8_t> got;
            got.reserve(isize + 600);
            uint32_t lp = 0;
            for (uint32_t k = 0; k < R.n_ent && ok; ++k) {
                const uint32_t e = ent[k], lr = e >> 24, len = e & 511u, dist = ((e >> 9) & 0x7FFFu) + 1u;
                for (uint32_t i = 0; i < lr; ++i) got.push_back(lit_al[lp++]);
                if (len) {
                    if (dist > got.size()) { ok = false; break; }
                    const size_t src = got.size() - dist;
                    for (uint32_t i = 0; i < len; ++i) got.push_back(got[src + i]);
                }
                if (got.size() > isize) ok = false;
            }
            ok = ok && lp == n_lit && got.size() == isize && memcmp(got.data(), want.data(), isize) == 0;
            if (!ok) {
                ++n_bad;
                if (n_bad < 5) fprintf(stderr, "MISMATCH block %ld at file offset %zu (isize %u, n_lit %u, n_ent %u, got %zu bytes)\n", n_blocks - 1, pos, isize, n_lit, R.n_ent, got.size());
            }
        }
        pos += bsize;
    }
    printf("%ld %ld %ld %ld\n", n_blocks, n_fast, n_general, n_bad);
    return n_bad == 0 ? 0 : 1;
}
