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

// K1b variant 2 (inflate.hip, kJump) step by step on the CPU -- the algorithm, not the HIP code: batches of <= 64 entries and
// <= kSpan bytes, a window that keeps >= kHist bytes below the batch, literal and far-match bytes put in place first (phase A), then
// every near-match byte resolved through origin pointers, positions taken 64 at a time in increasing order, the 64 lanes of a
// group in lockstep (all reads of a step before its writes).  Returns false on an inconsistency.
static long g_far_bytes = 0;
static bool resolve_jump(const uint32_t* ent, uint32_t n_ent, const uint8_t* lit, uint32_t isize, std::vector<uint8_t>& out,
                         long* n_lookups, long* n_positions, long* n_groups = nullptr, long* n_steps = nullptr) {
    static const uint32_t kHist = getenv("SBX_EMU_HIST") ? (uint32_t)atoi(getenv("SBX_EMU_HIST")) : 2048u;     // (the kernel: 2048)
    constexpr uint32_t kSpan = 1024;
    const uint32_t kCap = kHist + 1024 + kSpan;
    out.assign(isize, 0);
    std::vector<uint8_t> buf(kCap + 64, 0xDD);
    std::vector<uint16_t> org(kSpan, 0);
    uint32_t opos = 0, lpos = 0, base = 0;
    for (uint32_t e0 = 0; e0 < n_ent;) {
        // the batch: <= 64 entries whose inclusive sums stay <= kSpan (at least one entry: <= 513 bytes)
        uint32_t take = 0, span = 0, lspan = 0;
        uint32_t lr[64], len[64], dist[64], eo[64], el[64];
        while (take < 64 && e0 + take < n_ent) {
            const uint32_t e = ent[e0 + take], l = e >> 24, n = e & 511u;
            if (span + l + n > kSpan) break;
            lr[take] = l; len[take] = n; dist[take] = ((e >> 9) & 0x7FFFu) + 1u;
            eo[take] = opos + span; el[take] = lpos + lspan;
            span += l + n; lspan += l;
            ++take;
        }
        if (!take) return false;
        if (opos + span > isize) return false;
        // slide
        if (opos - base + kSpan > kCap) {
            const uint32_t nb = (opos - kHist) & ~15u, delta = nb - base, keep = opos - nb;
            memmove(buf.data(), buf.data() + delta, keep);
            base = nb;
        }
        const uint32_t lo = opos - base;
        // phase A: literals and far matches (source below the window: final output)
        bool near[64];
        for (uint32_t t = 0; t < take; ++t) {
            for (uint32_t i = 0; i < lr[t]; ++i) buf[eo[t] - base + i] = lit[el[t] + i];
            const uint32_t dst = eo[t] + lr[t];
            near[t] = false;
            if (len[t]) {
                if (dist[t] > dst) return false;
                const uint32_t src = dst - dist[t];
                if (src < base) { for (uint32_t i = 0; i < len[t]; ++i) buf[dst - base + i] = out[src + i]; g_far_bytes += len[t]; }      // (never self-overlapping: kHist > 258)
                else near[t] = true;
            }
        }
        // phase B: marks -> owners (last mark at or before a position)
        const uint32_t k = (span + 63) / 64;
        for (uint32_t q = 0; q < 64 * k; ++q) org[q] = 0xFFFF;
        for (uint32_t t = 0; t < take; ++t) if (near[t]) org[eo[t] + lr[t] - opos] = (uint16_t)t;
        { uint32_t run = 0xFFFF; for (uint32_t q = 0; q < 64 * k; ++q) { if (org[q] != 0xFFFF) run = org[q]; org[q] = (uint16_t)run; } }
        for (uint32_t g = 0; g < k; ++g) {
            const uint32_t g0 = 64 * g;
            uint32_t o[64];
            bool open[64], cov[64];
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t q = g0 + l, e = org[q];
                cov[l] = q < span && e != 0xFFFF && q < eo[e & 63] + lr[e & 63] - opos + len[e & 63];
                o[l] = cov[l] ? q + lo - dist[e & 63] : q + lo;
                open[l] = cov[l] && o[l] >= lo;
            }
            for (uint32_t l = 0; l < 64; ++l) org[g0 + l] = (uint16_t)o[l];
            for (int guard = 0;; ++guard) {
                bool any = false;
                for (uint32_t l = 0; l < 64; ++l) any |= open[l];
                if (!any) break;
                if (guard > 2000) return false;
                if (n_steps) ++*n_steps;
                uint32_t o2[64];
                for (uint32_t l = 0; l < 64; ++l) if (open[l]) { o2[l] = org[o[l] - lo]; ++*n_lookups; }        // reads of the step
                for (uint32_t l = 0; l < 64; ++l) {                                                              // ... then its writes
                    if (!open[l]) continue;
                    const uint32_t sl = o[l] - lo;
                    if (sl < g0) { o[l] = o2[l]; open[l] = false; }
                    else if (o2[l] == o[l]) open[l] = false;
                    else { o[l] = o2[l]; open[l] = o[l] >= lo; }
                    org[g0 + l] = (uint16_t)o[l];
                }
            }
            if (n_groups) ++*n_groups;
            uint8_t v[64];
            for (uint32_t l = 0; l < 64; ++l) if (cov[l]) { v[l] = buf[o[l]]; ++*n_positions; }
            for (uint32_t l = 0; l < 64; ++l) if (cov[l]) buf[g0 + l + lo] = v[l];
        }
        // phase C
        memcpy(out.data() + opos, buf.data() + lo, span);
        opos += span; lpos += lspan; e0 += take;
    }
    return opos == isize;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE [lane] [--jump]\n", argv[0]); return 2; }
    const uint32_t lane = argc > 2 ? (uint32_t)atoi(argv[2]) : 0u;
    const bool jump = argc > 3 && !strcmp(argv[3], "--jump");
    long n_lookups = 0, n_positions = 0, n_groups = 0, n_steps = 0;
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
            if (ok && jump) {
                std::vector<uint8_t> got2;
                ok = resolve_jump(ent.data(), R.n_ent, lit_al, isize, got2, &n_lookups, &n_positions, &n_groups, &n_steps) && memcmp(got2.data(), want.data(), isize) == 0;
            }
            if (!ok) {
                ++n_bad;
                if (n_bad < 5) fprintf(stderr, "MISMATCH block %ld at file offset %zu (isize %u, n_lit %u, n_ent %u, got %zu bytes)\n", n_blocks - 1, pos, isize, n_lit, R.n_ent, got.size());
            }
        }
        pos += bsize;
    }
    printf("%ld %ld %ld %ld\n", n_blocks, n_fast, n_general, n_bad);
    if (jump) fprintf(stderr, "jump: %ld far-match bytes (phase A, from global memory)\n", g_far_bytes);
    if (jump)
        fprintf(stderr, "jump: %ld near-match bytes, %.2f pointer lookups per byte; %ld groups of 64 positions, %.2f lockstep steps per group\n", n_positions,
                n_positions ? (double)n_lookups / n_positions : 0.0, n_groups, n_groups ? (double)n_steps / n_groups : 0.0);
    return n_bad == 0 ? 0 : 1;
}
