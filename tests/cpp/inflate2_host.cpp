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
#include "../../sambamba_amd/csrc/lz77_copy.hpp"

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

// The copy primitives of K1b (lz77_copy.hpp; round 6: two overlapping 8-byte words instead of four collapsing dwords) on the CPU: every
// length at every source / destination alignment against memcpy -- nothing outside [0, n) of the destination may change --, and the
// byte-permute expansion of a short periodic match for every (length, period) against the byte loop.
static bool copy_primitives_ok() {
    uint8_t src[96], dst[96], ref[96];
    for (uint32_t n = 0; n <= 16; ++n)
        for (uint32_t sa = 0; sa < 9; ++sa)
            for (uint32_t da = 0; da < 9; ++da) {
                for (int i = 0; i < 96; ++i) { src[i] = (uint8_t)(i * 37 + 11); dst[i] = ref[i] = (uint8_t)(200 - i); }
                sbx::lz::Short16 c;
                c.load(src + 16 + sa, n);
                c.store(dst + 32 + da, n);
                memcpy(ref + 32 + da, src + 16 + sa, n);
                if (memcmp(dst, ref, 96)) { fprintf(stderr, "Short16: n=%u src+%u dst+%u differs from memcpy\n", n, sa, da); return false; }
            }
    for (uint32_t len = 2; len <= 16; ++len)
        for (uint32_t dist = 1; dist <= 8 && dist < len; ++dist) {
            for (int i = 0; i < 96; ++i) { src[i] = (uint8_t)(i * 29 + 5); dst[i] = ref[i] = 0xEE; }
            const sbx::lz::W2 xx = sbx::lz::ld64(src + 3);
            sbx::lz::Short16 ws;
            sbx::lz::periodic16(xx.x, xx.y, len, sbx::lz::period_selector(dist, 0), sbx::lz::period_selector(dist, 1),
                                sbx::lz::period_selector(dist, 2), sbx::lz::period_selector(dist, 3), &ws);
            ws.store(dst + 40, len);
            for (uint32_t k = 0; k < len; ++k) ref[40 + k] = src[3 + k % dist];
            if (memcmp(dst, ref, 96)) { fprintf(stderr, "periodic16: len=%u dist=%u differs from the byte loop\n", len, dist); return false; }
        }
    return true;
}

// K1b's near-match resolution with the EXACT readiness rule (inflate.hip, kExact: the default kernel since round 5) step by step on the
// CPU -- the algorithm, not the HIP code: batches of <= 64 entries and <= kSpan bytes; literal runs and far matches (source below the
// window base) put in place first (phase A); then rounds: a pending match is READY when no pending match writes into its source range
// [src, src + min(len, dist)) -- the matches that could are a contiguous range of lanes [jlo, jhi), found by the same two branch-free
// binary searches over the lanes' {start, end} the kernel runs, tested against the set of pending lanes -- and all ready matches of a
// round are copied in lockstep (every read of the round before its writes).  Returns false on an inconsistency; counts the rounds.
static bool resolve_exact(const uint32_t* ent, uint32_t n_ent, const uint8_t* lit, uint32_t isize, std::vector<uint8_t>& out,
                          long* n_batches, long* n_rounds) {
    constexpr uint32_t kHist = 2048, kSpan = 1536, kCap = kHist + 1024 + kSpan;
    out.assign((size_t)isize + 64, 0);        // (the primitives may read a dword / write nothing beyond a copy: 64 bytes of slack as on the device)
    uint32_t opos = 0, lpos = 0, base = 0;
    for (uint32_t e0 = 0; e0 < n_ent;) {
        uint32_t take = 0, span = 0, lspan = 0;
        uint32_t lr[64], len[64], dist[64], dst[64];
        while (take < 64 && e0 + take < n_ent) {
            const uint32_t e = ent[e0 + take], l = e >> 24, n = e & 511u;
            if (span + l + n > kSpan) break;
            lr[take] = l; len[take] = n; dist[take] = ((e >> 9) & 0x7FFFu) + 1u;
            dst[take] = opos + span + l;
            span += l + n; lspan += l;
            ++take;
        }
        if (!take || opos + span > isize) return false;
        if (opos - base + kSpan > kCap) base = (opos - kHist) & ~15u;          // the window slides: what lies below `base` is "far"
        // phase A: literal runs; far matches (their source was final a batch ago)
        uint32_t lp = lpos;
        uint64_t pending = 0;
        uint32_t start[64], end[64];
        for (uint32_t j = 0; j < 64; ++j) {
            if (j < take) {
                if (lr[j] <= 32) {             // own-lane copy: two steps of Short16 (the kernel's phase A)
                    for (uint32_t h = 0; h < 2; ++h) {
                        const uint32_t nn = lr[j] > 16 * h ? (lr[j] - 16 * h < 16 ? lr[j] - 16 * h : 16) : 0;
                        sbx::lz::Short16 c;
                        c.load(lit + lp + 16 * h, nn);
                        c.store(out.data() + dst[j] - lr[j] + 16 * h, nn);
                    }
                    lp += lr[j];
                } else
                    for (uint32_t k = 0; k < lr[j]; ++k) out[dst[j] - lr[j] + k] = lit[lp++];
                if (len[j]) {
                    if (dist[j] > dst[j]) return false;
                    const uint32_t src = dst[j] - dist[j];
                    if (src < base) {
                        if (len[j] <= 32 && dist[j] >= len[j]) {
                            for (uint32_t h = 0; h < 2; ++h) {
                                const uint32_t nn = len[j] > 16 * h ? (len[j] - 16 * h < 16 ? len[j] - 16 * h : 16) : 0;
                                sbx::lz::Short16 c;
                                c.load(out.data() + src + 16 * h, nn);
                                c.store(out.data() + dst[j] + 16 * h, nn);
                            }
                        } else
                            for (uint32_t k = 0; k < len[j]; ++k) out[dst[j] + k] = out[src + k];
                    }
                    else pending |= 1ull << j;
                }
                start[j] = dst[j] - base; end[j] = dst[j] + len[j] - base;
            } else {
                start[j] = end[j] = opos + span - base;     // (lanes behind the batch: monotone, never pending)
            }
        }
        // dependency masks: two branch-free binary searches per lane, as in the kernel
        uint64_t dep[64];
        for (uint32_t i = 0; i < 64; ++i) {
            dep[i] = 0;
            if (!(pending >> i & 1)) continue;
            const uint32_t srco = dst[i] - dist[i] - base, s_hio = srco + (len[i] < dist[i] ? len[i] : dist[i]);
            uint32_t jlo = 0, jhi = 0;
            for (uint32_t step = 32; step; step >>= 1) {
                if (end[jlo + step - 1] <= srco) jlo += step;
                if (start[jhi + step - 1] < s_hio) jhi += step;
            }
            if (jhi > jlo) dep[i] = ((jhi >= 64 ? ~0ull : (1ull << jhi) - 1ull) & ~((1ull << jlo) - 1ull));
        }
        ++*n_batches;
        for (int guard = 0; pending; ++guard) {
            if (guard > 64) return false;                    // (a round finishes at least the first pending match)
            ++*n_rounds;
            uint64_t ready = 0;
            for (uint32_t i = 0; i < 64; ++i) if ((pending >> i & 1) && !(dep[i] & pending)) ready |= 1ull << i;
            if (!ready) return false;
            std::vector<std::pair<uint32_t, uint8_t>> writes;
            // the kernel's paths: plain matches of up to 32 bytes in two Short16 steps (all lanes load, then all lanes store, step by
            // step), periodic ones of up to 16 bytes with a period of up to 8 through the byte permutes; everything else byte by byte
            sbx::lz::Short16 cp[64];
            for (uint32_t h = 0; h < 2; ++h) {
                for (uint32_t i = 0; i < 64; ++i)
                    if ((ready >> i & 1) && dist[i] >= len[i] && len[i] <= 32) {
                        const uint32_t nn = len[i] > 16 * h ? (len[i] - 16 * h < 16 ? len[i] - 16 * h : 16) : 0;
                        cp[i].load(out.data() + dst[i] - dist[i] + 16 * h, nn);
                    }
                for (uint32_t i = 0; i < 64; ++i)
                    if ((ready >> i & 1) && dist[i] >= len[i] && len[i] <= 32) {
                        const uint32_t nn = len[i] > 16 * h ? (len[i] - 16 * h < 16 ? len[i] - 16 * h : 16) : 0;
                        cp[i].store(out.data() + dst[i] + 16 * h, nn);
                    }
            }
            for (uint32_t i = 0; i < 64; ++i) {
                if (!(ready >> i & 1) || (dist[i] >= len[i] && len[i] <= 32)) continue;
                const uint32_t src = dst[i] - dist[i];
                if (dist[i] < len[i] && len[i] <= 16 && dist[i] <= 8) {
                    const sbx::lz::W2 xx = sbx::lz::ld64(out.data() + src);
                    sbx::lz::Short16 ws;
                    sbx::lz::periodic16(xx.x, xx.y, len[i], sbx::lz::period_selector(dist[i], 0), sbx::lz::period_selector(dist[i], 1),
                                        sbx::lz::period_selector(dist[i], 2), sbx::lz::period_selector(dist[i], 3), &ws);
                    ws.store(out.data() + dst[i], len[i]);
                    continue;
                }
                for (uint32_t k = 0; k < len[i]; ++k) writes.push_back({dst[i] + k, out[src + k % dist[i]]});      // reads [src, dst) only
            }
            for (auto& w : writes) out[w.first] = w.second;
            pending &= ~ready;
        }
        opos += span; lpos += lspan; e0 += take;
    }
    out.resize(isize);
    return opos == isize;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE [lane] [--exact]\n", argv[0]); return 2; }
    const uint32_t lane = argc > 2 ? (uint32_t)atoi(argv[2]) : 0u;
    const bool exact = argc > 3 && !strcmp(argv[3], "--exact");
    if (!copy_primitives_ok()) return 1;
    long n_batches = 0, n_rounds = 0;
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
            if (ok && exact) {
                std::vector<uint8_t> got2;
                ok = resolve_exact(ent.data(), R.n_ent, lit_al, isize, got2, &n_batches, &n_rounds) && memcmp(got2.data(), want.data(), isize) == 0;
            }
            if (!ok) {
                ++n_bad;
                if (n_bad < 5) fprintf(stderr, "MISMATCH block %ld at file offset %zu (isize %u, n_lit %u, n_ent %u, got %zu bytes)\n", n_blocks - 1, pos, isize, n_lit, R.n_ent, got.size());
            }
        }
        pos += bsize;
    }
    printf("%ld %ld %ld %ld\n", n_blocks, n_fast, n_general, n_bad);
    if (exact) fprintf(stderr, "exact rule: %ld batches, %.2f rounds per batch\n", n_batches, n_batches ? (double)n_rounds / n_batches : 0.0);
    return n_bad == 0 ? 0 : 1;
}
