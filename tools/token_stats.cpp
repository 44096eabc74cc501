// token_stats -- development tool: what the LZ77 token streams of a BAM's BGZF blocks look like to K1b.
//
// Inflates every BGZF block of a file with a minimal RFC 1951 decoder that keeps the tokens (literal runs and
// {length, distance} matches -- the streams K1a hands to K1b), cuts them into K1b's batches (<= 64 entries and
// <= 1536 output bytes) and reports, per batch: entries, near / far matches (source inside / below the 2 KiB LDS
// history), and how many dependency rounds the near matches need
//   (a) under K1b's rule: a match may start once everything below its source end is final, "final" being everything
//       below the start of the first unfinished match (one readlane per round);
//   (b) under the exact rule: a match may start once no unfinished match writes into its source range.
// Host-only, no device code; g++ -O2 -o token_stats token_stats.cpp.  Numbers quoted in DESIGN.md sections 3 (K1a, K1b) and 10.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

struct Bits {
    const uint8_t* p;
    size_t n, pos = 0;
    uint64_t buf = 0;
    int cnt = 0;
    void fill() { while (cnt <= 56 && pos < n) { buf |= (uint64_t)p[pos++] << cnt; cnt += 8; } }
    uint32_t take(int k) { if (cnt < k) fill(); uint32_t v = (uint32_t)(buf & ((1ull << k) - 1)); buf >>= k; cnt -= k; return v; }
    void align() { int k = cnt & 7; buf >>= k; cnt -= k; }
};

struct Huff {
    uint16_t count[16], symbol[320];
    void build(const uint8_t* len, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int max_len() const { int m = 0; for (int l = 1; l <= 15; ++l) if (count[l]) m = l; return m; }
    int decode(Bits& b, uint64_t* use = nullptr) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= (int)b.take(1);
            int c = count[l];
            if (code - c < first) { if (use) ++use[l]; return symbol[index + (code - first)]; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        throw std::runtime_error("bad code");
    }
};

// code-length statistics of the dynamic codes (what K1a's canonical decode works through): longest code per block, and the
// lengths of the codes actually decoded
static uint64_t g_maxL[16], g_maxD[16], g_useL[16], g_useD[16];

struct Entry { uint32_t lit_run, len, dist; };     // lit_run literals, then a match (len == 0: literals only)

static void tokens_of(const uint8_t* p, size_t n, std::vector<Entry>* out) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    Bits b{p, n};
    uint32_t run = 0;
    for (bool last = false; !last;) {
        last = b.take(1);
        int type = (int)b.take(2);
        if (type == 0) {
            b.align();
            uint32_t len = b.take(16);
            b.take(16);
            for (uint32_t i = 0; i < len; ++i) b.take(8);
            run += len;
            continue;
        }
        Huff L, D;
        uint8_t lens[320];
        if (type == 1) {
            for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            L.build(lens, 288);
            for (int s = 0; s < 30; ++s) lens[s] = 5;
            D.build(lens, 30);
        } else {
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            int nlen = (int)b.take(5) + 257, ndist = (int)b.take(5) + 1, ncode = (int)b.take(4) + 4;
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)b.take(3);
            Huff C;
            C.build(cl, 19);
            for (int i = 0; i < nlen + ndist;) {
                int sym = C.decode(b);
                if (sym < 16) lens[i++] = (uint8_t)sym;
                else {
                    int prev = 0, rep;
                    if (sym == 16) { prev = lens[i - 1]; rep = 3 + (int)b.take(2); }
                    else if (sym == 17) rep = 3 + (int)b.take(3);
                    else rep = 11 + (int)b.take(7);
                    while (rep--) lens[i++] = (uint8_t)prev;
                }
            }
            L.build(lens, nlen);
            D.build(lens + nlen, ndist);
            ++g_maxL[L.max_len()];
            ++g_maxD[D.max_len()];
        }
        for (;;) {
            int sym = L.decode(b, g_useL);
            if (sym < 256) { ++run; continue; }
            if (sym == 256) break;
            sym -= 257;
            uint32_t len = lbase[sym] + b.take(lext[sym]);
            int ds = D.decode(b, g_useD);
            uint32_t dist = dbase[ds] + b.take(dext[ds]);
            while (run > 255) { out->push_back({255, 0, 1}); run -= 255; }
            out->push_back({run, len, dist});
            run = 0;
        }
    }
    while (run > 255) { out->push_back({255, 0, 1}); run -= 255; }
    if (run) out->push_back({run, 0, 1});
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: token_stats file.bam [max_blocks]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    size_t max_blocks = argc > 2 ? (size_t)atol(argv[2]) : 2000;
    std::vector<uint8_t> file;
    {
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        file.resize((size_t)n);
        if (fread(file.data(), 1, (size_t)n, f) != (size_t)n) return 1;
        fclose(f);
    }
    const uint32_t kHist = 2048, kSpan = 1536;
    uint64_t blocks = 0, entries = 0, out_bytes = 0, lit_bytes = 0, matches = 0, nearm = 0, farm = 0, batches = 0;
    uint64_t rounds_a = 0, rounds_b = 0, near_len = 0, far_len = 0, far_short = 0;
    uint64_t hist_a[40] = {0}, hist_b[40] = {0};
    // which paths of k_lz77_resolve a batch executes (a wave pays for a path when ANY of its lanes takes it)
    uint64_t pa_lit_long = 0, pa_far_long = 0, pa_lit_long_tasks = 0, pa_far_long_tasks = 0;
    uint64_t rb_plain_short = 0, rb_plain_long = 0, rb_plain_long_tasks = 0, rb_per_short = 0, rb_per_long_tasks = 0;
    uint64_t rb_per_short_d124 = 0, per_short_n = 0, per_short_d1 = 0, per_short_d124 = 0, per_short_d8 = 0, rb_per_short_gt8 = 0;
    uint64_t plain_long_hist[4] = {0, 0, 0, 0};      // 17-32, 33-64, 65-128, 129+
    uint64_t rb_plain_gt32 = 0;
    size_t off = 0;
    while (off + 18 <= file.size() && blocks < max_blocks + 1) {
        const uint8_t* p = file.data() + off;
        uint32_t xlen = p[10] | (p[11] << 8), bsize = p[16] | (p[17] << 8);
        size_t cdata = (size_t)bsize - xlen - 19;
        std::vector<Entry> ent;
        if (blocks > 0 && cdata > 2) tokens_of(p + 12 + xlen, cdata, &ent);       // (skip the header block)
        off += (size_t)bsize + 1;
        ++blocks;
        if (ent.empty()) continue;
        // positions
        uint32_t opos = 0, base = 0;
        for (size_t e0 = 0; e0 < ent.size();) {
            size_t e1 = e0;
            uint32_t span = 0;
            while (e1 < ent.size() && e1 - e0 < 64 && span + ent[e1].lit_run + ent[e1].len <= kSpan) { span += ent[e1].lit_run + ent[e1].len; ++e1; }
            if (e1 == e0) { span = ent[e1].lit_run + ent[e1].len; ++e1; }
            if (opos - base + kSpan > kHist + 1024 + kSpan) base = (opos - kHist) & ~15u;
            // entries of the batch: dst / src ranges
            struct M { uint32_t dst, src, shi, len; bool near_; uint32_t dist; };
            std::vector<M> ms;
            uint32_t o = opos;
            for (size_t e = e0; e < e1; ++e) {
                o += ent[e].lit_run;
                lit_bytes += ent[e].lit_run;
                if (ent[e].len) {
                    uint32_t dst = o, src = o - ent[e].dist;
                    bool nr = src >= base;
                    ms.push_back({dst, src, src + std::min(ent[e].len, ent[e].dist), ent[e].len, nr, ent[e].dist});
                    ++matches;
                    if (nr) { ++nearm; near_len += ent[e].len; } else { ++farm; far_len += ent[e].len; if (ent[e].len <= 16) ++far_short; }
                }
                o += ent[e].len;
            }
            {   // phase A: long literal runs / long far matches go through the cooperative copy loop (four tasks per iteration)
                uint32_t ll = 0, fl = 0;
                for (size_t e = e0; e < e1; ++e) if (ent[e].lit_run > 16) ++ll;
                for (auto& m : ms) if (!m.near_ && m.len > 16) ++fl;
                pa_lit_long += ll != 0; pa_far_long += fl != 0; pa_lit_long_tasks += ll; pa_far_long_tasks += fl;
            }
            // (a) frontier rule
            {
                std::vector<char> pend(ms.size());
                size_t np = 0;
                for (size_t i = 0; i < ms.size(); ++i) { pend[i] = ms[i].near_; np += pend[i]; }
                uint32_t r = 0;
                while (np) {
                    uint32_t F = 0;
                    for (size_t i = 0; i < ms.size(); ++i) if (pend[i]) { F = ms[i].dst; break; }
                    bool ps = false, pl = false, qs = false, qs_other = false, qs_gt8 = false, pl32 = false;
                    for (size_t i = 0; i < ms.size(); ++i) if (pend[i] && ms[i].shi <= F) {
                        pend[i] = 0; --np;
                        const bool per = ms[i].dist < ms[i].len;
                        if (!per) {
                            if (ms[i].len <= 16) ps = true;
                            else {
                                pl = true; ++rb_plain_long_tasks;
                                plain_long_hist[ms[i].len <= 32 ? 0 : ms[i].len <= 64 ? 1 : ms[i].len <= 128 ? 2 : 3]++;
                                if (ms[i].len > 32) pl32 = true;
                            }
                        }
                        else if (ms[i].len <= 16) {
                            qs = true; ++per_short_n;
                            if (ms[i].dist == 1) ++per_short_d1;
                            if (ms[i].dist == 1 || ms[i].dist == 2 || ms[i].dist == 4) ++per_short_d124; else qs_other = true;
                            if (ms[i].dist <= 8) ++per_short_d8; else qs_gt8 = true;
                        } else ++rb_per_long_tasks;
                    }
                    rb_plain_short += ps; rb_plain_long += pl; rb_per_short += qs; rb_per_short_d124 += qs && !qs_other;
                    rb_per_short_gt8 += qs_gt8; rb_plain_gt32 += pl32;
                    ++r;
                }
                rounds_a += r;
                hist_a[std::min<uint32_t>(r, 39)]++;
            }
            // (b) exact rule: ready when no pending match's destination intersects the source range
            {
                std::vector<char> pend(ms.size());
                size_t np = 0;
                for (size_t i = 0; i < ms.size(); ++i) { pend[i] = ms[i].near_; np += pend[i]; }
                uint32_t r = 0;
                while (np) {
                    std::vector<char> ready(ms.size(), 0);
                    for (size_t i = 0; i < ms.size(); ++i) {
                        if (!pend[i]) continue;
                        bool ok = true;
                        for (size_t j = 0; j < i && ok; ++j)
                            if (pend[j] && ms[j].dst < ms[i].shi && ms[j].dst + ms[j].len > ms[i].src) ok = false;
                        ready[i] = ok;
                    }
                    for (size_t i = 0; i < ms.size(); ++i) if (ready[i]) { pend[i] = 0; --np; }
                    ++r;
                }
                rounds_b += r;
                hist_b[std::min<uint32_t>(r, 39)]++;
            }
            entries += e1 - e0;
            out_bytes += span;
            opos += span;
            ++batches;
            e0 = e1;
        }
    }
    printf("blocks %llu  entries %llu  output %llu B  literals %.1f %%  matches %llu (near %.1f %%, far %.1f %%)\n", (unsigned long long)blocks,
           (unsigned long long)entries, (unsigned long long)out_bytes, 100.0 * lit_bytes / out_bytes, (unsigned long long)matches,
           100.0 * nearm / matches, 100.0 * farm / matches);
    printf("per batch: %.1f entries, %.0f output bytes; mean match length near %.1f far %.1f (far <= 16 B: %.1f %%)\n", (double)entries / batches,
           (double)out_bytes / batches, (double)near_len / std::max<uint64_t>(1, nearm), (double)far_len / std::max<uint64_t>(1, farm),
           100.0 * far_short / std::max<uint64_t>(1, farm));
    printf("dependency rounds per batch: frontier rule %.2f, exact rule %.2f\n", (double)rounds_a / batches, (double)rounds_b / batches);
    printf("phase A per batch: long literal runs in %.2f of the batches (%.2f tasks), long far matches in %.2f (%.2f tasks)\n",
           (double)pa_lit_long / batches, (double)pa_lit_long_tasks / batches, (double)pa_far_long / batches, (double)pa_far_long_tasks / batches);
    printf("phase B per batch (frontier rounds that execute a path): plain short %.2f, plain long %.2f (%.2f tasks), periodic short %.2f "
           "(of which only dist 1/2/4: %.2f), periodic long tasks %.2f\n", (double)rb_plain_short / batches, (double)rb_plain_long / batches,
           (double)rb_plain_long_tasks / batches, (double)rb_per_short / batches, (double)rb_per_short_d124 / batches, (double)rb_per_long_tasks / batches);
    printf("periodic short matches: %.2f per batch, dist 1: %.1f %%, dist 1/2/4: %.1f %%\n", (double)per_short_n / batches,
           100.0 * per_short_d1 / std::max<uint64_t>(1, per_short_n), 100.0 * per_short_d124 / std::max<uint64_t>(1, per_short_n));
    printf("periodic short with dist <= 8: %.1f %%; rounds per batch that still hold one with dist > 8: %.3f\n",
           100.0 * per_short_d8 / std::max<uint64_t>(1, per_short_n), (double)rb_per_short_gt8 / batches);
    printf("plain long near matches by length: 17-32 %.1f %%, 33-64 %.1f %%, 65-128 %.1f %%, 129+ %.1f %%; rounds per batch with one > 32: %.2f\n",
           100.0 * plain_long_hist[0] / std::max<uint64_t>(1, rb_plain_long_tasks), 100.0 * plain_long_hist[1] / std::max<uint64_t>(1, rb_plain_long_tasks),
           100.0 * plain_long_hist[2] / std::max<uint64_t>(1, rb_plain_long_tasks), 100.0 * plain_long_hist[3] / std::max<uint64_t>(1, rb_plain_long_tasks),
           (double)rb_plain_gt32 / batches);
    printf("code lengths (K1a): longest literal/length code per block | longest distance code per block | decoded literal/length codes | decoded distance codes\n");
    {
        uint64_t tb = 0, tl = 0, td = 0;
        for (int l = 0; l < 16; ++l) { tb += g_maxL[l]; tl += g_useL[l]; td += g_useD[l]; }
        for (int l = 1; l < 16; ++l)
            printf("  %2d  %6.2f %%  %6.2f %%  %6.2f %%  %6.2f %%\n", l, 100.0 * g_maxL[l] / std::max<uint64_t>(1, tb), 100.0 * g_maxD[l] / std::max<uint64_t>(1, tb),
                   100.0 * g_useL[l] / std::max<uint64_t>(1, tl), 100.0 * g_useD[l] / std::max<uint64_t>(1, td));
    }
    printf("rounds histogram (frontier | exact):\n");
    for (int r = 0; r < 16; ++r) printf("  %2d  %6.2f %%  %6.2f %%\n", r, 100.0 * hist_a[r] / batches, 100.0 * hist_b[r] / batches);
    return 0;
}
