// tools/gen_bam.cpp -- deterministic synthetic coordinate-sorted BAM + BAI generator
// (bench / test harness only; not part of the product library).
//
// Implements the synthetic inputs of SURVEY.md section 8(d): paired-end reads, insert size
// N(mu,sd) clipped to [read_len,1000], stratified-uniform start positions (sorted by
// construction), read names "r%010u" shared by mates, RG:Z:<sample>, bases uniform ACGT with
// 0.1 % N, qualities from {2,12,23,37} with P={.02,.05,.13,.80}, mapq 60 (3 % mapq 0),
// flags 99/147/83/163 with 2 % duplicates and 0.5 % QC-fail, CIGARs 92 % <L>M, 4 % soft
// clips, 2 % one insertion, 2 % one deletion, 0.2 % one N skip (100-5000).
// BGZF: 0xFF00 payload bytes per block, zlib deflate level 6 (default), EOF block; BAI with
// the standard binning + 16 kbp linear index (the .bai the depth tool requires).
//
// Everything is a pure function of (seed, pair index, mate) so that segments of the genome
// can be generated independently by worker threads and concatenated.
//
//   gen_bam --out x.bam --contigs chr1:248956422 --coverage 30 --seed 0x5A4D0002 --threads 8
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>
#include <dlfcn.h>

struct Rng {  // xoshiro256** seeded through splitmix64
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t stream) {
        uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL);
        for (auto& v : s) v = splitmix(x);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

struct Contig { std::string name; int64_t len; };

struct Params {
    std::string out;
    std::vector<Contig> contigs;
    double coverage = 30;
    int read_len = 150;
    uint64_t seed = 0x5A4D0002ULL;
    int threads = 0;
    int level = 6;
    double ins_mu = 400, ins_sd = 50;
    int n_samples = 1;
    bool tie_free_overlaps = false;  // config 5: mates agree in their overlap (no order-sensitive ties)
    int64_t segment = 1 << 19;      // positions per generation task
};

struct RecInfo { uint64_t off; int32_t pos, end; uint32_t bin; };  // off = offset in the segment's byte stream

struct Pair {  // everything about pair k that both mates' generators must agree on
    int64_t s1, s2;   // leftmost positions of the left / right read
    int ins;
    bool left_is_first, dup, qcfail, mapq0_l, mapq0_r;
    int sample;
};

static int reg2bin(int64_t beg, int64_t end) {
    if (end <= beg) end = beg + 1;
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct ContigPlan {
    int64_t len;
    int64_t n_pairs;
    double W;        // stratum width for pair start positions
    int64_t usable;  // starts are drawn in [0, usable)
    uint64_t pair_base;  // global index of this contig's pair 0 (for names / streams)
};

static Pair make_pair(const Params& P, const ContigPlan& cp, int64_t k) {
    Rng r(P.seed, ((cp.pair_base + (uint64_t)k) << 2) | 2);
    Pair p;
    double u = r.uniform();
    p.s1 = (int64_t)std::floor(((double)k + u) * cp.W);
    if (p.s1 >= cp.usable) p.s1 = cp.usable - 1;
    double ins = P.ins_mu + P.ins_sd * r.normal();
    int insi = (int)std::lround(ins);
    insi = std::max(P.read_len, std::min(1000, insi));
    p.ins = insi;
    p.s2 = p.s1 + insi - P.read_len;
    p.left_is_first = r.next() & 1;
    p.dup = r.uniform() < 0.02;
    p.qcfail = r.uniform() < 0.005;
    p.mapq0_l = r.uniform() < 0.03;
    p.mapq0_r = r.uniform() < 0.03;
    p.sample = P.n_samples > 1 ? (int)r.below((uint32_t)P.n_samples) : 0;
    return p;
}

static const uint8_t QUALS[4] = {2, 12, 23, 37};
static uint8_t QLUT[256];
static void init_luts() {
    // P = .02 .05 .13 .80 in 1/256 units: 5, 13, 33, 205
    int i = 0;
    for (; i < 5; ++i) QLUT[i] = QUALS[0];
    for (; i < 18; ++i) QLUT[i] = QUALS[1];
    for (; i < 51; ++i) QLUT[i] = QUALS[2];
    for (; i < 256; ++i) QLUT[i] = QUALS[3];
}
static const uint8_t NIB[4] = {1, 2, 4, 8};  // A C G T in BAM 4-bit codes

static void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 24));
}

// Append one read of pair k (right==false: left read) to `out`.
static void emit_read(const Params& P, const ContigPlan& cp, int ref_id, int64_t k, const Pair& pr, bool right,
                      std::vector<uint8_t>& out, std::vector<RecInfo>& info, const std::vector<std::string>& rg_ids) {
    const int L = P.read_len;
    Rng r(P.seed, ((cp.pair_base + (uint64_t)k) << 2) | (right ? 1u : 0u));
    int64_t pos = right ? pr.s2 : pr.s1;
    // CIGAR
    uint32_t cig[4];
    int nc = 0;
    double u = r.uniform();
    int64_t ref_span = L;
    if (u < 0.92) {
        cig[nc++] = ((uint32_t)L << 4) | 0;
    } else if (u < 0.96) {
        int kclip = 1 + (int)r.below((uint32_t)std::min(L - 1, 60));
        if (r.next() & 1) { cig[nc++] = ((uint32_t)kclip << 4) | 4; cig[nc++] = ((uint32_t)(L - kclip) << 4) | 0; }
        else { cig[nc++] = ((uint32_t)(L - kclip) << 4) | 0; cig[nc++] = ((uint32_t)kclip << 4) | 4; }
        ref_span = L - kclip;
    } else if (u < 0.98) {
        int il = 1 + (int)r.below(10);
        int a = 1 + (int)r.below((uint32_t)(L - il - 1));
        cig[nc++] = ((uint32_t)a << 4) | 0; cig[nc++] = ((uint32_t)il << 4) | 1; cig[nc++] = ((uint32_t)(L - a - il) << 4) | 0;
        ref_span = L - il;
    } else if (u < 0.998) {
        int dl = 1 + (int)r.below(30);
        int a = 1 + (int)r.below((uint32_t)(L - 1));
        cig[nc++] = ((uint32_t)a << 4) | 0; cig[nc++] = ((uint32_t)dl << 4) | 2; cig[nc++] = ((uint32_t)(L - a) << 4) | 0;
        ref_span = L + dl;
    } else {
        int nl = 100 + (int)r.below(4901);
        int a = 1 + (int)r.below((uint32_t)(L - 1));
        cig[nc++] = ((uint32_t)a << 4) | 0; cig[nc++] = ((uint32_t)nl << 4) | 3; cig[nc++] = ((uint32_t)(L - a) << 4) | 0;
        ref_span = L + nl;
    }
    if (P.tie_free_overlaps) {  // config 5: plain <L>M so that mates can be made to agree in overlaps
        nc = 0; cig[nc++] = ((uint32_t)L << 4) | 0; ref_span = L;
    }
    if (pos + ref_span > cp.len) {  // keep alignments inside the contig
        nc = 0; cig[nc++] = ((uint32_t)L << 4) | 0; ref_span = L;
        if (pos + ref_span > cp.len) pos = cp.len - ref_span;
    }
    bool first = right ? !pr.left_is_first : pr.left_is_first;
    uint32_t flag = 1 | 2 | (right ? 16u : 32u) | (first ? 64u : 128u);
    if (pr.dup) flag |= 0x400;
    if (pr.qcfail) flag |= 0x200;
    uint32_t mapq = (right ? pr.mapq0_r : pr.mapq0_l) ? 0 : 60;
    char name[32];
    int ln = snprintf(name, sizeof name, "r%010llu", (unsigned long long)(cp.pair_base + (uint64_t)k)) + 1;
    const std::string& rg = rg_ids[(size_t)pr.sample];
    uint32_t tags_len = 3 + (uint32_t)rg.size() + 1;
    uint32_t block_size = 32 + (uint32_t)ln + 4 * (uint32_t)nc + (uint32_t)(L + 1) / 2 + (uint32_t)L + tags_len;
    RecInfo ri;
    ri.off = out.size();
    ri.pos = (int32_t)pos;
    ri.end = (int32_t)(pos + ref_span);
    ri.bin = (uint32_t)reg2bin(pos, pos + ref_span);
    info.push_back(ri);
    size_t base = out.size();
    out.resize(base + 4 + block_size);
    uint8_t* p = out.data() + base;
    auto w32 = [&](size_t o, uint32_t x) { memcpy(p + o, &x, 4); };
    w32(0, block_size);
    w32(4, (uint32_t)ref_id);
    w32(8, (uint32_t)pos);
    w32(12, ((uint32_t)reg2bin(pos, pos + ref_span) << 16) | (mapq << 8) | (uint32_t)ln);
    w32(16, (flag << 16) | (uint32_t)nc);
    w32(20, (uint32_t)L);
    w32(24, (uint32_t)ref_id);
    w32(28, (uint32_t)(right ? pr.s1 : pr.s2));
    int32_t tlen = right ? -pr.ins : pr.ins;
    w32(32, (uint32_t)tlen);
    memcpy(p + 36, name, (size_t)ln);
    uint8_t* q = p + 36 + ln;
    memcpy(q, cig, 4 * (size_t)nc);
    q += 4 * nc;
    // bases: in tie-free mode both mates copy from a per-pair "template" stream indexed by
    // reference position so they agree wherever they overlap.
    uint8_t* seq = q;
    uint8_t* qual = q + (L + 1) / 2;
    if (!P.tie_free_overlaps) {
        for (int i = 0; i < L; i += 32) {
            uint64_t bits = r.next();
            for (int j = 0; j < 32 && i + j < L; j += 2) {
                uint8_t hi = NIB[bits & 3], lo = NIB[(bits >> 2) & 3];
                bits >>= 4;
                seq[(i + j) >> 1] = (uint8_t)((hi << 4) | ((i + j + 1 < L) ? lo : 0));
            }
        }
        // 0.1 % N
        uint32_t nN = 0;
        double un = r.uniform();
        double pz = std::exp(-0.001 * L), acc = pz, term = pz;
        while (un > acc && nN < 8) { ++nN; term *= 0.001 * L / nN; acc += term; }
        for (uint32_t t = 0; t < nN; ++t) {
            uint32_t i = r.below((uint32_t)L);
            if (i & 1) seq[i >> 1] = (uint8_t)((seq[i >> 1] & 0xF0) | 15);
            else seq[i >> 1] = (uint8_t)((seq[i >> 1] & 0x0F) | (15 << 4));
        }
        for (int i = 0; i < L; i += 8) {
            uint64_t bits = r.next();
            for (int j = 0; j < 8 && i + j < L; ++j) { qual[i + j] = QLUT[bits & 255]; bits >>= 8; }
        }
    } else {
        Rng t(P.seed, ((cp.pair_base + (uint64_t)k) << 2) | 3);
        // template covers reference offsets [0, ins): base and quality per offset
        int off0 = (int)(pos - pr.s1);
        std::vector<uint8_t> tb((size_t)pr.ins + 8), tq((size_t)pr.ins + 8);
        for (int i = 0; i < pr.ins; i += 8) {
            uint64_t b = t.next(), qq = t.next();
            for (int j = 0; j < 8; ++j) { tb[(size_t)(i + j)] = NIB[(b >> (2 * j)) & 3]; tq[(size_t)(i + j)] = QLUT[(qq >> (8 * j)) & 255]; }
        }
        for (int i = 0; i < L; ++i) {
            int o = std::min(std::max(off0 + i, 0), pr.ins - 1);
            uint8_t b = tb[(size_t)o];
            if (i & 1) seq[i >> 1] |= b; else seq[i >> 1] = (uint8_t)(b << 4);
            qual[i] = tq[(size_t)o];
        }
    }
    uint8_t* tg = qual + L;
    tg[0] = 'R'; tg[1] = 'G'; tg[2] = 'Z';
    memcpy(tg + 3, rg.c_str(), rg.size() + 1);
}

// Generate every read whose leftmost position lies in [p0, p1) of contig `ref_id`, sorted.
static void gen_segment(const Params& P, const ContigPlan& cp, int ref_id, int64_t p0, int64_t p1,
                        std::vector<uint8_t>& out, std::vector<RecInfo>& info, const std::vector<std::string>& rg_ids) {
    if (cp.n_pairs == 0) return;
    // pairs whose left read can start in [p0 - 1000, p1)
    int64_t k_lo = std::max<int64_t>(0, (int64_t)std::floor((double)(p0 - 1001) / cp.W) - 1);
    int64_t k_hi = std::min<int64_t>(cp.n_pairs, (int64_t)std::ceil((double)p1 / cp.W) + 1);
    struct Item { int64_t pos; int64_t k; bool right; Pair pr; };
    std::vector<Item> items;
    items.reserve((size_t)(k_hi - k_lo) * 2);
    for (int64_t k = k_lo; k < k_hi; ++k) {
        Pair pr = make_pair(P, cp, k);
        if (pr.s1 >= p0 && pr.s1 < p1) items.push_back({pr.s1, k, false, pr});
        // the right read may be shifted left to stay inside the contig (see emit_read); use the
        // same clamp here so that ownership is decided on the final position
        int64_t s2 = std::min<int64_t>(pr.s2, cp.len - P.read_len);
        if (s2 >= p0 && s2 < p1) items.push_back({s2, k, true, pr});
    }
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.pos < b.pos; });
    for (auto& it : items) {
        size_t before = info.size();
        emit_read(P, cp, ref_id, it.k, it.pr, it.right, out, info, rg_ids);
        // emit_read may clamp the position for reads with long ref spans near the contig end;
        // keep the stream sorted by falling back to a plain <L>M at the owned position.
        if (info[before].pos != it.pos) {
            // re-sort locally is not needed: positions only ever decrease by < 5000 at the very end
            // of a contig; handled by the final monotonic fix-up below.
        }
    }
    // monotonic fix-up (only triggers in the last few kb of a contig)
    bool sorted = true;
    for (size_t i = 1; i < info.size(); ++i) if (info[i].pos < info[i - 1].pos) { sorted = false; break; }
    if (!sorted) {
        std::vector<size_t> idx(info.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return info[a].pos < info[b].pos; });
        std::vector<uint8_t> o2;
        std::vector<RecInfo> i2;
        o2.reserve(out.size());
        for (size_t j : idx) {
            uint64_t b = info[j].off, e = j + 1 < info.size() ? info[j + 1].off : out.size();
            RecInfo ri = info[j];
            ri.off = o2.size();
            o2.insert(o2.end(), out.begin() + (long)b, out.begin() + (long)e);
            i2.push_back(ri);
        }
        out.swap(o2);
        info.swap(i2);
    }
}

static const uint8_t EOF_BLOCK[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                      0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// Optional libdeflate backend (runtime .so only in this image: bound through dlopen).
struct LibDeflate {
    void* (*alloc)(int) = nullptr;
    size_t (*compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    void (*free_)(void*) = nullptr;
    bool ok = false;
    LibDeflate() {
        void* h = dlopen("libdeflate.so.0", RTLD_NOW);
        if (!h) return;
        alloc = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
        compress = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_deflate_compress");
        free_ = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
        ok = alloc && compress && free_;
    }
};
static LibDeflate g_ld;
static bool g_use_libdeflate = false;

// Optional device backend: the product's BGZF writer (libsbx_depth.so: sbx_bgzf_compress, one lane per block on the GPU), bound
// through dlopen so that this harness tool does not link the product.  A whole wave of the stream is compressed per call.
struct DeviceCodec {
    int (*compress)(const uint8_t*, size_t, int, int, int, uint8_t*, size_t, size_t*, char*, size_t) = nullptr;
    bool ok = false;
    void open(const char* argv0) {
        std::string dir = argv0;
        size_t k = dir.rfind('/');
        dir = k == std::string::npos ? "." : dir.substr(0, k);
        const std::string cands[3] = {dir + "/../sambamba_amd/csrc/libsbx_depth.so", "sambamba_amd/csrc/libsbx_depth.so", "libsbx_depth.so"};
        for (auto& c : cands) {
            void* h = dlopen(c.c_str(), RTLD_NOW);
            if (!h) continue;
            compress = (decltype(compress))dlsym(h, "sbx_bgzf_compress");
            ok = compress != nullptr;
            if (ok) return;
        }
    }
};
static DeviceCodec g_dev;
static bool g_use_device = false;

static void bgzf_compress(const uint8_t* src, uint32_t n, int level, std::vector<uint8_t>& dst) {
    dst.resize(18 + compressBound(n) + 8 + 64);
    if (g_use_libdeflate) {
        thread_local void* comp = nullptr;
        thread_local int comp_level = -1;
        if (!comp || comp_level != level) { if (comp) g_ld.free_(comp); comp = g_ld.alloc(level); comp_level = level; }
        size_t clen = g_ld.compress(comp, src, n, dst.data() + 18, dst.size() - 18 - 8);
        if (clen == 0) { fprintf(stderr, "libdeflate compress failed\n"); exit(1); }
        uint32_t total = 18 + (uint32_t)clen + 8;
        if (total > 65536) { fprintf(stderr, "block too large\n"); exit(1); }
        static const uint8_t hdr[12] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0};
        memcpy(dst.data(), hdr, 12);
        dst[12] = 'B'; dst[13] = 'C'; dst[14] = 2; dst[15] = 0;
        dst[16] = (uint8_t)((total - 1) & 0xff); dst[17] = (uint8_t)((total - 1) >> 8);
        uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), src, n);
        memcpy(dst.data() + 18 + clen, &crc, 4);
        memcpy(dst.data() + 18 + clen + 4, &n, 4);
        dst.resize(total);
        return;
    }
    // one deflate state per thread, reset per block: deflateInit2 allocates ~270 KB, which glibc serves with mmap --
    // 256 threads doing that once per 64 KiB block serialise on the address-space lock (44 s for chr1 in round 2)
    struct ZState {
        z_stream zs;
        int level = -100;
        ~ZState() { if (level != -100) deflateEnd(&zs); }
    };
    thread_local ZState st;
    z_stream& zs = st.zs;
    if (st.level != level) {
        if (st.level != -100) deflateEnd(&zs);
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { fprintf(stderr, "deflateInit2 failed\n"); exit(1); }
        st.level = level;
    } else {
        deflateReset(&zs);
    }
    zs.next_in = (Bytef*)src;
    zs.avail_in = n;
    zs.next_out = dst.data() + 18;
    zs.avail_out = (uInt)(dst.size() - 18 - 8);
    int rc = deflate(&zs, Z_FINISH);
    if (rc != Z_STREAM_END) { fprintf(stderr, "deflate failed\n"); exit(1); }
    uint32_t clen = (uint32_t)zs.total_out;
    uint32_t total = 18 + clen + 8;
    if (total > 65536) { fprintf(stderr, "block too large\n"); exit(1); }
    static const uint8_t hdr[12] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0};
    memcpy(dst.data(), hdr, 12);
    dst[12] = 'B'; dst[13] = 'C'; dst[14] = 2; dst[15] = 0;
    dst[16] = (uint8_t)((total - 1) & 0xff); dst[17] = (uint8_t)((total - 1) >> 8);
    uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), src, n);
    memcpy(dst.data() + 18 + clen, &crc, 4);
    memcpy(dst.data() + 18 + clen + 4, &n, 4);
    dst.resize(total);
}

int main(int argc, char** argv) {
    Params P;
    std::string contigs = "chr1:248956422";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--out") P.out = val();
        else if (a == "--contigs") contigs = val();
        else if (a == "--coverage") P.coverage = atof(val().c_str());
        else if (a == "--read-len") P.read_len = atoi(val().c_str());
        else if (a == "--seed") P.seed = strtoull(val().c_str(), nullptr, 0);
        else if (a == "--threads") P.threads = atoi(val().c_str());
        else if (a == "--level") P.level = atoi(val().c_str());
        else if (a == "--insert-mean") P.ins_mu = atof(val().c_str());
        else if (a == "--insert-sd") P.ins_sd = atof(val().c_str());
        else if (a == "--samples") P.n_samples = atoi(val().c_str());
        else if (a == "--tie-free-overlaps") P.tie_free_overlaps = true;
        else if (a == "--segment") P.segment = atoll(val().c_str());
        else if (a == "--codec") {
            std::string c = val();
            g_use_libdeflate = (c == "libdeflate");
            if (g_use_libdeflate && !g_ld.ok) { fprintf(stderr, "libdeflate.so.0 not available, using zlib\n"); g_use_libdeflate = false; }
            if (c == "device") {
                g_dev.open(argv[0]);
                if (!g_dev.ok) { fprintf(stderr, "libsbx_depth.so (sbx_bgzf_compress) not found\n"); return 1; }
                g_use_device = true;
            }
        }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (P.out.empty()) { fprintf(stderr, "usage: gen_bam --out x.bam [--contigs n:len,n:len] [--coverage 30] ...\n"); return 2; }
    if (P.threads <= 0) {
        // the CPUs this process may actually use: a container's cgroup quota is often far below the CPUs it can see (16 of 256 on
        // the GPU boxes), and hundreds of compressing threads on 16 CPUs of quota mostly wait for each other
        P.threads = (int)std::max(1u, std::thread::hardware_concurrency());
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long long quota = atoll(q);
                if (quota > 0) P.threads = (int)std::max<long long>(1, std::min<long long>(P.threads, (quota + period - 1) / period + 2));
            }
            fclose(f);
        }
    }
    init_luts();
    {
        size_t p = 0;
        while (p < contigs.size()) {
            size_t e = contigs.find(',', p);
            if (e == std::string::npos) e = contigs.size();
            std::string item = contigs.substr(p, e - p);
            size_t c = item.rfind(':');
            P.contigs.push_back({item.substr(0, c), atoll(item.substr(c + 1).c_str())});
            p = e + 1;
        }
    }
    auto t0 = std::chrono::steady_clock::now();
    // plans
    std::vector<ContigPlan> plans;
    uint64_t pair_base = 0;
    for (auto& c : P.contigs) {
        ContigPlan cp;
        cp.len = c.len;
        cp.usable = std::max<int64_t>(1, c.len - 1000 - P.read_len);
        if (c.len < 2 * P.read_len + 1100) cp.usable = std::max<int64_t>(1, c.len - P.read_len);
        cp.n_pairs = (int64_t)std::llround(P.coverage * (double)c.len / (2.0 * P.read_len));
        if (c.len < P.read_len * 2) cp.n_pairs = 0;
        cp.W = cp.n_pairs ? (double)cp.usable / (double)cp.n_pairs : 1.0;
        cp.pair_base = pair_base;
        pair_base += (uint64_t)cp.n_pairs;
        plans.push_back(cp);
    }
    std::vector<std::string> rg_ids;
    for (int s = 0; s < P.n_samples; ++s) rg_ids.push_back("S" + std::to_string(s + 1));
    // header
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (auto& c : P.contigs) text += "@SQ\tSN:" + c.name + "\tLN:" + std::to_string(c.len) + "\n";
    for (auto& g : rg_ids) text += "@RG\tID:" + g + "\tSM:" + g + "\n";
    std::vector<uint8_t> carry;
    carry.insert(carry.end(), {'B', 'A', 'M', 1});
    put32(carry, (uint32_t)text.size());
    carry.insert(carry.end(), text.begin(), text.end());
    put32(carry, (uint32_t)P.contigs.size());
    for (auto& c : P.contigs) {
        put32(carry, (uint32_t)c.name.size() + 1);
        carry.insert(carry.end(), c.name.begin(), c.name.end());
        carry.push_back(0);
        put32(carry, (uint32_t)c.len);
    }

    FILE* fo = fopen(P.out.c_str(), "wb");
    if (!fo) { fprintf(stderr, "can't open %s\n", P.out.c_str()); return 1; }
    uint64_t file_off = 0;     // compressed offset written so far
    uint64_t stream_off = 0;   // uncompressed offset of carry[0]
    // BAI accumulators
    struct BinChunks { std::vector<std::pair<uint64_t, uint64_t>> chunks; };
    std::vector<std::vector<BinChunks>> bins(P.contigs.size(), std::vector<BinChunks>(37450));
    std::vector<std::vector<uint64_t>> lin(P.contigs.size());
    uint64_t n_reads = 0, n_bases = 0;

    // task list: (contig, p0, p1)
    struct Task { int ref; int64_t p0, p1; };
    std::vector<Task> tasks;
    for (size_t ci = 0; ci < P.contigs.size(); ++ci)
        for (int64_t p = 0; p < P.contigs[ci].len; p += P.segment) tasks.push_back({(int)ci, p, std::min(P.contigs[ci].len, p + P.segment)});

    const uint32_t BLK = 0xFF00;
    double tg=0,tc=0,tz=0,tb=0,tw=0; auto now=[]{return std::chrono::steady_clock::now();}; auto dt=[](auto a, auto b){return std::chrono::duration<double>(b-a).count();};
    size_t wave = (size_t)P.threads * 2;
    std::vector<std::vector<uint8_t>> segs(wave);
    std::vector<std::vector<RecInfo>> infos(wave);
    std::vector<uint8_t> stream;
    std::vector<std::vector<uint8_t>> cb;
    struct RecV { int ref; int32_t pos, end; uint32_t bin; uint64_t ubeg, uend; };
    std::vector<RecV> recs;
    for (size_t t0i = 0; t0i < tasks.size() || !carry.empty(); t0i += wave) {
        size_t t1i = std::min(tasks.size(), t0i + wave);
        size_t nt = t1i > t0i ? t1i - t0i : 0;
        for (size_t i = 0; i < wave; ++i) { segs[i].clear(); infos[i].clear(); }
        auto T0=now();
        {
            std::atomic<size_t> next(0);
            std::vector<std::thread> th;
            for (int w = 0; w < P.threads; ++w)
                th.emplace_back([&] {
                    for (;;) {
                        size_t i = next++;
                        if (i >= nt) break;
                        const Task& tk = tasks[t0i + i];
                        gen_segment(P, plans[(size_t)tk.ref], tk.ref, tk.p0, tk.p1, segs[i], infos[i], rg_ids);
                    }
                });
            for (auto& x : th) x.join();
        }
        auto T1=now(); tg+=dt(T0,T1);
        // rope view: carry + segs (no concatenation); blocks are compressed straight out of the
        // piece that holds them, or through a 64 KiB stitch buffer when they straddle pieces
        std::vector<const std::vector<uint8_t>*> pieces;
        std::vector<uint64_t> piece_off;  // offset of each piece inside this wave's stream
        uint64_t total_len = 0;
        pieces.push_back(&carry); piece_off.push_back(0); total_len += carry.size();
        std::vector<uint64_t> task_base(nt);
        for (size_t i = 0; i < nt; ++i) {
            task_base[i] = stream_off + total_len;
            pieces.push_back(&segs[i]); piece_off.push_back(total_len); total_len += segs[i].size();
        }
        piece_off.push_back(total_len);
        auto copy_range = [&](uint64_t from, uint64_t n, uint8_t* dst) {
            size_t pi = (size_t)(std::upper_bound(piece_off.begin(), piece_off.end(), from) - piece_off.begin()) - 1;
            while (n) {
                uint64_t in = from - piece_off[pi];
                uint64_t k = std::min<uint64_t>(n, pieces[pi]->size() - in);
                if (k) memcpy(dst, pieces[pi]->data() + in, k);
                dst += k; from += k; n -= k; ++pi;
            }
        };
        auto T2=now(); tc+=dt(T1,T2);
        bool last = t1i >= tasks.size();
        size_t nblk = last ? (size_t)((total_len + BLK - 1) / BLK) : (size_t)(total_len / BLK);
        if (cb.size() < nblk) cb.resize(nblk);
        if (g_use_device && nblk) {
            // the whole wave through the product's writer: one contiguous buffer in, BGZF blocks out (cut apart by their BSIZE)
            const uint64_t bytes = std::min<uint64_t>(total_len, (uint64_t)nblk * BLK);
            std::vector<uint8_t> flat((size_t)bytes), comp((size_t)bytes + (size_t)bytes / 2048 + 4096 + 64 * nblk);
            copy_range(0, bytes, flat.data());
            size_t clen = 0;
            char err[256] = {0};
            if (g_dev.compress(flat.data(), flat.size(), P.level, 0, -1, comp.data(), comp.size(), &clen, err, sizeof err) != 0) {
                fprintf(stderr, "sbx_bgzf_compress: %s\n", err);
                return 1;
            }
            size_t o = 0;
            for (size_t b = 0; b < nblk; ++b) {
                if (o + 18 > clen) { fprintf(stderr, "device writer returned too few blocks\n"); return 1; }
                const size_t bs = (size_t)(comp[o + 16] | (comp[o + 17] << 8)) + 1;
                cb[b].assign(comp.begin() + (long)o, comp.begin() + (long)(o + bs));
                o += bs;
            }
            if (o != clen) { fprintf(stderr, "device writer returned a different number of blocks\n"); return 1; }
        } else {
            std::atomic<size_t> next(0);
            std::vector<std::thread> th;
            for (int w = 0; w < P.threads; ++w)
                th.emplace_back([&] {
                    std::vector<uint8_t> stitch(BLK);
                    for (;;) {
                        size_t bi = next++;
                        if (bi >= nblk) break;
                        uint64_t from = (uint64_t)bi * BLK;
                        uint32_t n = (uint32_t)std::min<uint64_t>(BLK, total_len - from);
                        size_t pi = (size_t)(std::upper_bound(piece_off.begin(), piece_off.end(), from) - piece_off.begin()) - 1;
                        uint64_t in = from - piece_off[pi];
                        if (in + n <= pieces[pi]->size()) {
                            bgzf_compress(pieces[pi]->data() + in, n, P.level, cb[bi]);
                        } else {
                            copy_range(from, n, stitch.data());
                            bgzf_compress(stitch.data(), n, P.level, cb[bi]);
                        }
                    }
                });
            for (auto& x : th) x.join();
        }
        auto T3=now(); tz+=dt(T2,T3);
        // block table for voffsets
        std::vector<uint64_t> blk_coff(nblk + 1);
        uint64_t fo_off = file_off;
        for (size_t b = 0; b < nblk; ++b) { blk_coff[b] = fo_off; fo_off += cb[b].size(); }
        blk_coff[nblk] = fo_off;
        uint64_t consumed = std::min<uint64_t>(total_len, (uint64_t)nblk * BLK);
        auto voff = [&](uint64_t u) -> uint64_t {  // uncompressed stream offset -> virtual offset
            uint64_t rel = u - stream_off;
            size_t bq = (size_t)(rel / BLK);
            uint64_t in = rel % BLK;
            if (bq >= nblk) { bq = nblk; in = rel - (uint64_t)nblk * BLK; }  // lies in the carry: next wave's first block
            return (blk_coff[bq] << 16) | in;
        };
        for (size_t i = 0; i < nt; ++i) {
            int ref = tasks[t0i + i].ref;
            auto& li = lin[(size_t)ref];
            for (size_t j = 0; j < infos[i].size(); ++j) {
                const RecInfo& rv = infos[i][j];
                uint64_t ub = task_base[i] + rv.off;
                uint64_t ue = task_base[i] + (j + 1 < infos[i].size() ? infos[i][j + 1].off : segs[i].size());
                uint64_t vb = voff(ub), ve = voff(ue);
                auto& ch = bins[(size_t)ref][rv.bin].chunks;
                if (!ch.empty() && (ch.back().second >> 16) == (vb >> 16)) ch.back().second = ve;  // same block: extend
                else if (!ch.empty() && ch.back().second == vb) ch.back().second = ve;
                else ch.push_back({vb, ve});
                size_t w0 = (size_t)(rv.pos >> 14), w1 = (size_t)((std::max(rv.end, rv.pos + 1) - 1) >> 14);
                if (li.size() <= w1) li.resize(w1 + 1, 0);
                for (size_t w = w0; w <= w1; ++w) if (li[w] == 0 || vb < li[w]) li[w] = vb;
                ++n_reads;
                n_bases += (uint64_t)P.read_len;
            }
        }
        auto T4=now(); tb+=dt(T3,T4);
        for (size_t b = 0; b < nblk; ++b) fwrite(cb[b].data(), 1, cb[b].size(), fo);
        tw+=dt(T4,now());
        file_off = fo_off;
        stream_off += consumed;
        { std::vector<uint8_t> nc((size_t)(total_len - consumed)); if (!nc.empty()) copy_range(consumed, total_len - consumed, nc.data()); carry.swap(nc); }
        if (last) { carry.clear(); break; }
    }
    if (getenv("GEN_TIMING")) fprintf(stderr,"gen %.2f concat %.2f compress %.2f bai %.2f write %.2f\n",tg,tc,tz,tb,tw);
    fwrite(EOF_BLOCK, 1, 28, fo);
    fclose(fo);
    // BAI
    {
        FILE* fi = fopen((P.out + ".bai").c_str(), "wb");
        fwrite("BAI\1", 1, 4, fi);
        uint32_t n_ref = (uint32_t)P.contigs.size();
        fwrite(&n_ref, 4, 1, fi);
        for (size_t r = 0; r < P.contigs.size(); ++r) {
            uint32_t n_bin = 0;
            for (auto& b : bins[r]) n_bin += !b.chunks.empty();
            fwrite(&n_bin, 4, 1, fi);
            for (uint32_t id = 0; id < bins[r].size(); ++id) {
                auto& bc = bins[r][id];
                if (bc.chunks.empty()) continue;
                uint32_t nch = (uint32_t)bc.chunks.size();
                fwrite(&id, 4, 1, fi);
                fwrite(&nch, 4, 1, fi);
                for (auto& c : bc.chunks) { fwrite(&c.first, 8, 1, fi); fwrite(&c.second, 8, 1, fi); }
            }
            // fill empty linear-index slots with the next non-empty value to the left (htslib style)
            auto& li = lin[r];
            for (size_t w = 1; w < li.size(); ++w) if (li[w] == 0) li[w] = li[w - 1];
            uint32_t n_intv = (uint32_t)li.size();
            fwrite(&n_intv, 4, 1, fi);
            for (auto v : li) fwrite(&v, 8, 1, fi);
        }
        fclose(fi);
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"reads\": %llu, \"bases\": %llu, \"compressed_bytes\": %llu, \"uncompressed_bytes\": %llu, \"seconds\": %.2f, \"threads\": %d}\n",
           (unsigned long long)n_reads, (unsigned long long)n_bases, (unsigned long long)(file_off + 28),
           (unsigned long long)stream_off, secs, P.threads);
    return 0;
}
