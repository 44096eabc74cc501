// host_io.hpp -- host-side file format plumbing of libsbx_depth: BGZF block table, BAM header,
// BAI, BED / region strings and the -F filter compiler.  None of this is on the hot path; it
// restates the parts of the reference that stay on the host:
//   BGZF header     BioD/bio/core/bgzf/inputstream.d:54-199, constants.d:26-61
//   BAM header      BioD/bio/std/hts/bam/reader.d:101-125,580-598 ; sam/header.d (@HD SO, @RG ID/SM)
//   BAI             BioD/bio/std/hts/bam/baifile.d:75-80,126-169 ; bai/bin.d:56-76
//   chunk selection BioD/bio/std/hts/bam/randomaccessmanager.d:209-338
//   BED / region    sambamba/utils/common/bed.d:37-152 ; BioD/bio/core/region.d:97-246
//   -F filter       sambamba/utils/common/queryparser.d:232-483, filtering.d:86-214
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "common.hpp"

#include "regex_nfa.hpp"

namespace sbx {

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

struct FileMap {
    const uint8_t* data = nullptr;
    size_t size = 0;
    int fd = -1;
    std::string path;
    void open(const std::string& p) {
        path = p;
        fd = ::open(p.c_str(), O_RDONLY);
        if (fd < 0) throw Error(SBX_EIO, "can't open file " + p);
        struct stat st;
        if (fstat(fd, &st) != 0) throw Error(SBX_EIO, "can't stat " + p);
        size = (size_t)st.st_size;
        if (size) {
            void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) throw Error(SBX_EIO, "can't mmap " + p);
            data = (const uint8_t*)m;
        }
    }
    ~FileMap() {
        if (data) munmap((void*)data, size);
        if (fd >= 0) ::close(fd);
    }
};

// ---- BGZF block table ----------------------------------------------------------------------------
struct BlockTable {
    std::vector<uint64_t> coffset;    // file offset of each block
    std::vector<uint64_t> comp_off;   // file offset of its deflate payload
    std::vector<uint32_t> comp_len;
    std::vector<uint32_t> isize;
    std::vector<uint64_t> out_off;    // prefix sum of isize (size n+1)
    size_t size() const { return comp_len.size(); }
};

// Scans block headers from file offset `off` until the EOF block / end of file / `limit` (the stream stops at the first
// empty block, inputstream.d:393-394).  Returns the offset where the scan ended; *hit_eof: it ended at an empty block or at
// the end of the file rather than at `limit`.
inline uint64_t scan_bgzf_range(const uint8_t* f, size_t n, uint64_t off, uint64_t limit, BlockTable* tp, bool* hit_eof) {
    BlockTable& t = *tp;
    uint64_t uo = 0;
    *hit_eof = true;
    while (off < n) {
        if (off >= limit) { *hit_eof = false; break; }
        if (n - off < 4) break;  // short read of the magic == end of stream (inputstream.d:75-80)
        auto fail = [&](const std::string& m) {
            throw Error(SBX_EFORMAT, "Error reading BGZF block starting from offset " + std::to_string(off) + ": " + m);
        };
        const uint8_t* p = f + off;
        if (!(p[0] == 0x1f && p[1] == 0x8b && p[2] == 0x08 && p[3] == 0x04)) fail("wrong BGZF magic");
        if (n - off < 12) fail("unexpected end of file");
        uint32_t xlen = rd16(p + 10);
        if (n - off < 12 + (uint64_t)xlen) fail("unexpected end of file");
        bool found = false;
        uint32_t bsize = 0, len = 0;
        while (len < xlen) {
            if (len + 4 > xlen) fail("malformed gzip extra field");
            uint8_t si1 = p[12 + len], si2 = p[13 + len];
            uint32_t slen = rd16(p + 14 + len);
            if (si1 == 66 && si2 == 67) {
                if (slen != 2) fail("wrong BC subfield length: " + std::to_string(slen) + "; expected 2");
                if (found) fail("duplicate field with block size");
                bsize = rd16(p + 16 + len);
                found = true;
            }
            len += 4 + slen;
        }
        if (len != xlen) fail("total length of subfields in bytes (" + std::to_string(len) +
                              ") is not equal to gzip_extra_length (" + std::to_string(xlen) + ")");
        if (!found) fail("block size was not found in any subfield");
        int64_t cdata = (int64_t)bsize - (int64_t)xlen - 19;
        if (cdata < 0) fail("invalid block size");
        if (n - off < (uint64_t)bsize + 1) fail("unexpected end of file");
        uint32_t isz = rd32(p + 12 + xlen + cdata + 4);
        if (isz > 65536) fail("uncompressed block size exceeds 65536");
        if (isz == 0) break;  // EOF block
        t.coffset.push_back(off);
        t.comp_off.push_back(off + 12 + xlen);
        t.comp_len.push_back((uint32_t)cdata);
        t.isize.push_back(isz);
        t.out_off.push_back(uo);
        uo += isz;
        off += (uint64_t)bsize + 1;
    }
    t.out_off.push_back(uo);
    return off;
}

// The whole file.  `hints` (optional): file offsets that are known block starts -- the index names thousands of them (the
// coffset part of every BAI virtual offset) -- cut the serial header chain into pieces that are scanned by separate
// threads; a piece that does not end exactly where the next one starts (a stale index) falls back to the serial scan, so
// the hints can change the speed of the scan, never its result.
inline BlockTable scan_bgzf(const uint8_t* f, size_t n, const std::vector<uint64_t>* hints = nullptr) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    std::vector<uint64_t> cuts;
    size_t min_bytes = 64u << 20;
    if (const char* e = getenv("SBX_SCAN_PARALLEL_MIN")) min_bytes = (size_t)strtoull(e, nullptr, 10);      // (tests)
    if (hints && n > min_bytes && hw > 1) {
        std::vector<uint64_t> h;
        for (uint64_t x : *hints) if (x > 0 && x < n) h.push_back(x);
        std::sort(h.begin(), h.end());
        h.erase(std::unique(h.begin(), h.end()), h.end());
        const size_t pieces = std::min<size_t>({(size_t)16, (size_t)hw, h.size() + 1});
        for (size_t k = 1; k < pieces; ++k) {      // the hint closest to k/pieces of the file
            const uint64_t want = (uint64_t)((double)n * k / pieces);
            auto it = std::lower_bound(h.begin(), h.end(), want);
            if (it == h.end()) break;
            if (cuts.empty() || *it > cuts.back()) cuts.push_back(*it);
        }
    }
    if (!cuts.empty()) {
        const size_t P = cuts.size() + 1;
        std::vector<BlockTable> part(P);
        std::vector<uint64_t> ended(P, 0);
        std::vector<char> eof(P, 0), failed(P, 0);
        std::vector<std::thread> th;
        for (size_t k = 0; k < P; ++k)
            th.emplace_back([&, k] {
                try {
                    bool e = false;
                    ended[k] = scan_bgzf_range(f, n, k ? cuts[k - 1] : 0, k + 1 < P ? cuts[k] : ~0ull, &part[k], &e);
                    eof[k] = e ? 1 : 0;
                } catch (const std::exception&) { failed[k] = 1; }
            });
        for (auto& t : th) t.join();
        bool ok = true;
        size_t last = P - 1;          // the piece in which the stream ends
        for (size_t k = 0; k < P && ok; ++k) {
            if (failed[k]) { ok = false; break; }
            if (eof[k]) { last = k; break; }
            if (k + 1 < P && ended[k] != cuts[k]) ok = false;
        }
        if (ok) {
            BlockTable t;
            uint64_t uo = 0;
            size_t total = 0;
            for (size_t k = 0; k <= last; ++k) total += part[k].size();
            t.coffset.reserve(total); t.comp_off.reserve(total); t.comp_len.reserve(total); t.isize.reserve(total); t.out_off.reserve(total + 1);
            for (size_t k = 0; k <= last; ++k) {
                const BlockTable& q = part[k];
                t.coffset.insert(t.coffset.end(), q.coffset.begin(), q.coffset.end());
                t.comp_off.insert(t.comp_off.end(), q.comp_off.begin(), q.comp_off.end());
                t.comp_len.insert(t.comp_len.end(), q.comp_len.begin(), q.comp_len.end());
                t.isize.insert(t.isize.end(), q.isize.begin(), q.isize.end());
                for (size_t i = 0; i < q.size(); ++i) t.out_off.push_back(uo + q.out_off[i]);
                uo += q.out_off.back();
            }
            t.out_off.push_back(uo);
            return t;
        }
        // (a piece failed or the pieces do not join: the serial scan reports what is wrong, exactly as before)
    }
    BlockTable t;
    bool e = false;
    scan_bgzf_range(f, n, 0, ~0ull, &t, &e);
    return t;
}

// ---- BAM header ----------------------------------------------------------------------------------
struct RefSeq { std::string name; int32_t length = 0; };
struct ReadGroup { std::string id, sample; };
struct BamHeaderInfo {
    std::string text;
    std::vector<RefSeq> refs;
    std::string sorting_order = "unknown";
    std::vector<ReadGroup> read_groups;
    std::vector<std::string> sample_names;        // depth.d:1170-1181
    std::vector<uint16_t> rg_sample;              // sample id per read group
    uint64_t first_record_off = 0;
    int find_ref(const std::string& n) const {
        for (size_t i = 0; i < refs.size(); ++i) if (refs[i].name == n) return (int)i;
        return -1;
    }
};

// SamHeaderMerger.mergeSequenceDictionaries (BioD/bio/std/hts/utils/samheadermerger.d:127-177) for MultiBamReader
// (multireader.d:218-236): the dictionaries of the files become a directed graph -- one node per name in order of first
// appearance, one edge from every @SQ line to the next one of the same file, repeated edges kept -- and the merged dictionary
// is its topological order as DirectedGraph.topologicalSort produces it (utils/graph.d:57-87: Kahn's algorithm with a FIFO
// queue seeded with the nodes without predecessor in node order, successors visited in edge order).  Two lines with one name
// and different lengths cannot be merged; a cycle (two files listing two contigs in opposite orders) sends the reference to a
// strategy MultiBamReader does not implement ("NYI").  own_to_merged[f][id of file f] = id in the merged dictionary.
inline void merge_dictionaries(const std::vector<const std::vector<RefSeq>*>& dicts, std::vector<RefSeq>* merged,
                               std::vector<std::vector<int32_t>>* own_to_merged) {
    std::vector<RefSeq> nodes;
    std::map<std::string, size_t> index;
    std::vector<std::vector<size_t>> edges;
    auto node = [&](const RefSeq& r) -> size_t {
        auto it = index.find(r.name);
        if (it != index.end()) {
            if (nodes[it->second].length != r.length)
                throw Error(SBX_EINVAL, "can't merge SAM headers: one of references with name " + r.name + " has length " +
                                            std::to_string(nodes[it->second].length) + " while another one with the same name has length " +
                                            std::to_string(r.length));
            return it->second;
        }
        nodes.push_back(r);
        edges.emplace_back();
        return index[r.name] = nodes.size() - 1;
    };
    for (const std::vector<RefSeq>* d : dicts) {
        if (d->empty()) continue;
        size_t prev = node((*d)[0]);
        for (size_t k = 1; k < d->size(); ++k) {
            const size_t cur = node((*d)[k]);
            edges[prev].push_back(cur);
            prev = cur;
        }
    }
    std::vector<size_t> pred(nodes.size(), 0), queue;
    for (auto& e : edges) for (size_t v : e) ++pred[v];
    for (size_t v = 0; v < nodes.size(); ++v) if (!pred[v]) queue.push_back(v);
    std::vector<int32_t> new_id(nodes.size(), -1);
    merged->clear();
    for (size_t head = 0; head < queue.size(); ++head) {
        const size_t v = queue[head];
        new_id[v] = (int32_t)merged->size();
        merged->push_back(nodes[v]);
        for (size_t w : edges[v]) if (--pred[w] == 0) queue.push_back(w);
    }
    if (merged->size() < nodes.size())
        throw Error(SBX_EUNSUPPORTED, "the BAM files list their reference sequences in orders that contradict each other (the reference: NYI)");
    own_to_merged->assign(dicts.size(), {});
    for (size_t f = 0; f < dicts.size(); ++f)
        for (const RefSeq& r : *dicts[f]) (*own_to_merged)[f].push_back(new_id[index[r.name]]);
}

// Parses from the head of the inflated stream; returns false if `n` bytes were not enough
// (caller fetches more and retries).
inline bool parse_bam_header(const uint8_t* u, uint64_t n, uint64_t total, BamHeaderInfo* h) {
    auto need = [&](uint64_t k) -> bool { if (k > total) throw Error(SBX_EFORMAT, "BAM header is truncated"); return k <= n; };
    if (!need(12)) return false;
    if (memcmp(u, "BAM\1", 4) != 0) throw Error(SBX_EFORMAT, "Invalid file format: expected BAM\\1");
    uint64_t l_text = rd32(u + 4);
    if (!need(12 + l_text)) return false;
    h->text.assign((const char*)u + 8, l_text);
    size_t z = h->text.find('\0');
    if (z != std::string::npos) h->text.resize(z);
    uint64_t o = 8 + l_text;
    uint32_t n_ref = rd32(u + o);
    o += 4;
    h->refs.clear();
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!need(o + 4)) return false;
        uint64_t l_name = rd32(u + o);
        if (!need(o + 4 + l_name + 4)) return false;
        RefSeq r;
        r.name.assign((const char*)u + o + 4, l_name);
        while (!r.name.empty() && r.name.back() == '\0') r.name.pop_back();
        r.length = (int32_t)rd32(u + o + 4 + l_name);
        h->refs.push_back(r);
        o += 8 + l_name;
    }
    h->first_record_off = o;
    // text: @HD SO, @RG ID/SM (insertion order, first ID wins -- sam/header.d:345-354)
    h->sorting_order = "unknown";
    h->read_groups.clear();
    size_t p = 0;
    while (p < h->text.size()) {
        size_t e = h->text.find('\n', p);
        if (e == std::string::npos) e = h->text.size();
        std::string line = h->text.substr(p, e - p);
        p = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.size() < 3 || line[0] != '@') continue;
        std::string ty = line.substr(1, 2);
        if (ty != "HD" && ty != "RG") continue;
        std::map<std::string, std::string> kv;
        size_t q = 3;
        while (q < line.size()) {
            if (line[q] == '\t') { ++q; continue; }
            size_t t = line.find('\t', q);
            if (t == std::string::npos) t = line.size();
            if (t - q >= 3 && line[q + 2] == ':') {
                std::string k = line.substr(q, 2);
                if (!kv.count(k)) kv[k] = line.substr(q + 3, t - q - 3);
            }
            q = t;
        }
        if (ty == "HD") { if (kv.count("SO")) h->sorting_order = kv["SO"]; }
        else {
            ReadGroup g;
            g.id = kv.count("ID") ? kv["ID"] : "";
            g.sample = kv.count("SM") ? kv["SM"] : "";
            bool dup = false;
            for (auto& x : h->read_groups) dup |= x.id == g.id;
            if (!dup) h->read_groups.push_back(g);
        }
    }
    h->sample_names.clear();
    h->rg_sample.clear();
    std::map<std::string, uint16_t> sm2id;
    for (auto& g : h->read_groups) {
        if (!sm2id.count(g.sample)) { sm2id[g.sample] = (uint16_t)h->sample_names.size(); h->sample_names.push_back(g.sample); }
        h->rg_sample.push_back(sm2id[g.sample]);
    }
    if (h->sample_names.empty()) h->sample_names.push_back("*");
    return true;
}

// ---- BAI -----------------------------------------------------------------------------------------
struct BaiChunk { uint64_t beg, end; };
struct BaiBin { uint32_t id; std::vector<BaiChunk> chunks; };
struct BaiRef {
    std::vector<BaiBin> bins;
    std::vector<uint64_t> ioffsets;
    uint64_t min_offset(int64_t position) const {   // baifile.d:75-80
        int64_t pos = std::max<int64_t>(0, position);
        int64_t i = std::min<int64_t>(pos / 16384, (int64_t)ioffsets.size() - 1);
        return i < 0 ? 0 : ioffsets[(size_t)i];
    }
};
struct BaiIndex { std::vector<BaiRef> refs; bool loaded = false; };

inline bool load_bai(const std::string& bam_path, BaiIndex* out) {
    std::string cands[2] = {bam_path + ".bai", bam_path.size() > 4 ? bam_path.substr(0, bam_path.size() - 4) + ".bai" : bam_path + ".bai"};
    for (auto& c : cands) {
        if (access(c.c_str(), R_OK) != 0) continue;
        FileMap f;
        f.open(c);
        size_t p = 0;
        auto need = [&](size_t k) { if (p + k > f.size) throw Error(SBX_EFORMAT, "BAI file is truncated"); };
        need(8);
        if (memcmp(f.data, "BAI\1", 4) != 0) throw Error(SBX_EFORMAT, "Invalid file format: expected BAI\\1");
        int32_t n_ref = (int32_t)rd32(f.data + 4);
        p = 8;
        out->refs.assign((size_t)std::max(0, n_ref), BaiRef());
        for (auto& r : out->refs) {
            need(4);
            int32_t n_bin = (int32_t)rd32(f.data + p); p += 4;
            r.bins.resize((size_t)std::max(0, n_bin));
            for (auto& b : r.bins) {
                need(8);
                b.id = rd32(f.data + p);
                int32_t n_chunk = (int32_t)rd32(f.data + p + 4); p += 8;
                b.chunks.resize((size_t)std::max(0, n_chunk));
                for (auto& c2 : b.chunks) { need(16); c2.beg = rd64(f.data + p); c2.end = rd64(f.data + p + 8); p += 16; }
            }
            need(4);
            int32_t n_intv = (int32_t)rd32(f.data + p); p += 4;
            r.ioffsets.resize((size_t)std::max(0, n_intv));
            for (auto& o : r.ioffsets) { need(8); o = rd64(f.data + p); p += 8; }
        }
        out->loaded = true;
        return true;
    }
    return false;
}

// RandomAccessManager.getGroupChunks (randomaccessmanager.d:247-294): regions on one reference,
// sorted + merged.  Bin ids > 37448 (samtools' metadata pseudo-bin 37450) are skipped explicitly.
inline std::vector<BaiChunk> group_chunks(const BaiIndex& bai, const std::vector<sbx_region>& regs) {
    std::vector<bool> bits(37449, false);
    bits[0] = true;
    for (auto& r : regs) {
        uint32_t beg = r.start, end = r.end - 1, k;
        for (k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) bits[k] = true;
        for (k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) bits[k] = true;
        for (k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) bits[k] = true;
        for (k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) bits[k] = true;
        for (k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) if (k < bits.size()) bits[k] = true;
    }
    uint32_t ref = regs.front().ref_id;
    std::vector<BaiChunk> out;
    if (ref >= bai.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
    const BaiRef& ix = bai.refs[ref];
    uint64_t mo = ix.min_offset(regs.front().start);
    for (auto& b : ix.bins) {
        if (b.id >= bits.size() || !bits[b.id]) continue;
        for (BaiChunk c : b.chunks) if (c.end > mo) { if (c.beg < mo) c.beg = mo; out.push_back(c); }
    }
    std::sort(out.begin(), out.end(), [](const BaiChunk& a, const BaiChunk& b) { return a.beg != b.beg ? a.beg < b.beg : a.end < b.end; });
    std::vector<BaiChunk> merged;
    for (auto& c : out) {
        if (!merged.empty() && merged.back().end >= c.beg) merged.back().end = std::max(merged.back().end, c.end);
        else merged.push_back(c);
    }
    return merged;
}

// ---- BED / region strings ------------------------------------------------------------------------
struct BedInterval { std::string chr; long beg = 0, end = 0; };

inline std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) ++i;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) ++j;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

// readIntervals (bed.d:59-98).  Returns false if the file cannot be read or a coordinate does not
// parse (the reference then treats the argument as a region string, depth.d:1194-1208).
inline bool read_bed_file(const std::string& path, std::vector<BedInterval>* ivs, std::vector<std::string>* lines,
                          std::vector<size_t>* line_of_iv = nullptr) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
    fclose(fp);
    auto to_long = [](const std::string& s, long* v) {
        if (s.empty()) return false;
        size_t i = 0; bool neg = false;
        if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; }
        if (i >= s.size()) return false;
        long x = 0;
        for (; i < s.size(); ++i) { if (s[i] < '0' || s[i] > '9') return false; x = x * 10 + (s[i] - '0'); }
        *v = neg ? -x : x;
        return true;
    };
    size_t p = 0;
    while (p <= text.size()) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        std::string str = text.substr(p, e - p);
        p = e + 1;
        auto f = split_ws(str);
        if (f.size() < 2) continue;
        BedInterval iv;
        iv.chr = f[0];
        if (!to_long(f[1], &iv.beg)) return false;
        if (f.size() >= 3) { if (!to_long(f[2], &iv.end)) return false; } else iv.end = iv.beg + 1;
        if (iv.beg == iv.end) iv.end = iv.beg + 1;
        if (iv.beg < iv.end) { ivs->push_back(iv); if (line_of_iv) line_of_iv->push_back(lines->size()); }
        lines->push_back(str);
    }
    return true;
}

// parseBed non_overlapping=true (bed.d:37-55,128-141): per contig sort by beg, merge when
// cur.end >= next.beg; contigs missing from the BAM are dropped; result sorted.
inline std::vector<sbx_region> bed_merged(const std::vector<BedInterval>& ivs, const BamHeaderInfo& h) {
    std::map<std::string, std::vector<BedInterval>> by;
    for (auto& iv : ivs) by[iv.chr].push_back(iv);
    std::vector<sbx_region> regs;
    for (auto& kv : by) {
        int id = h.find_ref(kv.first);
        if (id < 0) continue;
        auto& v = kv.second;
        std::stable_sort(v.begin(), v.end(), [](const BedInterval& a, const BedInterval& b) { return a.beg < b.beg; });
        BedInterval cur = v[0];
        for (size_t i = 1; i < v.size(); ++i) {
            if (cur.end >= v[i].beg) cur.end = std::max(cur.end, v[i].end);
            else { regs.push_back({(uint32_t)id, (uint32_t)cur.beg, (uint32_t)cur.end}); cur = v[i]; }
        }
        regs.push_back({(uint32_t)id, (uint32_t)cur.beg, (uint32_t)cur.end});
    }
    std::sort(regs.begin(), regs.end(), [](const sbx_region& a, const sbx_region& b) {
        if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
        if (a.start != b.start) return a.start < b.start;
        return a.end < b.end;
    });
    return regs;
}
inline std::vector<sbx_region> bed_raw(const std::vector<BedInterval>& ivs, const BamHeaderInfo& h) {
    std::vector<sbx_region> regs;
    for (auto& iv : ivs) {
        int id = h.find_ref(iv.chr);
        if (id < 0) continue;
        regs.push_back({(uint32_t)id, (uint32_t)iv.beg, (uint32_t)iv.end});
    }
    return regs;
}

// parseRegion (BioD/bio/core/region.d:97-246): "ref[:beg[-end]]", commas allowed in numbers,
// beg 1-based -> 0-based, default [0, uint.max).
struct RegionString { std::string reference; uint32_t beg = 0, end = 0xFFFFFFFFu; };
inline RegionString parse_region_string(const std::string& s) {
    RegionString r;
    auto is_num = [](const std::string& t) {
        bool digit = false;
        for (char c : t) { if (c >= '0' && c <= '9') digit = true; else if (c != ',') return false; }
        return digit;
    };
    auto num = [](const std::string& t) { long v = 0; for (char c : t) if (c != ',') v = v * 10 + (c - '0'); return v; };
    size_t colon = s.rfind(':');
    if (colon != std::string::npos) {
        std::string tail = s.substr(colon + 1);
        size_t dash = tail.find('-');
        std::string a = dash == std::string::npos ? tail : tail.substr(0, dash);
        std::string b = dash == std::string::npos ? "" : tail.substr(dash + 1);
        if (is_num(a) && (dash == std::string::npos || is_num(b))) {
            r.reference = s.substr(0, colon);
            r.beg = (uint32_t)(num(a) - 1);
            if (dash != std::string::npos) r.end = (uint32_t)num(b);
            return r;
        }
    }
    r.reference = s;
    return r;
}

// ---- -F filter compiler --------------------------------------------------------------------------
// Pratt parser with the reference's binding powers (comparison 110 > not 100 > and 80 > or 60,
// queryparser.d:424-483) emitting the postfix program of sbx_filter.
class FilterCompiler {
public:
    FilterCompiler(const std::string& s, sbx_filter* out) : s_(s), out_(out) { out_->n_ops = 0; }
    void compile() {
        expr(0);
        skip();
        if (p_ != s_.size()) throw Error(SBX_EUNSUPPORTED, "filter: unexpected input at '" + s_.substr(p_) + "'");
    }

private:
    std::string s_;
    size_t p_ = 0;
    sbx_filter* out_;
    // 'text' with \' as the only escape (queryparser.d:343-377) -> (offset | length << 32) into the string pool
    int64_t string_literal() {
        skip();
        if (p_ >= s_.size() || s_[p_] != '\'') throw Error(SBX_EUNSUPPORTED, "filter: string literal expected");
        ++p_;
        std::string v;
        for (;;) {
            if (p_ >= s_.size()) throw Error(SBX_EUNSUPPORTED, "filter: unterminated string literal");
            char ch = s_[p_++];
            if (ch == '\\' && p_ < s_.size() && s_[p_] == '\'') { v.push_back('\''); ++p_; continue; }   // the only escape (queryparser.d:343-377)
            if (ch == '\'') break;
            v.push_back(ch);
        }
        if (pool_ + v.size() > SBX_FILTER_STRINGS) throw Error(SBX_EUNSUPPORTED, "filter: string literals too long for the device program");
        memcpy(out_->strings + pool_, v.data(), v.size());
        const int64_t r = (int64_t)pool_ | ((int64_t)v.size() << 32);
        pool_ += v.size();
        return r;
    }
    size_t pool_ = 0;
    // `=~ /pattern/options` after a string field or a tag (queryparser.d:427-476): emits a REGEX op
    bool regex_condition(uint8_t field, uint32_t key) {
        skip();
        if (s_.compare(p_, 2, "=~") != 0) return false;
        p_ += 2;
        skip();
        if (p_ >= s_.size() || s_[p_] != '/') throw Error(SBX_EUNSUPPORTED, "filter: regular expression literal /.../ expected after =~");
        size_t i = p_ + 1;
        std::string pat;
        for (;; ++i) {
            if (i >= s_.size()) throw Error(SBX_EUNSUPPORTED, "filter: unterminated regular expression");
            if (s_[i] == '\\' && i + 1 < s_.size() && s_[i + 1] == '/') { pat += "\\/"; ++i; continue; }
            if (s_[i] == '/') break;
            pat.push_back(s_[i]);
        }
        ++i;
        bool icase = false;
        while (i < s_.size() && !isspace((unsigned char)s_[i]) && s_[i] != ')') {
            if (s_[i] == 'i') icase = true;
            else throw Error(SBX_EUNSUPPORTED, std::string("filter: regular expression option '") + s_[i] + "' is not supported on the device path");
            ++i;
        }
        p_ = i;
        if (out_->n_regex >= SBX_FILTER_REGEXES) throw Error(SBX_EUNSUPPORTED, "filter: more than two regular expressions");
        RegexCompiler rc(pat, icase, &out_->regex[out_->n_regex]);
        rc.compile();
        emit(15, key, field, 0, out_->n_regex);
        out_->n_regex += 1;
        return true;
    }
    int cmp_op() {      // 0 > 1 < 2 >= 3 <= 4 == 5 !=, or -1
        static const char* ops[] = {">=", "<=", "==", "!=", ">", "<"};
        static const int opid[] = {2, 3, 4, 5, 0, 1};
        for (int k = 0; k < 6; ++k) if (eat(ops[k], false)) return opid[k];
        return -1;
    }
    void emit(uint8_t kind, uint32_t mask = 0, uint8_t field = 0, uint8_t cmp = 0, int64_t value = 0) {
        if (out_->n_ops >= SBX_FILTER_MAX_OPS) throw Error(SBX_EUNSUPPORTED, "filter: expression too long for the device program");
        sbx_filter_op& o = out_->ops[out_->n_ops++];
        o.kind = kind; o.field = field; o.cmp = cmp; o.pad = 0; o.mask = mask; o.value = value;
    }
    void skip() { while (p_ < s_.size() && isspace((unsigned char)s_[p_])) ++p_; }
    bool eat(const char* w, bool word) {
        skip();
        size_t n = strlen(w);
        if (s_.compare(p_, n, w) != 0) return false;
        if (word && p_ + n < s_.size() && (isalnum((unsigned char)s_[p_ + n]) || s_[p_ + n] == '_')) return false;
        p_ += n;
        return true;
    }
    void primary() {
        skip();
        if (eat("(", false)) { expr(0); if (!eat(")", false)) throw Error(SBX_EUNSUPPORTED, "filter: missing ')'"); return; }
        if (eat("not", true)) { expr(100); emit(5); return; }
        static const struct { const char* name; uint32_t mask; } flags[] = {
            {"proper_pair", 0x2}, {"paired", 0x1}, {"unmapped", 0x4}, {"mate_is_unmapped", 0x8},
            {"mate_is_reverse_strand", 0x20}, {"reverse_strand", 0x10}, {"first_of_pair", 0x40},
            {"second_of_pair", 0x80}, {"secondary_alignment", 0x100}, {"failed_quality_control", 0x200},
            {"duplicate", 0x400}, {"supplementary", 0x800}};
        for (auto& f : flags) if (eat(f.name, true)) { emit(0, f.mask); return; }
        if (eat("chimeric", true)) { emit(1); return; }
        static const char* fields[] = {"ref_id", "position", "mapping_quality", "sequence_length",
                                       "mate_ref_id", "mate_position", "template_length", "avg_base_quality"};
        for (int i = 0; i < 8; ++i)
            if (eat(fields[i], true)) {
                static const char* ops[] = {">=", "<=", "==", "!=", ">", "<"};
                static const uint8_t opid[] = {2, 3, 4, 5, 0, 1};
                for (int k = 0; k < 6; ++k)
                    if (eat(ops[k], false)) {
                        skip();
                        size_t q = p_;
                        if (q < s_.size() && (s_[q] == '-' || s_[q] == '+')) ++q;
                        size_t d0 = q;
                        while (q < s_.size() && isdigit((unsigned char)s_[q])) ++q;
                        if (q == d0) throw Error(SBX_EUNSUPPORTED, "filter: integer expected");
                        emit(2, 0, (uint8_t)i, opid[k], atoll(s_.substr(p_, q - p_).c_str()));
                        p_ = q;
                        return;
                    }
                throw Error(SBX_EUNSUPPORTED, "filter: comparison operator expected");
            }
        // string fields (StringFieldFilter, filtering.d:255-273)
        if (eat("read_name", true)) {
            if (regex_condition(0, 0)) return;
            const int op = cmp_op();
            if (op < 0) throw Error(SBX_EUNSUPPORTED, "filter: regex conditions are outside the device-compilable subset");
            emit(10, 0, 0, (uint8_t)op, string_literal());
            return;
        }
        for (int which = 0; which < 2; ++which)
            if (eat(which ? "cigar" : "sequence", true)) {      // compared as text (cmp(a.sequence, v), a.cigarString())
                if (regex_condition((uint8_t)(1 + which), 0)) return;
                const int op = cmp_op();
                if (op < 0) throw Error(SBX_EUNSUPPORTED, "filter: regex conditions are outside the device-compilable subset");
                emit((uint8_t)(13 + which), 0, 0, (uint8_t)op, string_literal());
                return;
            }
        for (int which = 0; which < 2; ++which)
            if (eat(which ? "mate_ref_name" : "ref_name", true)) {
                if (regex_condition((uint8_t)(4 + which), 0)) return;
                const int op = cmp_op();
                if (op != 4 && op != 5) throw Error(SBX_EUNSUPPORTED, "filter: reference names can be compared with == and != on the device path");
                emit(11, 0, (uint8_t)which, (uint8_t)op, string_literal());
                return;
            }
        if (eat("strand", true)) {          // a.strand is '+' or '-' (read.d strand property)
            const int op = cmp_op();
            if (op != 4 && op != 5) throw Error(SBX_EUNSUPPORTED, "filter: strand can be compared with == and != on the device path");
            const int64_t lit = string_literal();
            const size_t off = (size_t)(lit & 0xFFFFFFFF), len = (size_t)(lit >> 32);
            const char ch = len ? out_->strings[off] : 0;      // the reference compares with the first character
            if (ch == '-') emit(0, 0x10);
            else if (ch == '+') { emit(0, 0x10); emit(5); }
            else emit(12);
            if (op == 5) emit(5);
            return;
        }
        if (eat("[", false)) {      // [XX] op integer | [XX] == null | [XX] != null (queryparser.d:285-300)
            if (p_ + 3 > s_.size() || s_[p_ + 2] != ']') throw Error(SBX_EUNSUPPORTED, "filter: tag name of two characters expected");
            const uint32_t key = (uint8_t)s_[p_] | ((uint32_t)(uint8_t)s_[p_ + 1] << 8);
            p_ += 3;
            if (regex_condition(3, key)) return;
            static const char* ops[] = {">=", "<=", "==", "!=", ">", "<"};
            static const uint8_t opid[] = {2, 3, 4, 5, 0, 1};
            for (int k = 0; k < 6; ++k)
                if (eat(ops[k], false)) {
                    if (eat("null", true)) {
                        if (opid[k] != 4 && opid[k] != 5) throw Error(SBX_EUNSUPPORTED, "filter: only == and != can be used with null");
                        emit(8, key, 0, opid[k], 0);
                        return;
                    }
                    skip();
                    if (p_ < s_.size() && s_[p_] == '\'') { emit(9, key, 0, opid[k], string_literal()); return; }   // StringTagFilter
                    size_t q = p_;
                    if (q < s_.size() && (s_[q] == '-' || s_[q] == '+')) ++q;
                    size_t d0 = q;
                    while (q < s_.size() && isdigit((unsigned char)s_[q])) ++q;
                    if (q == d0) throw Error(SBX_EUNSUPPORTED, "filter: regex tag comparisons are outside the device-compilable subset");
                    emit(7, key, 0, opid[k], atoll(s_.substr(p_, q - p_).c_str()));
                    p_ = q;
                    return;
                }
            throw Error(SBX_EUNSUPPORTED, "filter: comparison operator expected");
        }
        throw Error(SBX_EUNSUPPORTED, "filter: '" + s_.substr(p_) + "' is outside the device-compilable subset "
                                      "(flags, integer fields, integer tags, tag existence, and/or/not)");
    }
    void expr(int rbp) {
        primary();
        for (;;) {
            skip();
            size_t save = p_;
            if (rbp < 80 && eat("and", true)) { expr(80); emit(3); }
            else if (rbp < 60 && eat("or", true)) { expr(60); emit(4); }
            else { p_ = save; return; }
        }
    }
};

}  // namespace sbx
