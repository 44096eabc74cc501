// ============================================================================
// oracle/bamio.hpp -- TEST INFRASTRUCTURE ONLY (CPU restatement, not the product)
//
// BGZF / BAM / BAI / BED readers used by the CPU oracle (depth_oracle.cpp).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  The product (sambamba_amd/csrc) never includes,
// links or calls this code.
//
// Each function cites the reference file:line (relative to /root/reference)
// whose behaviour it restates.  Nothing here is copied from the D sources: the
// reference is D, this is an independent C++ restatement of its behaviour.
// ============================================================================
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace orc {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

static inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

// ---------------------------------------------------------------------------
// Whole-file read-only mapping
// ---------------------------------------------------------------------------
struct MappedFile {
    const uint8_t* data = nullptr;
    size_t size = 0;
    int fd = -1;
    explicit MappedFile(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error("can't open file " + path);
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); throw Error("can't stat " + path); }
        size = (size_t)st.st_size;
        if (size) {
            void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) { ::close(fd); throw Error("can't mmap " + path); }
            data = (const uint8_t*)p;
        }
    }
    ~MappedFile() {
        if (data) munmap((void*)data, size);
        if (fd >= 0) ::close(fd);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
};

// ---------------------------------------------------------------------------
// BGZF block header (BioD/bio/core/bgzf/inputstream.d:54-199, constants.d:26-61)
//   1F 8B 08 04 | mtime u32 | xfl | os | xlen u16 | subfields ... | cdata | crc32 | isize
//   'B','C',slen=2 subfield carries bsize = total block size - 1;
//   cdata_size = bsize - xlen - 19.
// ---------------------------------------------------------------------------
struct BgzfBlock {
    uint64_t coffset = 0;     // file offset of the block (start_offset, block.d:42-93)
    uint32_t total = 0;       // bsize + 1
    uint32_t cdata_off = 0;   // offset of deflate payload inside the block
    uint32_t cdata_size = 0;
    uint32_t crc32 = 0;
    uint32_t isize = 0;       // input_size (uncompressed length, <= 65536)
};

// Returns false at clean end of file (inputstream.d:72-80: zero bytes available).
static inline bool parse_bgzf_header(const uint8_t* file, size_t file_size, uint64_t off, BgzfBlock* out) {
    if (off >= file_size) return false;
    auto fail = [&](const std::string& msg) -> bool {
        throw Error("Error reading BGZF block starting from offset " + std::to_string(off) + ": " + msg);
    };
    if (file_size - off < 4) return false;  // inputstream.d:75-80 (short read of magic => EOF)
    const uint8_t* p = file + off;
    if (!(p[0] == 0x1f && p[1] == 0x8b && p[2] == 0x08 && p[3] == 0x04)) fail("wrong BGZF magic");
    if (file_size - off < 12) fail("unexpected end of file");
    uint16_t xlen = le16(p + 10);
    if (file_size - off < (uint64_t)12 + xlen) fail("unexpected end of file");
    bool found = false;
    uint16_t bsize = 0;
    size_t len = 0;
    const uint8_t* x = p + 12;
    while (len < xlen) {  // inputstream.d:104-141
        if (len + 4 > xlen) fail("malformed extra field");
        uint8_t si1 = x[len], si2 = x[len + 1];
        uint16_t slen = le16(x + len + 2);
        if (si1 == 66 && si2 == 67) {
            if (slen != 2) fail("wrong BC subfield length: " + std::to_string(slen) + "; expected 2");
            if (found) fail("duplicate field with block size");
            bsize = le16(x + len + 4);
            found = true;
        }
        len += 4 + (size_t)slen;
    }
    if (len != xlen) fail("total length of subfields in bytes (" + std::to_string(len) +
                          ") is not equal to gzip_extra_length (" + std::to_string(xlen) + ")");
    if (!found) fail("block size was not found in any subfield");
    // inputstream.d:155-160: cdata_size = bsize - gzip_extra_length - 19
    int cdata = (int)bsize - (int)xlen - 19;
    if (cdata < 0 || cdata > (int)bsize) fail("invalid block size");
    if (file_size - off < (uint64_t)bsize + 1) fail("unexpected end of file");
    out->coffset = off;
    out->total = (uint32_t)bsize + 1;
    out->cdata_off = 12u + xlen;
    out->cdata_size = (uint32_t)cdata;
    out->crc32 = le32(p + out->cdata_off + cdata);
    out->isize = le32(p + out->cdata_off + cdata + 4);
    if (out->isize > 65536) fail("uncompressed block size exceeds 65536");
    return true;
}

// decompressBgzfBlock (BioD/bio/core/bgzf/block.d:127-216): raw inflate, windowBits -15,
// Z_FINISH; CRC is only asserted in debug builds (block.d:187), i.e. not checked.
static inline void inflate_block(const uint8_t* cdata, uint32_t cdata_size, uint8_t* out, uint32_t isize) {
    if (isize == 0) return;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    zs.next_in = (Bytef*)cdata;
    zs.avail_in = cdata_size;
    int err = inflateInit2(&zs, -15);
    if (err != Z_OK) throw Error("zlib inflateInit2 failed");
    zs.next_out = out;
    zs.avail_out = isize;
    err = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    if (err != Z_STREAM_END) throw Error("zlib inflate failed (" + std::to_string(err) + ")");
}

// ---------------------------------------------------------------------------
// Work list of blocks with skip_start / skip_end: restates
//   StreamSupplier        (inputstream.d:216-249)  -- whole file, sequential
//   StreamChunksSupplier  (inputstream.d:257-345)  -- BAI chunk list
// ---------------------------------------------------------------------------
struct BlockJob {
    BgzfBlock blk;
    uint32_t skip_start = 0;
    uint32_t skip_end = 0;
};

struct Chunk {  // BioD/bio/core/bgzf/chunk.d:28-41
    uint64_t beg = 0, end = 0;  // virtual offsets: coffset<<16 | uoffset (virtualoffset.d:29-60)
    bool operator<(const Chunk& o) const { return beg != o.beg ? beg < o.beg : end < o.end; }
};

static inline std::vector<BlockJob> jobs_whole_file(const MappedFile& f, uint64_t start = 0) {
    std::vector<BlockJob> jobs;
    uint64_t off = start;
    BgzfBlock b;
    while (parse_bgzf_header(f.data, f.size, off, &b)) {
        // BgzfInputStream.fillNextBlock stops at the first empty block (inputstream.d:393-394).
        if (b.isize == 0) break;
        BlockJob j;
        j.blk = b;
        jobs.push_back(j);
        off += b.total;
    }
    return jobs;
}

static inline std::vector<BlockJob> jobs_from_chunks(const MappedFile& f, std::vector<Chunk> chunks) {
    std::vector<BlockJob> jobs;
    size_t ci = 0;
    uint64_t pos = 0;
    // moveToNextChunk (inputstream.d:262-276): among chunks starting in the same block keep
    // the last one but with the first one's beginning.
    auto move_to_next = [&]() {
        if (ci >= chunks.size()) return;
        size_t i = ci + 1;
        uint64_t beg = chunks[ci].beg;
        for (; i < chunks.size(); ++i)
            if ((chunks[i].beg >> 16) > (chunks[ci].beg >> 16)) break;
        ci = i - 1;
        chunks[ci].beg = beg;
        pos = chunks[ci].beg >> 16;
    };
    move_to_next();
    while (ci < chunks.size()) {
        BgzfBlock b;
        if (!parse_bgzf_header(f.data, f.size, pos, &b)) break;
        pos += b.total;
        uint64_t offset = b.coffset;
        BlockJob j;
        j.blk = b;
        j.skip_start = (offset == (chunks[ci].beg >> 16)) ? (uint32_t)(chunks[ci].beg & 0xFFFF) : 0;
        long skip_end = 0;  // may equal 65536 (inputstream.d:316-321)
        if (offset == (chunks[ci].end >> 16)) skip_end = (long)b.isize - (long)(chunks[ci].end & 0xFFFF);
        j.skip_end = (uint32_t)(uint16_t)skip_end;  // cast(ushort) in the reference
        if (offset >= (chunks[ci].end >> 16)) {
            ++ci;
            move_to_next();
        }
        // inputstream.d:330-335: chunk ended exactly on a block edge -> skip that block
        if (b.isize > 0 && skip_end == (long)b.isize) continue;
        if (b.isize == 0) break;  // empty block ends the stream (inputstream.d:393-394)
        jobs.push_back(j);
    }
    return jobs;
}

// ---------------------------------------------------------------------------
// In-order decompressed byte stream over a job list with T inflate workers:
// restates BgzfInputStream (inputstream.d:349-541): a ring of inflate tasks on a
// thread pool, consumed strictly in order by a single reader thread.
// With n_threads == 0 everything runs on the caller (depth.d:1081,1154).
// ---------------------------------------------------------------------------
class InflateStream {
public:
    InflateStream(const MappedFile& f, std::vector<BlockJob> jobs, int n_threads)
        : file_(f), jobs_(std::move(jobs)), n_threads_(std::max(0, n_threads)) {
        ring_ = std::max<size_t>(2 * std::max(n_threads_, 1), 8);
        slots_.resize(ring_);
        for (auto& s : slots_) s.buf.resize(65536);
        if (n_threads_ > 0) {
            for (int t = 0; t < n_threads_; ++t) workers_.emplace_back([this] { worker(); });
        }
    }
    ~InflateStream() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        cv_done_.notify_all();
        for (auto& w : workers_) w.join();
    }
    // read up to n bytes; returns bytes read (0 at end of stream)
    size_t read(uint8_t* dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (cur_avail_ == 0 && !next_block()) break;
            size_t k = std::min(n - got, cur_avail_);
            memcpy(dst + got, cur_ptr_, k);
            cur_ptr_ += k;
            cur_avail_ -= k;
            got += k;
        }
        return got;
    }
    uint64_t blocks_consumed() const { return consumed_; }

private:
    struct Slot {
        std::vector<uint8_t> buf;
        bool ready = false;
        std::string err;
    };
    void do_job(size_t idx) {
        Slot& s = slots_[idx % ring_];
        const BlockJob& j = jobs_[idx];
        try {
            inflate_block(file_.data + j.blk.coffset + j.blk.cdata_off, j.blk.cdata_size, s.buf.data(), j.blk.isize);
        } catch (const std::exception& e) {
            s.err = e.what();
        }
    }
    void worker() {
        for (;;) {
            size_t idx;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || (next_job_ < jobs_.size() && next_job_ < consumed_ + ring_); });
                if (stop_) return;
                idx = next_job_++;
            }
            do_job(idx);
            {
                std::lock_guard<std::mutex> lk(mu_);
                slots_[idx % ring_].ready = true;
            }
            cv_done_.notify_all();
        }
    }
    bool next_block() {
        for (;;) {
            if (consumed_ >= jobs_.size()) return false;
            size_t idx = consumed_;
            Slot& s = slots_[idx % ring_];
            if (n_threads_ == 0) {
                do_job(idx);
            } else {
                std::unique_lock<std::mutex> lk(mu_);
                cv_done_.wait(lk, [&] { return s.ready; });
                s.ready = false;
            }
            if (!s.err.empty()) throw Error(s.err);
            const BlockJob& j = jobs_[idx];
            // the buffer must stay valid until the next call: copy into cur_ (one block)
            size_t lo = j.skip_start, hi = (size_t)j.blk.isize - std::min<size_t>(j.skip_end, j.blk.isize);
            if (hi < lo) hi = lo;
            cur_.assign(s.buf.begin() + lo, s.buf.begin() + hi);
            cur_ptr_ = cur_.data();
            cur_avail_ = cur_.size();
            {
                std::lock_guard<std::mutex> lk(mu_);
                ++consumed_;
            }
            cv_work_.notify_all();
            if (cur_avail_ > 0) return true;
        }
    }

    const MappedFile& file_;
    std::vector<BlockJob> jobs_;
    int n_threads_;
    size_t ring_;
    std::vector<Slot> slots_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    bool stop_ = false;
    size_t next_job_ = 0;
    size_t consumed_ = 0;
    std::vector<uint8_t> cur_;
    const uint8_t* cur_ptr_ = nullptr;
    size_t cur_avail_ = 0;
};

// ---------------------------------------------------------------------------
// BAM header (BioD/bio/std/hts/bam/reader.d:101-125,580-598) and the parts of the
// SAM text header depth needs (sam/header.d: @HD SO, @RG ID/SM; RG dictionary keeps
// insertion order and ignores repeated IDs -- header.d:345-354).
// ---------------------------------------------------------------------------
struct RefSeq {
    std::string name;
    int32_t length = 0;
};
struct ReadGroup {
    std::string id, sample;
};
struct BamHeader {
    std::string text;
    std::vector<RefSeq> refs;
    std::string sorting_order = "unknown";
    std::vector<ReadGroup> read_groups;
    uint64_t first_record_uoffset = 0;  // offset of the first record in the uncompressed stream
    int ref_id(const std::string& name) const {
        for (size_t i = 0; i < refs.size(); ++i)
            if (refs[i].name == name) return (int)i;
        return -1;
    }
};

static inline void parse_sam_text(BamHeader& h) {
    size_t p = 0;
    while (p < h.text.size()) {
        size_t e = h.text.find('\n', p);
        if (e == std::string::npos) e = h.text.size();
        std::string line = h.text.substr(p, e - p);
        p = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.size() < 3 || line[0] != '@') continue;
        std::string type = line.substr(1, 2);
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
        if (type == "HD") {
            if (kv.count("SO")) h.sorting_order = kv["SO"];
        } else if (type == "RG") {
            ReadGroup rg;
            rg.id = kv.count("ID") ? kv["ID"] : "";
            rg.sample = kv.count("SM") ? kv["SM"] : "";
            bool dup = false;
            for (auto& o : h.read_groups) dup |= (o.id == rg.id);
            if (!dup) h.read_groups.push_back(rg);
        }
    }
}

// ---------------------------------------------------------------------------
// BAI (BioD/bio/std/hts/bam/baifile.d:126-169; bai/bin.d:56-76; constants.d:35-37)
// ---------------------------------------------------------------------------
struct BaiBin {
    uint32_t id = 0;
    std::vector<Chunk> chunks;
};
struct BaiRef {
    std::vector<BaiBin> bins;
    std::vector<uint64_t> ioffsets;
    // Index.getMinimumOffset (baifile.d:75-80)
    uint64_t min_offset(int position) const {
        int pos = std::max(0, position);
        int i = std::min(pos / 16384, (int)ioffsets.size() - 1);
        return i == -1 ? 0 : ioffsets[(size_t)i];
    }
};
struct Bai {
    std::vector<BaiRef> refs;
};

static inline Bai parse_bai(const std::string& path) {
    MappedFile f(path);
    Bai bai;
    size_t p = 0;
    auto need = [&](size_t n) {
        if (p + n > f.size) throw Error("BAI file is truncated");
    };
    need(8);
    if (memcmp(f.data, "BAI\1", 4) != 0) throw Error("Invalid file format: expected BAI\\1");
    int32_t n_ref = (int32_t)le32(f.data + 4);
    p = 8;
    bai.refs.resize((size_t)std::max(0, n_ref));
    for (auto& r : bai.refs) {
        need(4);
        int32_t n_bin = (int32_t)le32(f.data + p);
        p += 4;
        r.bins.resize((size_t)std::max(0, n_bin));
        for (auto& b : r.bins) {
            need(8);
            b.id = le32(f.data + p);
            int32_t n_chunk = (int32_t)le32(f.data + p + 4);
            p += 8;
            b.chunks.resize((size_t)std::max(0, n_chunk));
            for (auto& c : b.chunks) {
                need(16);
                c.beg = le64(f.data + p);
                c.end = le64(f.data + p + 8);
                p += 16;
            }
        }
        need(4);
        int32_t n_intv = (int32_t)le32(f.data + p);
        p += 4;
        r.ioffsets.resize((size_t)std::max(0, n_intv));
        for (auto& o : r.ioffsets) {
            need(8);
            o = le64(f.data + p);
            p += 8;
        }
    }
    return bai;  // trailing n_no_coor (if any) is never read by the reference
}

struct Region {  // BioD/bio/std/hts/bam/region.d:28-65
    uint32_t ref_id = 0, start = 0, end = 0;
    bool operator<(const Region& o) const {
        if (ref_id != o.ref_id) return ref_id < o.ref_id;
        if (start != o.start) return start < o.start;
        return end < o.end;
    }
    bool overlaps(uint32_t r, uint32_t pos) const { return ref_id == r && start <= pos && pos < end; }
    bool fully_left_of(uint32_t r, uint32_t pos) const { return ref_id < r || (ref_id == r && end <= pos); }
};

// nonOverlapping (BioD/bio/core/utils/algo.d:95-162): merge sorted [beg,end) items when
// prev.end >= next.beg.
template <class T, class B, class E>
static inline std::vector<T> merge_sorted(const std::vector<T>& v, B beg, E end) {
    std::vector<T> out;
    for (T x : v) {
        if (!out.empty() && end(out.back()) >= beg(x)) {
            if (end(x) > end(out.back())) end(out.back()) = end(x);
        } else {
            out.push_back(x);
        }
    }
    return out;
}

// RandomAccessManager.getGroupChunks (randomaccessmanager.d:247-294); regions are on one
// reference, sorted and merged.  Bin ids beyond 37448 are skipped explicitly (the
// reference indexes a 37449-long bitset with them).
static inline std::vector<Chunk> group_chunks(const Bai& bai, const std::vector<Region>& regions) {
    if (regions.empty()) throw Error("empty region group");
    std::vector<bool> bitset(37449, false);
    bitset[0] = true;
    for (const Region& rg : regions) {
        uint32_t beg = rg.start, end = rg.end;
        if (!(beg < end)) throw Error("Enforcement failed");
        --end;
        uint32_t k;
        for (k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) bitset[k] = true;
        for (k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) bitset[k] = true;
        for (k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) bitset[k] = true;
        for (k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) bitset[k] = true;
        for (k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k)
            if (k < bitset.size()) bitset[k] = true;
    }
    uint32_t ref_id = regions.front().ref_id;
    if (ref_id >= bai.refs.size()) throw Error("Invalid reference sequence index");
    const BaiRef& ix = bai.refs[ref_id];
    uint64_t min_off = ix.min_offset((int)regions.front().start);
    std::vector<Chunk> chunks;
    for (const BaiBin& b : ix.bins) {
        if (b.id >= bitset.size() || !bitset[b.id]) continue;
        for (Chunk c : b.chunks) {  // appendChunks (randomaccessmanager.d:209-220)
            if (c.end > min_off) {
                if (c.beg < min_off) c.beg = min_off;
                chunks.push_back(c);
            }
        }
    }
    std::sort(chunks.begin(), chunks.end());
    return merge_sorted(chunks, [](Chunk& c) -> uint64_t& { return c.beg; }, [](Chunk& c) -> uint64_t& { return c.end; });
}

// ---------------------------------------------------------------------------
// BED file / region string (sambamba/utils/common/bed.d:37-152; BioD/bio/core/region.d:97-246)
// ---------------------------------------------------------------------------
struct BedInterval {
    std::string chr;
    long beg = 0, end = 0;
};

static inline std::vector<std::string> split_ws(const std::string& s) {
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

static inline long to_long_strict(const std::string& s) {
    if (s.empty()) throw Error("Unexpected end of input when converting from type string to type long");
    size_t i = 0;
    bool neg = false;
    if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; }
    if (i >= s.size()) throw Error("Unexpected '" + s + "' when converting from type string to type long");
    long v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') throw Error("Unexpected '" + std::string(1, s[i]) + "' when converting from type string to type long");
        v = v * 10 + (s[i] - '0');
    }
    return neg ? -v : v;
}

// readIntervals (bed.d:59-98).  `lines` gets every line with >=2 fields (even if the
// interval is dropped); `ivs` gets the kept intervals in file order.
static inline bool read_bed(const std::string& path, std::vector<BedInterval>* ivs, std::vector<std::string>* lines) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
    fclose(fp);
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
        iv.beg = to_long_strict(f[1]);
        iv.end = f.size() >= 3 ? to_long_strict(f[2]) : iv.beg + 1;
        if (iv.beg == iv.end) iv.end = iv.beg + 1;
        if (iv.beg < iv.end && ivs) ivs->push_back(iv);
        if (lines) lines->push_back(str);
    }
    return true;
}

// parseBed (bed.d:128-152), non_overlapping=true flavour: per contig sort by beg, merge when
// cur.end >= next.beg (bed.d:37-55), drop contigs absent from the BAM, sort regions.
static inline std::vector<Region> bed_merged(const std::vector<BedInterval>& ivs, const BamHeader& h) {
    std::map<std::string, std::vector<BedInterval>> by_chr;
    for (auto& iv : ivs) by_chr[iv.chr].push_back(iv);
    std::vector<Region> regs;
    for (auto& kv : by_chr) {
        int id = h.ref_id(kv.first);
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
    std::sort(regs.begin(), regs.end());
    return regs;
}
// non_overlapping=false flavour: file order, contigs absent from the BAM dropped.
static inline std::vector<Region> bed_raw(const std::vector<BedInterval>& ivs, const BamHeader& h) {
    std::vector<Region> regs;
    for (auto& iv : ivs) {
        int id = h.ref_id(iv.chr);
        if (id < 0) continue;
        regs.push_back({(uint32_t)id, (uint32_t)iv.beg, (uint32_t)iv.end});
    }
    return regs;
}

// parseRegion (BioD/bio/core/region.d:97-246): "ref", "ref:beg", "ref:beg-end"; digits may
// contain commas; beg is 1-based (-> beg-1), end kept; defaults beg=0, end=uint.max.
struct RegionStr {
    std::string reference;
    uint32_t beg = 0, end = 0xFFFFFFFFu;
};
static inline RegionStr parse_region_string(const std::string& s) {
    RegionStr r;
    size_t colon = s.rfind(':');
    // the grammar is ref (':' num ('-' num)?)? ; a ':' not followed by digits belongs to the name
    auto is_num = [](const std::string& t) {
        if (t.empty()) return false;
        bool digit = false;
        for (char c : t) {
            if (c >= '0' && c <= '9') digit = true;
            else if (c != ',') return false;
        }
        return digit;
    };
    auto num = [](const std::string& t) {
        long v = 0;
        for (char c : t)
            if (c != ',') v = v * 10 + (c - '0');
        return v;
    };
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

}  // namespace orc
