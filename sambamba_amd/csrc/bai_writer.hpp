// bai_writer.hpp -- BAI index of a coordinate-sorted BAM from the device's record descriptors.
//
// Restates IndexBuilder (BioD/bio/std/hts/bam/bai/indexing.d:52-346), the engine of `sambamba index`: reads arrive in
// file order with their virtual offsets; consecutive reads of one reference with the same stored `bin` form a chunk
// (a chunk that starts in the BGZF block the previous chunk of its bin ended in is merged into it, :215-240), the linear
// index keeps the start offset of the first read of every 16 kbp window a read overlaps (:128-156, empty windows
// repeat the last non-empty one when written, :158-170), every reference ends with samtools' metadata pseudo-bin 37450
// (:182-188) and the file with the number of reads without coordinates (:341).  sbx_build_index computes all of this on the
// device (bai_parallel.hpp: one step per record, no loop-carried state); this file is the loop as the reference has it -- the
// checker of that formulation (tests/native/bai_host.cpp) and the path for input it calls irregular (unsorted reads, whose error is
// worded here; SBX_BAI_HOST=1), fed record by record from descriptors copied back from the device.
// Bins are written in ascending id order (the reference iterates a D associative array: its order is unspecified).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "common.hpp"

namespace sbx {

struct BaiRecord {
    int32_t ref_id, position, end_position;     // end = position + basesCovered()
    uint32_t bin;
    bool is_unmapped;
    uint64_t start_vo, end_vo;
};

// Virtual offsets (compressed block start << 16 | offset inside the inflated block) of positions of the inflated stream, for
// queries that never go backwards.  The reference takes them from its reader: BgzfInputStream sets up the next block as soon
// as the current one is exhausted (inputstream.d:497-530), so the position BEHIND the last byte of a block is offset 0 of the
// block that starts there -- even an empty one: the end of the last read of a file is the start of its EOF block (this is
// what the .bai files of the reference's test-suite hold) -- while the position OF a byte is the block that holds it.
class VoffCursor {
  public:
    // block i: file offset coffset[i], inflated bytes [ustart[i], ustart[i + 1]); file_end = offset behind the last block
    VoffCursor(const uint64_t* coffset, const uint64_t* ustart, size_t n_blocks, uint64_t file_end)
        : coff_(coffset), ustart_(ustart), n_(n_blocks), file_end_(file_end) {}
    uint64_t of_byte(uint64_t u) {          // start of a record
        while (bi_ < n_ && ustart_[bi_ + 1] <= u) ++bi_;
        return bi_ < n_ ? (coff_[bi_] << 16) | (u - ustart_[bi_]) : file_end_ << 16;
    }
    uint64_t behind(uint64_t u) {           // end of a record whose last byte is u - 1
        while (bi_ < n_ && ustart_[bi_ + 1] <= u && ustart_[bi_] != u) ++bi_;
        return bi_ < n_ ? (coff_[bi_] << 16) | (u - ustart_[bi_]) : file_end_ << 16;
    }

  private:
    const uint64_t* coff_;
    const uint64_t* ustart_;
    size_t n_, bi_ = 0;
    uint64_t file_end_;
};

class BaiBuilder {
  public:
    BaiBuilder(int n_refs) : n_refs_(n_refs), linear_(37450 - 4681 + 1, 0) {
        out_.insert(out_.end(), {'B', 'A', 'I', 1});
        put32((uint32_t)n_refs);
    }
    // IndexBuilder.put (:262-300)
    void put(const BaiRecord& r) {
        check_sorted(r);
        struct Meta { BaiBuilder* b; const BaiRecord& r; ~Meta() { b->update_metadata(r); } } meta{this, r};      // scope(exit)
        if (r.ref_id < 0) return;
        if (r.position < 0) return;
        if (first_) {
            prev_ = r;
            first_ = false;
            chunk_beg_ = r.start_vo;
            for (int i = 0; i < r.ref_id; ++i) write_empty_reference();
            return;
        }
        if (r.ref_id > prev_.ref_id) {
            update_linear_index();
            update_chunks();
            dump_current_reference();
            for (int i = prev_.ref_id + 1; i < r.ref_id; ++i) write_empty_reference();
        }
        if (r.ref_id == prev_.ref_id) {
            update_linear_index();
            if (r.bin != prev_.bin) update_chunks();
        }
        prev_ = r;
    }
    // IndexBuilder.finish (:302-316)
    const std::vector<uint8_t>& finish() {
        if (!first_) {
            update_linear_index();
            update_chunks();
            dump_current_reference();
        }
        for (int i = (first_ ? -1 : prev_.ref_id) + 1; i < n_refs_; ++i) write_empty_reference();
        put64(no_coord_);
        return out_;
    }

  private:
    int n_refs_;
    std::vector<uint8_t> out_;
    std::vector<uint64_t> linear_;
    size_t linear_len_ = 0;
    BaiRecord prev_{-1, 0, 0, 0, false, 0, 0};
    bool first_ = true;
    uint64_t no_coord_ = 0, beg_vo_ = ~0ull, end_vo_ = 0, unmapped_ = 0, mapped_ = 0;
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> chunks_;
    uint64_t chunk_beg_ = 0;

    void put32(uint32_t v) { for (int k = 0; k < 4; ++k) out_.push_back((uint8_t)(v >> (8 * k))); }
    void put64(uint64_t v) { for (int k = 0; k < 8; ++k) out_.push_back((uint8_t)(v >> (8 * k))); }
    static size_t lin_off(int32_t position) { return position < 0 ? 0 : (size_t)(position / 16384); }
    void write_empty_reference() { put32(0); put32(0); }
    void check_sorted(const BaiRecord& r) {          // checkThatInputIsSorted (:242-257)
        if (first_) return;
        if (r.ref_id == -1) return;
        if (prev_.ref_id < r.ref_id) return;
        if (!(r.ref_id == prev_.ref_id && r.position >= prev_.position))
            throw Error(SBX_ENOTSORTED, "BAM file is not coordinate-sorted: read at " + std::to_string(r.ref_id) + ":" + std::to_string(r.position) +
                                            " must be after read at " + std::to_string(prev_.ref_id) + ":" + std::to_string(prev_.position));
    }
    void update_metadata(const BaiRecord& r) {       // :107-122
        if (r.ref_id == -1) { ++no_coord_; return; }
        if (r.is_unmapped) ++unmapped_; else ++mapped_;
        if (beg_vo_ == ~0ull) beg_vo_ = r.start_vo;
        end_vo_ = r.end_vo;
    }
    void update_linear_index() {                     // :124-156
        size_t beg = lin_off(prev_.position), end = beg;
        if (!prev_.is_unmapped) end = lin_off(prev_.position + (prev_.end_position - prev_.position) - 1);
        for (size_t i = beg; i < end + 1 && i < linear_.size(); ++i)
            if (linear_[i] == 0) linear_[i] = prev_.start_vo;
        if (end + 1 > linear_len_) linear_len_ = end + 1;
    }
    void update_chunks() {                           // :207-240
        const uint64_t cur_end = prev_.end_vo;
        auto& cs = chunks_[prev_.bin];
        if (cs.empty() || (cs.back().second >> 16) != (chunk_beg_ >> 16)) cs.push_back({chunk_beg_, cur_end});
        else cs.back().second = cur_end;
        chunk_beg_ = cur_end;
    }
    void dump_current_reference() {                  // :172-205
        put32((uint32_t)chunks_.size() + 1);
        for (auto& kv : chunks_) {
            if (kv.second.empty()) continue;
            put32(kv.first);
            put32((uint32_t)kv.second.size());
            for (auto& c : kv.second) { put64(c.first); put64(c.second); }
        }
        put32(37450);
        put32(2);
        put64(beg_vo_); put64(end_vo_); put64(mapped_); put64(unmapped_);
        const size_t n = std::min(linear_len_, linear_.size());
        put32((uint32_t)n);
        uint64_t last = 0;
        for (size_t i = 0; i < n; ++i) {
            uint64_t v = linear_[i];
            if (v == 0) v = last; else last = v;
            put64(v);
        }
        std::fill(linear_.begin(), linear_.end(), 0);
        linear_len_ = 0;
        chunks_.clear();
        chunk_beg_ = prev_.end_vo;
        beg_vo_ = end_vo_ = chunk_beg_;
        unmapped_ = mapped_ = 0;
    }
};

}  // namespace sbx
