// kernels.hpp -- device data structures and launchers of the HIP kernels of libsbx_depth.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/sbx_depth.h"

namespace sbx {

// ---------------------------------------------------------------------------------------------
// Data layout in HBM
//   comp      : the BAM file bytes as on disk (BGZF blocks back to back)
//   U         : the inflated BAM byte stream, block b at U[out_off[b] .. +isize[b])
//   RecDesc[] : one 32-byte descriptor per BAM record, in file order
//   tiles     : the concatenated reference is cut into tiles of kTilePos positions, every contig
//               starting on a tile boundary; tile t covers positions [t*T - base, ...) of its contig
//   counters  : per *active* tile (>= 1 admitted read overlaps it): u32[T][n_samples][7]
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kOffUnknown = ~0ULL;       // "no value yet" in the record-chain arrays
constexpr uint64_t kOffInvalid = ~0ULL - 1;   // walking from a guess ran into garbage

struct RecDesc {            // 32 bytes, one per BAM record (read.d:907-1003 fields the path needs)
    uint64_t rec_off;       // offset of the record's block_size field in U
    int32_t pos;            // 0-based leftmost position (BamRead.position)
    int32_t end;            // pos + basesCovered() (read.d:255-262); == pos when the read is not admitted
    uint32_t l_seq;
    uint16_t n_cigar;
    uint8_t l_name;
    uint8_t mapq;
    uint16_t flag;
    uint16_t sample;        // CustomBamRead.sample_id (depth.d:240-250); 0 when --combined / single sample
    uint16_t q_start;       // kind==1: query offset of the first aligned base
    uint8_t kind;           // 0 not admitted, 1 one run of M/=/X (fast path), 2 general CIGAR
    uint8_t pad;
};
static_assert(sizeof(RecDesc) == 32, "RecDesc must be 32 bytes");

constexpr int kMaxThresholds = 16;   // -T options handled per launch

struct RangeChunk {         // a piece (<= 16384 positions) of a region or one window
    uint32_t ref_id, start, end, id;
};
struct SortedRegion {       // regions of the raw BED list sorted by (ref, start)
    uint32_t start, end, id;
};

struct DeviceFilter {       // compiled -F program (sbx_filter), evaluated per record
    int32_t n_ops;
    sbx_filter_op ops[SBX_FILTER_MAX_OPS];
    char strings[SBX_FILTER_STRINGS];
    sbx_regex regex[SBX_FILTER_REGEXES];
    const uint8_t* ref_sets;    // REFSET ops: one byte per reference id + 1 ("*" = id -1 first), at op.value
    int32_t n_ref;
};

// operations eval_filter_simple (index.hip) knows: flag tests, mate_is_on_another_chromosome-style flag logic, integer fields (not the
// average base quality: it reads the qualities), and / or / not, the constants
inline bool filter_op_is_simple(uint32_t kind, uint32_t field) {
    return kind == 0 || kind == 1 || (kind == 2 && field != 7) || kind == 3 || kind == 4 || kind == 5 || kind == 6 || kind == 12;
}

struct RefTable {           // per-reference device arrays
    const int32_t* ref_len;        // [n_ref]
    const uint32_t* tile_base;     // [n_ref + 1] index of the contig's first tile
    int32_t n_ref;
    // -L read selection (bam.getReadsOverlapping, randomaccessmanager.d:397-461): merged regions sorted by start,
    // sel_first[ref] .. sel_first[ref + 1]; sel == nullptr: every read
    const SortedRegion* sel;
    const uint32_t* sel_first;
    // several BAMs with compatible but different @SQ dictionaries (MultiBamReader, multireader.d:174-215): the arrays above are
    // those of the MERGED dictionary, a record's reference id is translated when it is read: own_to_merged[id], id < n_ref_own.
    // nullptr: the file's own dictionary is the merged one (n_ref_own == n_ref).
    const int32_t* own_to_merged;
    int32_t n_ref_own;
};

struct RgTable {            // read-group id strings -> sample id (depth.d:1170-1181)
    const char* ids;               // concatenated NUL-terminated ids
    const uint32_t* id_off;        // [n_rg] offsets into ids
    const uint16_t* sample_of;     // [n_rg]
    int32_t n_rg;
    int32_t lookup;                // 0: every read is sample 0 and RG tags are not inspected
    uint32_t ids_bytes;            // bytes of `ids`
};

// ---- K1: BGZF inflate (inflate.hip) -------------------------------------------------------
// Two launches: K1a huffman_decode (lane per block -> literal + match-entry streams) and
// K1b lz77_resolve (wave per block -> inflated bytes).  The per-block arrays are indexed from 0 for
// the n_blocks blocks of this launch; block0 is the index of the first one in the whole file (it
// only positions the blocks' slices inside the token streams).
// offset of block b's 64-byte aligned slice (capacity >= isize[b] + 49 bytes) in the literal stream; a pure function of
// (out_off[b], b).  K2 reuses the slice, dead once the block is inflated, for the block's record offsets.
__host__ __device__ __forceinline__ uint64_t inflate_lit_offset(uint64_t out_off_b, uint32_t b) { return (out_off_b + 112ull * b + 63ull) & ~63ull; }
size_t inflate_scratch_bytes(uint32_t n_blocks);
size_t inflate_lit_bytes(uint64_t total_out, uint32_t n_blocks);
size_t inflate_ent_words(uint64_t total_out, uint32_t n_blocks);
void launch_bgzf_inflate(const uint8_t* d_comp, const uint64_t* d_comp_off, const uint32_t* d_comp_len,
                         const uint32_t* d_isize, const uint64_t* d_out_off, uint8_t* d_out, uint32_t n_blocks,
                         uint32_t block0, uint8_t* d_scratch, uint8_t* d_lit, uint32_t* d_ent, uint32_t* d_nent,
                         uint32_t* d_status, hipStream_t stream, hipEvent_t ev_mid = nullptr,
                         unsigned long long* d_tok_bytes = nullptr);     // [64] += literal bytes + 4 x match entries (accounting; may be null)
const char* inflate_status_string(uint32_t s);

// ---- K2: record index (index.hip) ---------------------------------------------------------
// One run of the device work list: a stretch [u_beg, u_end) of the (compacted) inflated stream that holds a
// whole number of BAM records -- the whole file from its first record on, or a group of merged BAI chunks
// (chunk boundaries are record boundaries, randomaccessmanager.d:247-294).  Blocks are launch-local indices.
// open_end != 0 (the batches of sbx_build_index): u_end is a block boundary, not a record boundary -- the chain stops in front
// of the first record that does not end inside the run, and the exit of the run's last block is where the next batch starts.
struct ChainRun {
    uint64_t u_beg, u_end;
    uint32_t blk_first, blk_last;      // inclusive
    uint32_t open_end, reserved;
};

constexpr uint32_t kIndexStatSlots = 64;     // IndexArgs.stats points at this many accumulators (power of two); the host adds them up
struct IndexStats {         // device-side accumulators of the describe pass
    unsigned long long n_records, n_admitted, n_bad, n_unknown_rg;
    // bytes K3 has to read of the admitted records: CIGAR + packed sequence, and their base qualities (read only when -q > 0)
    unsigned long long adm_seq_bytes, adm_qual_bytes;
    unsigned int max_span;                  // longest alignment (reference positions) among the admitted records
    unsigned int over_tiles;                // position tiles an admitted alignment reaches beyond the spare tiles of its contig (slot 0 only)
};

struct IndexArgs {
    const uint8_t* U;               // inflated stream of this launch (offset 0 = first byte of block 0)
    uint64_t u_alloc;               // readable bytes of U rounded up to 16 (the allocation is >= u_alloc + 64)
    const uint64_t* out_off;        // [n_blocks] offset of every block in U
    const uint32_t* isize;          // [n_blocks]
    const uint32_t* run_of;         // [n_blocks] index into runs
    const ChainRun* runs;
    uint32_t n_blocks;
    const uint32_t* inflate_status; // [n_blocks] K1 status
    const uint64_t* entry_in;       // nullptr: guess the first record start of every block; else the repaired chain
    uint32_t simple_filter;         // the -F program holds flag tests, integer fields and and / or / not only (index.hip: eval_filter_simple)
    uint64_t* entry;                // [n_blocks] out: first record start at or after the block's first byte (may lie beyond it)
    uint64_t* exit_;                // [n_blocks] out: where the chain leaves the block
    uint32_t* count;                // [n_blocks] out: records starting inside the block
    uint64_t* state;                // [n_blocks] out: 2 << 62 | records in blocks 0..b
    uint8_t* scratch;               // K1's literal stream (launch-local block indices): block b's record offsets go to its slice
    RefTable refs;
    const DeviceFilter* filt;
    RgTable rg;
    uint32_t tile_pos;
    RecDesc* desc;
    int32_t* rec_ref;
    uint64_t* name_hash;            // may be null
    uint64_t desc_cap;              // capacity of desc / rec_ref / name_hash in records
    uint32_t* tile_lo;
    uint32_t* tile_hi;
    IndexStats* stats;
    // read ownership (sbx_run_interval_owned): own_ref >= 0 admits only records whose leftmost position lies in
    // [own_beg, own_end) of that contig -- reads are then partitioned, not clipped, and per-position sums over the
    // owners' runs equal the whole
    int32_t own_ref;
    uint32_t own_beg, own_end;
    unsigned long long* scan_part;  // kScanPartWords words of scratch for the multi-workgroup scans (nullptr: single-workgroup kernels)
    uint32_t* flags;                // [0] lowest block with an inconsistent chain (0xFFFFFFFF: none), [1] lowest block whose
                                    // inflate failed, [2] != 0: desc_cap was too small (nothing useful was written)
};
constexpr uint32_t kScanPartWords = 4096;      // 16 M blocks / tiles per launch of the multi-workgroup scans
void launch_index_blocks(const IndexArgs& a, hipStream_t stream);
// parallel repair round: blocks not entered where their predecessor was left are walked again from there
// (*d_n_changed += blocks re-walked; reads exit_[b-1] of the previous round: launch until it stays 0)
void launch_rewalk_mismatched(const IndexArgs& a, uint32_t* d_n_changed, hipStream_t stream);
// serial (one wave) repair of the recorded chain from block `from` on; *d_n_rewalked += blocks re-walked
void launch_chain_repair(const uint8_t* d_U, const uint64_t* d_out_off, const uint32_t* d_isize, const uint32_t* d_run_of,
                         const ChainRun* d_runs, uint32_t n_blocks, uint32_t from, uint64_t* d_entry, uint64_t* d_exit,
                         uint32_t* d_count, uint32_t* d_n_rewalked, hipStream_t stream);
// exclusive scan of per-chunk counts -> d_base[n + 1] (used by K6)
void launch_count_scan(const uint32_t* d_count, uint32_t n_blocks, uint64_t* d_base, void* d_tmp, size_t tmp_bytes,
                       hipStream_t stream);
size_t count_scan_tmp_bytes(uint32_t n_blocks);
// compact the tiles that have work: active[] = tile ids, slot_of[t] = index into active or ~0u;
// d_n_active[0] = number of active tiles, [1] = how many of them have >= 65536 records
void launch_tile_compact(const uint32_t* d_tile_lo, const uint32_t* d_tile_hi, uint32_t n_tiles, uint32_t deep_thr, uint32_t* d_active,
                         uint32_t* d_slot_of, uint32_t* d_n_active, hipStream_t stream, unsigned long long* d_scan_part = nullptr);

// ---- K3: decode + accumulate (depth.hip) -------------------------------------------------
// n_deep: active tiles with >= deep_thr (<= 65536) records, counted by tile_compact; they keep 32-bit LDS counters
constexpr uint32_t kDeepTileRecords = 65536;
void launch_accumulate(const uint8_t* d_U, const RecDesc* d_desc, const uint32_t* d_tile_lo, const uint32_t* d_tile_hi,
                       const uint32_t* d_active, uint32_t n_active, uint32_t n_deep, uint32_t deep_thr, const uint32_t* d_tile_base,
                       int32_t n_ref, uint32_t tile_pos, uint32_t n_samples, uint32_t min_bq, uint32_t* d_counters, uint32_t* d_span,
                       hipStream_t stream, bool compact = false);
// compact == true (region / window modes without -m, no deep tile): d_counters holds ONE word per tile position and sample,
// {bases counted (codes 0..4) : 16 | depth (all 7 counters) : 16}, instead of the seven counters -- what launch_range_reduce needs

// ---- K7: --fix-mate-overlaps, base mode (mates.hip) ---------------------------------------
// d_mate[i] = index of the single overlapping same-name record of i (0xFFFFFFFF: none),
// d_n_partners[i] = how many were found (> 1 is outside the supported scope)
void launch_find_mates(const uint8_t* d_U, const RecDesc* d_desc, const uint64_t* d_hash, const int32_t* d_rec_ref, uint64_t n_records,
                       uint32_t* d_mate, uint32_t* d_n_partners, hipStream_t stream);
// second pass when some record has more than one partner: up to three partners per record in d_ext[3 i ..] (d_n_partners is
// counted again and must be zero on entry)
// several files with -m: does an admitted record of file A share name and sample with an overlapping admitted record of file B?  (*d_flag |= 1)
void launch_cross_file_mates(const uint8_t* d_Ua, const RecDesc* d_desc_a, const uint64_t* d_hash_a, const int32_t* d_ref_a, uint64_t n_a,
                             const uint8_t* d_Ub, const RecDesc* d_desc_b, const uint64_t* d_hash_b, const int32_t* d_ref_b, uint64_t n_b,
                             uint32_t max_span_b, uint32_t* d_flag, hipStream_t stream);
void launch_find_partners(const uint8_t* d_U, const RecDesc* d_desc, const uint64_t* d_hash, const int32_t* d_rec_ref, uint64_t n_records,
                          uint32_t* d_ext, uint32_t* d_n_partners, hipStream_t stream);
// d_ext == nullptr: every record has at most one partner (d_mate); otherwise records with two or three partners take them
// from d_ext, and *d_too_many is set when four or more same-name records cover one column
void launch_accumulate_mates(const uint8_t* d_U, const RecDesc* d_desc, const uint32_t* d_mate, const uint32_t* d_tile_lo,
                             const uint32_t* d_tile_hi, const uint32_t* d_active, uint32_t n_active, const uint32_t* d_tile_base,
                             int32_t n_ref, uint32_t tile_pos, uint32_t n_samples, uint32_t min_bq, const uint32_t* d_ext,
                             const uint32_t* d_n_partners, uint32_t* d_too_many, uint32_t* d_counters, uint32_t* d_span, hipStream_t stream);
// max over records of n_partners (single block reduction into *d_out)
void launch_max_u32(const uint32_t* d_in, uint64_t n, uint32_t* d_out, hipStream_t stream);

// ---- K5: region / window statistics (reduce.hip) --------------------------------------------
void launch_range_reduce(const RangeChunk* d_chunks, uint32_t n_chunks, const uint32_t* d_counters, const uint32_t* d_span,
                         const uint32_t* d_slot_of, const uint32_t* d_tile_base, uint32_t T, uint32_t S,
                         const uint32_t* d_thresholds, uint32_t n_thr, uint32_t* d_n_bases, uint32_t* d_cov_counts,
                         uint32_t* d_seen, hipStream_t stream, bool compact = false);
// window mode, every window of the run in ONE launch and without a chunk list: window k of contig r has id d_win_base[r] + k
// (d_win_base = running sum of d_n_win over the contigs), n_windows = their total
void launch_range_reduce_windows(const uint64_t* d_win_base, const uint64_t* d_n_win, uint32_t n_ref, uint32_t window, uint32_t n_windows,
                                 const uint32_t* d_counters, const uint32_t* d_span, const uint32_t* d_slot_of, const uint32_t* d_tile_base,
                                 uint32_t T, uint32_t S, const uint32_t* d_thresholds, uint32_t n_thr, uint32_t* d_n_bases,
                                 uint32_t* d_cov_counts, uint32_t* d_seen, hipStream_t stream, bool compact);
void launch_count_reads_windows(const uint8_t* d_U, const RecDesc* d_desc, uint64_t n_records, const int32_t* d_rec_ref,
                                uint32_t window, const uint64_t* d_win_base, const uint64_t* d_n_win, uint32_t S,
                                uint32_t min_bq, uint32_t* d_n_reads, hipStream_t stream);
void launch_count_reads_regions(const uint8_t* d_U, const RecDesc* d_desc, uint64_t n_records, const int32_t* d_rec_ref,
                                const SortedRegion* d_regs, const uint32_t* d_pmax_end, const uint32_t* d_ref_first, uint32_t S,
                                uint32_t min_bq, uint32_t* d_n_reads, const uint32_t* d_min_start, uint32_t* d_n_bases,
                                hipStream_t stream);

// region / window statistics with --fix-mate-overlaps (mates.hip: per-column quantities; reduce.hip: the rest)
void launch_mates_columns(const uint8_t* d_U, const RecDesc* d_desc, const uint32_t* d_mate, const uint32_t* d_tile_lo,
                          const uint32_t* d_tile_hi, const uint32_t* d_active, uint32_t n_active, const uint32_t* d_tile_base,
                          int32_t n_ref, uint32_t tile_pos, uint32_t n_samples, uint32_t min_bq, uint32_t* d_covm, uint32_t* d_addm,
                          uint32_t* d_span, hipStream_t stream);
void launch_range_first(const RangeChunk* d_chunks, uint32_t n_chunks, const uint32_t* d_span, const uint32_t* d_slot_of,
                        const uint32_t* d_tile_base, uint32_t T, uint32_t* d_first, hipStream_t stream);
void launch_range_reduce_m(const RangeChunk* d_chunks, uint32_t n_chunks, const uint32_t* d_covm, const uint32_t* d_addm,
                           const uint32_t* d_span, const uint32_t* d_slot_of, const uint32_t* d_tile_base, uint32_t T, uint32_t S,
                           const uint32_t* d_thresholds, uint32_t n_thr, uint32_t* d_n_bases, uint32_t* d_cov_counts, uint32_t* d_seen,
                           hipStream_t stream);
void launch_count_reads_mates(const uint8_t* d_U, const RecDesc* d_desc, uint64_t n_records, const int32_t* d_rec_ref,
                              const uint32_t* d_mate, const SortedRegion* d_regs, const uint32_t* d_pmax_end, const uint32_t* d_ref_first,
                              const SortedRegion* d_union, const uint32_t* d_union_first, bool everywhere, const uint32_t* d_first,
                              uint32_t S, uint32_t min_bq, uint32_t* d_n_bases, uint32_t* d_n_reads, hipStream_t stream);

// multi-BAM: add the tile slots of one file's run into the merged tile set
void launch_merge_tiles(const uint32_t* d_src, const uint32_t* d_src_active, uint32_t n_src_active, const uint32_t* d_dst_slot_of,
                        uint32_t per_tile, uint32_t* d_dst, hipStream_t stream);

// ---- BGZF writer (deflate.hip): one lane per <= 0xFF00-byte block -> 64 KiB slots -> packed stream; record bins for the BAI
size_t deflate_table_entries(uint32_t n_blocks);
size_t deflate_work_bytes(uint32_t n_blocks);          // the dynamic-Huffman tables of the blocks in flight (deflate_core.hpp DynWork)
void launch_bgzf_deflate(const uint8_t* d_in, uint64_t n_bytes, uint32_t n_blocks, int level, uint8_t* d_slots, uint16_t* d_tables,
                         uint8_t* d_work, uint32_t* d_block_len, hipStream_t stream);
void launch_pack_blocks(const uint8_t* d_slots, const uint32_t* d_block_len, const uint64_t* d_offset, uint32_t n_blocks, uint8_t* d_out,
                        hipStream_t stream);
struct BaiArgs;
struct BaiCarry;
void launch_bai_records(const BaiArgs& a, BaiCarry* d_carry_out, hipStream_t stream);      // bai_parallel.hpp
void launch_gather_bins(const uint8_t* d_U, const RecDesc* d_desc, uint64_t n_records, uint16_t* d_bins, hipStream_t stream);

// K6 format_base_rows (format.hip): text of `depth base` for positions [beg, end) of one contig
struct FormatArgs {
    const uint32_t* counters;     // per active tile u32[T][S][7]
    const uint32_t* span;         // per active tile u32[T] or nullptr (then "column exists" == any counter != 0)
    const uint32_t* slot_of;      // tile -> slot, 0xFFFFFFFF = no admitted read touches the tile
    uint32_t tile_first, tile_end;    // tiles of this contig
    uint32_t T, S;
    uint32_t beg, end;
    uint64_t lo, hi;              // rows are printed for lo <= COV <= hi (or flagged y/n when annotate)
    uint32_t annotate, combined, zero_fill;
    const char* names;            // device blob: contig name, then the sample names
    uint32_t ref_name_len;
    const uint32_t* sample_off;   // [S + 1] offsets of the sample names inside `names`
    uint32_t max_sample_len;      // longest sample name (sizes the LDS of a chunk)
};
uint32_t format_chunk_positions();
void launch_format_measure(const FormatArgs& a, uint32_t n_chunks, uint32_t* d_chunk_len, hipStream_t stream);
void launch_format_write(const FormatArgs& a, uint32_t n_chunks, const uint64_t* d_chunk_off, uint8_t* d_text, hipStream_t stream);

}  // namespace sbx
