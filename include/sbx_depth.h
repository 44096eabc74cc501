/* ============================================================================
 * sbx_depth.h -- C ABI of libsbx_depth.so, the MI355X (gfx950) engine behind
 *                `sambamba depth base|region|window`.
 *
 * The reference (biod/sambamba, D) has no plugin API.  Its only FFI precedent is
 * the zlib binding (BioD/bio/core/utils/zlib.d:6,139-163: extern(C) prototypes,
 * caller-owned buffers, int return codes turned into exceptions).  This header
 * follows the same conventions and plugs into the two seams SURVEY.md 8(b) names:
 *
 *   codec seam   decompressBgzfBlock(BgzfBlock)            BioD/bio/core/bgzf/block.d:127
 *                (called from BgzfInputStream.fillNextBlock, inputstream.d:414-417)
 *                -> sbx_inflate_blocks()
 *
 *   engine seam  everything between "bam opened, filter/regions/mode known"
 *                (sambamba/depth.d:1163-1217) and "printer.push / printer.close"
 *                (depth.d:1218-1234)   -> sbx_open ... sbx_depth_* ... sbx_close
 *
 * Conventions: every function returns 0 on success or a negative SBX_E* code;
 * the message is available from sbx_last_error() (the D shim turns it into the
 * `sambamba-depth: <msg>` line of depth.d:1237-1244).  The caller owns every
 * output buffer; the library owns device memory and file mappings.  One sbx_ctx
 * is used from one host thread at a time; calls are blocking.  The library never
 * falls back to a CPU implementation: without a usable HIP device every compute
 * entry point fails with SBX_ENODEVICE.
 * ========================================================================== */
#ifndef SBX_DEPTH_H
#define SBX_DEPTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBX_OK            0
#define SBX_EINVAL       -1   /* bad argument */
#define SBX_EIO          -2   /* file missing / unreadable / truncated */
#define SBX_EFORMAT      -3   /* not BGZF / BAM / BAI, or corrupt deflate stream */
#define SBX_ENODEVICE    -4   /* no HIP device or HIP runtime failure */
#define SBX_EUNSUPPORTED -5   /* valid request outside the device path (e.g. regex -F filter) */
#define SBX_ENOTSORTED   -6   /* "All files must be coordinate-sorted" (depth.d:1164-1165) */
#define SBX_ENOINDEX     -7   /* "All files must be indexed"           (depth.d:1166)      */
#define SBX_ENOMEM       -8
#define SBX_ERG          -9   /* read group of a read is not in the header (depth.d:246-248) */

typedef struct sbx_ctx sbx_ctx;

/* sizeof() of the named struct of this header as the library was compiled ("sbx_filter", "sbx_regex",
 * "sbx_filter_op", "sbx_region", "sbx_region_stats", "sbx_header_info", "sbx_batch", "sbx_run_stats",
 * "sbx_regex_state", "sbx_shard"); 0 for an unknown name.  Lets a foreign-language binding (d/sbx_depth.d, the ctypes
 * binding) verify its struct layouts against the library it loaded. */
size_t sbx_abi_sizeof(const char* type_name);

/* Counter layout of one reference position of one sample, `depth base`
 * (PerBasePrinter.writeColumn, depth.d:495-556): A, C, G, T, other (N/IUPAC/'='),
 * DEL, REFSKIP.  COV = sum of all seven (code 4 counts towards COV but has no column). */
#define SBX_NCOUNTERS 7

enum { SBX_MODE_BASE = 0, SBX_MODE_REGION = 1, SBX_MODE_WINDOW = 2 };   /* depth.d:101-105 */

typedef struct {
    uint32_t ref_id;   /* BamRegion (BioD/bio/std/hts/bam/region.d:28-31) */
    uint32_t start;    /* 0-based, inclusive */
    uint32_t end;      /* 0-based, exclusive */
} sbx_region;

typedef struct {
    int32_t n_ref;             /* reference_sequences.length (reader.d:580-598)            */
    int32_t n_samples;         /* sample_names.length, >= 1 ("*" when there is no @RG)      */
    int32_t n_read_groups;
    int32_t sorted_by_coordinate;  /* header.sorting_order == coordinate (depth.d:1164)     */
    int32_t has_index;         /* bam.has_index (depth.d:1166)                             */
    int32_t reserved;
    uint64_t n_bgzf_blocks;
    uint64_t compressed_bytes;
    uint64_t uncompressed_bytes;
} sbx_header_info;

/* Per region (or window) x sample statistics: PerSampleRegionData (depth.d:609-635).
 * The reference keeps these as 32-bit `uint` (wrap-around is reference behaviour). */
typedef struct {
    uint32_t n_reads;
    uint32_t n_bases;
} sbx_region_stats;

/* The compiled subset of the -F filter language (sambamba/utils/common/filtering.d:86-214,
 * queryparser.d:232-483): a postfix program over flag tests, integer-field comparisons,
 * integer-tag comparisons ([NM] <= 2), tag existence ([XS] == null), string comparisons of tags and of
 * read_name / ref_name / mate_ref_name / strand ([RG] == 'lane1', ref_name != 'chrM') and and / or / not.
 * Built by sbx_compile_filter() from the query string. */
#define SBX_FILTER_MAX_OPS 64
#define SBX_FILTER_STRINGS 512
#define SBX_FILTER_REGEXES 2
#define SBX_REGEX_STATES 64
#define SBX_REGEX_CLASSES 8
/* A regular expression (`read_name =~ /^chr[0-9]+/`) compiled to a Thompson NFA of at most 64 states, so that a
 * state set is one 64-bit mask: type 0 CHAR c, 1 ANY, 2 CLASS classes[c], 3 SPLIT a|b, 4 JMP a, 5 BOL, 6 EOL, 7 MATCH;
 * consuming states continue at a. */
typedef struct { uint8_t type, a, b, c; } sbx_regex_state;
typedef struct {
    uint8_t n_states, n_classes, start, reserved;
    sbx_regex_state states[SBX_REGEX_STATES];
    uint8_t classes[SBX_REGEX_CLASSES][32];
} sbx_regex;
typedef struct {
    uint8_t  kind;     /* 0 FLAG_ANY(mask)  1 CHIMERIC  2 INTCMP  3 AND  4 OR  5 NOT  6 TRUE
                          7 TAGCMP (mask = key chars c0 | c1 << 8; cmp; value)  8 TAGNULL (cmp 4: absent, 5: present)
                          9 TAGSTR (mask = key; cmp; value = string)  10 NAMESTR (read_name; cmp; value = string)
                          11 REFNAME (field 0 ref_name, 1 mate_ref_name; cmp 4 / 5; value = string; resolved against
                             the header when the filter is installed)  12 FALSE  13 SEQSTR / 14 CIGARSTR (sequence / cigar as text)
                          15 REGEX (field 0 read_name 1 sequence 2 cigar 3 tag [mask = key] 4 ref_name 5 mate_ref_name;
                             value = index into sbx_filter.regex)  16 REFSET (internal: 15 on a reference name, resolved
                             against the header when the filter is installed)
                          strings: value = offset into sbx_filter.strings | length << 32 */
    uint8_t  field;    /* INTCMP: 0 ref_id 1 position 2 mapping_quality 3 sequence_length
                                  4 mate_ref_id 5 mate_position 6 template_length 7 avg_base_quality */
    uint8_t  cmp;      /* INTCMP: 0 >  1 <  2 >=  3 <=  4 ==  5 !=                           */
    uint8_t  pad;
    uint32_t mask;
    int64_t  value;
} sbx_filter_op;
typedef struct {
    int32_t n_ops;
    int32_t reserved;
    sbx_filter_op ops[SBX_FILTER_MAX_OPS];
    char strings[SBX_FILTER_STRINGS];
    int32_t n_regex;
    int32_t reserved2;
    sbx_regex regex[SBX_FILTER_REGEXES];
} sbx_filter;

/* ---- codec seam -------------------------------------------------------------
 * Batched replacement of decompressBgzfBlock (block.d:127-216): raw-deflate (RFC 1951,
 * zlib windowBits = -15) payloads comp[comp_off[i] .. +comp_len[i]) are inflated on the
 * device into out[out_off[i] .. +isize[i]).  All pointers are host memory.  As in the
 * reference's release build the CRC32 is not verified (block.d:187).  Returns
 * SBX_EFORMAT if any payload is not a valid deflate stream of exactly isize[i] bytes. */
int sbx_inflate_blocks(const uint8_t* comp, const uint64_t* comp_off, const uint32_t* comp_len,
                       const uint32_t* isize, uint32_t n_blocks, uint8_t* out, const uint64_t* out_off,
                       char* err, size_t errlen);

/* The write side of the same seam: bgzfCompress (BioD/bio/core/bgzf/compress.d:34-103) for a whole buffer at once.  in[0, n) is
 * cut into 0xFF00-byte pieces (bgzf/constants.d:33), every piece becomes one BGZF block -- 18-byte header with the BC
 * subfield, raw deflate, CRC32, ISIZE -- compressed on the device (one lane per block; any RFC 1951 stream is valid for every
 * BGZF reader, the reference's included).  level is zlib's, as bgzfCompress hands it on: 0 = stored blocks; 1 .. 3 = the fixed
 * Huffman code over greedy hash-table matches; 4 .. 9 and -1 -- Z_DEFAULT_COMPRESSION, the reference's default
 * (bgzfCompress(chunk, level = -1), BamWriter(compression_level = -1)) -- = a dynamic Huffman code per block, from level 7 on over
 * four match candidates per position with lazy evaluation; anything else: SBX_EINVAL.  with_eof != 0 appends the 28-byte EOF
 * block (constants.d:37-49).  All pointers are host
 * memory; *out_len receives the size of the stream (also on SBX_ENOMEM, when cap is too small: n + n / 2048 + 64 is
 * always enough).  device: HIP ordinal or -1. */
int sbx_bgzf_compress(const uint8_t* in, size_t n, int level, int with_eof, int device, uint8_t* out, size_t cap, size_t* out_len,
                      char* err, size_t errlen);
/* BamWriter + BgzfOutputStream (BioD/bio/std/hts/bam/writer.d:67-287, bgzf/outputstream.d): `stream` is the uncompressed BAM
 * byte stream ("BAM\1", header text, references, records); it is written to `path` as BGZF with the EOF block, and -- with_index
 * != 0 -- indexed into path + ".bai" (sbx_build_index). */
int sbx_write_bam(const char* path, const uint8_t* stream, size_t n, int level, int with_index, int device, char* err, size_t errlen);
/* `sambamba index` (createIndex / IndexBuilder, BioD/bio/std/hts/bam/bai/indexing.d:52-366): the BAI of a coordinate-sorted BAM.
 * The file goes through the device pipeline once (inflate, record chain, field decode: position, basesCovered(), stored
 * bin, virtual offsets); chunks per bin, the 16 kbp linear index, the metadata pseudo-bin 37450 and the no-coordinate count
 * are assembled on the host as IndexBuilder.put / finish do.  Bins are written in ascending id order (the reference's order
 * is that of a D associative array).  Like IndexBuilder (one pass over a stream of records, indexing.d:262-316) it does not
 * need the file resident: the blocks go through the device in batches sized by the free device memory, a batch ends in front
 * of the record that straddles its last block boundary, the next one starts with it.
 * SBX_ENOTSORTED when the reads are not coordinate-sorted. */
int sbx_build_index(const char* bam_path, const char* bai_path, int device, char* err, size_t errlen);

/* ---- engine seam ------------------------------------------------------------ */

/* new MultiBamReader(filenames) (multireader.d:244) + BamReader header parse (reader.d:101-125).
 * Several BAMs are processed as the reference's merged stream would be: the pileup of the merge is the union
 * of the files' reads, so every file goes through the device pipeline on its own and the per-position results
 * are added up; samples are the union of the @RG SM values in order of first appearance (depth.d:1170-1181 over
 * the merged header) and every file keeps its own RG-id -> sample table.  Reference dictionaries that differ
 * but are compatible are merged the way SamHeaderMerger does it (samheadermerger.d:127-177: a topological order
 * of the union of the @SQ lists; every context then holds the MERGED dictionary, ref_id arguments and results of
 * this API are ids of the merged dictionary, and the records' own ids are translated when they are read, as
 * adjustTagsInRange does, multireader.d:174-190).  Dictionaries that cannot be merged -- one name with two
 * lengths, contradicting orders -- fail with SBX_EUNSUPPORTED and the reference's message.  With -m, mates
 * are paired within a file only.
 * device = HIP device ordinal (or -1: use LOCAL_RANK / 0). */
sbx_ctx* sbx_open(const char* const* bam_paths, int n_bams, int device, char* err, size_t errlen);
void sbx_close(sbx_ctx*);
const char* sbx_last_error(sbx_ctx*);

int sbx_header(sbx_ctx*, sbx_header_info* out);
/* reference_sequences[i].name / .length (depth.d:455,576,1041,1223) */
const char* sbx_ref_name(sbx_ctx*, int ref_id);
int64_t sbx_ref_length(sbx_ctx*, int ref_id);
int sbx_ref_id(sbx_ctx*, const char* name);           /* bam.hasReference / bam[name].id; -1 if absent */
/* printer.sample_names (depth.d:1170-1181) */
const char* sbx_sample_name(sbx_ctx*, int sample_id);
/* raw SAM header text (for the D shim's SamHeader) */
const char* sbx_header_text(sbx_ctx*, size_t* len);

/* createFilterFromQuery (filtering.d:40-51).  query == NULL compiles the default
 * "mapping_quality > 0 and not duplicate and not failed_quality_control" (depth.d:1159).
 * The whole query language compiles; regular expressions in the common core of D's std.regex and ECMAScript
 * (no back-references, look-around or word boundaries; option i only; at most 2 per filter and 64 NFA states each),
 * anything else returns SBX_EUNSUPPORTED. */
int sbx_compile_filter(const char* query, sbx_filter* out, char* err, size_t errlen);
int sbx_set_filter(sbx_ctx*, const sbx_filter* f);
/* Host-side evaluation of one `=~` condition (the same NFA simulation the device runs): 1 = the pattern matches
 * somewhere in text[0, n), 0 = it does not, < 0 = the pattern is outside the supported subset (message in err). */
int sbx_regex_search(const char* pattern, const char* options, const char* text, size_t n, char* err, size_t errlen);

/* -q, -m, --combined (depth.d:1127-1132); window/overlap and -T (depth.d:712-715,1015-1018). */
int sbx_set_params(sbx_ctx*, int mode, uint8_t min_base_quality, int fix_mate_overlaps, int combined,
                   uint32_t window_size, uint32_t overlap, const uint32_t* cov_thresholds, int n_thresholds);

/* -L: the merged, sorted region list used to fetch reads (getReadsOverlapping, depth.d:1211 ->
 * randomaccessmanager.d:316-338).  n == 0 means "all reads" (bam.reads, depth.d:1214). */
int sbx_set_regions(sbx_ctx*, const sbx_region* regions, size_t n);

/* The -L argument of depth_main turned into regions the way the reference does it (depth.d:1184-1208): a BED file
 * (readIntervals / parseBed, sambamba/utils/common/bed.d:59-152: two-column lines mean [beg, beg+1), empty intervals
 * become one base long, lines naming contigs the BAM does not have are dropped) or, when the argument cannot be read
 * as one, a region string "ref[:beg[-end]]" (BioD/bio/core/region.d:97-246).  Afterwards sbx_parsed_regions copies out
 * the merged, sorted list (merged != 0: what sbx_set_regions wants) or the raw list in input order (what region mode
 * reports on), and sbx_parsed_region_line returns the input line that precedes the rows of raw region i
 * (depth.d:902-906; for a region string the synthetic "ref\tstart\tend", depth.d:1203-1206). */
int sbx_parse_regions(sbx_ctx*, const char* bed_path_or_region, size_t* n_merged, size_t* n_raw);
int sbx_parsed_regions(sbx_ctx*, int merged, sbx_region* out, size_t cap);
const char* sbx_parsed_region_line(sbx_ctx*, size_t raw_index);

/* Run BGZF inflate -> record index -> decode+accumulate on the device for everything the
 * current filter/params/regions select.  Results stay resident in HBM until the next
 * sbx_run()/sbx_close(); the sbx_depth_* getters below copy them out.  */
int sbx_run(sbx_ctx*);

/* Streaming over contigs.  One pass needs ~4.5 bytes of HBM per inflated byte (stream, token streams,
 * descriptors, counter tiles), so a whole-genome BAM is processed as consecutive batches of contigs --
 * the device-side counterpart of the reference's streaming BamReadRange (readrange.d:118-173).
 * sbx_plan_batches splits contigs [0, n_ref) into the fewest consecutive batches whose estimated footprint
 * stays below budget_bytes (0 = 70 % of the free device memory); an over-sized single contig gets a batch of
 * its own.  sbx_run_batch is sbx_run() restricted to the reads of contigs [first_ref, first_ref + n_refs)
 * (and to the regions of sbx_set_regions that lie on them, if any): afterwards the getters below answer for
 * those contigs only.  Results of the previous run / batch are replaced. */
typedef struct {
    uint32_t first_ref, n_refs;
    uint64_t est_bytes;
} sbx_batch;
int sbx_plan_batches(sbx_ctx*, uint64_t budget_bytes, sbx_batch* out, size_t cap, size_t* n_out);
int sbx_run_batch(sbx_ctx*, uint32_t first_ref, uint32_t n_refs);
/* sbx_run() restricted to the reads overlapping [beg, end) of ref_id (intersected with the regions of
 * sbx_set_regions, if any) -- the unit of sub-contig streaming and of position sharding across GPUs
 * (the analogue of one element of pileupChunks, BioD/bio/std/hts/bam/pileup.d:1011-1015; the reads come
 * through the BAI exactly as bam[ref][beg .. end] fetches them, randomaccessmanager.d:247-338).  Afterwards
 * the getters answer for positions in [beg, end) only: counters there are complete (every read overlapping
 * the interval was seen), outside they are partial. */
int sbx_run_interval(sbx_ctx*, uint32_t ref_id, uint32_t beg, uint32_t end);
/* Upload stage of the next sbx_run_interval(ref_id, beg, end) on its own: the BGZF blocks of the interval's work list are read from
 * the file and copied to the device (pinned staging, the context's copy stream); the results of the previous run stay valid and
 * may be read meanwhile -- by sbx_stream_base_rows / sbx_depth_* on ANOTHER host thread (the one exception to "one context, one
 * thread at a time").  A following sbx_run_interval with the same arguments finds the blocks resident and starts with the
 * kernels.  Lets a caller overlap file -> device, the kernels and device -> text of consecutive slices (cli.cpp does). */
int sbx_prefetch_interval(sbx_ctx*, uint32_t ref_id, uint32_t beg, uint32_t end);
/* The same fetch, but the run keeps only the reads whose LEFTMOST position lies in [beg, end) and counts every position
 * they cover, also beyond `end`: reads are partitioned between the intervals instead of clipped to them (what the elements of
 * pileupChunks are, pileup.d:1011-1015: chunks of READS).  Per-position counters of such runs are partial sums; adding them up
 * over a set of intervals that tile the contig (an all-reduce across GPUs, see sambamba_amd.dist_depth --reduce allreduce) gives
 * the counters of the whole.  Only without --fix-mate-overlaps (a pair must be resolved by one owner: SBX_EUNSUPPORTED). */
int sbx_run_interval_owned(sbx_ctx*, uint32_t ref_id, uint32_t beg, uint32_t end);

/* depth base: counters[(pos-beg)*n_samples*7 + s*7 + k] for pos in [beg,end) of ref_id
 * (k = A,C,G,T,other,DEL,REFSKIP; n_samples = 1 when --combined).  Positions no admitted read
 * spans are all-zero.  covered (optional, may be NULL): 1 byte per position, non-zero iff >= 1
 * admitted read spans the position (a pileup column exists there, pileup.d:345-397) -- needed
 * to tell "column with COV 0" from "no column" when min_base_quality > 0.
 * counters may be NULL when only `covered` is wanted.  The seven counters per position exist after a `base` run (and after
 * any run with --fix-mate-overlaps or several files); a region / window run keeps what its statistics are sums of -- the bases
 * counted and the depth of a position -- and answers a request for counters with SBX_EINVAL (`covered` is always available). */
int sbx_depth_base_tile(sbx_ctx*, uint32_t ref_id, uint32_t beg, uint32_t end, uint32_t* counters,
                        uint8_t* covered);

/* The same counters copied into DEVICE memory of the context's GPU (d_counters: (end - beg) * n_samples * 7 uint32, e.g. the
 * storage of a torch tensor handed to an RCCL all-reduce): device-to-device, nothing crosses PCIe. */
int sbx_depth_base_tile_device(sbx_ctx*, uint32_t ref_id, uint32_t beg, uint32_t end, void* d_counters);

/* depth region: stats[r*n_samples + s] and cov_counts[(r*n_samples + s)*n_thresholds + t] for the
 * raw (unmerged, input-order) region list given here (PerBedRegionPrinter, depth.d:879-931).
 * seen[r] != 0 iff at least one column hit region r (is_first_occurrence cleared). */
int sbx_depth_region_stats(sbx_ctx*, const sbx_region* raw_regions, size_t n_regions,
                           sbx_region_stats* stats, uint32_t* cov_counts, uint8_t* seen);

/* The same with a per-region lower bound on the read start (min_start[r] != 0: only reads with position >= min_start[r]
 * are counted for region r, n_bases being the sum over those reads).  This is what the reference's first ring of
 * OVERLAPPING windows computes (is_first_occurrence starts out false, depth.d:1031-1032): the host composes
 * `window --overlap` from this call and plain region statistics (cli.cpp WindowPrinter). */
int sbx_depth_region_stats_from(sbx_ctx*, const sbx_region* raw_regions, size_t n_regions, const uint32_t* min_start,
                                sbx_region_stats* stats, uint32_t* cov_counts, uint8_t* seen);

/* depth window: the same for windows [k*step, k*step + window) of ref_id, k in
 * [first_win, first_win + n_win) (PerWindowPrinter, depth.d:933-1077). */
int sbx_depth_window_stats(sbx_ctx*, uint32_t ref_id, uint64_t first_win, uint64_t n_win,
                           sbx_region_stats* stats, uint32_t* cov_counts);

/* Device-side row formatting of `depth base` (writeColumn's text, depth.d:534-555) for
 * [beg,end) of ref_id into a caller buffer; returns bytes written through *out_len, or
 * SBX_ENOMEM if cap is too small (then *out_len = required size). */
int sbx_format_base_rows(sbx_ctx*, uint32_t ref_id, uint32_t beg, uint32_t end, double min_cov, double max_cov,
                         int annotate, char* out, size_t cap, size_t* out_len);

/* sbx_format_base_rows with the text left in DEVICE memory: d_out is a device pointer on the context's device with room for cap
 * bytes (null with cap 0: measure only; *out_len receives the size either way, SBX_ENOMEM when cap is too small).  For a consumer
 * that keeps working on the device -- and for bench.py's `device_text` figure (the pass including its text, nothing over PCIe). */
int sbx_format_base_rows_device(sbx_ctx*, uint32_t ref_id, uint32_t beg, uint32_t end, double min_cov, double max_cov,
                                int annotate, void* d_out, size_t cap, size_t* out_len);
/* The same rows handed to a writer piece by piece, in output order.  The device formats the next piece while the
 * previous one travels to pinned host memory and the writer consumes the one before; `write` returns 0 on success
 * (anything else aborts with SBX_EIO).  `data` is only valid during the call.  Replaces the per-column
 * output.write(...) calls of writeColumn / writeEmptyColumns (depth.d:452-487,534-555) for a whole interval. */
typedef int (*sbx_write_fn)(void* user, const char* data, size_t n);
int sbx_stream_base_rows(sbx_ctx*, uint32_t ref_id, uint32_t beg, uint32_t end, double min_cov, double max_cov,
                         int annotate, sbx_write_fn write, void* user);

/* Timing / accounting of the last sbx_run() (for bench.py): per-kernel milliseconds measured
 * with HIP events on the engine's stream, record counts, byte counts. */
typedef struct {
    double ms_inflate, ms_index, ms_accumulate, ms_reduce, ms_total, ms_h2d;
    uint64_t n_records, n_admitted, n_bgzf_blocks;
    uint64_t compressed_bytes, uncompressed_bytes, counter_bytes, covered_positions;
    uint64_t launches_inflate, launches_index, launches_accumulate;
    double ms_huffman, ms_lz77;   /* the two kernels of ms_inflate */
    uint64_t n_malformed;         /* records dropped as malformed (always 0 after a successful run: they raise SBX_EFORMAT) */
    uint64_t n_runs;              /* record-chain runs of the device work list (1 for a whole file, one per merged BAI chunk group with -L) */
    uint64_t uploaded_bytes;      /* compressed bytes resident in HBM for this run (the BGZF blocks of the work list only) */
    uint64_t accumulate_read_bytes;   /* bytes the accumulate kernel has to read: 32-byte descriptors of the records + CIGAR and packed
                                         sequence of the admitted ones (+ their base qualities when min_base_quality > 0) */
    uint64_t token_bytes;             /* bytes of the literal + match-entry streams the Huffman kernel wrote and the LZ77 kernel read */
    uint64_t max_alignment_span;      /* longest alignment (reference positions) among the admitted records of the run */
    uint64_t reserved2, reserved3;
} sbx_run_stats;
int sbx_last_run_stats(sbx_ctx*, sbx_run_stats* out);

/* Geometry of the result tiles of the last sbx_run(): positions per tile and the effective
 * number of samples (1 under --combined). */
int sbx_tile_info(sbx_ctx*, uint32_t* tile_pos, uint32_t* n_samples);
/* Next maximal run of tiles of ref_id holding >= 1 admitted read, at or after position `from`:
 * [*beg,*end) in contig coordinates (tile-aligned end), or *beg == *end == UINT64_MAX when there is
 * none.  Lets a caller walk a contig without copying the all-zero stretches. */
int sbx_next_active_range(sbx_ctx*, uint32_t ref_id, uint64_t from, uint64_t* beg, uint64_t* end);

/* ---- several devices -----------------------------------------
 * The path shards by POSITION: the outputs of disjoint position ranges are disjoint, so one process drives N contexts -- one
 * sbx_open(..., device = k, ...) per device, each from its own thread -- and every context runs the slices it was given
 * (sbx_run_interval) and hands over its share of the result; there is no data-path collective (the reference's analogue of the
 * cut is pileupChunks, BioD/bio/std/hts/bam/pileup.d:1011-1015).  `sbx-depth --gpus N` and d/sbx_depth.d (sbxDepthRunSharded)
 * are built on these two calls. */
int sbx_device_count(void);          /* HIP devices visible to the process; 0 when there is none (never negative) */
typedef struct {
    uint32_t shard;    /* 0 .. n_shards - 1, ascending in the array */
    uint32_t ref_id;
    uint32_t beg;      /* [beg, end) of the contig; cuts inside a contig are multiples of `align` */
    uint32_t end;
} sbx_shard;
/* Equal shares of the concatenated reference for n_shards devices: whole contigs and, where a contig is cut, position
 * intervals inside it -- so ONE long contig shards as well as a genome does.  Every position of every contig belongs to
 * exactly one shard; a shard's intervals are in genome order; `align` (the tile size 1024, or the window size) keeps tiles /
 * windows whole.  Host arithmetic only (no device, no context).  *n_out receives the number of intervals (also with
 * SBX_ENOMEM, when cap is too small: n_ref + n_shards is always enough). */
int sbx_plan_shards(const int64_t* ref_lengths, int32_t n_ref, int32_t n_shards, uint32_t align, sbx_shard* out, size_t cap,
                    size_t* n_out);

/* Keep the compressed file resident in HBM across sbx_run() calls (bench: "inputs already
 * resident in HBM when the timed region starts"). */
int sbx_preload(sbx_ctx*);

#ifdef __cplusplus
}
#endif
#endif /* SBX_DEPTH_H */
