/**
 * d/sbx_depth.d -- D binding of libsbx_depth.so for sambamba, and the replacement of the body of
 * depth_main (sambamba/depth.d:1163-1234) that drives it.
 *
 * Source only: the build image has no D compiler (SURVEY.md F1).  The declarations below are kept in
 * lock-step with include/sbx_depth.h by tests/test_d_binding.py, which parses the struct and function
 * declarations of this file, lays the structs out with the C ABI's rules and compares every size with
 * sbx_abi_sizeof() of the built library, and every function name with the header's.
 *
 * It follows the one FFI precedent in the reference, BioD/bio/core/utils/zlib.d:6,139-163
 * (extern(C) prototypes, caller-owned buffers, int status turned into an exception).
 * Link with:  -L-lsbx_depth  (plus -L-L<dir of libsbx_depth.so>).
 */
module sbx_depth;

import std.algorithm : map, max, min, sort;
import std.array : array, split;
import std.conv : to;
import std.exception : enforce;
import std.format : format;
import std.stdio : File, stderr, stdout;
import std.string : fromStringz, stripRight, toStringz;

enum SBX_OK = 0;
enum SBX_ENOMEM = -8;
enum SBX_NCOUNTERS = 7;
enum SBX_MODE_BASE = 0;
enum SBX_MODE_REGION = 1;
enum SBX_MODE_WINDOW = 2;
enum SBX_FILTER_MAX_OPS = 64;
enum SBX_FILTER_STRINGS = 512;
enum SBX_FILTER_REGEXES = 2;
enum SBX_REGEX_STATES = 64;
enum SBX_REGEX_CLASSES = 8;

extern (C) nothrow @nogc {
    struct sbx_ctx;
    struct sbx_region { uint ref_id; uint start; uint end; }
    struct sbx_region_stats { uint n_reads; uint n_bases; }
    struct sbx_header_info {
        int n_ref; int n_samples; int n_read_groups; int sorted_by_coordinate; int has_index; int reserved;
        ulong n_bgzf_blocks; ulong compressed_bytes; ulong uncompressed_bytes;
    }
    struct sbx_regex_state { ubyte type; ubyte a; ubyte b; ubyte c; }
    struct sbx_regex {
        ubyte n_states; ubyte n_classes; ubyte start; ubyte reserved;
        sbx_regex_state[SBX_REGEX_STATES] states;
        ubyte[32][SBX_REGEX_CLASSES] classes;      // C: uint8_t classes[SBX_REGEX_CLASSES][32]
    }
    struct sbx_filter_op { ubyte kind; ubyte field; ubyte cmp; ubyte pad; uint mask; long value; }
    struct sbx_filter {
        int n_ops; int reserved;
        sbx_filter_op[SBX_FILTER_MAX_OPS] ops;
        char[SBX_FILTER_STRINGS] strings;
        int n_regex; int reserved2;
        sbx_regex[SBX_FILTER_REGEXES] regex;
    }
    struct sbx_batch { uint first_ref; uint n_refs; ulong est_bytes; }
    struct sbx_run_stats {
        double ms_inflate; double ms_index; double ms_accumulate; double ms_reduce; double ms_total; double ms_h2d;
        ulong n_records; ulong n_admitted; ulong n_bgzf_blocks;
        ulong compressed_bytes; ulong uncompressed_bytes; ulong counter_bytes; ulong covered_positions;
        ulong launches_inflate; ulong launches_index; ulong launches_accumulate;
        double ms_huffman; double ms_lz77;
        ulong n_malformed; ulong n_runs; ulong uploaded_bytes; ulong accumulate_read_bytes;
        ulong token_bytes; ulong max_alignment_span; ulong reserved2; ulong reserved3;
    }

    size_t sbx_abi_sizeof(const(char)* type_name);
    int sbx_bgzf_compress(const(ubyte)* input, size_t n, int level, int with_eof, int device, ubyte* output, size_t cap, size_t* out_len,
                          char* err, size_t errlen);
    int sbx_write_bam(const(char)* path, const(ubyte)* stream, size_t n, int level, int with_index, int device, char* err, size_t errlen);
    int sbx_build_index(const(char)* bam_path, const(char)* bai_path, int device, char* err, size_t errlen);
    int sbx_inflate_blocks(const(ubyte)* comp, const(ulong)* comp_off, const(uint)* comp_len, const(uint)* isize,
                           uint n_blocks, ubyte* out_, const(ulong)* out_off, char* err, size_t errlen);
    sbx_ctx* sbx_open(const(char*)* bam_paths, int n_bams, int device, char* err, size_t errlen);
    void sbx_close(sbx_ctx*);
    const(char)* sbx_last_error(sbx_ctx*);
    int sbx_header(sbx_ctx*, sbx_header_info*);
    const(char)* sbx_ref_name(sbx_ctx*, int);
    long sbx_ref_length(sbx_ctx*, int);
    int sbx_ref_id(sbx_ctx*, const(char)*);
    const(char)* sbx_sample_name(sbx_ctx*, int);
    const(char)* sbx_header_text(sbx_ctx*, size_t* len);
    int sbx_compile_filter(const(char)* query, sbx_filter* out_, char* err, size_t errlen);
    int sbx_set_filter(sbx_ctx*, const(sbx_filter)*);
    int sbx_regex_search(const(char)* pattern, const(char)* options, const(char)* text, size_t n, char* err, size_t errlen);
    int sbx_set_params(sbx_ctx*, int mode, ubyte min_bq, int fix_mate_overlaps, int combined, uint window, uint overlap,
                       const(uint)* thresholds, int n_thresholds);
    int sbx_set_regions(sbx_ctx*, const(sbx_region)*, size_t);
    int sbx_parse_regions(sbx_ctx*, const(char)* bed_path_or_region, size_t* n_merged, size_t* n_raw);
    int sbx_parsed_regions(sbx_ctx*, int merged, sbx_region* out_regions, size_t cap);
    const(char)* sbx_parsed_region_line(sbx_ctx*, size_t raw_index);
    int sbx_run(sbx_ctx*);
    int sbx_plan_batches(sbx_ctx*, ulong budget_bytes, sbx_batch* out_batches, size_t cap, size_t* n_out);
    int sbx_run_batch(sbx_ctx*, uint first_ref, uint n_refs);
    int sbx_run_interval(sbx_ctx*, uint ref_id, uint beg, uint end);
    int sbx_prefetch_interval(sbx_ctx*, uint ref_id, uint beg, uint end);
    int sbx_run_interval_owned(sbx_ctx*, uint ref_id, uint beg, uint end);
    int sbx_depth_base_tile_device(sbx_ctx*, uint ref_id, uint beg, uint end, void* d_counters);
    int sbx_depth_base_tile(sbx_ctx*, uint ref_id, uint beg, uint end, uint* counters, ubyte* covered);
    int sbx_depth_region_stats(sbx_ctx*, const(sbx_region)*, size_t, sbx_region_stats*, uint* cov_counts, ubyte* seen);
    int sbx_depth_region_stats_from(sbx_ctx*, const(sbx_region)*, size_t, const(uint)* min_start, sbx_region_stats*,
                                    uint* cov_counts, ubyte* seen);
    int sbx_depth_window_stats(sbx_ctx*, uint ref_id, ulong first_win, ulong n_win, sbx_region_stats*, uint* cov_counts);
    int sbx_format_base_rows(sbx_ctx*, uint ref_id, uint beg, uint end, double min_cov, double max_cov, int annotate,
                             char* out_buf, size_t cap, size_t* out_len);
    alias sbx_write_fn = int function(void* user, const(char)* data, size_t n);
    int sbx_stream_base_rows(sbx_ctx*, uint ref_id, uint beg, uint end, double min_cov, double max_cov, int annotate,
                             sbx_write_fn write, void* user);
    int sbx_last_run_stats(sbx_ctx*, sbx_run_stats*);
    int sbx_next_active_range(sbx_ctx*, uint ref_id, ulong from, ulong* beg, ulong* end);
    int sbx_tile_info(sbx_ctx*, uint* tile_pos, uint* n_samples);
    int sbx_preload(sbx_ctx*);
}

/// Thrown exactly where depth.d would throw; depth_main's catch (depth.d:1237-1244) prints it.
void sbxEnforce(sbx_ctx* ctx, int rc) {
    enforce(rc == SBX_OK, fromStringz(sbx_last_error(ctx)).idup);
}

/// What depth_main has parsed from the command line before it opens the BAM files (depth.d:1093-1158).
struct SbxDepthOptions {
    int mode;                    // SBX_MODE_*
    string query;                // -F, null = the default filter of depth.d:1159
    double min_cov = 0, max_cov = 1e50;
    ubyte min_base_quality;
    bool annotate, combined, fix_mate_overlaps;
    uint window_size, overlap;   // window mode (depth.d:1015-1018)
    uint[] cov_thresholds;       // -T (depth.d:712-715)
    sbx_region[] merged_bed;     // -L, merged and sorted as parseBed(..., true) returns it (bed.d:128-141)
    sbx_region[] raw_bed;        // region mode: the unmerged list in input order (depth.d:1192)
    string[] raw_bed_lines;      // region mode: the input lines that go in front of every row (depth.d:902-906)
}

private string fmtG(float f) { return format("%g", f); }   // write(float), depth.d:859-864

/// printRegionStats (depth.d:847-876): one row of region / window statistics.
private void printRegionRow(File output, ref const SbxDepthOptions o, string prefix, uint length,
                            sbx_region_stats st, const(uint)[] cov, string sample) {
    const float mean = cast(float) st.n_bases / length;
    const bool ok = mean >= o.min_cov && mean <= o.max_cov;
    if (!ok && !o.annotate) return;
    output.write(prefix, st.n_reads, '\t', fmtG(mean));
    foreach (j, thr; o.cov_thresholds) {
        const float pct = thr == 0 ? 100.0f : cast(float) cov[j] * 100 / length;
        output.write('\t', fmtG(pct));
    }
    if (!o.combined) output.write('\t', sample);
    if (o.annotate) output.write('\t', ok ? 'y' : 'n');
    output.write('\n');
}

/**
 * Replacement of depth.d:1163-1234 ("new MultiBamReader ... printer.close()").  depth_main keeps its option
 * parsing and its BED parsing (parseBed / parseRegion need the reference dictionary: use sbxOpen first and
 * sbx_ref_id / sbx_ref_length in place of bam.hasReference / bam[name]) and calls this with the opened context.
 * The header lines (`REF\tPOS...` / `# chrom\t...`) are printed by the caller exactly as printer.init() does.
 *
 * Text rules are the reference's: base rows come formatted from the device (K6 reproduces writeColumn /
 * writeEmptyColumns byte for byte), region and window rows are printed here with printRegionStats' rules.
 * Two stateful corners stay with the reference's own CPU code path (return false -> the caller runs the old
 * body of depth_main): `base -L` together with `-c 0`, and `window --overlap > 0`; sambamba_amd/csrc/cli.cpp
 * shows how to drive the same ABI for them.
 */
bool sbxDepthRun(sbx_ctx* ctx, ref const SbxDepthOptions o, File output) {
    if (o.mode == SBX_MODE_BASE && o.merged_bed.length && o.min_cov <= 0) return false;
    if (o.mode == SBX_MODE_WINDOW && o.overlap != 0) return false;

    sbx_header_info hi;
    sbxEnforce(ctx, sbx_header(ctx, &hi));
    enforce(hi.sorted_by_coordinate != 0, "All files must be coordinate-sorted");       // depth.d:1164-1165
    enforce(hi.has_index != 0, "All files must be indexed");                            // depth.d:1166

    char[512] err;
    sbx_filter f;
    enforce(sbx_compile_filter(o.query is null ? null : o.query.toStringz, &f, err.ptr, err.length) == SBX_OK,
            fromStringz(err.ptr).idup);
    sbxEnforce(ctx, sbx_set_filter(ctx, &f));
    sbxEnforce(ctx, sbx_set_params(ctx, o.mode, o.min_base_quality, o.fix_mate_overlaps ? 1 : 0, o.combined ? 1 : 0,
                                   o.window_size, o.overlap, o.cov_thresholds.ptr, cast(int) o.cov_thresholds.length));
    if (o.merged_bed.length)
        sbxEnforce(ctx, sbx_set_regions(ctx, o.merged_bed.ptr, o.merged_bed.length));   // bam.getReadsOverlapping(bed)

    string[] samples;
    foreach (s; 0 .. hi.n_samples) samples ~= fromStringz(sbx_sample_name(ctx, s)).idup;
    const uint S = o.combined ? 1 : cast(uint) samples.length;
    const size_t n_thr = max(1, o.cov_thresholds.length);

    // the device takes the file in batches of contigs sized to its free memory (one batch unless whole-genome sized)
    size_t n_batches;
    sbxEnforce(ctx, sbx_plan_batches(ctx, 0, null, 0, &n_batches));
    auto plan = new sbx_batch[n_batches];
    if (n_batches) sbxEnforce(ctx, sbx_plan_batches(ctx, 0, plan.ptr, plan.length, &n_batches));

    // region mode gathers per batch and prints at the end, in input order (PerBedRegionPrinter.close, depth.d:925-930)
    auto r_st = new sbx_region_stats[o.raw_bed.length * S];
    auto r_cov = new uint[o.raw_bed.length * S * n_thr];
    auto r_seen = new ubyte[o.raw_bed.length];
    bool seen_columns = false;       // window mode: nothing is printed before the first pileup column (SURVEY App. B-12)

    foreach (b; plan) {
        if (plan.length == 1) sbxEnforce(ctx, sbx_run(ctx));
        else sbxEnforce(ctx, sbx_run_batch(ctx, b.first_ref, b.n_refs));
        foreach (r; b.first_ref .. b.first_ref + b.n_refs) {
            const name = fromStringz(sbx_ref_name(ctx, cast(int) r)).idup;
            const ulong len = cast(ulong) max(0L, sbx_ref_length(ctx, cast(int) r));
            stderr.writeln("Processing reference #", r + 1, " (", name, ")");           // depth.d:1225-1229
            final switch (o.mode) {
            case SBX_MODE_BASE:
                // rows of [p, q) formatted on the device; with -c > 0 only active stretches can hold rows
                // (sbx_stream_base_rows hands the text over piece by piece from pinned buffers while the device formats
                // and copies the next piece; sbx_format_base_rows into a caller buffer is the synchronous form)
                void emit(ulong p, ulong q) {
                    static extern (C) int sink(void* user, const(char)* data, size_t n) {
                        try { (cast(File*) user).rawWrite(data[0 .. n]); return 0; } catch (Exception) { return 1; }
                    }
                    sbxEnforce(ctx, sbx_stream_base_rows(ctx, r, cast(uint) p, cast(uint) q, o.min_cov, o.max_cov, o.annotate ? 1 : 0,
                                                         &sink, &output));
                }
                if (o.merged_bed.length) {
                    foreach (g; o.merged_bed) if (g.ref_id == r) emit(g.start, g.end);     // outputRequired, depth.d:558-565
                } else if (o.min_cov <= 0) {
                    emit(0, len);          // (contigs without columns between two with columns: see cli.cpp BasePrinter)
                } else {
                    ulong from = 0, rb, re;
                    for (;;) {
                        sbxEnforce(ctx, sbx_next_active_range(ctx, r, from, &rb, &re));
                        if (rb == ulong.max) break;
                        emit(rb, min(re, len + 1024));
                        from = re;
                    }
                }
                break;
            case SBX_MODE_REGION:
                size_t[] ids;
                sbx_region[] sub;
                foreach (i, g; o.raw_bed) if (g.ref_id == r) { ids ~= i; sub ~= g; }
                if (!sub.length) break;
                auto st = new sbx_region_stats[sub.length * S];
                auto cv = new uint[sub.length * S * n_thr];
                auto sn = new ubyte[sub.length];
                sbxEnforce(ctx, sbx_depth_region_stats(ctx, sub.ptr, sub.length, st.ptr, cv.ptr, sn.ptr));
                foreach (j, id; ids) {
                    r_seen[id] = sn[j];
                    foreach (s; 0 .. S) {
                        r_st[id * S + s] = st[j * S + s];
                        foreach (t; 0 .. o.cov_thresholds.length)
                            r_cov[(id * S + s) * n_thr + t] = cv[(j * S + s) * o.cov_thresholds.length + t];
                    }
                }
                break;
            case SBX_MODE_WINDOW:
                const ulong n_win = len / o.window_size;            // only full windows are printed (depth.d:1057,1071)
                ulong first_col, fc_end;
                sbxEnforce(ctx, sbx_next_active_range(ctx, r, 0, &first_col, &fc_end));
                if (first_col == ulong.max && !seen_columns) break;  // lazily created `samples`: nothing before the first column
                ulong k0 = 0;
                if (!seen_columns) { k0 = first_col / o.window_size; seen_columns = true; }   // (exact first column: cli.cpp first_column)
                enum ulong CHW = 1u << 20;
                for (ulong k = k0; k < n_win; k += CHW) {
                    const ulong n = min(CHW, n_win - k);
                    auto st = new sbx_region_stats[cast(size_t) n * S];
                    auto cv = new uint[cast(size_t) n * S * n_thr];
                    sbxEnforce(ctx, sbx_depth_window_stats(ctx, r, k, n, st.ptr, cv.ptr));
                    foreach (i; 0 .. n) foreach (s; 0 .. S) {
                        const beg = (k + i) * o.window_size;
                        printRegionRow(output, o, name ~ "\t" ~ beg.to!string ~ "\t" ~ (beg + o.window_size).to!string ~ "\t",
                                       o.window_size, st[cast(size_t)(i * S + s)],
                                       cv[cast(size_t)((i * S + s) * o.cov_thresholds.length) .. $], samples[s]);
                    }
                }
                break;
            }
        }
    }
    if (o.mode == SBX_MODE_REGION) {
        bool any = false;
        foreach (v; r_seen) any |= v != 0;
        if (any) foreach (id, g; o.raw_bed) {
            const prefix = o.raw_bed_lines[id].stripRight ~ "\t";                        // depth.d:902-906
            foreach (s; 0 .. S)
                printRegionRow(output, o, prefix, g.end - g.start, r_st[id * S + s], r_cov[(id * S + s) * n_thr .. $], samples[s]);
        }
    }
    return true;
}

/// `auto bam = new MultiBamReader(bam_filenames)` (depth.d:1163) -> the device context.
sbx_ctx* sbxOpen(string[] bam_filenames) {
    char[512] err;
    auto paths = bam_filenames.map!toStringz.array;
    auto ctx = sbx_open(paths.ptr, cast(int) paths.length, -1, err.ptr, err.length);
    enforce(ctx !is null, fromStringz(err.ptr).idup);
    return ctx;
}
