/**
 * d/sbx_depth.d -- D binding of libsbx_depth.so for sambamba (source only: there is no D
 * compiler in the build image, see SURVEY.md F1; this file is what a sambamba maintainer adds).
 *
 * It follows the one FFI precedent in the reference, BioD/bio/core/utils/zlib.d:6,139-163
 * (extern(C) prototypes, caller-owned buffers, int status turned into an exception).
 * Link with:  -L-lsbx_depth  (plus -L-L<dir of libsbx_depth.so>).
 */
module sbx_depth;

import std.exception : enforce;
import std.string : toStringz, fromStringz;

extern (C) nothrow @nogc {
    struct sbx_ctx;
    struct sbx_region { uint ref_id, start, end; }
    struct sbx_region_stats { uint n_reads, n_bases; }
    struct sbx_header_info {
        int n_ref, n_samples, n_read_groups, sorted_by_coordinate, has_index, reserved;
        ulong n_bgzf_blocks, compressed_bytes, uncompressed_bytes;
    }
    struct sbx_filter_op { ubyte kind, field, cmp, pad; uint mask; long value; }
    struct sbx_filter { int n_ops, reserved; sbx_filter_op[64] ops; }

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
    int sbx_compile_filter(const(char)* query, sbx_filter* out_, char* err, size_t errlen);
    int sbx_set_filter(sbx_ctx*, const(sbx_filter)*);
    int sbx_set_params(sbx_ctx*, int mode, ubyte min_bq, int fix_mate_overlaps, int combined, uint window, uint overlap,
                       const(uint)* thresholds, int n_thresholds);
    int sbx_set_regions(sbx_ctx*, const(sbx_region)*, size_t);
    int sbx_run(sbx_ctx*);
    struct sbx_batch { uint first_ref, n_refs; ulong est_bytes; }
    int sbx_plan_batches(sbx_ctx*, ulong budget_bytes, sbx_batch* out_batches, size_t cap, size_t* n_out);
    int sbx_run_batch(sbx_ctx*, uint first_ref, uint n_refs);
    int sbx_depth_base_tile(sbx_ctx*, uint ref_id, uint beg, uint end, uint* counters, ubyte* covered);
    int sbx_depth_region_stats(sbx_ctx*, const(sbx_region)*, size_t, sbx_region_stats*, uint* cov_counts, ubyte* seen);
    int sbx_depth_region_stats_from(sbx_ctx*, const(sbx_region)*, size_t, const(uint)* min_start, sbx_region_stats*, uint* cov_counts, ubyte* seen);
    int sbx_depth_window_stats(sbx_ctx*, uint ref_id, ulong first_win, ulong n_win, sbx_region_stats*, uint* cov_counts);
    int sbx_format_base_rows(sbx_ctx*, uint ref_id, uint beg, uint end, double min_cov, double max_cov, int annotate,
                             char* out_buf, size_t cap, size_t* out_len);
    int sbx_next_active_range(sbx_ctx*, uint ref_id, ulong from, ulong* beg, ulong* end);
    int sbx_tile_info(sbx_ctx*, uint* tile_pos, uint* n_samples);
}

/// Thrown exactly where depth.d would throw; depth_main's catch (depth.d:1237-1244) prints it.
void sbxEnforce(sbx_ctx* ctx, int rc) {
    enforce(rc == 0, fromStringz(sbx_last_error(ctx)).idup);
}

/**
 * Patch to sambamba/depth.d (depth_main, lines 1163-1234).  `printer` keeps its text formatting
 * methods; only the source of the numbers changes:
 *
 *   // was: auto bam = new MultiBamReader(bam_filenames); ... foreach (column; pileup) printer.push(column);
 *   char[512] err;
 *   auto paths = bam_filenames.map!toStringz.array;
 *   auto ctx = sbx_open(paths.ptr, cast(int) paths.length, -1, err.ptr, err.length);
 *   enforce(ctx !is null, fromStringz(err.ptr).idup);
 *   scope(exit) sbx_close(ctx);
 *   sbx_header_info hi; sbxEnforce(ctx, sbx_header(ctx, &hi));
 *   enforce(hi.sorted_by_coordinate, "All files must be coordinate-sorted");      // depth.d:1164
 *   enforce(hi.has_index, "All files must be indexed");                            // depth.d:1166
 *   sbx_filter f; enforce(sbx_compile_filter(query is null ? null : query.toStringz, &f, err.ptr, err.length) == 0, ...);
 *   sbxEnforce(ctx, sbx_set_filter(ctx, &f));
 *   sbxEnforce(ctx, sbx_set_params(ctx, mode, printer.min_base_quality, printer.fix_mate_overlaps, printer.combined,
 *                                  window_size, overlap, cov_thresholds.ptr, cast(int) cov_thresholds.length));
 *   if (bed.length) sbxEnforce(ctx, sbx_set_regions(ctx, cast(sbx_region*) bed.ptr, bed.length)); // BamRegion has the same layout
 *   sbxEnforce(ctx, sbx_run(ctx));
 *   // base mode: walk active ranges, fetch counters, call PerBasePrinter.writeColumn-equivalent on each covered position
 *   // region/window mode: sbx_depth_region_stats / sbx_depth_window_stats, then printRegionStats (depth.d:847-876)
 *
 * sambamba_amd/csrc/cli.cpp is that host logic written in C++ (the build image has no D compiler);
 * it is a line-for-line guide for the D version.
 */
