// engine.cpp -- the C ABI of libsbx_depth.so (include/sbx_depth.h) and the device pipeline
//   compressed BGZF blocks in HBM -> K1 inflate -> K2 record index -> K3 decode+accumulate -> counters in HBM.
// Host code here is orchestration only; every byte of BGZF payload, every record and every
// counter is produced on the device.  There is no CPU fallback: without a HIP device the compute
// entry points fail with SBX_ENODEVICE.
//
// The unit of work is a WORK LIST of chain runs (kernels.hpp ChainRun): the whole file from its first record on,
// or -- with -L, with sbx_run_batch / sbx_run_interval -- the BGZF block runs that hold the merged BAI chunks of the
// requested regions (RandomAccessManager.getChunks / getReads, randomaccessmanager.d:247-348; StreamChunksSupplier,
// inputstream.d:257-345), every run starting at a record boundary the index names.  Only those blocks are uploaded
// and inflated; the inflated pieces are laid out back to back in one device buffer.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>

#include "bai_writer.hpp"
#include "bai_parallel.hpp"
#include "common.hpp"
#include "deflate_core.hpp"
#include "host_io.hpp"
#include "kernels.hpp"

namespace sbx {

// Padding behind the compressed bytes on the device.  The fast K1a lane (inflate2_core.hpp) prefetches its input 16 bytes at a time and
// tests the end of the block's payload only between deflate blocks: on a corrupt stream a lane may go on decoding what follows its
// block until its output position passes ISIZE.  Every literal/length symbol it consumes (<= 15 bits, <= 48 with a match) produces at
// least one output byte of at most 65536, and it reads at most kMaxSeg block headers (<= 600 bytes each), so it can run at most
// ~128 KiB beyond its own block -- into the following blocks, or, for the last blocks of a batch, into this padding.  (The general
// kernel, which re-decodes every block the fast one flags, tests its input per symbol.)
constexpr size_t kCompPad = 192 * 1024;

void require_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        throw Error(SBX_ENODEVICE, std::string("no HIP device available (libsbx_depth has no CPU fallback): ") +
                                       (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0) {
        const char* lr = getenv("LOCAL_RANK");
        device = lr ? atoi(lr) % n : 0;
    }
    if (device >= n) throw Error(SBX_ENODEVICE, "HIP device ordinal " + std::to_string(device) + " out of range");
    SBX_HIP(hipSetDevice(device));
}

static uint32_t floor_pow2(uint32_t x) {
    uint32_t p = 1;
    while (p * 2 <= x) p *= 2;
    return p;
}

// One run of the work list in FILE coordinates: BGZF blocks [blk0, blk1), inflated-stream offsets [ub, ue).
struct FileRun {
    uint32_t blk0, blk1;
    uint64_t ub, ue;
    bool open_end = false;        // ue is a block boundary in the middle of the record stream (ChainRun::open_end)
    bool operator==(const FileRun& o) const { return blk0 == o.blk0 && blk1 == o.blk1 && ub == o.ub && ue == o.ue && open_end == o.open_end; }
};

// The work list of a launch: file runs, and the per-block tables of the launch in compacted coordinates
// (block i of the launch is file block file_blk[i]; its payload sits at comp_off[i] of d_comp, its inflated
// bytes at out_off[i] of d_U).
struct WorkList {
    std::vector<FileRun> runs;
    std::vector<uint32_t> file_blk, comp_len, isize, run_of;
    std::vector<uint64_t> comp_off, out_off;        // out_off has n + 1 entries
    std::vector<ChainRun> chain;
    struct Range { uint64_t file_off, len, dst; };
    std::vector<Range> ranges;                      // file bytes -> d_comp (only when the file is not preloaded)
    uint64_t comp_bytes = 0, u_bytes = 0;
    size_t n_blocks() const { return file_blk.size(); }
};

// result words of a pass, written by async copies into pinned host memory and read after one synchronisation
struct HostResults {
    uint32_t flags[4];
    uint32_t n_active, n_deep;
    IndexStats st[kIndexStatSlots];
    uint64_t last_state;
    uint32_t max_partners, n_rewalked;
    unsigned long long tok_bytes[64];
    uint64_t straddler;           // open-ended run: start of the record that continues behind it (kOffUnknown: none)
};

}  // namespace sbx

using namespace sbx;

constexpr size_t kStageBytes = 32u << 20;   // one pinned staging buffer of the upload
constexpr int kStages = 4;                   // ... of a ring of four

struct sbx_ctx {
    std::string last_error;
    std::mutex err_mu;                       // last_error: sbx_prefetch_interval may run on a second thread
    std::atomic<double> upload_ms{0.0};      // wall clock of the last upload (make_resident)
    // several BAMs (MultiBamReader, multireader.d:244): this context is the first file and owns the merged view;
    // every further file is a complete single-file context of its own
    std::vector<sbx_ctx*> members;
    bool in_group = false;                   // this context is one file of several (the primary or a member): its tiles get merged
    // the last run kept ONE word per tile position and sample, {bases counted : 16 | depth : 16}, instead of the seven counters
    // (region / window modes without -m: launch_accumulate `compact`); sbx_depth_base_tile then knows `covered` only
    bool compact_counters = false;
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr, text_stream = nullptr;     // compute; file bytes host -> device; text device -> host
    FileMap file;
    BlockTable blocks;
    BamHeaderInfo hdr;
    BaiIndex bai;
    bool has_index = false;

    // parameters
    int mode = SBX_MODE_BASE;
    uint32_t min_bq = 0;
    bool fix_mate = false, combined = false;
    uint32_t window = 0, overlap = 0;
    std::vector<uint32_t> thresholds;
    sbx_filter filter;
    std::vector<sbx_region> regions;
    bool index_mode = false;        // sbx_build_index: every record is described, no index / sort order / read group is required, no K3
    // read ownership of the next run (sbx_run_interval_owned): own_ref >= 0
    int32_t own_ref = -1;
    uint32_t own_beg = 0, own_end = 0;

    // compressed input on the device: the whole file (sbx_preload) or the blocks of the current work list
    bool preloaded = false;
    DevBuf<uint8_t> d_comp;
    WorkList wl;                    // the work list whose tables (and, unless preloaded, payload bytes) are resident
    bool wl_resident = false;
    DevBuf<uint64_t> d_comp_off, d_out_off;
    DevBuf<uint32_t> d_comp_len, d_isize, d_run_of, d_status;
    DevBuf<ChainRun> d_runs;
    // pinned staging for host -> device copies of file bytes
    // sbx_stream_base_rows: two pieces of text in flight (device buffer, pinned host buffer, events)
    DevBuf<uint8_t> d_fmt_text2[2];
    uint8_t* text_host[2] = {nullptr, nullptr};
    size_t text_host_cap[2] = {0, 0};
    hipEvent_t text_ev_fmt[2] = {nullptr, nullptr}, text_ev_copy[2] = {nullptr, nullptr};
    std::string h_fmt_blob;
    bool fmt_blob_on_device = false;      // d_fmt_names / d_fmt_soff hold h_fmt_blob / h_fmt_soff
    std::vector<uint32_t> h_fmt_soff;
    uint8_t* stage[kStages] = {};
    hipEvent_t stage_ev[kStages] = {};
    hipEvent_t upload_done = nullptr;
    HostResults* res = nullptr;     // pinned

    DevBuf<uint8_t> d_U, d_scratch, d_lit;
    uint64_t primary_records = 0;   // records of THIS file in the last run (stats.n_records is the sum over files after a merge)
    // several BAMs whose dictionaries differ (merge_dictionaries, host_io.hpp): hdr.refs is the MERGED dictionary in every file's
    // context; the records and the BAI of a file speak its own ids.  Empty: the file's own dictionary is the merged one.
    std::vector<int32_t> own_to_merged, merged_to_own;
    DevBuf<int32_t> d_own_to_merged;
    uint64_t index_straddler = kOffUnknown;      // index mode, open-ended batch: work-list offset of the record the next batch starts with
    const uint8_t* U() const { return d_U.p; }
    DevBuf<uint32_t> d_ent, d_nent;
    DevBuf<uint64_t> d_entry, d_exit, d_state;
    DevBuf<uint32_t> d_count, d_flag;
    DevBuf<RecDesc> d_desc;
    DevBuf<int32_t> d_rec_ref;
    DevBuf<uint64_t> d_name_hash;
    uint64_t desc_cap = 0;
    DevBuf<uint32_t> d_mate, d_n_partners, d_mate_ext;
    DevBuf<int32_t> d_ref_len;
    DevBuf<uint32_t> d_tile_base, d_tile_lo, d_tile_hi, d_active, d_slot_of, d_n_active;
    DevBuf<unsigned long long> d_scan_part;      // scratch of the multi-workgroup scans of K2 (index.hip)
    DevBuf<uint32_t> d_counters, d_span;
    DevBuf<uint32_t> d_covm, d_addm;     // per-column quantities of region/window runs with --fix-mate-overlaps
    DevBuf<DeviceFilter> d_filter;
    DevBuf<uint8_t> d_ref_sets;
    DevBuf<char> d_rg_ids;
    DevBuf<uint32_t> d_rg_off;
    DevBuf<uint16_t> d_rg_sample;
    DevBuf<IndexStats> d_stats;
    DevBuf<unsigned long long> d_tok;      // token bytes of the last inflate (accounting)
    DevBuf<SortedRegion> d_sel;
    DevBuf<uint32_t> d_sel_first;
    DevBuf<uint32_t> d_fmt_len, d_fmt_soff;
    DevBuf<uint64_t> d_fmt_off;
    DevBuf<uint8_t> d_fmt_text;
    DevBuf<char> d_fmt_names;
    // host images of the small tables (they must outlive the asynchronous copies that read them)
    std::vector<int32_t> h_ref_len;
    std::vector<uint32_t> h_tile_base_up, h_sel_first, h_rg_off;
    std::vector<SortedRegion> h_sel;
    std::vector<uint8_t> h_ref_sets;
    std::string h_rg_ids;
    DeviceFilter h_df;
    bool filter_is_simple = false;     // the -F program of the last upload_static: only operations eval_filter_simple knows

    // region / window statistics: buffers and the preprocessed range list survive between calls (a caller that asks for the
    // same BED after every run -- bench.py config 4, the CLI per batch -- pays for sorting, chunking and uploading it once)
    struct RangeCache {
        std::vector<sbx_region> ranges;
        std::vector<uint32_t> min_start;
        bool has_min_start = false, valid = false;
        size_t n_chunks = 0;
        DevBuf<RangeChunk> d_chunks;
        DevBuf<SortedRegion> d_regs;
        DevBuf<uint32_t> d_pmax, d_first, d_min_start;
        bool sorted_valid = false;
        DevBuf<uint32_t> d_nb, d_nr, d_cov, d_seen, d_thr;
        std::vector<uint32_t> h_nb, h_nr, h_seen;
    } rc;

    // window mode: the statistics of EVERY window of the resident run, computed by the first sbx_depth_window_stats call after the run
    // (one pass over the records, one reduction over the positions) and handed out contig by contig
    struct WindowCache {
        bool valid = false;
        uint64_t serial = 0;
        uint32_t window = 0, S = 0;
        std::vector<uint32_t> thr;
        std::vector<uint64_t> base, n_win;          // per contig
        std::vector<sbx_region_stats> st;           // [window id][S]
        std::vector<uint32_t> cov;                  // [window id][S][n_thr]
        DevBuf<uint64_t> d_base, d_nwin;
        DevBuf<uint32_t> d_nb, d_nr, d_cov, d_seen, d_thr;
        std::vector<uint32_t> h_nb, h_nr;
    } wc;
    uint64_t run_serial = 0;                        // counts the runs of this context (have_run = true)
    // The same selection run again (a bench's passes, a caller that re-runs with other parameters): the BAI query and the grouping of
    // its chunks into runs (build_runs: two sorts of the region list, the bins of every contig) and the read-selection table of K2 are
    // functions of (selection, index) alone and are kept.  Config 4 (200 k regions, 90 k runs): profiles/round5/README.md.
    struct RunsCache {
        bool valid = false, restricted = false;
        std::vector<sbx_region> sel;
        std::vector<FileRun> runs;
    } runs_cache;
    std::vector<sbx_region> sel_uploaded;           // the selection whose table sits in d_sel / d_sel_first
    bool sel_uploaded_valid = false;

    // result of the last sbx_parse_regions
    std::vector<sbx_region> parsed_merged, parsed_raw;
    std::vector<std::string> parsed_lines;

    // results of the last run
    bool have_run = false;
    uint32_t tile_pos = 0, n_samples_eff = 1, n_tiles = 0, n_active = 0;
    // spare position tiles behind the last position of every contig, for alignments hanging over its end; enlarged (and kept) when a
    // pass meets an alignment that reaches beyond them (run_impl); spare_of_run: what the last pass was laid out with
    uint32_t spare_tiles = 1, spare_of_run = 0;
    // > 0 while sbx_stream_base_rows is handing text out.  The one call allowed on another thread meanwhile is
    // sbx_prefetch_interval: it touches the work list, the compressed bytes and the block tables (make_resident) -- nothing the text path
    // reads (counters, slot_of, tile_base, span, the format buffers).  A RUN replaces exactly those: it refuses while text is streaming.
    std::atomic<int> text_streaming{0};
    bool span_valid = false;
    std::vector<uint32_t> h_tile_base, h_slot_of;
    sbx_run_stats stats{};

    ~sbx_ctx() {
        for (int i = 0; i < 2; ++i) {
            if (text_host[i]) (void)hipHostFree(text_host[i]);
            if (text_ev_fmt[i]) (void)hipEventDestroy(text_ev_fmt[i]);
            if (text_ev_copy[i]) (void)hipEventDestroy(text_ev_copy[i]);
            if (stage[i]) (void)hipHostFree(stage[i]);
            if (stage_ev[i]) (void)hipEventDestroy(stage_ev[i]);
        }
        if (upload_done) (void)hipEventDestroy(upload_done);
        if (res) (void)hipHostFree(res);
    }
};

namespace {



// the files of a (possibly multi-BAM) context, the primary first
std::vector<sbx_ctx*> files_of(sbx_ctx* c) {
    std::vector<sbx_ctx*> v{c};
    v.insert(v.end(), c->members.begin(), c->members.end());
    return v;
}

template <class F>
int guarded(sbx_ctx* c, F&& f) {
    try {
        // several contexts on several devices, driven from any thread: the device is a property of the context, not of the thread
        if (c) SBX_HIP(hipSetDevice(c->device));
        f();
        return SBX_OK;
    } catch (const Error& e) {
        if (c) { std::lock_guard<std::mutex> g(c->err_mu); c->last_error = e.what(); }
        return e.code;
    } catch (const std::exception& e) {
        if (c) { std::lock_guard<std::mutex> g(c->err_mu); c->last_error = e.what(); }
        return SBX_EINVAL;
    }
}

void set_err(char* err, size_t n, const std::string& m) {
    if (err && n) snprintf(err, n, "%s", m.c_str());
}

// ---- work list ---------------------------------------------------------------------------------------
// virtual offset -> (file block, offset in the inflated stream of the file)
uint64_t voffset_to_stream(const sbx_ctx* c, uint64_t v, uint32_t* blk) {
    const uint32_t nb = (uint32_t)c->blocks.size();
    const uint64_t co = v >> 16, uo = v & 0xFFFF;
    const size_t bi = (size_t)(std::lower_bound(c->blocks.coffset.begin(), c->blocks.coffset.end(), co) - c->blocks.coffset.begin());
    if (bi >= nb) { *blk = nb; return c->blocks.out_off.back(); }     // at / beyond the EOF block
    if (c->blocks.coffset[bi] != co) throw Error(SBX_EFORMAT, "BAI virtual offset does not point at a BGZF block");
    *blk = (uint32_t)bi;
    return c->blocks.out_off[bi] + uo;
}

std::vector<sbx_region> sorted_regions(const std::vector<sbx_region>& sel) {
    std::vector<sbx_region> regs = sel;
    std::sort(regs.begin(), regs.end(), [](const sbx_region& a, const sbx_region& b) {
        if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
        if (a.start != b.start) return a.start < b.start;
        return a.end < b.end;
    });
    return regs;
}

// The runs of a pass.  restricted == false: every record of the file.  Otherwise: per contig, the merged BAI chunks
// of its merged regions (getGroupChunks, randomaccessmanager.d:247-294); chunks that share a BGZF block or are at
// most one block apart are joined into one run (what lies between two chunks is a whole number of records, which the
// read selection of K2 drops again), everything else stays a run of its own -- so a sparse BED touches only the
// blocks its chunks live in.
std::vector<FileRun> build_runs(const sbx_ctx* c, const std::vector<sbx_region>& sel, bool restricted) {
    std::vector<FileRun> runs;
    const uint32_t nb = (uint32_t)c->blocks.size();
    const uint64_t total = c->blocks.out_off.back(), first = c->hdr.first_record_off;
    if (!restricted) {
        if (first < total) {
            const uint32_t b0 = (uint32_t)(std::upper_bound(c->blocks.out_off.begin(), c->blocks.out_off.end(), first) - c->blocks.out_off.begin()) - 1;
            runs.push_back({b0, nb, first, total});
        }
        return runs;
    }
    const std::vector<sbx_region> regs = sorted_regions(sel);
    for (size_t i = 0; i < regs.size();) {
        size_t j = i;
        std::vector<sbx_region> group;
        while (j < regs.size() && regs[j].ref_id == regs[i].ref_id) {
            if (!group.empty() && group.back().end >= regs[j].start) group.back().end = std::max(group.back().end, regs[j].end);
            else group.push_back(regs[j]);
            ++j;
        }
        // (the index of a file speaks the file's own reference ids)
        int64_t own = regs[i].ref_id;
        if (!c->merged_to_own.empty()) own = regs[i].ref_id < c->merged_to_own.size() ? c->merged_to_own[regs[i].ref_id] : -1;
        for (auto& g : group) g.ref_id = (uint32_t)std::max<int64_t>(own, 0);
        if (own >= 0 && (size_t)own < c->bai.refs.size())
            for (auto& ch : group_chunks(c->bai, group)) {
                if (ch.beg >= ch.end) continue;
                uint32_t bb = 0, be = 0;
                uint64_t ub = voffset_to_stream(c, ch.beg, &bb), ue = voffset_to_stream(c, ch.end, &be);
                ub = std::max(ub, first);
                ue = std::min(ue, total);
                if (ub >= ue || bb >= nb) continue;
                while (bb + 1 < nb && c->blocks.out_off[bb + 1] <= ub) ++bb;      // (a chunk start at the very end of a block)
                const uint32_t b1 = (be < nb && ue > c->blocks.out_off[be]) ? be + 1 : be;
                runs.push_back({bb, std::max(b1, bb + 1), ub, ue});
            }
        i = j;
    }
    std::sort(runs.begin(), runs.end(), [](const FileRun& a, const FileRun& b) { return a.ub != b.ub ? a.ub < b.ub : a.ue < b.ue; });
    std::vector<FileRun> merged;
    for (auto& r : runs) {
        if (!merged.empty() && r.blk0 <= merged.back().blk1 + 1 && r.ub >= merged.back().ub) {
            FileRun& m = merged.back();
            m.ue = std::max(m.ue, r.ue);
            m.blk1 = std::max(m.blk1, r.blk1);
        } else merged.push_back(r);
    }
    return merged;
}

void build_worklist(const sbx_ctx* c, std::vector<FileRun> runs, bool file_resident, WorkList* w) {
    *w = WorkList();
    w->runs = std::move(runs);
    uint64_t uo = 0, co = 0;
    for (size_t ri = 0; ri < w->runs.size(); ++ri) {
        const FileRun& r = w->runs[ri];
        const uint32_t first_local = (uint32_t)w->file_blk.size();
        const uint64_t cbase = c->blocks.coffset[r.blk0];
        const uint64_t cend = c->blocks.comp_off[r.blk1 - 1] + c->blocks.comp_len[r.blk1 - 1] + 8;     // + CRC32, ISIZE
        if (!file_resident) w->ranges.push_back({cbase, cend - cbase, co});
        for (uint32_t b = r.blk0; b < r.blk1; ++b) {
            w->file_blk.push_back(b);
            w->comp_off.push_back(file_resident ? c->blocks.comp_off[b] : co + (c->blocks.comp_off[b] - cbase));
            w->comp_len.push_back(c->blocks.comp_len[b]);
            w->isize.push_back(c->blocks.isize[b]);
            w->run_of.push_back((uint32_t)ri);
            w->out_off.push_back(uo + (c->blocks.out_off[b] - c->blocks.out_off[r.blk0]));
        }
        const uint64_t ubase = c->blocks.out_off[r.blk0];
        w->chain.push_back({uo + (r.ub - ubase), uo + (r.ue - ubase), first_local, (uint32_t)w->file_blk.size() - 1, r.open_end ? 1u : 0u, 0u});
        uo += c->blocks.out_off[r.blk1] - ubase;
        co += (cend - cbase + 15) & ~15ull;
    }
    w->out_off.push_back(uo);
    w->u_bytes = uo;
    w->comp_bytes = file_resident ? c->file.size : co;
}

// ---- host -> device copies of file bytes: a pool of threads preads 2 MiB pieces into a ring of pinned staging buffers, the
// calling thread sends every buffer that is complete with an asynchronous DMA on the copy stream (two DMAs in flight while the
// other two buffers are being filled).  The page cache -> pinned copy is what bounds the upload (PCIe takes 57 GB/s, one
// thread copies ~3 GB/s), so the pieces are small and claimed in order: all threads work on the oldest incomplete buffer.
void ensure_staging(sbx_ctx* c) {
    for (int i = 0; i < kStages; ++i) {
        if (!c->stage[i]) SBX_HIP(hipHostMalloc((void**)&c->stage[i], kStageBytes, hipHostMallocDefault));
        if (!c->stage_ev[i]) SBX_HIP(hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
    }
    if (!c->upload_done) SBX_HIP(hipEventCreateWithFlags(&c->upload_done, hipEventDisableTiming));
}

void read_file_piece(const sbx_ctx* c, uint64_t off, size_t n, uint8_t* dst) {
    size_t lo = 0;
    while (lo < n) {
        ssize_t k = pread(c->file.fd, dst + lo, n - lo, (off_t)(off + lo));
        if (k <= 0) { memcpy(dst + lo, c->file.data + off + lo, n - lo); break; }     // (the mapping always works)
        lo += (size_t)k;
    }
}

unsigned upload_threads() {
    static const unsigned n = [] {
        if (const char* e = getenv("SBX_UPLOAD_THREADS")) return (unsigned)std::max(1, atoi(e));
        return std::min(8u, std::max(1u, std::thread::hardware_concurrency()));      // (4 .. 16 measured: 8 is the best by a little)
    }();
    return n;
}

// copies the ranges into d_comp; the compute stream waits for the last DMA (no host synchronisation here)
void upload_ranges(sbx_ctx* c, const std::vector<WorkList::Range>& ranges) {
    ensure_staging(c);
    struct Chunk { uint64_t file_off, dst; size_t n; };
    struct Piece { uint32_t chunk; uint32_t off, n; };
    constexpr size_t kPiece = 2u << 20;
    std::vector<Chunk> chunks;
    std::vector<Piece> pieces;
    for (auto& r : ranges)
        for (uint64_t done = 0; done < r.len;) {
            const size_t n = (size_t)std::min<uint64_t>(kStageBytes, r.len - done);
            for (size_t o = 0; o < n; o += kPiece) pieces.push_back({(uint32_t)chunks.size(), (uint32_t)o, (uint32_t)std::min(kPiece, n - o)});
            chunks.push_back({r.file_off + done, r.dst + done, n});
            done += n;
        }
    if (chunks.size() == 1 && pieces.size() <= 2) {       // a small transfer: no pool
        read_file_piece(c, chunks[0].file_off, chunks[0].n, c->stage[0]);
        SBX_HIP(hipMemcpyAsync(c->d_comp.p + chunks[0].dst, c->stage[0], chunks[0].n, hipMemcpyHostToDevice, c->copy_stream));
        SBX_HIP(hipEventRecord(c->stage_ev[0], c->copy_stream));
        SBX_HIP(hipEventSynchronize(c->stage_ev[0]));      // (the buffer may be refilled by the next call)
        SBX_HIP(hipEventRecord(c->upload_done, c->copy_stream));
        return;
    }
    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint32_t> left(chunks.size(), 0);
    for (auto& p : pieces) ++left[p.chunk];
    size_t avail = kStages;                 // chunks [0, avail) may be filled: the buffer of chunk x is free once chunk x - kStages has left it
    std::atomic<size_t> next{0};
    bool abort_all = false;
    auto worker = [&] {
        for (;;) {
            const size_t p = next.fetch_add(1);
            if (p >= pieces.size()) return;
            const Piece& pc = pieces[p];
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return abort_all || pc.chunk < avail; });
                if (abort_all) return;
            }
            read_file_piece(c, chunks[pc.chunk].file_off + pc.off, pc.n, c->stage[pc.chunk % kStages] + pc.off);
            std::lock_guard<std::mutex> g(mu);
            if (--left[pc.chunk] == 0) cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    const size_t n_thr = std::min<size_t>(upload_threads(), pieces.size());
    for (size_t t = 0; t < n_thr; ++t) pool.emplace_back(worker);
    struct Stop {       // an error on the way out must not leave the pool waiting
        std::mutex& mu; std::condition_variable& cv; bool& abort_all; std::vector<std::thread>& pool;
        ~Stop() { { std::lock_guard<std::mutex> g(mu); abort_all = true; } cv.notify_all(); for (auto& t : pool) t.join(); }
    } stop{mu, cv, abort_all, pool};
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
        { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return left[ci] == 0; }); }
        const int slot = (int)(ci % kStages);
        SBX_HIP(hipMemcpyAsync(c->d_comp.p + chunks[ci].dst, c->stage[slot], chunks[ci].n, hipMemcpyHostToDevice, c->copy_stream));
        SBX_HIP(hipEventRecord(c->stage_ev[slot], c->copy_stream));
        if (ci + 2 >= (size_t)kStages) {       // two DMAs stay in flight; the buffer of the one before them is free again
            const size_t j = ci + 2 - kStages;
            SBX_HIP(hipEventSynchronize(c->stage_ev[j % kStages]));
            std::lock_guard<std::mutex> g(mu);
            avail = j + kStages + 1;
            cv.notify_all();
        }
    }
    // the buffers must be free when the next call starts to fill them
    for (int i = 0; i < kStages; ++i) SBX_HIP(hipEventSynchronize(c->stage_ev[i]));
    SBX_HIP(hipEventRecord(c->upload_done, c->copy_stream));      // (the callers wait for the copy stream on the host)
}

// makes `runs` the resident work list: per-block tables on the device and (unless the file is preloaded) the payload bytes
void make_resident(sbx_ctx* c, std::vector<FileRun> runs) {
    if (c->wl_resident && c->wl.runs == runs) return;
    c->wl_resident = false;
    build_worklist(c, std::move(runs), c->preloaded, &c->wl);
    const WorkList& w = c->wl;
    const size_t n = w.n_blocks();
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    c->d_comp_off.ensure(n + 1);
    c->d_comp_len.ensure(n + 1);
    c->d_isize.ensure(n + 1);
    c->d_run_of.ensure(n + 1);
    c->d_out_off.ensure(n + 1);
    c->d_runs.ensure(w.chain.size() + 1);
    hipStream_t s = c->stream;
    if (n) {
        SBX_HIP(hipMemcpyAsync(c->d_comp_off.p, w.comp_off.data(), n * 8, hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(c->d_comp_len.p, w.comp_len.data(), n * 4, hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(c->d_isize.p, w.isize.data(), n * 4, hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(c->d_run_of.p, w.run_of.data(), n * 4, hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(c->d_runs.p, w.chain.data(), w.chain.size() * sizeof(ChainRun), hipMemcpyHostToDevice, s));
    }
    SBX_HIP(hipMemcpyAsync(c->d_out_off.p, w.out_off.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
    if (!c->preloaded) {
        c->d_comp.ensure((size_t)w.comp_bytes + kCompPad);
        upload_ranges(c, w.ranges);
        SBX_HIP(hipStreamSynchronize(c->copy_stream));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        c->upload_ms.store((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6, std::memory_order_relaxed);
    }
    c->wl_resident = true;
}

// Inflate the blocks of the resident work list into d_U.
void inflate_worklist(sbx_ctx* c, hipEvent_t ev_mid) {
    const WorkList& w = c->wl;
    const uint32_t n = (uint32_t)w.n_blocks();
    c->d_U.ensure((size_t)w.u_bytes + 128);
    c->d_status.ensure(n + 1);
    c->d_nent.ensure(n + 1);
    c->d_scratch.ensure(inflate_scratch_bytes(n));
    c->d_lit.ensure(inflate_lit_bytes(w.u_bytes, n));
    c->d_ent.ensure(inflate_ent_words(w.u_bytes, n));
    c->d_tok.ensure(64);
    SBX_HIP(hipMemsetAsync(c->d_tok.p, 0, 64 * 8, c->stream));
    launch_bgzf_inflate(c->d_comp.p, c->d_comp_off.p, c->d_comp_len.p, c->d_isize.p, c->d_out_off.p, c->d_U.p, n, 0, c->d_scratch.p,
                        c->d_lit.p, c->d_ent.p, c->d_nent.p, c->d_status.p, c->stream, ev_mid, c->d_tok.p);
}

// inflates the first k BGZF blocks of the file into host memory (BAM header at open)
void inflate_prefix(sbx_ctx* c, uint32_t k, std::vector<uint8_t>* host) {
    const BlockTable& bt = c->blocks;
    const uint64_t in_end = bt.comp_off[k - 1] + bt.comp_len[k - 1], out_end = bt.out_off[k];
    DevBuf<uint8_t> d_in(in_end + kCompPad), d_out(out_end + 128), d_scr(inflate_scratch_bytes(k)), d_lit(inflate_lit_bytes(out_end, k));
    DevBuf<uint32_t> d_ent(inflate_ent_words(out_end, k)), d_nent(k), d_clen(k), d_isz(k), d_st(k);
    DevBuf<uint64_t> d_coff(k), d_ooff(k);
    SBX_HIP(hipMemset(d_in.p + in_end, 0, 64));
    SBX_HIP(hipMemcpy(d_in.p, c->file.data, in_end, hipMemcpyHostToDevice));
    SBX_HIP(hipMemcpy(d_coff.p, bt.comp_off.data(), k * 8ull, hipMemcpyHostToDevice));
    SBX_HIP(hipMemcpy(d_ooff.p, bt.out_off.data(), k * 8ull, hipMemcpyHostToDevice));
    SBX_HIP(hipMemcpy(d_clen.p, bt.comp_len.data(), k * 4ull, hipMemcpyHostToDevice));
    SBX_HIP(hipMemcpy(d_isz.p, bt.isize.data(), k * 4ull, hipMemcpyHostToDevice));
    launch_bgzf_inflate(d_in.p, d_coff.p, d_clen.p, d_isz.p, d_ooff.p, d_out.p, k, 0, d_scr.p, d_lit.p, d_ent.p, d_nent.p, d_st.p, c->stream);
    std::vector<uint32_t> st(k);
    SBX_HIP(hipStreamSynchronize(c->stream));
    SBX_HIP(hipMemcpy(st.data(), d_st.p, k * 4ull, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < k; ++i)
        if (st[i] != 0)
            throw Error(SBX_EFORMAT, "Error inflating BGZF block starting from offset " + std::to_string(bt.coffset[i]) + ": " +
                                         inflate_status_string(st[i]));
    host->resize(out_end);
    SBX_HIP(hipMemcpy(host->data(), d_out.p, out_end, hipMemcpyDeviceToHost));
}

void parse_header_on_device(sbx_ctx* c) {
    uint64_t total = c->blocks.out_off.back();
    if (total < 12) throw Error(SBX_EFORMAT, "BAM header is truncated");
    uint32_t nb = (uint32_t)c->blocks.size();
    uint32_t k = std::min<uint32_t>(nb, 4);
    std::vector<uint8_t> host;
    for (;;) {
        inflate_prefix(c, k, &host);
        if (parse_bam_header(host.data(), host.size(), total, &c->hdr)) break;
        if (k == nb) throw Error(SBX_EFORMAT, "BAM header is truncated");
        k = std::min<uint32_t>(nb, k * 4);
    }
}

void default_filter(sbx_filter* f) {
    FilterCompiler fc("mapping_quality > 0 and not duplicate and not failed_quality_control", f);  // depth.d:1159
    fc.compile();
}

}  // namespace

extern "C" {

size_t sbx_abi_sizeof(const char* name) {
    if (!name) return 0;
    const std::string n = name;
    if (n == "sbx_region") return sizeof(sbx_region);
    if (n == "sbx_region_stats") return sizeof(sbx_region_stats);
    if (n == "sbx_header_info") return sizeof(sbx_header_info);
    if (n == "sbx_regex_state") return sizeof(sbx_regex_state);
    if (n == "sbx_regex") return sizeof(sbx_regex);
    if (n == "sbx_filter_op") return sizeof(sbx_filter_op);
    if (n == "sbx_filter") return sizeof(sbx_filter);
    if (n == "sbx_batch") return sizeof(sbx_batch);
    if (n == "sbx_run_stats") return sizeof(sbx_run_stats);
    if (n == "sbx_shard") return sizeof(sbx_shard);
    return 0;
}

int sbx_inflate_blocks(const uint8_t* comp, const uint64_t* comp_off, const uint32_t* comp_len, const uint32_t* isize,
                       uint32_t n_blocks, uint8_t* out, const uint64_t* out_off, char* err, size_t errlen) {
    try {
        require_device(-1);
        if (n_blocks == 0) return SBX_OK;
        uint64_t in_end = 0, out_end = 0;
        for (uint32_t i = 0; i < n_blocks; ++i) {
            in_end = std::max(in_end, comp_off[i] + comp_len[i]);
            out_end = std::max(out_end, out_off[i] + isize[i]);
        }
        DevBuf<uint8_t> d_in(in_end + kCompPad), d_out(out_end + 64), d_scr(inflate_scratch_bytes(n_blocks));
        DevBuf<uint8_t> d_lit(inflate_lit_bytes(out_end, n_blocks));
        DevBuf<uint32_t> d_ent(inflate_ent_words(out_end, n_blocks)), d_nent(n_blocks);
        DevBuf<uint64_t> d_coff(n_blocks), d_ooff(n_blocks);
        DevBuf<uint32_t> d_clen(n_blocks), d_isz(n_blocks), d_st(n_blocks);
        SBX_HIP(hipMemset(d_in.p + in_end, 0, 64));
        SBX_HIP(hipMemcpy(d_in.p, comp, in_end, hipMemcpyHostToDevice));
        SBX_HIP(hipMemcpy(d_coff.p, comp_off, n_blocks * 8ull, hipMemcpyHostToDevice));
        SBX_HIP(hipMemcpy(d_ooff.p, out_off, n_blocks * 8ull, hipMemcpyHostToDevice));
        SBX_HIP(hipMemcpy(d_clen.p, comp_len, n_blocks * 4ull, hipMemcpyHostToDevice));
        SBX_HIP(hipMemcpy(d_isz.p, isize, n_blocks * 4ull, hipMemcpyHostToDevice));
        launch_bgzf_inflate(d_in.p, d_coff.p, d_clen.p, d_isz.p, d_ooff.p, d_out.p, n_blocks, 0, d_scr.p, d_lit.p, d_ent.p,
                            d_nent.p, d_st.p, nullptr);
        std::vector<uint32_t> st(n_blocks);
        SBX_HIP(hipMemcpy(st.data(), d_st.p, n_blocks * 4ull, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_blocks; ++i)
            if (st[i]) throw Error(SBX_EFORMAT, "block " + std::to_string(i) + ": " + inflate_status_string(st[i]));
        // copy out exactly the produced ranges (blocks may be sparse in `out`)
        for (uint32_t i = 0; i < n_blocks; ++i)
            if (isize[i]) SBX_HIP(hipMemcpy(out + out_off[i], d_out.p + out_off[i], isize[i], hipMemcpyDeviceToHost));
        return SBX_OK;
    } catch (const Error& e) {
        set_err(err, errlen, e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_err(err, errlen, e.what());
        return SBX_EINVAL;
    }
}

int sbx_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

// shares of the concatenated reference (sambamba_amd/shard.py plan_position_shards states the same rule; tests/test_shard_plan_cpu.py
// holds the two against each other)
int sbx_plan_shards(const int64_t* ref_lengths, int32_t n_ref, int32_t n_shards, uint32_t align, sbx_shard* out, size_t cap, size_t* n_out) {
    if (n_out) *n_out = 0;
    if (n_ref < 0 || n_shards < 1 || align == 0 || (n_ref && !ref_lengths)) return SBX_EINVAL;
    std::vector<uint64_t> lens((size_t)n_ref), starts((size_t)n_ref);
    uint64_t total = 0;
    for (int32_t r = 0; r < n_ref; ++r) {
        lens[(size_t)r] = ref_lengths[r] > 0 ? (uint64_t)std::min<int64_t>(ref_lengths[r], 0x7FFFFFFF) : 0;
        starts[(size_t)r] = total;
        total += lens[(size_t)r];
    }
    if (total == 0) return SBX_OK;
    struct Cut { uint32_t ref; uint64_t pos; };
    auto less = [](const Cut& a, const Cut& b) { return a.ref != b.ref ? a.ref < b.ref : a.pos < b.pos; };
    std::vector<Cut> bounds((size_t)n_shards + 1);
    bounds[0] = {0, 0};
    for (int32_t k = 1; k < n_shards; ++k) {
        const uint64_t g = (uint64_t)((unsigned __int128)total * (uint64_t)k / (uint64_t)n_shards);      // 0 <= g < total
        const size_t r = (size_t)(std::upper_bound(starts.begin(), starts.end(), g) - starts.begin()) - 1;
        bounds[(size_t)k] = {(uint32_t)r, (g - starts[r]) / align * align};
        if (less(bounds[(size_t)k], bounds[(size_t)k - 1])) bounds[(size_t)k] = bounds[(size_t)k - 1];   // monotone (tiny contigs, more shards than tiles)
    }
    bounds[(size_t)n_shards] = {(uint32_t)n_ref, 0};
    size_t n = 0;
    for (int32_t k = 0; k < n_shards; ++k) {
        const Cut a = bounds[(size_t)k], b = bounds[(size_t)k + 1];
        for (uint32_t r = a.ref; r <= std::min<uint32_t>(b.ref, (uint32_t)n_ref - 1); ++r) {
            const uint64_t beg = r == a.ref ? a.pos : 0, end = r == b.ref ? b.pos : lens[r];
            if (end > beg) {
                if (out && n < cap) out[n] = {(uint32_t)k, r, (uint32_t)beg, (uint32_t)end};
                ++n;
            }
        }
    }
    if (n_out) *n_out = n;
    return (out && n <= cap) || (!out && cap == 0) ? SBX_OK : SBX_ENOMEM;
}

sbx_ctx* sbx_open(const char* const* bam_paths, int n_bams, int device, char* err, size_t errlen) {
    std::unique_ptr<sbx_ctx> c(new sbx_ctx());
    try {
        if (n_bams < 1 || !bam_paths || !bam_paths[0]) throw Error(SBX_EINVAL, "no input files");
        const bool timing = getenv("SBX_TIMING") != nullptr;
        auto now = [] { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
        const double t0 = now();
        // the host side of opening -- mapping the file, the BAI, the scan of the BGZF headers -- runs next to the bring-up of the
        // HIP runtime (80 ms for a process's first HIP call)
        std::exception_ptr host_err;
        bool host_joined = false;
        std::thread host([&] {
            try {
                c->file.open(bam_paths[0]);
                c->has_index = load_bai(c->file.path, &c->bai);
                // every virtual offset of the index names a BGZF block start: the header chain is scanned in pieces
                std::vector<uint64_t> hints;
                for (auto& r : c->bai.refs) {
                    for (uint64_t v : r.ioffsets) hints.push_back(v >> 16);
                    for (auto& b : r.bins) for (auto& ch : b.chunks) hints.push_back(ch.beg >> 16);
                }
                c->blocks = scan_bgzf(c->file.data, c->file.size, hints.empty() ? nullptr : &hints);
            } catch (...) { host_err = std::current_exception(); }
        });
        struct Joiner { std::thread& t; bool& done; ~Joiner() { if (!done && t.joinable()) t.join(); } } joiner{host, host_joined};
        require_device(device);
        SBX_HIP(hipGetDevice(&c->device));
        SBX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        SBX_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        SBX_HIP(hipStreamCreateWithFlags(&c->text_stream, hipStreamNonBlocking));
        const double t1 = now();
        host.join();
        host_joined = true;
        if (host_err) std::rethrow_exception(host_err);
        const double t2 = t1;
        default_filter(&c->filter);
        const double t3 = now();
        parse_header_on_device(c.get());
        if (timing)
            fprintf(stderr, "[sbx] open %s: device (with the BAI and the BGZF scan of %zu blocks next to it) %.3f s, wait for the scan %.3f s, header %.3f s\n",
                    bam_paths[0], c->blocks.size(), t1 - t0, t3 - t2, now() - t3);
        // further files: MultiBamReader semantics that matter for depth -- identical reference dictionaries
        // (the reference merges compatible ones, multireader.d:174-215; anything else is rejected here), samples =
        // union of the @RG SM values in order of first appearance (depth.d:1170-1181 over the merged header), every
        // file keeps its own RG-id -> sample table (so colliding RG ids need no renaming)
        for (int i = 1; i < n_bams; ++i) {
            if (!bam_paths[i]) throw Error(SBX_EINVAL, "null path");
            const char* one[1] = {bam_paths[i]};
            char e2[512] = {0};
            sbx_ctx* m = sbx_open(one, 1, c->device, e2, sizeof e2);
            if (!m) throw Error(SBX_EIO, e2);
            c->members.push_back(m);
            m->in_group = true;
            c->in_group = true;
            if (m->hdr.sorting_order != "coordinate") c->hdr.sorting_order = m->hdr.sorting_order;
            if (!m->has_index) c->has_index = false;
        }
        if (!c->members.empty()) {
            // the reference dictionary of the merged header (SamHeaderMerger); files whose own dictionary differs from it translate
            // the reference ids of their records on the device (RefTable::own_to_merged) and the ids of BAI queries on the host
            {
                const auto files = files_of(c.get());
                std::vector<const std::vector<RefSeq>*> dicts;
                for (sbx_ctx* f : files) dicts.push_back(&f->hdr.refs);
                std::vector<RefSeq> merged;
                std::vector<std::vector<int32_t>> maps;
                merge_dictionaries(dicts, &merged, &maps);
                for (size_t k = 0; k < files.size(); ++k) {
                    sbx_ctx* f = files[k];
                    bool same = f->hdr.refs.size() == merged.size();
                    for (size_t r = 0; same && r < merged.size(); ++r) same = maps[k][r] == (int32_t)r;
                    if (same) continue;
                    f->own_to_merged = maps[k];
                    f->merged_to_own.assign(merged.size(), -1);
                    for (size_t r = 0; r < maps[k].size(); ++r) f->merged_to_own[(size_t)maps[k][r]] = (int32_t)r;
                    f->hdr.refs = merged;
                }
            }
            std::vector<std::string> names;
            auto id_of = [&](const std::string& sm) -> uint16_t {
                for (size_t k = 0; k < names.size(); ++k) if (names[k] == sm) return (uint16_t)k;
                names.push_back(sm);
                return (uint16_t)(names.size() - 1);
            };
            for (sbx_ctx* f : files_of(c.get())) {
                f->hdr.rg_sample.clear();
                for (auto& g : f->hdr.read_groups) f->hdr.rg_sample.push_back(id_of(g.sample));
            }
            if (names.empty()) names.push_back("*");
            for (sbx_ctx* f : files_of(c.get())) f->hdr.sample_names = names;
        }
        return c.release();
    } catch (const std::exception& e) {
        set_err(err, errlen, e.what());
        if (c) for (sbx_ctx* m : c->members) sbx_close(m);
        if (c && c->stream) (void)hipStreamDestroy(c->stream);
        if (c && c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        if (c && c->text_stream) (void)hipStreamDestroy(c->text_stream);
        return nullptr;
    }
}

void sbx_close(sbx_ctx* c) {
    if (!c) return;
    for (sbx_ctx* m : c->members) sbx_close(m);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->text_stream) { (void)hipStreamSynchronize(c->text_stream); (void)hipStreamDestroy(c->text_stream); }
    delete c;
}

const char* sbx_last_error(sbx_ctx* c) {
    if (!c) return "null context";
    // (a prefetch on a second thread may set the message while this one reads it: the caller gets its own copy)
    static thread_local std::string copy;
    std::lock_guard<std::mutex> g(c->err_mu);
    copy = c->last_error;
    return copy.c_str();
}

int sbx_header(sbx_ctx* c, sbx_header_info* out) {
    if (!c || !out) return SBX_EINVAL;
    out->n_ref = (int32_t)c->hdr.refs.size();
    out->n_samples = (int32_t)c->hdr.sample_names.size();
    out->n_read_groups = (int32_t)c->hdr.read_groups.size();
    out->sorted_by_coordinate = c->hdr.sorting_order == "coordinate";
    out->has_index = c->has_index ? 1 : 0;
    out->reserved = 0;
    out->n_bgzf_blocks = c->blocks.size();
    out->compressed_bytes = c->file.size;
    out->uncompressed_bytes = c->blocks.out_off.back();
    return SBX_OK;
}
const char* sbx_ref_name(sbx_ctx* c, int r) { return (c && r >= 0 && (size_t)r < c->hdr.refs.size()) ? c->hdr.refs[(size_t)r].name.c_str() : nullptr; }
int64_t sbx_ref_length(sbx_ctx* c, int r) { return (c && r >= 0 && (size_t)r < c->hdr.refs.size()) ? c->hdr.refs[(size_t)r].length : -1; }
int sbx_ref_id(sbx_ctx* c, const char* name) { return (c && name) ? c->hdr.find_ref(name) : -1; }
const char* sbx_sample_name(sbx_ctx* c, int s) { return (c && s >= 0 && (size_t)s < c->hdr.sample_names.size()) ? c->hdr.sample_names[(size_t)s].c_str() : nullptr; }
const char* sbx_header_text(sbx_ctx* c, size_t* len) {
    if (!c) return nullptr;
    if (len) *len = c->hdr.text.size();
    return c->hdr.text.c_str();
}

int sbx_compile_filter(const char* query, sbx_filter* out, char* err, size_t errlen) {
    if (!out) return SBX_EINVAL;
    try {
        memset(out, 0, sizeof *out);
        if (!query) default_filter(out);
        else { FilterCompiler fc(query, out); fc.compile(); }
        return SBX_OK;
    } catch (const Error& e) {
        set_err(err, errlen, e.what());
        return e.code;
    }
}

int sbx_regex_search(const char* pattern, const char* options, const char* text, size_t n, char* err, size_t errlen) {
    if (!pattern || (!text && n)) return SBX_EINVAL;
    try {
        bool icase = false;
        for (const char* o = options; o && *o; ++o) {
            if (*o == 'i') icase = true;
            else throw Error(SBX_EUNSUPPORTED, std::string("filter: regular expression option '") + *o + "' is not supported on the device path");
        }
        sbx_regex re;
        RegexCompiler rc(pattern, icase, &re);
        rc.compile();
        return re_search(re, (uint32_t)n, [&](uint32_t k) { return (uint8_t)text[k]; }) ? 1 : 0;
    } catch (const Error& e) {
        set_err(err, errlen, e.what());
        return e.code;
    }
}

int sbx_set_filter(sbx_ctx* c, const sbx_filter* f) {
    if (!c || !f || f->n_ops < 0 || f->n_ops > SBX_FILTER_MAX_OPS) return SBX_EINVAL;
    for (sbx_ctx* m : files_of(c)) { m->filter = *f; m->have_run = false; }
    return SBX_OK;
}

int sbx_set_params(sbx_ctx* c, int mode, uint8_t min_bq, int fix_mate, int combined, uint32_t window, uint32_t overlap,
                   const uint32_t* thr, int n_thr) {
    return guarded(c, [&] {
        if (!c) throw Error(SBX_EINVAL, "null context");
        if (mode < 0 || mode > 2) throw Error(SBX_EINVAL, "unknown mode");
        if (mode == SBX_MODE_WINDOW) {
            if (window == 0) throw Error(SBX_EINVAL, "positive window size must be specified");        // depth.d:1020-1021
            if (overlap >= window) throw Error(SBX_EINVAL, "specified overlap is larger than window size");  // depth.d:1023-1024
        }
        for (sbx_ctx* m : files_of(c)) {
            m->mode = mode;
            m->min_bq = min_bq;
            m->fix_mate = fix_mate != 0;
            m->combined = combined != 0;
            m->window = window;
            m->overlap = overlap;
            m->thresholds.assign(thr, thr + (n_thr > 0 ? n_thr : 0));
            m->have_run = false;
        }
    });
}

int sbx_set_regions(sbx_ctx* c, const sbx_region* r, size_t n) {
    return guarded(c, [&] {
        if (!c) throw Error(SBX_EINVAL, "null context");
        for (size_t i = 0; i < n; ++i) {
            if (r[i].ref_id >= c->hdr.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
            if (!(r[i].start < r[i].end)) throw Error(SBX_EINVAL, "Enforcement failed");   // randomaccessmanager.d:256
        }
        for (sbx_ctx* m : files_of(c)) { m->regions.assign(r, r + n); m->have_run = false; }
    });
}

// -L argument -> regions, exactly as depth_main does it (depth.d:1184-1208): a BED file (bed.d:59-152), or, when it
// cannot be read as one, a region string (BioD/bio/core/region.d:97-246).
int sbx_parse_regions(sbx_ctx* c, const char* arg, size_t* n_merged, size_t* n_raw) {
    return guarded(c, [&] {
        if (!c || !arg) throw Error(SBX_EINVAL, "null argument");
        c->parsed_merged.clear(); c->parsed_raw.clear(); c->parsed_lines.clear();
        std::vector<BedInterval> ivs;
        std::vector<std::string> lines;
        std::vector<size_t> line_of;
        if (read_bed_file(arg, &ivs, &lines, &line_of)) {
            c->parsed_merged = bed_merged(ivs, c->hdr);
            // every kept region keeps its own input line (the reference pairs them by index, which misaligns when a
            // line names a contig the BAM does not have -- SURVEY App. B-6)
            for (size_t i = 0; i < ivs.size(); ++i) {
                const int id = c->hdr.find_ref(ivs[i].chr);
                if (id < 0) continue;
                c->parsed_raw.push_back({(uint32_t)id, (uint32_t)ivs[i].beg, (uint32_t)ivs[i].end});
                c->parsed_lines.push_back(lines[line_of[i]]);
            }
        } else {
            const RegionString rs = parse_region_string(arg);
            const int id = c->hdr.find_ref(rs.reference);
            if (id < 0) throw Error(SBX_EINVAL, std::string("couldn't open file ") + arg + " or find reference " + rs.reference);
            sbx_region g{(uint32_t)id, rs.beg, rs.end};
            if (g.end == 0xFFFFFFFFu) g.end = (uint32_t)c->hdr.refs[(size_t)id].length;
            c->parsed_merged.push_back(g);
            c->parsed_raw.push_back(g);
            c->parsed_lines.push_back(rs.reference + "\t" + std::to_string(g.start) + "\t" + std::to_string(g.end));
        }
        if (n_merged) *n_merged = c->parsed_merged.size();
        if (n_raw) *n_raw = c->parsed_raw.size();
    });
}
int sbx_parsed_regions(sbx_ctx* c, int merged, sbx_region* out, size_t cap) {
    if (!c || (!out && cap)) return SBX_EINVAL;
    const auto& v = merged ? c->parsed_merged : c->parsed_raw;
    if (cap < v.size()) return SBX_ENOMEM;
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return SBX_OK;
}
const char* sbx_parsed_region_line(sbx_ctx* c, size_t raw_index) {
    return (c && raw_index < c->parsed_lines.size()) ? c->parsed_lines[raw_index].c_str() : nullptr;
}

int sbx_preload(sbx_ctx* c) {
    return guarded(c, [&] {
        if (!c) throw Error(SBX_EINVAL, "null context");
        SBX_HIP(hipSetDevice(c->device));
        for (sbx_ctx* m : files_of(c)) {
            if (m->preloaded) continue;
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            m->d_comp.alloc(m->file.size + kCompPad);
            SBX_HIP(hipMemsetAsync(m->d_comp.p + m->file.size, 0, 64, m->copy_stream));
            upload_ranges(m, {{0, m->file.size, 0}});
            SBX_HIP(hipStreamSynchronize(m->copy_stream));
            clock_gettime(CLOCK_MONOTONIC, &t1);
            m->upload_ms.store((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6, std::memory_order_relaxed);
            m->preloaded = true;
            m->wl_resident = false;
        }
    });
}

// small tables of a pass that depend on the header, the filter and the read selection only
static void upload_static(sbx_ctx* c, const std::vector<sbx_region>& sel, bool restricted, uint32_t T, uint64_t* n_tiles,
                          RefTable* refs_out, RgTable* rg_out) {
    hipStream_t s = c->stream;
    const int32_t n_ref = (int32_t)c->hdr.refs.size();
    c->h_ref_len.assign((size_t)n_ref, 0);
    c->h_tile_base_up.assign((size_t)n_ref + 1, 0);
    uint64_t nt = 0;
    for (int32_t r = 0; r < n_ref; ++r) {
        c->h_ref_len[(size_t)r] = c->hdr.refs[(size_t)r].length;
        c->h_tile_base_up[(size_t)r] = (uint32_t)nt;
        // spare tiles per contig for alignments hanging over the contig end
        nt += ((uint64_t)std::max(0, c->hdr.refs[(size_t)r].length) + T - 1) / T + c->spare_tiles;
        if (nt > 0xFFFFFFF0ull) throw Error(SBX_EUNSUPPORTED, "too many position tiles");
    }
    c->h_tile_base_up[(size_t)n_ref] = (uint32_t)nt;
    *n_tiles = nt;
    c->d_ref_len.ensure((size_t)n_ref + 1);
    c->d_tile_base.ensure((size_t)n_ref + 1);
    if (n_ref) SBX_HIP(hipMemcpyAsync(c->d_ref_len.p, c->h_ref_len.data(), (size_t)n_ref * 4, hipMemcpyHostToDevice, s));
    SBX_HIP(hipMemcpyAsync(c->d_tile_base.p, c->h_tile_base_up.data(), ((size_t)n_ref + 1) * 4, hipMemcpyHostToDevice, s));
    // -L: merged, start-sorted regions per contig for the read selection in K2
    const bool sel_same = restricted && c->sel_uploaded_valid && c->sel_uploaded.size() == sel.size() && c->d_sel.n && c->d_sel_first.n &&
                          (sel.empty() || memcmp(c->sel_uploaded.data(), sel.data(), sel.size() * sizeof(sbx_region)) == 0);
    if (restricted && !sel_same) {
        c->sel_uploaded_valid = false;
        const std::vector<sbx_region> regs = sorted_regions(sel);
        c->h_sel.clear();
        c->h_sel_first.assign((size_t)n_ref + 1, 0);
        size_t j = 0;
        for (int32_t r = 0; r < n_ref; ++r) {
            c->h_sel_first[(size_t)r] = (uint32_t)c->h_sel.size();
            bool open = false;
            while (j < regs.size() && regs[j].ref_id == (uint32_t)r) {
                if (open && c->h_sel.back().end >= regs[j].start) c->h_sel.back().end = std::max(c->h_sel.back().end, regs[j].end);
                else { c->h_sel.push_back({regs[j].start, regs[j].end, 0}); open = true; }
                ++j;
            }
        }
        c->h_sel_first[(size_t)n_ref] = (uint32_t)c->h_sel.size();
        c->d_sel.ensure(c->h_sel.size() + 1);
        c->d_sel_first.ensure((size_t)n_ref + 2);
        if (!c->h_sel.empty()) SBX_HIP(hipMemcpyAsync(c->d_sel.p, c->h_sel.data(), c->h_sel.size() * sizeof(SortedRegion), hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(c->d_sel_first.p, c->h_sel_first.data(), ((size_t)n_ref + 1) * 4, hipMemcpyHostToDevice, s));
        SBX_HIP(hipStreamSynchronize(s));           // (the table is kept: the host copies may change before the next run needs them)
        c->sel_uploaded = sel;
        c->sel_uploaded_valid = true;
    }
    if (!c->own_to_merged.empty() && !c->d_own_to_merged.n) {
        c->d_own_to_merged.alloc(c->own_to_merged.size() + 1);
        SBX_HIP(hipMemcpyAsync(c->d_own_to_merged.p, c->own_to_merged.data(), c->own_to_merged.size() * 4, hipMemcpyHostToDevice, s));
    }
    *refs_out = RefTable{c->d_ref_len.p, c->d_tile_base.p, n_ref, restricted ? c->d_sel.p : nullptr, restricted ? c->d_sel_first.p : nullptr,
                         c->own_to_merged.empty() ? nullptr : c->d_own_to_merged.p,
                         c->own_to_merged.empty() ? n_ref : (int32_t)c->own_to_merged.size()};
    // filter
    c->d_filter.ensure(1);
    DeviceFilter& df = c->h_df;
    memset(&df, 0, sizeof df);
    df.n_ops = c->filter.n_ops;
    memcpy(df.ops, c->filter.ops, sizeof(sbx_filter_op) * (size_t)c->filter.n_ops);
    memcpy(df.strings, c->filter.strings, sizeof df.strings);
    memcpy(df.regex, c->filter.regex, sizeof df.regex);
    df.n_ref = n_ref;
    c->h_ref_sets.clear();
    for (int i = 0; i < df.n_ops; ++i) {
        // ref_name / mate_ref_name == 'x' becomes a comparison of the reference id ("*" is the name of id -1)
        sbx_filter_op& o = df.ops[i];
        if (o.kind == 15 && o.field >= 4) {
            // ref_name =~ /re/: one byte per reference id + 1 ("*", the name of id -1, first)
            const sbx_regex& re = df.regex[o.value & 1];
            const size_t at0 = c->h_ref_sets.size();
            auto hit = [&](const std::string& nm) { return re_search(re, (uint32_t)nm.size(), [&](uint32_t k) { return (uint8_t)nm[k]; }) ? 1 : 0; };
            c->h_ref_sets.push_back((uint8_t)hit("*"));
            for (auto& r : c->hdr.refs) c->h_ref_sets.push_back((uint8_t)hit(r.name));
            o.kind = 16;
            o.field = (uint8_t)(o.field - 4);
            o.value = (int64_t)at0;
            continue;
        }
        if (o.kind != 11) continue;
        const size_t off = (size_t)(o.value & 0xFFFFFFFF), len = (size_t)(o.value >> 32);
        const std::string name(df.strings + std::min(off, sizeof df.strings), std::min(len, sizeof df.strings - std::min(off, sizeof df.strings)));
        const int id = name == "*" ? -1 : c->hdr.find_ref(name);
        if (id < 0 && name != "*") { o.kind = (o.cmp == 4) ? 12 : 6; continue; }     // unknown name: never equal
        o.kind = 2;
        o.field = o.field ? 4 : 0;
        o.value = id;
    }
    c->d_ref_sets.ensure(c->h_ref_sets.size() + 1);
    if (!c->h_ref_sets.empty()) SBX_HIP(hipMemcpyAsync(c->d_ref_sets.p, c->h_ref_sets.data(), c->h_ref_sets.size(), hipMemcpyHostToDevice, s));
    df.ref_sets = c->d_ref_sets.p;
    c->filter_is_simple = true;
    for (int i = 0; i < df.n_ops; ++i) c->filter_is_simple = c->filter_is_simple && filter_op_is_simple(df.ops[i].kind, df.ops[i].field);
    if (const char* e = getenv("SBX_K2_SIMPLE_FILTER")) c->filter_is_simple = c->filter_is_simple && atoi(e) != 0;      // (A/B: 0 = the interpreter always)
    SBX_HIP(hipMemcpyAsync(c->d_filter.p, &df, sizeof df, hipMemcpyHostToDevice, s));
    // read groups
    *rg_out = RgTable{nullptr, nullptr, nullptr, 0, 0, 0};
    if (!c->hdr.read_groups.empty()) {
        c->h_rg_ids.clear();
        c->h_rg_off.clear();
        for (auto& g : c->hdr.read_groups) { c->h_rg_off.push_back((uint32_t)c->h_rg_ids.size()); c->h_rg_ids += g.id; c->h_rg_ids.push_back('\0'); }
        c->d_rg_ids.ensure(c->h_rg_ids.size());
        c->d_rg_off.ensure(c->h_rg_off.size());
        c->d_rg_sample.ensure(c->h_rg_off.size());
        SBX_HIP(hipMemcpyAsync(c->d_rg_ids.p, c->h_rg_ids.data(), c->h_rg_ids.size(), hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(c->d_rg_off.p, c->h_rg_off.data(), c->h_rg_off.size() * 4, hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(c->d_rg_sample.p, c->hdr.rg_sample.data(), c->h_rg_off.size() * 2, hipMemcpyHostToDevice, s));
        *rg_out = RgTable{c->d_rg_ids.p, c->d_rg_off.p, c->d_rg_sample.p, (int32_t)c->h_rg_off.size(), 1, (uint32_t)c->h_rg_ids.size()};
    }
}

// The whole device pipeline for the reads selected by `sel` (restricted == false: every read of the file).
static void run_impl(sbx_ctx* c, const std::vector<sbx_region>& sel, bool restricted, const std::vector<FileRun>* given_runs = nullptr) {
    if (!c->index_mode) {
        if (c->hdr.sorting_order != "coordinate") throw Error(SBX_ENOTSORTED, "All files must be coordinate-sorted");
        if (!c->has_index) throw Error(SBX_ENOINDEX, "All files must be indexed");
    }
    SBX_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    c->have_run = false;
    c->stats = sbx_run_stats{};
    if (!c->res) SBX_HIP(hipHostMalloc((void**)&c->res, sizeof(HostResults), hipHostMallocDefault));
    HostResults& R = *c->res;

    // ---- work list, tables, compressed bytes ----
    struct timespec tp0, tp1, tp2;
    clock_gettime(CLOCK_MONOTONIC, &tp0);
    if (given_runs) make_resident(c, *given_runs);
    else {
        sbx_ctx::RunsCache& rcache = c->runs_cache;
        // (SBX_RUNS_CACHE=0: every run builds its work list, as a one-shot command does -- bench.py times config 4 both ways)
        const char* rc_env = getenv("SBX_RUNS_CACHE");
        if (rc_env && atoi(rc_env) == 0) { rcache.valid = false; c->sel_uploaded_valid = false; }
        const bool hit = rcache.valid && rcache.restricted == restricted && rcache.sel.size() == sel.size() &&
                         (sel.empty() || memcmp(rcache.sel.data(), sel.data(), sel.size() * sizeof(sbx_region)) == 0);
        if (!hit) {
            rcache.valid = false;
            rcache.runs = build_runs(c, sel, restricted);
            rcache.sel = sel;
            rcache.restricted = restricted;
            rcache.valid = true;
        }
        clock_gettime(CLOCK_MONOTONIC, &tp1);
        make_resident(c, rcache.runs);
    }
    clock_gettime(CLOCK_MONOTONIC, &tp2);
    if (getenv("SBX_TIMING") && !given_runs)
        fprintf(stderr, "[sbx] run: work list %.1f ms, resident %.1f ms\n", (tp1.tv_sec - tp0.tv_sec) * 1e3 + (tp1.tv_nsec - tp0.tv_nsec) * 1e-6,
                (tp2.tv_sec - tp1.tv_sec) * 1e3 + (tp2.tv_nsec - tp1.tv_nsec) * 1e-6);
    c->stats.ms_h2d = c->upload_ms.load(std::memory_order_relaxed);      // (the upload may have been a prefetch on another thread)
    const WorkList& w = c->wl;
    const uint32_t nb = (uint32_t)w.n_blocks();
    EventTimer t_all, t1, t1m, t2, t3;
    t_all.start(s);

    // ---- K1 ----
    t1.start(s);
    inflate_worklist(c, t1m.b);
    t1.stop(s);

    // ---- K2 ----
    const int32_t n_ref = (int32_t)c->hdr.refs.size();
    const uint32_t S = c->combined ? 1u : (uint32_t)c->hdr.sample_names.size();
    const uint32_t T = std::max<uint32_t>(16, floor_pow2(std::max<uint32_t>(1, 1024u / std::max<uint32_t>(1, S))));
    if ((size_t)448 * S + 64 > 160u * 1024)
        throw Error(SBX_EUNSUPPORTED, "too many samples for the device path (" + std::to_string(S) + "): the counters of a position tile no longer fit "
                                      "the LDS of a compute unit; use --combined");
    uint64_t nt = 0;
    RefTable refs{};
    RgTable rg{};
    upload_static(c, sel, restricted, T, &nt, &refs, &rg);
    if (c->index_mode) rg.lookup = 0;

    c->d_entry.ensure(nb + 1);
    c->d_exit.ensure(nb + 1);
    c->d_state.ensure(nb + 1);
    c->d_count.ensure(nb + 1);
    c->d_flag.ensure(12);
    c->d_tile_lo.ensure((size_t)nt + 1);
    c->d_tile_hi.ensure((size_t)nt + 1);
    c->d_active.ensure((size_t)nt + 1);
    c->d_slot_of.ensure((size_t)nt + 1);
    c->d_n_active.ensure(4);
    c->d_scan_part.ensure(kScanPartWords);
    c->d_stats.ensure(kIndexStatSlots);
    // descriptor capacity: sized for records of >= 160 bytes on average; K2 reports an overflow and the pass is repeated
    // with the exact number (short-read fixtures, amplicon data with tiny records)
    uint64_t want_cap = std::max<uint64_t>(c->desc_cap, w.u_bytes / 160 + 4096);
    if (want_cap > c->desc_cap) want_cap += (uint64_t)((double)want_cap * devbuf_slack_pct().load(std::memory_order_relaxed) / 100.0);
    const bool dbg = getenv("SBX_DEBUG") != nullptr;
    const char* force = getenv("SBX_FORCE_REPAIR");   // debug hook (tests/test_gpu_repair.py)
    // tiles with this many records or more keep 32-bit LDS counters in K3 (debug hook: a small value sends ordinary tiles
    // down that path, tests/test_gpu_depth.py)
    uint32_t deep_thr = kDeepTileRecords;
    if (const char* e = getenv("SBX_DEEP_TILE_RECORDS")) { const long v = atol(e); if (v >= 1 && v < (long)kDeepTileRecords) deep_thr = (uint32_t)v; }
    uint32_t n_rewalked = 0;
    uint64_t n_records = 0;
    bool entries_given = false, spare_retried = false;
    t2.start(s);
    for (int attempt = 0;; ++attempt) {
        if (attempt > 4) throw Error(SBX_EFORMAT, "BAM record chain does not converge");
        if (want_cap > 0xFFFFFFF0ull) throw Error(SBX_EUNSUPPORTED, "more than 2^32 records in one batch");
        if (want_cap > c->desc_cap) {
            c->d_desc.release(); c->d_rec_ref.release();
            c->d_desc.alloc((size_t)want_cap + 64);
            c->d_rec_ref.alloc((size_t)want_cap + 64);
            if (c->d_name_hash.n) { c->d_name_hash.release(); }
            c->desc_cap = want_cap;
        }
        if (c->fix_mate) c->d_name_hash.ensure((size_t)c->desc_cap + 64);
        SBX_HIP(hipMemsetAsync(c->d_tile_lo.p, 0xFF, (size_t)nt * 4, s));
        SBX_HIP(hipMemsetAsync(c->d_tile_hi.p, 0, (size_t)nt * 4, s));
        SBX_HIP(hipMemsetAsync(c->d_stats.p, 0, sizeof(IndexStats) * kIndexStatSlots, s));
        SBX_HIP(hipMemsetAsync(c->d_state.p, 0, ((size_t)nb + 1) * 8, s));
        SBX_HIP(hipMemsetAsync(c->d_flag.p, 0xFF, 8, s));           // [0] first inconsistent block, [1] first failed inflate
        SBX_HIP(hipMemsetAsync(c->d_flag.p + 2, 0, 24, s));         // [2] overflow, [3] ticket, [4] rewalked, [5] max partners, [6] K3 -m overflow
        SBX_HIP(hipMemsetAsync(c->d_flag.p + 8, 0xFF, 8, s));       // [8..9] start of the record behind an open-ended run (64 bits)
        IndexArgs a{};
        a.U = c->d_U.p;
        a.u_alloc = (w.u_bytes + 15) & ~15ull;
        a.out_off = c->d_out_off.p; a.isize = c->d_isize.p; a.run_of = c->d_run_of.p; a.runs = c->d_runs.p;
        a.n_blocks = nb;
        a.inflate_status = c->d_status.p;
        a.entry_in = entries_given ? c->d_entry.p : nullptr;
        a.entry = c->d_entry.p; a.exit_ = c->d_exit.p; a.count = c->d_count.p;
        a.state = c->d_state.p; a.scratch = c->d_lit.p;
        a.refs = refs; a.filt = c->d_filter.p; a.rg = rg; a.tile_pos = T;
        a.simple_filter = c->filter_is_simple ? 1u : 0u;
        a.desc = c->d_desc.p; a.rec_ref = c->d_rec_ref.p; a.name_hash = c->fix_mate ? c->d_name_hash.p : nullptr;
        a.desc_cap = c->desc_cap;
        a.tile_lo = c->d_tile_lo.p; a.tile_hi = c->d_tile_hi.p; a.stats = c->d_stats.p; a.flags = c->d_flag.p;
        a.scan_part = c->d_scan_part.p;
        a.own_ref = c->own_ref; a.own_beg = c->own_beg; a.own_end = c->own_end;
        launch_index_blocks(a, s);
        launch_tile_compact(c->d_tile_lo.p, c->d_tile_hi.p, (uint32_t)nt, deep_thr, c->d_active.p, c->d_slot_of.p, c->d_n_active.p, s,
                            c->d_scan_part.p);
        if (attempt == 0) t2.stop(s);
        R.last_state = 0;
        SBX_HIP(hipMemcpyAsync(R.flags, c->d_flag.p, 16, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipMemcpyAsync(&R.n_active, c->d_n_active.p, 8, hipMemcpyDeviceToHost, s));      // n_active, n_deep
        SBX_HIP(hipMemcpyAsync(R.st, c->d_stats.p, sizeof(IndexStats) * kIndexStatSlots, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipMemcpyAsync(R.tok_bytes, c->d_tok.p, 64 * 8, hipMemcpyDeviceToHost, s));
        if (nb) SBX_HIP(hipMemcpyAsync(&R.last_state, c->d_state.p + (nb - 1), 8, hipMemcpyDeviceToHost, s));
        if (c->index_mode) SBX_HIP(hipMemcpyAsync(&R.straddler, c->d_flag.p + 8, 8, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipStreamSynchronize(s));                            // ---- host synchronisation 1 of 2 ----
        if (R.flags[1] != 0xFFFFFFFFu) {
            uint32_t st = 0;
            SBX_HIP(hipMemcpy(&st, c->d_status.p + R.flags[1], 4, hipMemcpyDeviceToHost));
            throw Error(SBX_EFORMAT, "Error inflating BGZF block starting from offset " +
                                         std::to_string(c->blocks.coffset[w.file_blk[R.flags[1]]]) + ": " + inflate_status_string(st));
        }
        n_records = R.last_state & ((1ull << 62) - 1);
        uint32_t first_bad = R.flags[0];
        bool forced = false;
        if (force && attempt == 0 && !entries_given) { const uint32_t f = (uint32_t)atoi(force); if (f < first_bad && f < nb) { first_bad = f; forced = true; } }
        if (dbg) fprintf(stderr, "[sbx]   index attempt %d: records=%llu first_bad=%u overflow=%u cap=%llu\n", attempt,
                         (unsigned long long)n_records, first_bad, R.flags[2], (unsigned long long)c->desc_cap);
        if (dbg && first_bad != 0xFFFFFFFFu && first_bad < nb) {
            const uint32_t b0 = first_bad >= 2 ? first_bad - 2 : 0, b1 = std::min<uint32_t>(nb, first_bad + 3);
            std::vector<uint64_t> he(b1 - b0), hx(b1 - b0);
            std::vector<uint32_t> hc(b1 - b0);
            SBX_HIP(hipMemcpy(he.data(), c->d_entry.p + b0, (b1 - b0) * 8ull, hipMemcpyDeviceToHost));
            SBX_HIP(hipMemcpy(hx.data(), c->d_exit.p + b0, (b1 - b0) * 8ull, hipMemcpyDeviceToHost));
            SBX_HIP(hipMemcpy(hc.data(), c->d_count.p + b0, (b1 - b0) * 4ull, hipMemcpyDeviceToHost));
            for (uint32_t b = b0; b < b1; ++b)
                fprintf(stderr, "[sbx]     block %u: out_off=%llu isize=%u entry=%lld exit=%lld count=%u\n", b, (unsigned long long)w.out_off[b], w.isize[b],
                        (long long)he[b - b0], (long long)hx[b - b0], hc[b - b0]);
        }
        if (first_bad != 0xFFFFFFFFu && first_bad < nb) {
            // a guessed entry was wrong (or a block holds no record start): follow the chain serially from there and
            // launch again with the entries given; a chain that is still inconsistent then is a corrupt file
            if (entries_given && !forced) throw Error(SBX_EFORMAT, "BAM record chain is broken (truncated or corrupt record)");
            // Wrong guesses are isolated, so they are repaired in parallel first: every block that is not entered where
            // its predecessor was left is walked again from there, round after round until nothing changes.  What is
            // left after a few rounds (a long stretch of blocks without record starts, a corrupt file) -- and the test
            // hook -- goes to the serial repair, which follows the chain from the first inconsistent block on.
            bool settled = false;
            if (!forced) {
                for (int round = 0; round < 8 && !settled; ++round) {
                    SBX_HIP(hipMemsetAsync(c->d_flag.p + 4, 0, 4, s));
                    launch_rewalk_mismatched(a, c->d_flag.p + 4, s);
                    SBX_HIP(hipMemcpyAsync(&R.n_rewalked, c->d_flag.p + 4, 4, hipMemcpyDeviceToHost, s));
                    SBX_HIP(hipStreamSynchronize(s));
                    n_rewalked += R.n_rewalked;
                    settled = R.n_rewalked == 0 && round > 0;
                    if (R.n_rewalked == 0) break;
                }
            }
            if (!settled) {
                SBX_HIP(hipMemsetAsync(c->d_flag.p + 4, 0, 4, s));
                launch_chain_repair(c->d_U.p, c->d_out_off.p, c->d_isize.p, c->d_run_of.p, c->d_runs.p, nb, first_bad, c->d_entry.p, c->d_exit.p,
                                    c->d_count.p, c->d_flag.p + 4, s);
                SBX_HIP(hipMemcpyAsync(&R.n_rewalked, c->d_flag.p + 4, 4, hipMemcpyDeviceToHost, s));
                SBX_HIP(hipStreamSynchronize(s));
                n_rewalked += R.n_rewalked;
            }
            // whatever the repair rounds produced is only a proposal: the pass is repeated with these entries, and the chain
            // check of that pass (k_check_scan, every block against its predecessor) is what accepts or rejects it
            entries_given = true;
            continue;
        }
        if (R.flags[2]) { want_cap = n_records + 1024; continue; }
        if (R.st[0].over_tiles && !c->index_mode) {
            // an admitted alignment reaches beyond the spare tiles of its contig (the reference prints every column a read covers,
            // pileup.d:345-397): lay the tiles out with room for it and repeat the pass -- the chain is known by now
            if (spare_retried) throw Error(SBX_EFORMAT, "internal error: alignments beyond the enlarged spare tiles");
            spare_retried = true;
            const uint64_t want = (uint64_t)c->spare_tiles + R.st[0].over_tiles;
            if (want > 0x7FFFFFFFull / T + 2) throw Error(SBX_EFORMAT, "malformed BAM record (an alignment ends beyond position 2^31)");
            c->spare_tiles = (uint32_t)want;
            upload_static(c, sel, restricted, T, &nt, &refs, &rg);
            c->d_tile_lo.ensure((size_t)nt + 1);
            c->d_tile_hi.ensure((size_t)nt + 1);
            c->d_active.ensure((size_t)nt + 1);
            c->d_slot_of.ensure((size_t)nt + 1);
            entries_given = true;
            attempt = 0;
            continue;
        }
        break;
    }
    c->spare_of_run = c->spare_tiles;
    IndexStats ist{};
    for (uint32_t k = 0; k < kIndexStatSlots; ++k) {
        const IndexStats& x = R.st[k];
        ist.n_records += x.n_records; ist.n_admitted += x.n_admitted; ist.n_bad += x.n_bad; ist.n_unknown_rg += x.n_unknown_rg;
        ist.adm_seq_bytes += x.adm_seq_bytes; ist.adm_qual_bytes += x.adm_qual_bytes;
        ist.max_span = std::max(ist.max_span, x.max_span);
    }
    const uint32_t n_active = R.n_active;
    if (ist.n_records != n_records)
        throw Error(SBX_EFORMAT, "internal error: record chain (" + std::to_string(n_records) + ") and describe pass (" +
                                     std::to_string(ist.n_records) + ") disagree on the number of records");
    if (c->index_mode) {              // the descriptors are the result
        c->primary_records = n_records;
        c->index_straddler = R.straddler;
        c->stats.n_records = ist.n_records;
        c->stats.n_bgzf_blocks = nb;
        c->stats.ms_inflate = t1.ms();
        c->stats.ms_index = t2.ms();
        return;
    }
    if (ist.n_unknown_rg)
        throw Error(SBX_ERG, "error in read: read group is not present in the header (" + std::to_string(ist.n_unknown_rg) + " reads)");
    if (ist.n_bad)
        throw Error(SBX_EFORMAT, "malformed BAM record (" + std::to_string(ist.n_bad) + " records whose lengths are inconsistent with block_size, "
                                 "whose reference id is out of range, or which start beyond the end of their contig)");

    // ---- K3 ----
    const bool want_span = c->min_bq > 0 || (c->fix_mate && c->mode != SBX_MODE_BASE);
    // region / window statistics are sums over {bases counted, depth} per position (reduce.hip): without -m, for a single file and
    // without a tile of 2^16 records or more K3 writes that one word per position instead of seven counters (SBX_COMPACT=0: never)
    static const bool compact_ok = [] { const char* e = getenv("SBX_COMPACT"); return !e || atoi(e) != 0; }();
    const bool compact = compact_ok && c->mode != SBX_MODE_BASE && !c->fix_mate && !c->in_group && R.n_deep == 0;
    c->compact_counters = compact;
    size_t per_tile = compact ? (size_t)T * S : (size_t)T * S * SBX_NCOUNTERS;
    c->d_counters.ensure((size_t)n_active * per_tile + 4);
    if (want_span) c->d_span.ensure((size_t)n_active * T + 4);
    if (c->fix_mate) {
        c->d_mate.ensure((size_t)n_records + 64);
        c->d_n_partners.ensure((size_t)n_records + 64);
        SBX_HIP(hipMemsetAsync(c->d_mate.p, 0xFF, (size_t)n_records * 4, s));
        SBX_HIP(hipMemsetAsync(c->d_n_partners.p, 0, (size_t)n_records * 4, s));
    }
    t3.start(s);
    if (c->fix_mate) {
        launch_find_mates(c->U(), c->d_desc.p, c->d_name_hash.p, c->d_rec_ref.p, n_records, c->d_mate.p, c->d_n_partners.p, s);
        launch_max_u32(c->d_n_partners.p, n_records, c->d_flag.p + 5, s);
        SBX_HIP(hipMemcpyAsync(&R.max_partners, c->d_flag.p + 5, 4, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipStreamSynchronize(s));
        const char* many_msg = "--fix-mate-overlaps: four or more overlapping records with the same name cover one position (or, in region / "
                               "window mode, a record overlaps two or more such records); the reference's result then depends on the hash "
                               "order of unrelated reads (depth.d:343-377) and is not on the device path";
        if (R.max_partners > 1 && c->mode != SBX_MODE_BASE) throw Error(SBX_EUNSUPPORTED, many_msg);
        if (c->mode == SBX_MODE_BASE) {
            const bool multi = R.max_partners > 1;
            if (multi) {
                // groups of more than two same-name records: list up to three partners per record
                c->d_mate_ext.ensure(3 * (size_t)n_records + 64);
                SBX_HIP(hipMemsetAsync(c->d_mate_ext.p, 0xFF, 3 * (size_t)n_records * 4, s));
                SBX_HIP(hipMemsetAsync(c->d_n_partners.p, 0, (size_t)n_records * 4, s));
                launch_find_partners(c->U(), c->d_desc.p, c->d_name_hash.p, c->d_rec_ref.p, n_records, c->d_mate_ext.p, c->d_n_partners.p, s);
            }
            launch_accumulate_mates(c->U(), c->d_desc.p, c->d_mate.p, c->d_tile_lo.p, c->d_tile_hi.p, c->d_active.p, n_active,
                                    c->d_tile_base.p, n_ref, T, S, c->min_bq, multi ? c->d_mate_ext.p : nullptr, c->d_n_partners.p,
                                    c->d_flag.p + 6, c->d_counters.p, want_span ? c->d_span.p : nullptr, s);
            if (multi) {
                SBX_HIP(hipMemcpyAsync(&R.max_partners, c->d_flag.p + 6, 4, hipMemcpyDeviceToHost, s));
                SBX_HIP(hipStreamSynchronize(s));
                if (R.max_partners) throw Error(SBX_EUNSUPPORTED, many_msg);
            }
        } else {
            // region / window: the statistics come from per-column quantities, not from the base counters
            c->d_covm.ensure((size_t)n_active * T * S + 1);
            c->d_addm.ensure((size_t)n_active * T * S + 1);
            if (n_active) SBX_HIP(hipMemsetAsync(c->d_counters.p, 0, (size_t)n_active * per_tile * 4, s));
            launch_mates_columns(c->U(), c->d_desc.p, c->d_mate.p, c->d_tile_lo.p, c->d_tile_hi.p, c->d_active.p, n_active,
                                 c->d_tile_base.p, n_ref, T, S, c->min_bq, c->d_covm.p, c->d_addm.p, c->d_span.p, s);
        }
    } else
        launch_accumulate(c->U(), c->d_desc.p, c->d_tile_lo.p, c->d_tile_hi.p, c->d_active.p, n_active, R.n_deep, deep_thr, c->d_tile_base.p,
                          n_ref, T, S, c->min_bq, c->d_counters.p, want_span ? c->d_span.p : nullptr, s, compact);
    t3.stop(s);
    t_all.stop(s);
    c->h_slot_of.resize((size_t)nt);
    if (nt) SBX_HIP(hipMemcpyAsync(c->h_slot_of.data(), c->d_slot_of.p, (size_t)nt * 4, hipMemcpyDeviceToHost, s));
    SBX_HIP(hipStreamSynchronize(s));                                // ---- host synchronisation 2 of 2 ----

    c->h_tile_base = c->h_tile_base_up;
    c->tile_pos = T;
    c->n_samples_eff = S;
    c->n_tiles = (uint32_t)nt;
    c->n_active = n_active;
    c->span_valid = want_span;
    c->stats.ms_inflate = t1.ms();
    {
        float f = 0;
        SBX_HIP(hipEventElapsedTime(&f, t1.a, t1m.b));
        c->stats.ms_huffman = f;
        c->stats.ms_lz77 = c->stats.ms_inflate - f;
    }
    c->stats.ms_index = t2.ms();
    c->stats.ms_accumulate = t3.ms();
    c->stats.ms_total = t_all.ms();
    c->stats.n_records = ist.n_records;
    c->stats.n_admitted = ist.n_admitted;
    c->stats.n_malformed = ist.n_bad;
    c->stats.n_bgzf_blocks = nb;
    c->stats.n_runs = w.runs.size();
    c->stats.uploaded_bytes = w.comp_bytes;
    {
        uint64_t cb = 0;
        for (auto& r : w.runs) cb += c->blocks.comp_off[r.blk1 - 1] + c->blocks.comp_len[r.blk1 - 1] + 8 - c->blocks.coffset[r.blk0];
        c->stats.compressed_bytes = cb;
    }
    c->stats.uncompressed_bytes = w.u_bytes;
    c->stats.counter_bytes = (uint64_t)n_active * per_tile * 4 + (want_span ? (uint64_t)n_active * T * 4 : 0);
    c->stats.token_bytes = 0;
    for (int k = 0; k < 64; ++k) c->stats.token_bytes += R.tok_bytes[k];
    c->stats.max_alignment_span = ist.max_span;
    c->stats.accumulate_read_bytes = 32ull * ist.n_records + ist.adm_seq_bytes + (c->min_bq > 0 || c->fix_mate ? ist.adm_qual_bytes : 0);
    c->stats.covered_positions = (uint64_t)n_active * T;
    c->stats.launches_inflate = 1;
    c->stats.launches_index = 2 + (n_rewalked ? 2 : 0);
    if (dbg)
        fprintf(stderr, "[sbx] blocks=%u runs=%zu records=%llu rewalked_blocks=%u tiles=%llu active=%u T=%u\n", nb, w.runs.size(),
                (unsigned long long)n_records, n_rewalked, (unsigned long long)nt, n_active, T);
    c->stats.launches_accumulate = 1;
    if (getenv("SBX_TIMING")) fprintf(stderr, "[sbx] hipMalloc/hipFree so far: %.3f s\n", alloc_seconds());
    c->have_run = true;
    ++c->run_serial;
}

// Several BAMs: every file has been through the pipeline on its own; the per-position results are sums over the
// files (the pileup of the merged stream is the union of the reads), so the primary's tile set becomes the union
// of the files' tile sets with the counters added up.  Per-read work that needs a file's records (read counts of
// regions / windows) is done file by file at query time.
static void merge_members(sbx_ctx* c) {
    if (c->members.empty()) return;
    SBX_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const auto files = files_of(c);
    const uint32_t T = c->tile_pos, S = c->n_samples_eff;
    const size_t nt = c->h_slot_of.size();
    for (sbx_ctx* m : files)
        if (m->tile_pos != T || m->n_samples_eff != S || m->h_slot_of.size() != nt) throw Error(SBX_EINVAL, "internal: tile grids differ");
    std::vector<uint32_t> slot(nt, 0xFFFFFFFFu);
    uint32_t n_active = 0;
    for (size_t t = 0; t < nt; ++t) {
        bool on = false;
        for (sbx_ctx* m : files) on |= m->h_slot_of[t] != 0xFFFFFFFFu;
        if (on) slot[t] = n_active++;
    }
    const size_t per_tile = (size_t)T * S * SBX_NCOUNTERS;
    const bool region_m = c->fix_mate && c->mode != SBX_MODE_BASE;
    DevBuf<uint32_t> d_slot(nt + 1), cnt((size_t)n_active * per_tile + 4), spn, covm, addm;
    if (nt) SBX_HIP(hipMemcpyAsync(d_slot.p, slot.data(), nt * 4, hipMemcpyHostToDevice, s));
    SBX_HIP(hipMemsetAsync(cnt.p, 0, cnt.bytes(), s));
    if (c->span_valid) { spn.alloc((size_t)n_active * T + 4); SBX_HIP(hipMemsetAsync(spn.p, 0, spn.bytes(), s)); }
    if (region_m) {
        covm.alloc((size_t)n_active * T * S + 1); addm.alloc((size_t)n_active * T * S + 1);
        SBX_HIP(hipMemsetAsync(covm.p, 0, covm.bytes(), s));
        SBX_HIP(hipMemsetAsync(addm.p, 0, addm.bytes(), s));
    }
    sbx_run_stats sum{};
    for (sbx_ctx* m : files) {
        SBX_HIP(hipStreamSynchronize(m->stream));
        if (m->compact_counters) throw Error(SBX_EINVAL, "internal: a member file ran with compact counters");
        launch_merge_tiles(m->d_counters.p, m->d_active.p, m->n_active, d_slot.p, (uint32_t)per_tile, cnt.p, s);
        if (c->span_valid) launch_merge_tiles(m->d_span.p, m->d_active.p, m->n_active, d_slot.p, T, spn.p, s);
        if (region_m) {
            launch_merge_tiles(m->d_covm.p, m->d_active.p, m->n_active, d_slot.p, T * S, covm.p, s);
            launch_merge_tiles(m->d_addm.p, m->d_active.p, m->n_active, d_slot.p, T * S, addm.p, s);
        }
        const sbx_run_stats& a = m->stats;
        sum.ms_inflate += a.ms_inflate; sum.ms_index += a.ms_index; sum.ms_accumulate += a.ms_accumulate; sum.ms_total += a.ms_total;
        sum.ms_h2d += a.ms_h2d; sum.ms_huffman += a.ms_huffman; sum.ms_lz77 += a.ms_lz77;
        sum.n_records += a.n_records; sum.n_admitted += a.n_admitted; sum.n_bgzf_blocks += a.n_bgzf_blocks;
        sum.compressed_bytes += a.compressed_bytes; sum.uncompressed_bytes += a.uncompressed_bytes;
        sum.accumulate_read_bytes += a.accumulate_read_bytes; sum.token_bytes += a.token_bytes;
        sum.max_alignment_span = std::max(sum.max_alignment_span, a.max_alignment_span);
        sum.launches_inflate += a.launches_inflate; sum.launches_index += a.launches_index; sum.launches_accumulate += a.launches_accumulate;
    }
    SBX_HIP(hipStreamSynchronize(s));
    // the primary now answers for the merged tile set (its own per-file results were folded in above)
    c->primary_records = c->stats.n_records;
    std::swap(c->d_counters, cnt);
    if (c->span_valid) std::swap(c->d_span, spn);
    if (region_m) { std::swap(c->d_covm, covm); std::swap(c->d_addm, addm); }
    std::swap(c->d_slot_of, d_slot);
    c->h_slot_of = slot;
    c->n_active = n_active;
    sum.counter_bytes = (uint64_t)n_active * per_tile * 4;
    sum.covered_positions = (uint64_t)n_active * T;
    c->stats = sum;
}

static void run_files(sbx_ctx* c, const std::vector<sbx_region>& sel, bool restricted) {
    if (c->text_streaming.load() > 0)
        throw Error(SBX_EINVAL, "a run was started while sbx_stream_base_rows is handing out text of this context (only sbx_prefetch_interval may run next to it)");
    const auto files = files_of(c);
    for (sbx_ctx* m : files) run_impl(m, sel, restricted);
    // several files share one tile grid: a file that had to enlarge its spare tiles (run_impl) makes the others follow
    for (bool again = files.size() > 1; again;) {
        again = false;
        uint32_t spare = 1;
        for (sbx_ctx* m : files) spare = std::max(spare, m->spare_tiles);
        for (sbx_ctx* m : files)
            if (m->spare_of_run != spare) { m->spare_tiles = spare; run_impl(m, sel, restricted); again = true; }
    }
    if (c->fix_mate && files.size() > 1) {
        // The reference pairs same-name, same-sample records of a column across files (it merges the files before the pileup,
        // multireader.d:265-268, depth.d:338-377); the files went through the pipeline one by one and were paired within themselves.
        // That is the same thing unless such a pair exists -- which is checked here, and refused rather than printed differently.
        hipStream_t s = c->stream;
        SBX_HIP(hipSetDevice(c->device));
        for (sbx_ctx* m : files) SBX_HIP(hipStreamSynchronize(m->stream));
        SBX_HIP(hipMemsetAsync(c->d_flag.p + 7, 0, 4, s));
        for (size_t x = 0; x < files.size(); ++x)
            for (size_t y = x + 1; y < files.size(); ++y) {
                sbx_ctx *a = files[x], *b = files[y];
                launch_cross_file_mates(a->U(), a->d_desc.p, a->d_name_hash.p, a->d_rec_ref.p, a->stats.n_records, b->U(), b->d_desc.p,
                                        b->d_name_hash.p, b->d_rec_ref.p, b->stats.n_records, (uint32_t)b->stats.max_alignment_span,
                                        c->d_flag.p + 7, s);
            }
        uint32_t hit = 0;
        SBX_HIP(hipMemcpyAsync(&hit, c->d_flag.p + 7, 4, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipStreamSynchronize(s));
        if (hit)
            throw Error(SBX_EUNSUPPORTED, "--fix-mate-overlaps with several BAM files: overlapping records with the same name and sample lie in "
                                          "different files; the reference pairs them across files (depth.d:338-377 on the merged stream), the "
                                          "device path pairs within a file -- merge the files first");
    }
    merge_members(c);
    ++c->run_serial;
}

int sbx_run(sbx_ctx* c) {
    return guarded(c, [&] {
        if (!c) throw Error(SBX_EINVAL, "null context");
        run_files(c, c->regions, !c->regions.empty());
    });
}

// ---- streaming over contigs -----------------------------------------------------------------------
// BGZF block range [b0, b1) holding every read of contig r (from the BAI; empty contigs: b0 == b1)
static void contig_blocks(sbx_ctx* c, uint32_t r, uint32_t* b0, uint32_t* b1) {
    *b0 = *b1 = 0;
    if (!c->merged_to_own.empty()) {       // (the index speaks the file's own reference ids)
        if (r >= c->merged_to_own.size() || c->merged_to_own[r] < 0) return;
        r = (uint32_t)c->merged_to_own[r];
    }
    if (r >= c->bai.refs.size()) return;
    std::vector<sbx_region> whole{{r, 0u, 0x7FFFFFFFu}};
    uint64_t vbeg = ~0ull, vend = 0;
    for (auto& ch : group_chunks(c->bai, whole)) { vbeg = std::min(vbeg, ch.beg); vend = std::max(vend, ch.end); }
    if (vbeg >= vend) return;
    const auto& co = c->blocks.coffset;
    const uint32_t nb = (uint32_t)c->blocks.size();
    size_t i0 = (size_t)(std::lower_bound(co.begin(), co.end(), vbeg >> 16) - co.begin());
    size_t i1 = (size_t)(std::lower_bound(co.begin(), co.end(), vend >> 16) - co.begin());
    if (i0 >= nb) return;
    if (i1 < nb && (vend & 0xFFFF)) ++i1;
    *b0 = (uint32_t)i0;
    *b1 = (uint32_t)std::min<size_t>(std::max(i1, i0 + 1), nb);
}

// estimated device bytes of a run over BGZF blocks [b0, b1) covering `positions` reference positions:
// compressed payload + inflated stream + literal and entry token streams (~2.4x) + descriptors + counter tiles
static uint64_t footprint(sbx_ctx* c, uint32_t b0, uint32_t b1, uint64_t positions) {
    if (b1 <= b0) return 0;
    const uint64_t u = c->blocks.out_off[b1] - c->blocks.out_off[b0];
    const uint64_t comp = c->preloaded ? 0 : c->blocks.coffset[b1 - 1] - c->blocks.coffset[b0] + 65536;
    const uint32_t S = c->combined ? 1u : (uint32_t)std::max<size_t>(1, c->hdr.sample_names.size());
    return comp + u + (u + 48ull * (b1 - b0)) + 4 * (u / 3 + u / 255 + 12ull * (b1 - b0)) + u / 4 + positions * (28ull * S + 4);
}

int sbx_plan_batches(sbx_ctx* c, uint64_t budget_bytes, sbx_batch* out, size_t cap, size_t* n_out) {
    return guarded(c, [&] {
        if (!c || !n_out) throw Error(SBX_EINVAL, "null argument");
        if (!c->has_index) throw Error(SBX_ENOINDEX, "All files must be indexed");
        SBX_HIP(hipSetDevice(c->device));
        if (budget_bytes == 0) {
            size_t free_b = 0, total_b = 0;
            SBX_HIP(hipMemGetInfo(&free_b, &total_b));
            // what this context already holds (the compressed file, buffers of an earlier run) is reused
            budget_bytes = (uint64_t)((double)free_b * 0.7) + c->d_U.bytes() + c->d_lit.bytes() + c->d_ent.bytes() + c->d_counters.bytes() +
                           c->d_desc.bytes() + c->d_rec_ref.bytes() + (c->preloaded ? 0 : c->d_comp.bytes());
        }
        const uint32_t n_ref = (uint32_t)c->hdr.refs.size();
        std::vector<sbx_batch> plan;
        uint32_t first = 0, lo = 0, hi = 0;       // current batch: contigs [first, r), blocks [lo, hi)
        uint64_t pos = 0;
        for (uint32_t r = 0; r < n_ref; ++r) {
            uint32_t b0, b1;
            contig_blocks(c, r, &b0, &b1);       // (several BAMs: sized by the first file times the number of files)
            const uint64_t len = (uint64_t)std::max(0, c->hdr.refs[r].length);
            uint32_t nlo = lo, nhi = hi;
            if (b1 > b0) { nlo = hi > lo ? std::min(lo, b0) : b0; nhi = hi > lo ? std::max(hi, b1) : b1; }
            if (r > first && footprint(c, nlo, nhi, pos + len) * (1 + c->members.size()) > budget_bytes) {
                plan.push_back({first, r - first, footprint(c, lo, hi, pos)});
                first = r; pos = 0;
                nlo = b0; nhi = b1;
            }
            lo = nlo; hi = nhi; pos += len;
        }
        if (n_ref > first) plan.push_back({first, n_ref - first, footprint(c, lo, hi, pos)});
        *n_out = plan.size();
        if (out) for (size_t i = 0; i < plan.size() && i < cap; ++i) out[i] = plan[i];
        if (out && plan.size() > cap) throw Error(SBX_ENOMEM, "batch array too small");
    });
}

int sbx_run_batch(sbx_ctx* c, uint32_t first_ref, uint32_t n_refs) {
    return guarded(c, [&] {
        if (!c) throw Error(SBX_EINVAL, "null context");
        if ((uint64_t)first_ref + n_refs > c->hdr.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
        std::vector<sbx_region> sel;
        if (c->regions.empty()) {
            for (uint32_t r = first_ref; r < first_ref + n_refs; ++r) sel.push_back({r, 0u, 0x7FFFFFFFu});
        } else {
            for (auto& g : c->regions)
                if (g.ref_id >= first_ref && g.ref_id < first_ref + n_refs) sel.push_back(g);
        }
        run_files(c, sel, true);
    });
}

int sbx_run_interval(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end) {
    return guarded(c, [&] {
        if (!c) throw Error(SBX_EINVAL, "null context");
        if (ref_id >= c->hdr.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
        if (!(beg < end)) throw Error(SBX_EINVAL, "empty interval");
        std::vector<sbx_region> sel;
        if (c->regions.empty()) sel.push_back({ref_id, beg, end});
        else
            for (auto& g : c->regions)
                if (g.ref_id == ref_id && g.start < end && g.end > beg) sel.push_back({ref_id, std::max(g.start, beg), std::min(g.end, end)});
        run_files(c, sel, true);
    });
}

int sbx_prefetch_interval(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end) {
    return guarded(c, [&] {
        if (!c) throw Error(SBX_EINVAL, "null context");
        if (ref_id >= c->hdr.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
        if (!(beg < end)) throw Error(SBX_EINVAL, "empty interval");
        if (!c->has_index) throw Error(SBX_ENOINDEX, "All files must be indexed");
        {   // slices of similar size follow: no buffer should have to grow twice
            int cur = devbuf_slack_pct().load(std::memory_order_relaxed);
            while (cur < 8 && !devbuf_slack_pct().compare_exchange_weak(cur, 8, std::memory_order_relaxed)) {}
        }
        std::vector<sbx_region> sel;
        if (c->regions.empty()) sel.push_back({ref_id, beg, end});
        else
            for (auto& g : c->regions)
                if (g.ref_id == ref_id && g.start < end && g.end > beg) sel.push_back({ref_id, std::max(g.start, beg), std::min(g.end, end)});
        for (sbx_ctx* m : files_of(c)) {
            SBX_HIP(hipSetDevice(m->device));
            make_resident(m, build_runs(m, sel, true));
        }
    });
}

int sbx_run_interval_owned(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end) {
    return guarded(c, [&] {
        if (!c) throw Error(SBX_EINVAL, "null context");
        if (ref_id >= c->hdr.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
        if (!(beg < end)) throw Error(SBX_EINVAL, "empty interval");
        if (c->fix_mate) throw Error(SBX_EUNSUPPORTED, "sbx_run_interval_owned: --fix-mate-overlaps needs both mates of a pair in one run");
        std::vector<sbx_region> sel;
        if (c->regions.empty()) sel.push_back({ref_id, beg, end});
        else
            for (auto& g : c->regions)
                if (g.ref_id == ref_id && g.start < end && g.end > beg) sel.push_back({ref_id, std::max(g.start, beg), std::min(g.end, end)});
        struct Own {      // the restriction lasts for this run only
            std::vector<sbx_ctx*> f;
            ~Own() { for (sbx_ctx* m : f) m->own_ref = -1; }
        } own{files_of(c)};
        for (sbx_ctx* m : own.f) { m->own_ref = (int32_t)ref_id; m->own_beg = beg; m->own_end = end; }
        run_files(c, sel, true);
    });
}

// ---- the write side: BGZF compression, BAM files, BAI ---------------------------------------------------------------
extern "C++" {
namespace {
const uint8_t kEofBlock[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// compresses in[0, n) piece by piece on the device; sink(data, len) receives consecutive pieces of the BGZF stream
template <class Sink>
void bgzf_compress_stream(const uint8_t* in, size_t n, int level, Sink&& sink) {
    hipStream_t s = nullptr;
    SBX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Guard { hipStream_t s; ~Guard() { (void)hipStreamDestroy(s); } } guard{s};
    const size_t piece_blocks = 32768;
    const size_t n_blocks_total = (n + kBgzfPayload - 1) / kBgzfPayload;
    const uint32_t cap_blocks = (uint32_t)std::min<size_t>(piece_blocks, std::max<size_t>(1, n_blocks_total));
    DevBuf<uint8_t> d_in((size_t)cap_blocks * kBgzfPayload + 64), d_slots((size_t)cap_blocks * kBgzfSlot), d_out((size_t)cap_blocks * kBgzfSlot);
    DevBuf<uint16_t> d_tab(deflate_table_entries(cap_blocks));
    DevBuf<uint8_t> d_work(deflate_work_bytes(cap_blocks));
    DevBuf<uint32_t> d_len(cap_blocks + 1);
    DevBuf<uint64_t> d_off((size_t)cap_blocks + 2);
    std::vector<uint8_t> host;
    const bool timing = getenv("SBX_TIMING") != nullptr;
    EventTimer t_def, t_pack;
    double ms_h2d = 0, ms_def = 0, ms_pack = 0, ms_d2h = 0;
    uint64_t out_total = 0;
    for (size_t done = 0; done < n;) {
        const size_t bytes = std::min<size_t>(n - done, (size_t)cap_blocks * kBgzfPayload);
        const uint32_t nb = (uint32_t)((bytes + kBgzfPayload - 1) / kBgzfPayload);
        const double w0 = wall_now();
        SBX_HIP(hipMemcpyAsync(d_in.p, in + done, bytes, hipMemcpyHostToDevice, s));
        if (timing) SBX_HIP(hipStreamSynchronize(s));
        const double w1 = wall_now();
        t_def.start(s);
        launch_bgzf_deflate(d_in.p, bytes, nb, level, d_slots.p, d_tab.p, d_work.p, d_len.p, s);
        t_def.stop(s);
        t_pack.start(s);
        launch_count_scan(d_len.p, nb, d_off.p, nullptr, 0, s);
        launch_pack_blocks(d_slots.p, d_len.p, d_off.p, nb, d_out.p, s);
        t_pack.stop(s);
        uint64_t total = 0;
        SBX_HIP(hipMemcpyAsync(&total, d_off.p + nb, 8, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipStreamSynchronize(s));
        const double w2 = wall_now();
        host.resize((size_t)total);
        SBX_HIP(hipMemcpy(host.data(), d_out.p, (size_t)total, hipMemcpyDeviceToHost));
        if (timing) { ms_h2d += (w1 - w0) * 1e3; ms_def += t_def.ms(); ms_pack += t_pack.ms(); ms_d2h += (wall_now() - w2) * 1e3; out_total += total; }
        sink(host.data(), (size_t)total);
        done += bytes;
    }
    if (timing)
        fprintf(stderr, "[sbx] bgzf_compress: %zu bytes -> %llu in %zu blocks: host -> device %.1f ms (pageable), deflate kernel %.1f ms (%.1f GB/s of input), "
                        "scan + pack %.1f ms, device -> host %.1f ms\n", n, (unsigned long long)out_total, n_blocks_total, ms_h2d, ms_def,
                ms_def > 0 ? (double)n / ms_def / 1e6 : 0.0, ms_pack, ms_d2h);
}
}  // namespace
}  // extern "C++"

int sbx_bgzf_compress(const uint8_t* in, size_t n, int level, int with_eof, int device, uint8_t* out, size_t cap, size_t* out_len,
                      char* err, size_t errlen) {
    try {
        if ((!in && n) || !out_len) throw Error(SBX_EINVAL, "null argument");
        if (level < -1 || level > 9) throw Error(SBX_EINVAL, "compression level must be -1 (default) or 0 .. 9");
        require_device(device);
        size_t pos = 0;
        bgzf_compress_stream(in, n, level, [&](const uint8_t* p, size_t k) {
            if (pos + k > cap || !out) { pos += k; return; }
            memcpy(out + pos, p, k);
            pos += k;
        });
        if (with_eof) {
            if (out && pos + 28 <= cap) memcpy(out + pos, kEofBlock, 28);
            pos += 28;
        }
        *out_len = pos;
        if (pos > cap || !out) throw Error(SBX_ENOMEM, "output buffer too small for the BGZF stream");
        return SBX_OK;
    } catch (const Error& e) {
        set_err(err, errlen, e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_err(err, errlen, e.what());
        return SBX_EINVAL;
    }
}

int sbx_build_index(const char* bam_path, const char* bai_path, int device, char* err, size_t errlen) {
    sbx_ctx* c = nullptr;
    try {
        if (!bam_path || !bai_path) throw Error(SBX_EINVAL, "null argument");
        const char* one[1] = {bam_path};
        char e2[512] = {0};
        c = sbx_open(one, 1, device, e2, sizeof e2);
        if (!c) throw Error(SBX_EIO, e2);
        c->index_mode = true;
        memset(&c->filter, 0, sizeof c->filter);         // no filter: every record is described
        c->mode = SBX_MODE_BASE;
        c->fix_mate = false;
        // IndexBuilder is one pass over a stream of records (bai/indexing.d:262-316), and so is this: the file goes through the
        // device in batches of whole BGZF blocks -- inflate, record chain, descriptors --, a batch ends in front of the record that
        // straddles its last block boundary (ChainRun::open_end) and the next batch starts with that record.  The batch size follows
        // the free device memory (a batch holds its compressed bytes, its inflated bytes, the token streams and the descriptors: about
        // five times its inflated size); SBX_INDEX_BATCH_BYTES overrides it (tests).
        const BlockTable& bt = c->blocks;
        const size_t nbk = bt.size();
        const uint64_t total = bt.out_off.back(), first = c->hdr.first_record_off;
        const uint64_t file_end_coff = nbk ? bt.comp_off[nbk - 1] + bt.comp_len[nbk - 1] + 8 : 0;
        uint64_t batch_u = 0;
        if (const char* e = getenv("SBX_INDEX_BATCH_BYTES")) batch_u = strtoull(e, nullptr, 10);
        if (!batch_u) {
            size_t free_b = 0, total_b = 0;
            SBX_HIP(hipMemGetInfo(&free_b, &total_b));
            batch_u = std::max<uint64_t>(64ull << 20, (uint64_t)((double)free_b * 0.7 / 5.0));
        }
        hipStream_t s = c->stream;
        const int n_ref = (int)c->hdr.refs.size();
        // What consumes the records of a batch: the device (bai_parallel.hpp -- one lane per record; only the run heads, about one
        // record in fifty, and a few per-reference arrays come back) or, for input that formulation calls irregular (unsorted reads:
        // the reference's error is worded by the serial builder; reads far beyond the end of their reference) and with
        // SBX_BAI_HOST=1, IndexBuilder's loop restated on the host (bai_writer.hpp) over descriptors copied back record by record.
        std::vector<uint8_t> bytes;
        uint32_t n_batches = 0;
        auto pass = [&](bool on_device) -> bool {
            // serial consumer
            VoffCursor vc(bt.coffset.data(), bt.out_off.data(), nbk, file_end_coff);
            BaiBuilder bb(n_ref);
            BaiRecord held;                 // the last record of the batch before: it ends where the next batch starts
            bool have_held = false;
            DevBuf<uint16_t> d_bins;
            std::vector<RecDesc> desc;
            std::vector<int32_t> ref;
            std::vector<uint16_t> bins;
            // device consumer
            BaiHostResults R;
            DevBuf<uint64_t> d_coff, d_ustart, d_lin, d_meta;      // d_meta: meta_end | n_mapped | n_unmapped, n_ref + 1 each
            DevBuf<uint32_t> d_lin_off, d_lin_len;
            DevBuf<unsigned long long> d_scalars;
            DevBuf<BaiRun> d_runs;
            DevBuf<BaiCarry> d_carry(1);
            BaiCarry carry{-1, 0, 0, 0, 0};
            uint64_t rec_base = 0;
            if (on_device) {
                R.lin_off.assign((size_t)n_ref + 1, 0);
                for (int r = 0; r < n_ref; ++r) R.lin_off[(size_t)r + 1] = R.lin_off[(size_t)r] + bai_windows_for(c->hdr.refs[(size_t)r].length);
                d_coff.alloc(nbk + 1); d_ustart.alloc(nbk + 2);
                d_lin.alloc((size_t)R.lin_off[(size_t)n_ref] + 1); d_lin_off.alloc((size_t)n_ref + 2); d_lin_len.alloc((size_t)n_ref + 1);
                d_meta.alloc(3 * ((size_t)n_ref + 1)); d_scalars.alloc(kBaiScalars);
                if (nbk) SBX_HIP(hipMemcpyAsync(d_coff.p, bt.coffset.data(), nbk * 8, hipMemcpyHostToDevice, s));
                SBX_HIP(hipMemcpyAsync(d_ustart.p, bt.out_off.data(), (nbk + 1) * 8, hipMemcpyHostToDevice, s));
                SBX_HIP(hipMemcpyAsync(d_lin_off.p, R.lin_off.data(), ((size_t)n_ref + 1) * 4, hipMemcpyHostToDevice, s));
                SBX_HIP(hipMemsetAsync(d_lin.p, 0xFF, d_lin.bytes(), s));
                SBX_HIP(hipMemsetAsync(d_lin_len.p, 0, d_lin_len.bytes(), s));
                SBX_HIP(hipMemsetAsync(d_meta.p, 0, d_meta.bytes(), s));
                SBX_HIP(hipMemsetAsync(d_scalars.p, 0, d_scalars.bytes(), s));
                SBX_HIP(hipMemsetAsync(d_scalars.p + kBaiFirstVo, 0xFF, 8, s));
            }
            uint64_t bu = batch_u;
            n_batches = 0;
            for (uint64_t cur = first; cur < total;) {
                const uint32_t b0 = (uint32_t)(std::upper_bound(bt.out_off.begin(), bt.out_off.end(), cur) - bt.out_off.begin()) - 1;
                uint32_t b1 = (uint32_t)(std::lower_bound(bt.out_off.begin() + b0, bt.out_off.end(), bt.out_off[b0] + bu) - bt.out_off.begin());
                b1 = std::min<uint32_t>(std::max(b1, b0 + 1), (uint32_t)nbk);
                if (bt.out_off[b1] >= total) b1 = (uint32_t)nbk;          // (whatever follows holds no bytes: EOF blocks)
                const bool last = b1 == nbk;
                const std::vector<FileRun> runs{FileRun{b0, b1, cur, last ? total : bt.out_off[b1], !last}};
                run_impl(c, {}, false, &runs);
                const uint64_t nrec = c->primary_records;
                const uint64_t base = bt.out_off[b0];            // work-list offsets count from the batch's first block
                const uint64_t next = last ? total : c->index_straddler != kOffUnknown ? base + c->index_straddler : bt.out_off[b1];
                if (!last && next == cur) {                      // not one whole record in the batch: a longer batch
                    bu *= 2;
                    continue;
                }
                ++n_batches;
                if (on_device) {
                    const uint64_t cap = nrec / 4 + 4096;
                    d_runs.ensure((size_t)cap);
                    SBX_HIP(hipMemsetAsync(d_scalars.p + kBaiNumRuns, 0, 8, s));
                    BaiArgs a{};
                    a.U = c->U(); a.desc = c->d_desc.p; a.rec_ref = c->d_rec_ref.p; a.n = nrec; a.rec_base = rec_base;
                    a.u_base = base; a.u_next = next;
                    a.coff = d_coff.p; a.ustart = d_ustart.p; a.n_blocks = (uint32_t)nbk; a.file_end = file_end_coff;
                    a.carry = carry; a.n_ref = n_ref;
                    a.lin = d_lin.p; a.lin_off = d_lin_off.p; a.lin_len = d_lin_len.p;
                    a.meta_end = d_meta.p; a.n_mapped = d_meta.p + (n_ref + 1); a.n_unmapped = d_meta.p + 2 * ((size_t)n_ref + 1);
                    a.scalars = d_scalars.p; a.runs = d_runs.p; a.runs_cap = cap;
                    launch_bai_records(a, d_carry.p, s);
                    unsigned long long sc[kBaiScalars];
                    SBX_HIP(hipMemcpyAsync(sc, d_scalars.p, sizeof sc, hipMemcpyDeviceToHost, s));
                    SBX_HIP(hipMemcpyAsync(&carry, d_carry.p, sizeof carry, hipMemcpyDeviceToHost, s));
                    SBX_HIP(hipStreamSynchronize(s));
                    if (sc[kBaiIrregular] || sc[kBaiNumRuns] > cap) return false;
                    const size_t at = R.runs.size(), nr = (size_t)sc[kBaiNumRuns];
                    R.runs.resize(at + nr);
                    if (nr) SBX_HIP(hipMemcpy(R.runs.data() + at, d_runs.p, nr * sizeof(BaiRun), hipMemcpyDeviceToHost));
                    rec_base += nrec;
                } else {
                    d_bins.ensure((size_t)nrec + 1);
                    launch_gather_bins(c->U(), c->d_desc.p, nrec, d_bins.p, s);
                    desc.resize((size_t)nrec); ref.resize((size_t)nrec); bins.resize((size_t)nrec);
                    SBX_HIP(hipStreamSynchronize(s));
                    if (nrec) {
                        SBX_HIP(hipMemcpy(desc.data(), c->d_desc.p, (size_t)nrec * sizeof(RecDesc), hipMemcpyDeviceToHost));
                        SBX_HIP(hipMemcpy(ref.data(), c->d_rec_ref.p, (size_t)nrec * 4, hipMemcpyDeviceToHost));
                        SBX_HIP(hipMemcpy(bins.data(), d_bins.p, (size_t)nrec * 2, hipMemcpyDeviceToHost));
                    }
                    for (uint64_t i = 0; i < nrec; ++i) {
                        const uint64_t at = base + desc[(size_t)i].rec_off;
                        if (have_held) { held.end_vo = vc.behind(at); bb.put(held); }
                        held.ref_id = ref[(size_t)i];
                        held.position = desc[(size_t)i].pos;
                        held.end_position = desc[(size_t)i].end;
                        held.bin = bins[(size_t)i];
                        held.is_unmapped = (desc[(size_t)i].flag & 0x4) != 0;
                        held.start_vo = vc.of_byte(at);
                        have_held = true;
                    }
                }
                cur = next;
            }
            if (on_device) {
                const size_t m = (size_t)n_ref + 1;
                R.lin.resize(d_lin.n); R.lin_len.resize(m); R.meta_end.resize(m); R.n_mapped.resize(m); R.n_unmapped.resize(m);
                unsigned long long sc[kBaiScalars];
                SBX_HIP(hipMemcpy(R.lin.data(), d_lin.p, d_lin.n * 8, hipMemcpyDeviceToHost));
                SBX_HIP(hipMemcpy(R.lin_len.data(), d_lin_len.p, m * 4, hipMemcpyDeviceToHost));
                SBX_HIP(hipMemcpy(R.meta_end.data(), d_meta.p, m * 8, hipMemcpyDeviceToHost));
                SBX_HIP(hipMemcpy(R.n_mapped.data(), d_meta.p + m, m * 8, hipMemcpyDeviceToHost));
                SBX_HIP(hipMemcpy(R.n_unmapped.data(), d_meta.p + 2 * m, m * 8, hipMemcpyDeviceToHost));
                SBX_HIP(hipMemcpy(sc, d_scalars.p, sizeof sc, hipMemcpyDeviceToHost));
                for (int k = 0; k < (int)kBaiScalars; ++k) R.scalars[k] = sc[k];
                R.last = carry;
                bytes = bai_assemble(n_ref, R);
            } else {
                if (have_held) { held.end_vo = vc.behind(total); bb.put(held); }
                bytes = bb.finish();
            }
            return true;
        };
        const bool host_only = getenv("SBX_BAI_HOST") != nullptr;
        bool on_device = !host_only;
        if (on_device && !pass(true)) on_device = false;
        if (!on_device) pass(false);
        if (getenv("SBX_TIMING"))
            fprintf(stderr, "[sbx] build_index: %u batch(es) of <= %llu inflated bytes, records consumed %s\n", n_batches, (unsigned long long)batch_u,
                    on_device ? "on the device" : "by the serial builder on the host");
        FILE* f = fopen(bai_path, "wb");
        if (!f) throw Error(SBX_EIO, std::string("cannot write ") + bai_path);
        const bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
        if (fclose(f) != 0 || !ok) throw Error(SBX_EIO, std::string("error writing ") + bai_path);
        sbx_close(c);
        return SBX_OK;
    } catch (const Error& e) {
        set_err(err, errlen, e.what());
        if (c) sbx_close(c);
        return e.code;
    } catch (const std::exception& e) {
        set_err(err, errlen, e.what());
        if (c) sbx_close(c);
        return SBX_EINVAL;
    }
}

int sbx_write_bam(const char* path, const uint8_t* stream, size_t n, int level, int with_index, int device, char* err, size_t errlen) {
    try {
        if (!path || (!stream && n)) throw Error(SBX_EINVAL, "null argument");
        if (level < -1 || level > 9) throw Error(SBX_EINVAL, "compression level must be -1 (default) or 0 .. 9");
        require_device(device);
        FILE* f = fopen(path, "wb");
        if (!f) throw Error(SBX_EIO, std::string("cannot write ") + path);
        bool ok = true;
        try {
            bgzf_compress_stream(stream, n, level, [&](const uint8_t* p, size_t k) { ok = ok && fwrite(p, 1, k, f) == k; });
        } catch (...) { fclose(f); throw; }
        ok = ok && fwrite(kEofBlock, 1, 28, f) == 28;
        if (fclose(f) != 0 || !ok) throw Error(SBX_EIO, std::string("error writing ") + path);
    } catch (const Error& e) {
        set_err(err, errlen, e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_err(err, errlen, e.what());
        return SBX_EINVAL;
    }
    if (with_index) return sbx_build_index(path, (std::string(path) + ".bai").c_str(), device, err, errlen);
    return SBX_OK;
}

int sbx_last_run_stats(sbx_ctx* c, sbx_run_stats* out) {
    if (!c || !out) return SBX_EINVAL;
    *out = c->stats;
    return SBX_OK;
}

int sbx_depth_base_tile(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end, uint32_t* counters, uint8_t* covered) {
    return guarded(c, [&] {
        if (!c || (!counters && !covered)) throw Error(SBX_EINVAL, "null argument");
        if (!c->have_run) throw Error(SBX_EINVAL, "sbx_run() has not been called");
        if (ref_id >= c->hdr.refs.size() || beg > end) throw Error(SBX_EINVAL, "bad interval");
        if (counters && c->compact_counters)
            throw Error(SBX_EINVAL, "per-position base counters are kept by `base` runs only: a region / window run keeps the bases counted "
                                    "and the depth of a position (pass counters = NULL for `covered` alone)");
        SBX_HIP(hipSetDevice(c->device));
        const uint32_t T = c->tile_pos, S = c->n_samples_eff;
        const size_t row = c->compact_counters ? (size_t)S : (size_t)S * SBX_NCOUNTERS;
        const uint32_t t_first = c->h_tile_base[ref_id], t_end = c->h_tile_base[ref_id + 1];
        std::vector<uint32_t> own;          // counters == NULL: `covered` alone
        if (!counters) { own.assign((size_t)(end - beg) * row + 1, 0u); counters = own.data(); }
        else memset(counters, 0, (size_t)(end - beg) * row * 4);
        if (covered) memset(covered, 0, (size_t)(end - beg));
        std::vector<uint32_t> spn;
        for (uint64_t p = beg; p < end;) {
            uint32_t t = t_first + (uint32_t)(p / T);
            uint64_t tile_start = (uint64_t)(t - t_first) * T;
            if (t >= t_end || c->h_slot_of[t] == 0xFFFFFFFFu) { p = std::min<uint64_t>(end, tile_start + T); continue; }
            // a run of consecutive active tiles occupies consecutive slots: one copy for the whole run
            uint32_t t2 = t + 1;
            while (t2 < t_end && (uint64_t)(t2 - t_first) * T < end && c->h_slot_of[t2] == c->h_slot_of[t] + (t2 - t)) ++t2;
            uint64_t stop = std::min<uint64_t>(end, (uint64_t)(t2 - t_first) * T);
            size_t slot = c->h_slot_of[t];
            uint32_t* dst = counters + (size_t)(p - beg) * row;
            SBX_HIP(hipMemcpy(dst, c->d_counters.p + slot * T * row + (size_t)(p - tile_start) * row, (size_t)(stop - p) * row * 4,
                              hipMemcpyDeviceToHost));
            if (covered) {
                if (c->span_valid) {
                    spn.resize((size_t)(stop - p));
                    SBX_HIP(hipMemcpy(spn.data(), c->d_span.p + slot * T + (size_t)(p - tile_start), (size_t)(stop - p) * 4,
                                      hipMemcpyDeviceToHost));
                    for (uint64_t q = p; q < stop; ++q) covered[q - beg] = spn[(size_t)(q - p)] ? 1 : 0;
                } else {
                    for (uint64_t q = p; q < stop; ++q) {
                        const uint32_t* r = counters + (size_t)(q - beg) * row;
                        uint32_t any = 0;
                        for (size_t k = 0; k < row; ++k) any |= r[k];
                        covered[q - beg] = any ? 1 : 0;
                    }
                }
            }
            p = stop;
        }
    });
}

// Shared implementation of region / window statistics over an explicit list of ranges.
// ranges[i] gets id i; stats/cov are [i][S] / [i][S][n_thr]; seen[i] (optional).
// min_start (optional, per range): != 0 -> only reads starting at or after it are counted for that range, and its
// n_bases is the sum over those reads instead of the sum over the position counters.
static void range_stats(sbx_ctx* c, const std::vector<sbx_region>& ranges, bool windows, uint32_t window,
                        const std::vector<uint64_t>& win_base, const std::vector<uint64_t>& n_win, sbx_region_stats* stats,
                        uint32_t* cov_counts, uint8_t* seen, const uint32_t* min_start = nullptr) {
    if (!c->have_run) throw Error(SBX_EINVAL, "sbx_run() has not been called");
    SBX_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint32_t S = c->n_samples_eff, T = c->tile_pos;
    const uint32_t n_thr = (uint32_t)c->thresholds.size();
    if (n_thr > (uint32_t)kMaxThresholds) throw Error(SBX_EUNSUPPORTED, "more than 16 coverage thresholds");
    const size_t n = ranges.size();
    if (n > 0x7FFFFFF0ull) throw Error(SBX_EUNSUPPORTED, "too many regions / windows");
    // chunk list for the position reductions (cached: the same ranges as in the previous call need no new list)
    if (n > 0x3FFFFFF0ull) throw Error(SBX_EUNSUPPORTED, "too many regions / windows");
    sbx_ctx::RangeCache& rc = c->rc;
    const bool same = rc.valid && rc.ranges.size() == n && (n == 0 || memcmp(rc.ranges.data(), ranges.data(), n * sizeof(sbx_region)) == 0) &&
                      rc.has_min_start == (min_start != nullptr) &&
                      (!min_start || n == 0 || memcmp(rc.min_start.data(), min_start, n * 4) == 0);
    if (!same) {
        rc.valid = false;
        rc.sorted_valid = false;
        std::vector<RangeChunk> chunks;
        const uint32_t CH = 16384;
        for (size_t i = 0; i < n; ++i)
            for (uint64_t p = ranges[i].start; p < ranges[i].end; p += CH)
                chunks.push_back({ranges[i].ref_id, (uint32_t)p, (uint32_t)std::min<uint64_t>(ranges[i].end, p + CH),
                                  (uint32_t)i | ((min_start && min_start[i]) ? 0x40000000u : 0u)});
        rc.d_chunks.ensure(chunks.size() + 1);
        if (!chunks.empty()) SBX_HIP(hipMemcpy(rc.d_chunks.p, chunks.data(), chunks.size() * sizeof(RangeChunk), hipMemcpyHostToDevice));
        rc.n_chunks = chunks.size();
        rc.ranges = ranges;
        rc.has_min_start = min_start != nullptr;
        rc.min_start.assign(min_start ? min_start : nullptr, min_start ? min_start + n : nullptr);
        rc.valid = true;
    }
    const size_t n_chunks = rc.n_chunks;
    DevBuf<RangeChunk>& d_chunks = rc.d_chunks;
    DevBuf<uint32_t>&d_nb = rc.d_nb, &d_nr = rc.d_nr, &d_cov = rc.d_cov, &d_seen = rc.d_seen, &d_thr = rc.d_thr;
    d_nb.ensure(n * S + 1); d_nr.ensure(n * S + 1); d_cov.ensure(n * S * std::max<uint32_t>(1, n_thr) + 1); d_seen.ensure(n + 1); d_thr.ensure(n_thr + 1);
    if (n_thr) SBX_HIP(hipMemcpyAsync(d_thr.p, c->thresholds.data(), n_thr * 4, hipMemcpyHostToDevice, s));
    SBX_HIP(hipMemsetAsync(d_nb.p, 0, (n * S + 1) * 4, s));
    SBX_HIP(hipMemsetAsync(d_nr.p, 0, (n * S + 1) * 4, s));
    SBX_HIP(hipMemsetAsync(d_cov.p, 0, (n * S * std::max<uint32_t>(1, n_thr) + 1) * 4, s));
    SBX_HIP(hipMemsetAsync(d_seen.p, 0, (n + 1) * 4, s));
    EventTimer t;
    t.start(s);
    // per-read work goes file by file (records of every BAM stay resident after the run)
    const auto files = files_of(c);
    auto records_of = [&](sbx_ctx* f) -> uint64_t { return (f == c && !c->members.empty()) ? c->primary_records : f->stats.n_records; };
    if (c->fix_mate && c->mode != SBX_MODE_BASE) {
        // ---- --fix-mate-overlaps: closed form of depth.d:717-845 (reduce.hip) -------------------------------------
        const size_t n_ref = c->hdr.refs.size();
        DevBuf<uint32_t> d_firstcol(n + 1);
        SBX_HIP(hipMemsetAsync(d_firstcol.p, 0xFF, d_firstcol.bytes(), s));
        launch_range_first(d_chunks.p, (uint32_t)n_chunks, c->d_span.p, c->d_slot_of.p, c->d_tile_base.p, T, d_firstcol.p, s);
        launch_range_reduce_m(d_chunks.p, (uint32_t)n_chunks, c->d_covm.p, c->d_addm.p, c->d_span.p, c->d_slot_of.p,
                              c->d_tile_base.p, T, S, d_thr.p, n_thr, d_nb.p, d_cov.p, d_seen.p, s);
        // (ref, start)-sorted view with prefix maxima of the ends, and the union of the ranges (where pairs get "fixed")
        std::vector<uint32_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            if (ranges[x].ref_id != ranges[y].ref_id) return ranges[x].ref_id < ranges[y].ref_id;
            return ranges[x].start < ranges[y].start;
        });
        std::vector<SortedRegion> regs(n), un;
        std::vector<uint32_t> pmax(n), first(n_ref + 1, 0), un_first(n_ref + 1, 0);
        size_t j = 0;
        for (size_t r = 0; r < n_ref; ++r) {
            first[r] = (uint32_t)j;
            un_first[r] = (uint32_t)un.size();
            uint32_t mx = 0;
            bool open = false;
            while (j < n && ranges[order[j]].ref_id == r) {
                const sbx_region& g = ranges[order[j]];
                regs[j] = {g.start, g.end, order[j]};
                mx = std::max(mx, g.end);
                pmax[j] = mx;
                if (g.end > g.start) {
                    if (open && un.back().end >= g.start) un.back().end = std::max(un.back().end, g.end);
                    else { un.push_back({g.start, g.end, 0}); open = true; }
                }
                ++j;
            }
        }
        first[n_ref] = (uint32_t)j;
        un_first[n_ref] = (uint32_t)un.size();
        DevBuf<SortedRegion> d_regs2(n + 1), d_un(un.size() + 1);
        DevBuf<uint32_t> d_pmax2(n + 1), d_first2(n_ref + 2), d_unfirst(n_ref + 2);
        if (n) {
            SBX_HIP(hipMemcpyAsync(d_regs2.p, regs.data(), n * sizeof(SortedRegion), hipMemcpyHostToDevice, s));
            SBX_HIP(hipMemcpyAsync(d_pmax2.p, pmax.data(), n * 4, hipMemcpyHostToDevice, s));
        }
        if (!un.empty()) SBX_HIP(hipMemcpyAsync(d_un.p, un.data(), un.size() * sizeof(SortedRegion), hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(d_first2.p, first.data(), (n_ref + 1) * 4, hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(d_unfirst.p, un_first.data(), (n_ref + 1) * 4, hipMemcpyHostToDevice, s));
        for (sbx_ctx* f : files)
            launch_count_reads_mates(f->U(), f->d_desc.p, records_of(f), f->d_rec_ref.p, f->d_mate.p, d_regs2.p, d_pmax2.p, d_first2.p,
                                     d_un.p, d_unfirst.p, windows || c->mode == SBX_MODE_WINDOW /* every column lies in some window */, d_firstcol.p, S,
                                     c->min_bq, d_nb.p, d_nr.p, s);
        SBX_HIP(hipStreamSynchronize(s));   // host vectors above must outlive the async copies
    } else {
    launch_range_reduce(d_chunks.p, (uint32_t)n_chunks, c->d_counters.p, c->span_valid ? c->d_span.p : nullptr, c->d_slot_of.p,
                        c->d_tile_base.p, T, S, d_thr.p, n_thr, d_nb.p, d_cov.p, d_seen.p, s, c->compact_counters);
    DevBuf<uint64_t> d_wb, d_nw;
    if (windows) {
        d_wb.alloc(win_base.size() + 1);
        d_nw.alloc(n_win.size() + 1);
        SBX_HIP(hipMemcpyAsync(d_wb.p, win_base.data(), win_base.size() * 8, hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(d_nw.p, n_win.data(), n_win.size() * 8, hipMemcpyHostToDevice, s));
        for (sbx_ctx* f : files)
            launch_count_reads_windows(f->U(), f->d_desc.p, records_of(f), f->d_rec_ref.p, window, d_wb.p, d_nw.p, S, c->min_bq, d_nr.p, s);
    } else {
        // (ref, start)-sorted view + prefix max of ends per contig (cached with the range list)
        const size_t n_ref = c->hdr.refs.size();
        if (min_start && c->fix_mate) throw Error(SBX_EUNSUPPORTED, "--fix-mate-overlaps together with overlapping windows is not on the device path");
        if (!rc.sorted_valid) {
            std::vector<uint32_t> order(n);
            for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                if (ranges[x].ref_id != ranges[y].ref_id) return ranges[x].ref_id < ranges[y].ref_id;
                return ranges[x].start < ranges[y].start;
            });
            std::vector<SortedRegion> regs(n);
            std::vector<uint32_t> pmax(n), first(n_ref + 1, 0);
            size_t j = 0;
            for (size_t r = 0; r < n_ref; ++r) {
                first[r] = (uint32_t)j;
                uint32_t mx = 0;
                while (j < n && ranges[order[j]].ref_id == r) {
                    regs[j] = {ranges[order[j]].start, ranges[order[j]].end, order[j]};
                    mx = std::max(mx, ranges[order[j]].end);
                    pmax[j] = mx;
                    ++j;
                }
            }
            first[n_ref] = (uint32_t)j;
            rc.d_regs.ensure(n + 1);
            rc.d_pmax.ensure(n + 1);
            rc.d_first.ensure(n_ref + 2);
            if (n) {
                SBX_HIP(hipMemcpy(rc.d_regs.p, regs.data(), n * sizeof(SortedRegion), hipMemcpyHostToDevice));
                SBX_HIP(hipMemcpy(rc.d_pmax.p, pmax.data(), n * 4, hipMemcpyHostToDevice));
            }
            SBX_HIP(hipMemcpy(rc.d_first.p, first.data(), (n_ref + 1) * 4, hipMemcpyHostToDevice));
            if (min_start) {
                rc.d_min_start.ensure(n + 1);
                if (n) SBX_HIP(hipMemcpy(rc.d_min_start.p, min_start, n * 4, hipMemcpyHostToDevice));
            }
            rc.sorted_valid = true;
        }
        DevBuf<SortedRegion>& d_regs = rc.d_regs;
        DevBuf<uint32_t>&d_pmax = rc.d_pmax, &d_first = rc.d_first, &d_min_start = rc.d_min_start;
        for (sbx_ctx* f : files)
            launch_count_reads_regions(f->U(), f->d_desc.p, records_of(f), f->d_rec_ref.p, d_regs.p, d_pmax.p, d_first.p, S, c->min_bq,
                                       d_nr.p, min_start ? d_min_start.p : nullptr, d_nb.p, s);
    }
    }
    t.stop(s);
    std::vector<uint32_t>&h_nb = rc.h_nb, &h_nr = rc.h_nr, &h_seen = rc.h_seen;
    h_nb.resize(n * S); h_nr.resize(n * S); h_seen.resize(n);
    if (n) {
        SBX_HIP(hipMemcpyAsync(h_nb.data(), d_nb.p, n * S * 4, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipMemcpyAsync(h_nr.data(), d_nr.p, n * S * 4, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipMemcpyAsync(h_seen.data(), d_seen.p, n * 4, hipMemcpyDeviceToHost, s));
        if (n_thr && cov_counts) SBX_HIP(hipMemcpyAsync(cov_counts, d_cov.p, n * S * n_thr * 4, hipMemcpyDeviceToHost, s));
    }
    SBX_HIP(hipStreamSynchronize(s));
    c->stats.ms_reduce = t.ms();
    for (size_t i = 0; i < n * S; ++i) { stats[i].n_reads = h_nr[i]; stats[i].n_bases = h_nb[i]; }
    if (seen) for (size_t i = 0; i < n; ++i) seen[i] = h_seen[i] ? 1 : 0;
}

int sbx_depth_region_stats(sbx_ctx* c, const sbx_region* raw, size_t n, sbx_region_stats* stats, uint32_t* cov_counts,
                           uint8_t* seen) {
    return guarded(c, [&] {
        if (!c || (!raw && n) || !stats) throw Error(SBX_EINVAL, "null argument");
        std::vector<sbx_region> ranges(raw, raw + n);
        for (auto& r : ranges) {
            if (r.ref_id >= c->hdr.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
            if (r.end < r.start) r.end = r.start;
        }
        range_stats(c, ranges, false, 0, {}, {}, stats, cov_counts, seen);
    });
}

int sbx_depth_region_stats_from(sbx_ctx* c, const sbx_region* raw, size_t n, const uint32_t* min_start, sbx_region_stats* stats,
                                uint32_t* cov_counts, uint8_t* seen) {
    return guarded(c, [&] {
        if (!c || (!raw && n) || !stats) throw Error(SBX_EINVAL, "null argument");
        std::vector<sbx_region> ranges(raw, raw + n);
        for (auto& r : ranges) {
            if (r.ref_id >= c->hdr.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
            if (r.end < r.start) r.end = r.start;
        }
        range_stats(c, ranges, false, 0, {}, {}, stats, cov_counts, seen, min_start);
    });
}

// Every window of every contig that has data in the resident run, in two launches (round 5; VERDICT r4: window mode spent 86 ms of a
// whole-genome pass outside the pipeline kernels -- a chunk list of 3.1 M windows built on the host per contig call, and a pass over ALL
// records of the batch per contig call): count_reads_windows over the records once, range_reduce over the positions once with the
// windows generated from their id, the results kept until the next run and handed out by sbx_depth_window_stats.
static void window_stats_all(sbx_ctx* c) {
    sbx_ctx::WindowCache& wc = c->wc;
    SBX_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint32_t S = c->n_samples_eff, T = c->tile_pos, w = c->window;
    const uint32_t n_thr = (uint32_t)c->thresholds.size();
    if (n_thr > (uint32_t)kMaxThresholds) throw Error(SBX_EUNSUPPORTED, "more than 16 coverage thresholds");
    const size_t n_ref = c->hdr.refs.size();
    wc.valid = false;
    wc.base.assign(n_ref + 1, 0);
    wc.n_win.assign(n_ref + 1, 0);
    uint64_t total = 0;
    for (size_t r = 0; r < n_ref; ++r) {
        // contigs without an active tile have all-zero windows: they get no ids
        bool any = false;
        for (uint32_t t = c->h_tile_base[r]; t < c->h_tile_base[r + 1] && !any; ++t) any = c->h_slot_of[t] != 0xFFFFFFFFu;
        wc.base[r] = total;
        wc.n_win[r] = any ? (uint64_t)std::max(0, c->hdr.refs[r].length) / w : 0;
        total += wc.n_win[r];
    }
    wc.base[n_ref] = total;
    if (total > 0x3FFFFFF0ull) throw Error(SBX_EUNSUPPORTED, "too many windows");
    const size_t n = (size_t)total, ncov = n * S * std::max<uint32_t>(1, n_thr);
    wc.d_base.ensure(n_ref + 2); wc.d_nwin.ensure(n_ref + 2);
    wc.d_nb.ensure(n * S + 1); wc.d_nr.ensure(n * S + 1); wc.d_cov.ensure(ncov + 1); wc.d_seen.ensure(n + 1); wc.d_thr.ensure(n_thr + 1);
    SBX_HIP(hipMemcpyAsync(wc.d_base.p, wc.base.data(), (n_ref + 1) * 8, hipMemcpyHostToDevice, s));
    SBX_HIP(hipMemcpyAsync(wc.d_nwin.p, wc.n_win.data(), (n_ref + 1) * 8, hipMemcpyHostToDevice, s));
    if (n_thr) SBX_HIP(hipMemcpyAsync(wc.d_thr.p, c->thresholds.data(), n_thr * 4, hipMemcpyHostToDevice, s));
    SBX_HIP(hipMemsetAsync(wc.d_nb.p, 0, (n * S + 1) * 4, s));
    SBX_HIP(hipMemsetAsync(wc.d_nr.p, 0, (n * S + 1) * 4, s));
    SBX_HIP(hipMemsetAsync(wc.d_cov.p, 0, (ncov + 1) * 4, s));
    SBX_HIP(hipMemsetAsync(wc.d_seen.p, 0, (n + 1) * 4, s));
    EventTimer t;
    t.start(s);
    launch_range_reduce_windows(wc.d_base.p, wc.d_nwin.p, (uint32_t)n_ref, w, (uint32_t)n, c->d_counters.p, c->span_valid ? c->d_span.p : nullptr,
                                c->d_slot_of.p, c->d_tile_base.p, T, S, wc.d_thr.p, n_thr, wc.d_nb.p, wc.d_cov.p, wc.d_seen.p, s, c->compact_counters);
    for (sbx_ctx* f : files_of(c)) {
        const uint64_t nrec = (f == c && !c->members.empty()) ? c->primary_records : f->stats.n_records;
        launch_count_reads_windows(f->U(), f->d_desc.p, nrec, f->d_rec_ref.p, w, wc.d_base.p, wc.d_nwin.p, S, c->min_bq, wc.d_nr.p, s);
    }
    t.stop(s);
    wc.h_nb.resize(n * S); wc.h_nr.resize(n * S); wc.cov.assign(ncov, 0);
    if (n) {
        SBX_HIP(hipMemcpyAsync(wc.h_nb.data(), wc.d_nb.p, n * S * 4, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipMemcpyAsync(wc.h_nr.data(), wc.d_nr.p, n * S * 4, hipMemcpyDeviceToHost, s));
        if (n_thr) SBX_HIP(hipMemcpyAsync(wc.cov.data(), wc.d_cov.p, ncov * 4, hipMemcpyDeviceToHost, s));
    }
    SBX_HIP(hipStreamSynchronize(s));
    c->stats.ms_reduce = t.ms();
    wc.st.resize(n * S);
    for (size_t i = 0; i < n * S; ++i) { wc.st[i].n_reads = wc.h_nr[i]; wc.st[i].n_bases = wc.h_nb[i]; }
    wc.serial = c->run_serial; wc.window = w; wc.S = S; wc.thr = c->thresholds;
    wc.valid = true;
}

int sbx_depth_window_stats(sbx_ctx* c, uint32_t ref_id, uint64_t first_win, uint64_t n_win, sbx_region_stats* stats,
                           uint32_t* cov_counts) {
    return guarded(c, [&] {
        if (!c || !stats) throw Error(SBX_EINVAL, "null argument");
        if (ref_id >= c->hdr.refs.size()) throw Error(SBX_EINVAL, "Invalid reference sequence index");
        if (c->window == 0) throw Error(SBX_EINVAL, "positive window size must be specified");
        if (c->overlap != 0)
            throw Error(SBX_EUNSUPPORTED, "--overlap > 0 is not supported on the device path (the reference's overlapping-window "
                                          "bookkeeping is order dependent, see DESIGN.md section 6)");
        const uint64_t w = c->window;
        const uint64_t len = (uint64_t)std::max(0, c->hdr.refs[ref_id].length);
        const uint64_t total_win = len / w;                       // only full windows are ever printed (depth.d:1057,1071)
        if (first_win + n_win > total_win) throw Error(SBX_EINVAL, "window range exceeds the contig");
        static const bool all_at_once = [] { const char* e = getenv("SBX_WINDOWS_AT_ONCE"); return !e || atoi(e) != 0; }();
        // every window of the run at once -- unless there are too many of them to keep (small windows on a large genome: -w 1 on a
        // human genome is 3 G windows): then this call computes the windows it was asked for, as every call did before round 5
        bool at_once = all_at_once && !c->fix_mate;
        if (at_once) {
            if (!c->have_run) throw Error(SBX_EINVAL, "sbx_run() has not been called");
            static const uint64_t budget = [] { const char* e = getenv("SBX_WINDOW_CACHE_BYTES"); return e ? strtoull(e, nullptr, 10) : (1ull << 30); }();
            uint64_t total = 0;
            for (auto& r : c->hdr.refs) total += (uint64_t)std::max(0, r.length) / w;
            const uint64_t per_win = (uint64_t)c->n_samples_eff * (5 + 2 * std::max<uint64_t>(1, c->thresholds.size())) * 4 + 4;
            if (total > 0x3FFFFFF0ull || total * per_win > budget) at_once = false;
        }
        if (at_once) {
            sbx_ctx::WindowCache& wc = c->wc;
            const uint32_t S = c->n_samples_eff, n_thr = (uint32_t)c->thresholds.size();
            if (!(wc.valid && wc.serial == c->run_serial && wc.window == c->window && wc.S == S && wc.thr == c->thresholds)) window_stats_all(c);
            if (wc.n_win[ref_id] == 0) {            // a contig without data in this run: all-zero windows
                for (size_t i = 0; i < (size_t)n_win * S; ++i) stats[i] = sbx_region_stats{0, 0};
                if (cov_counts && n_thr) memset(cov_counts, 0, (size_t)n_win * S * n_thr * 4);
                return;
            }
            const size_t id0 = (size_t)(wc.base[ref_id] + first_win);
            memcpy(stats, wc.st.data() + id0 * S, (size_t)n_win * S * sizeof(sbx_region_stats));
            if (cov_counts && n_thr) memcpy(cov_counts, wc.cov.data() + id0 * S * n_thr, (size_t)n_win * S * n_thr * 4);
            return;
        }
        std::vector<sbx_region> ranges((size_t)n_win);
        for (uint64_t k = 0; k < n_win; ++k) ranges[(size_t)k] = {ref_id, (uint32_t)((first_win + k) * w), (uint32_t)((first_win + k + 1) * w)};
        // count_reads_windows indexes windows as win_base[ref] + k with k counted from 0 on the contig
        std::vector<uint64_t> wb(c->hdr.refs.size(), 0), nw(c->hdr.refs.size(), 0);
        // windows before first_win are not requested: shift the base so that id = k - first_win
        wb[ref_id] = (uint64_t)0 - first_win;
        nw[ref_id] = first_win + n_win;
        // records of other contigs see n_win == 0 and are skipped; windows k < first_win would get a
        // "negative" id: exclude them by temporarily treating them through a per-call clamp
        if (first_win != 0) {
            // simple and exact: compute from window 0 and copy the requested slice
            std::vector<sbx_region> all((size_t)(first_win + n_win));
            for (uint64_t k = 0; k < first_win + n_win; ++k) all[(size_t)k] = {ref_id, (uint32_t)(k * w), (uint32_t)((k + 1) * w)};
            wb[ref_id] = 0;
            const uint32_t S = c->n_samples_eff, n_thr = (uint32_t)c->thresholds.size();
            std::vector<sbx_region_stats> st(all.size() * S);
            std::vector<uint32_t> cv(all.size() * S * std::max<uint32_t>(1, n_thr));
            range_stats(c, all, true, (uint32_t)w, wb, nw, st.data(), cv.data(), nullptr);
            memcpy(stats, st.data() + first_win * S, (size_t)n_win * S * sizeof(sbx_region_stats));
            if (cov_counts && n_thr) memcpy(cov_counts, cv.data() + first_win * S * n_thr, (size_t)n_win * S * n_thr * 4);
            return;
        }
        range_stats(c, ranges, true, (uint32_t)w, wb, nw, stats, cov_counts, nullptr);
    });
}
int sbx_depth_base_tile_device(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end, void* d_out) {
    return guarded(c, [&] {
        if (!c || (!d_out && end > beg)) throw Error(SBX_EINVAL, "null argument");
        if (!c->have_run) throw Error(SBX_EINVAL, "sbx_run() has not been called");
        if (ref_id >= c->hdr.refs.size() || beg > end) throw Error(SBX_EINVAL, "bad interval");
        if (c->compact_counters) throw Error(SBX_EINVAL, "per-position base counters are kept by `base` runs only");
        SBX_HIP(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        const uint32_t T = c->tile_pos, S = c->n_samples_eff;
        const size_t row = (size_t)S * SBX_NCOUNTERS;
        const uint32_t t_first = c->h_tile_base[ref_id], t_end = c->h_tile_base[ref_id + 1];
        uint32_t* out = (uint32_t*)d_out;
        if (end > beg) SBX_HIP(hipMemsetAsync(out, 0, (size_t)(end - beg) * row * 4, s));
        for (uint64_t p = beg; p < end;) {
            const uint32_t t = t_first + (uint32_t)(p / T);
            const uint64_t tile_start = (uint64_t)(t - t_first) * T;
            if (t >= t_end || c->h_slot_of[t] == 0xFFFFFFFFu) { p = std::min<uint64_t>(end, tile_start + T); continue; }
            uint32_t t2 = t + 1;       // consecutive active tiles occupy consecutive slots: one copy per run of them
            while (t2 < t_end && (uint64_t)(t2 - t_first) * T < end && c->h_slot_of[t2] == c->h_slot_of[t] + (t2 - t)) ++t2;
            const uint64_t stop = std::min<uint64_t>(end, (uint64_t)(t2 - t_first) * T);
            const size_t slot = c->h_slot_of[t];
            SBX_HIP(hipMemcpyAsync(out + (size_t)(p - beg) * row, c->d_counters.p + slot * T * row + (size_t)(p - tile_start) * row,
                                   (size_t)(stop - p) * row * 4, hipMemcpyDeviceToDevice, s));
            p = stop;
        }
        SBX_HIP(hipStreamSynchronize(s));
    });
}

// K6: the text of `depth base` for [beg, end) of ref_id, formatted on the device (format.hip).
// FormatArgs of a `depth base` run for rows of ref_id (the names blob travels on the stream first); beg / end are set by the caller
static FormatArgs format_args(sbx_ctx* c, uint32_t ref_id, double min_cov, double max_cov, int annotate, hipStream_t s) {
    const uint32_t S = c->n_samples_eff;
    // names blob: contig name, then the sample names ("*" when the header has no read groups, as the CLI prints)
    // (the blob on the device is kept while the next call asks for the same contig and sample names: a caller that formats a contig
    //  piece by piece does not pay two copies and a synchronisation per piece)
    std::string blob = c->hdr.refs[ref_id].name;
    std::vector<uint32_t> soff;
    for (uint32_t i = 0; i < S; ++i) {
        soff.push_back((uint32_t)blob.size());
        if (!c->combined && i < c->hdr.sample_names.size()) blob += c->hdr.sample_names[i];
    }
    soff.push_back((uint32_t)blob.size());
    if (!c->fmt_blob_on_device || blob != c->h_fmt_blob || soff != c->h_fmt_soff) {
        c->fmt_blob_on_device = false;
        c->h_fmt_blob = blob;
        c->h_fmt_soff = soff;
        c->d_fmt_names.ensure(c->h_fmt_blob.size() + 1);
        c->d_fmt_soff.ensure(c->h_fmt_soff.size());
        SBX_HIP(hipMemcpyAsync(c->d_fmt_names.p, c->h_fmt_blob.data(), c->h_fmt_blob.size(), hipMemcpyHostToDevice, s));
        SBX_HIP(hipMemcpyAsync(c->d_fmt_soff.p, c->h_fmt_soff.data(), c->h_fmt_soff.size() * 4, hipMemcpyHostToDevice, s));
        SBX_HIP(hipStreamSynchronize(s));         // (the host copies may be changed by the next call)
        c->fmt_blob_on_device = true;
    }
    FormatArgs a{};
    a.counters = c->d_counters.p;
    a.span = c->span_valid ? c->d_span.p : nullptr;
    a.slot_of = c->d_slot_of.p;
    a.tile_first = c->h_tile_base[ref_id];
    a.tile_end = c->h_tile_base[ref_id + 1];
    a.T = c->tile_pos;
    a.S = S;
    // COV is an integer: the reference's double comparisons (depth.d:538) become integer bounds
    if (!(max_cov >= 0) || !(min_cov <= max_cov)) { a.lo = 1; a.hi = 0; }
    else {
        a.lo = min_cov <= 0 ? 0 : (min_cov >= 1.8e19 ? ~0ull : (uint64_t)std::ceil(min_cov));
        a.hi = max_cov >= 1.8e19 ? ~0ull : (uint64_t)std::floor(max_cov);
    }
    a.annotate = annotate ? 1u : 0u;
    a.combined = c->combined ? 1u : 0u;
    a.zero_fill = min_cov <= 0 ? 1u : 0u;
    a.names = c->d_fmt_names.p;
    a.ref_name_len = (uint32_t)c->hdr.refs[ref_id].name.size();
    a.sample_off = c->d_fmt_soff.p;
    a.max_sample_len = 0;
    for (size_t i = 0; i + 1 < c->h_fmt_soff.size(); ++i) a.max_sample_len = std::max(a.max_sample_len, c->h_fmt_soff[i + 1] - c->h_fmt_soff[i]);
    return a;
}

// measure the rows of [a.beg, a.end): chunk offsets on the device, total bytes on the host (one synchronisation)
static uint64_t format_measure(sbx_ctx* c, const FormatArgs& a, uint32_t* n_chunks_out, hipStream_t s) {
    const uint32_t per = format_chunk_positions();
    const uint32_t n_chunks = (uint32_t)(((uint64_t)(a.end - a.beg) + per - 1) / per);
    c->d_fmt_len.ensure(n_chunks);
    c->d_fmt_off.ensure((size_t)n_chunks + 1);
    launch_format_measure(a, n_chunks, c->d_fmt_len.p, s);
    launch_count_scan(c->d_fmt_len.p, n_chunks, c->d_fmt_off.p, nullptr, 0, s);
    if (!c->res) SBX_HIP(hipHostMalloc((void**)&c->res, sizeof(HostResults), hipHostMallocDefault));
    SBX_HIP(hipMemcpyAsync(&c->res->last_state, c->d_fmt_off.p + n_chunks, 8, hipMemcpyDeviceToHost, s));
    SBX_HIP(hipStreamSynchronize(s));
    *n_chunks_out = n_chunks;
    return c->res->last_state;
}

static void check_base_run(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end, const char* who) {
    if (!c->have_run) throw Error(SBX_EINVAL, "sbx_run() has not been called");
    if (c->mode != SBX_MODE_BASE) throw Error(SBX_EINVAL, std::string(who) + " needs a `depth base` run");
    // (the layout of d_counters belongs to the run, not to the current mode setting: a compact run holds one word per position)
    if (c->compact_counters) throw Error(SBX_EINVAL, std::string(who) + ": the last run kept {bases, depth} per position, not the seven counters");
    if (ref_id >= c->hdr.refs.size() || beg > end) throw Error(SBX_EINVAL, "bad interval");
}

int sbx_format_base_rows(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end, double min_cov, double max_cov, int annotate,
                         char* out, size_t cap, size_t* out_len) {
    return guarded(c, [&] {
        if (!c || !out_len) throw Error(SBX_EINVAL, "null argument");
        check_base_run(c, ref_id, beg, end, "sbx_format_base_rows");
        SBX_HIP(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        *out_len = 0;
        if (beg == end) return;
        FormatArgs a = format_args(c, ref_id, min_cov, max_cov, annotate, s);
        a.beg = beg;
        a.end = end;
        uint32_t n_chunks = 0;
        const uint64_t total = format_measure(c, a, &n_chunks, s);
        *out_len = (size_t)total;
        if (total > cap || (!out && total)) throw Error(SBX_ENOMEM, "output buffer too small for the formatted rows");
        if (!total) return;
        c->d_fmt_text.ensure((size_t)total + 64);
        launch_format_write(a, n_chunks, c->d_fmt_off.p, c->d_fmt_text.p, s);
        SBX_HIP(hipMemcpyAsync(out, c->d_fmt_text.p, (size_t)total, hipMemcpyDeviceToHost, s));
        SBX_HIP(hipStreamSynchronize(s));
    });
}

// The same text left in DEVICE memory (a consumer that compresses, checksums or ships it from there; bench.py's `device_text`): d_out
// is a device pointer of the context's device, or null to measure.
int sbx_format_base_rows_device(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end, double min_cov, double max_cov, int annotate,
                                void* d_out, size_t cap, size_t* out_len) {
    return guarded(c, [&] {
        if (!c || !out_len) throw Error(SBX_EINVAL, "null argument");
        check_base_run(c, ref_id, beg, end, "sbx_format_base_rows_device");
        hipStream_t s = c->stream;
        *out_len = 0;
        if (beg == end) return;
        FormatArgs a = format_args(c, ref_id, min_cov, max_cov, annotate, s);
        a.beg = beg;
        a.end = end;
        uint32_t n_chunks = 0;
        const uint64_t total = format_measure(c, a, &n_chunks, s);
        *out_len = (size_t)total;
        if (!d_out && cap == 0) return;
        if (total > cap || !d_out) throw Error(SBX_ENOMEM, "output buffer too small for the formatted rows");
        if (!total) return;
        launch_format_write(a, n_chunks, c->d_fmt_off.p, (uint8_t*)d_out, s);
        SBX_HIP(hipStreamSynchronize(s));
    });
}

// The same text handed to a writer piece by piece, in order: the device formats piece k + 1 while piece k travels to a
// pinned host buffer on the copy stream and the writer consumes piece k - 1 -- the D side passes the delegate that
// wraps its output File (sambamba/depth.d:1233-1234 flushes one in the reference).
int sbx_stream_base_rows(sbx_ctx* c, uint32_t ref_id, uint32_t beg, uint32_t end, double min_cov, double max_cov, int annotate,
                         sbx_write_fn write, void* user) {
    return guarded(c, [&] {
        if (!c || !write) throw Error(SBX_EINVAL, "null argument");
        check_base_run(c, ref_id, beg, end, "sbx_stream_base_rows");
        if (beg == end) return;
        struct Streaming {
            std::atomic<int>& n;
            explicit Streaming(std::atomic<int>& x) : n(x) { n.fetch_add(1); }
            ~Streaming() { n.fetch_sub(1); }
        } streaming_guard(c->text_streaming);
        hipStream_t s = c->stream;
        FormatArgs a = format_args(c, ref_id, min_cov, max_cov, annotate, s);
        uint64_t piece = 2u << 20;                // positions per piece (~55 MB of text at one sample, 30x: the two pinned buffers
                                                  // of a context are allocated on first use, 15 ms each at this size)
        if (const char* e = getenv("SBX_STREAM_PIECE")) { const long v = atol(e); if (v >= 256) piece = (uint64_t)v; }      // (tests)
        for (int i = 0; i < 2; ++i) {
            if (!c->text_ev_fmt[i]) SBX_HIP(hipEventCreateWithFlags(&c->text_ev_fmt[i], hipEventDisableTiming));
            if (!c->text_ev_copy[i]) SBX_HIP(hipEventCreateWithFlags(&c->text_ev_copy[i], hipEventDisableTiming));
        }
        size_t pending_len[2] = {0, 0};
        bool pending[2] = {false, false};
        auto drain = [&](int i) {
            if (!pending[i]) return;
            SBX_HIP(hipEventSynchronize(c->text_ev_copy[i]));
            pending[i] = false;
            if (pending_len[i] && write(user, (const char*)c->text_host[i], pending_len[i]) != 0)
                throw Error(SBX_EIO, "the output writer reported an error");
        };
        int k = 0;
        try {
        for (uint64_t p = beg; p < end; p += piece, k ^= 1) {
            a.beg = (uint32_t)p;
            a.end = (uint32_t)std::min<uint64_t>(end, p + piece);
            uint32_t n_chunks = 0;
            const uint64_t total = format_measure(c, a, &n_chunks, s);       // (synchronises the compute stream only)
            drain(k);                                                          // buffer k is free again once its piece is written
            if (total) {
                // (pieces differ in size by a few percent: a buffer that had to grow with every larger piece would be freed and
                //  allocated again and again, and hipFree waits for the whole device -- the copy of the previous piece included)
                if (c->d_fmt_text2[k].n < (size_t)total + 64) c->d_fmt_text2[k].alloc((size_t)total + (size_t)(total / 4) + (1u << 20));
                if (c->text_host_cap[k] < total) {
                    if (c->text_host[k]) SBX_HIP(hipHostFree(c->text_host[k]));
                    c->text_host[k] = nullptr;
                    c->text_host_cap[k] = (size_t)(total + total / 8 + (1u << 20));
                    SBX_HIP(hipHostMalloc((void**)&c->text_host[k], c->text_host_cap[k], hipHostMallocDefault));
                }
                launch_format_write(a, n_chunks, c->d_fmt_off.p, c->d_fmt_text2[k].p, s);
                SBX_HIP(hipEventRecord(c->text_ev_fmt[k], s));
                SBX_HIP(hipStreamWaitEvent(c->text_stream, c->text_ev_fmt[k], 0));
                SBX_HIP(hipMemcpyAsync(c->text_host[k], c->d_fmt_text2[k].p, (size_t)total, hipMemcpyDeviceToHost, c->text_stream));
                SBX_HIP(hipEventRecord(c->text_ev_copy[k], c->text_stream));
                pending[k] = true;
                pending_len[k] = (size_t)total;
                // d_fmt_off / d_fmt_len are reused by the next measure: format_write of this piece must have read them
                SBX_HIP(hipEventSynchronize(c->text_ev_fmt[k]));
            }
            drain(k ^ 1);                                                      // the previous piece: copied while this one was formatted
        }
        drain(0);
        drain(1);
        } catch (...) {      // leave nothing in flight on the buffers the next call reuses
            (void)hipStreamSynchronize(c->text_stream);
            (void)hipStreamSynchronize(s);
            throw;
        }
    });
}

// extent of the tile grid of a contig (positions) and activity of a tile -- used by the CLI to skip
// empty stretches without copying zeros.
int sbx_tile_info(sbx_ctx* c, uint32_t* tile_pos, uint32_t* n_samples) {
    if (!c || !c->have_run) return SBX_EINVAL;
    if (tile_pos) *tile_pos = c->tile_pos;
    if (n_samples) *n_samples = c->n_samples_eff;
    return SBX_OK;
}
// next active tile of ref_id at or after position `from`; returns its [beg,end) or beg==end==UINT32_MAX
int sbx_next_active_range(sbx_ctx* c, uint32_t ref_id, uint64_t from, uint64_t* beg, uint64_t* end) {
    if (!c || !c->have_run || ref_id >= c->hdr.refs.size() || !beg || !end) return SBX_EINVAL;
    const uint32_t T = c->tile_pos;
    const uint32_t t_first = c->h_tile_base[ref_id], t_end = c->h_tile_base[ref_id + 1];
    uint64_t t = t_first + from / T;
    while (t < t_end && c->h_slot_of[(size_t)t] == 0xFFFFFFFFu) ++t;
    if (t >= t_end) { *beg = *end = ~0ULL; return SBX_OK; }
    uint64_t t2 = t;
    while (t2 < t_end && c->h_slot_of[(size_t)t2] != 0xFFFFFFFFu) ++t2;
    *beg = std::max<uint64_t>(from, (t - t_first) * (uint64_t)T);
    *end = (t2 - t_first) * (uint64_t)T;
    return SBX_OK;
}

}  // extern "C"
