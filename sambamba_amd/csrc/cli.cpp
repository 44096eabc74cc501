// cli.cpp -- `sbx-depth`: the host side of `sambamba depth base|region|window` on top of the
// C ABI of libsbx_depth.so.  It mirrors depth_main (sambamba/depth.d:1079-1245): same
// sub-commands, same options (depth.d:59-98), same text output byte for byte, same error
// line ("sambamba-depth: <msg>", exit code 1).  All counting happens on the MI355X through
// sbx_run(); this file only parses arguments and prints.
//
//   sbx-depth base   [-F filter] [-o out] [-c min] [-C max] [-q bq] [-a] [--combined] [-L regions] [-z] in.bam
//   sbx-depth region -L regions [-T thr]... [common options] in.bam
//   sbx-depth window -w size [--overlap n] [-T thr]... [common options] in.bam
#include <cerrno>
#include <csignal>
#include <fcntl.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sbx_depth.h"
#include "host_io.hpp"

using namespace sbx;

namespace {

struct Options {
    std::string mode;
    std::vector<std::string> bams;
    std::string filter;
    bool has_filter = false;
    std::string output_fn;
    int n_threads = 0;
    double min_cov = 0.0, max_cov = 1e50;
    int min_bq = 0;
    bool annotate = false, combined = false, fix_mate = false;
    std::string regions;
    bool has_regions = false;
    bool report_zero = false;
    std::vector<uint32_t> thresholds;
    unsigned long long window = 0, overlap = 0;
    int gpus = 0;             // --gpus N: shard the job by position over N devices (an extension; SBX_DEVICES lists the ordinals)
};

void usage() {  // depth.d:50-99
    fputs("Usage: sambamba-depth region|window|base [options] input.bam  [input2.bam [...]]\n\n"
          "          All BAM files must be coordinate-sorted and indexed.\n\n"
          "          The tool has three modes: base, region, and window,\n"
          "          each name means per which unit to print the statistics.\n\n"
          "Common options:\n"
          "         -F, --filter=FILTER\n"
          "                    set custom filter for alignments; the default value is\n"
          "                    'mapping_quality > 0 and not duplicate and not failed_quality_control'\n"
          "         -o, --output-file=FILENAME\n"
          "                    output filename (by default /dev/stdout)\n"
          "         -t, --nthreads=NTHREADS\n"
          "                    maximum number of threads to use\n"
          "         -c, --min-coverage=MINCOVERAGE\n"
          "                    minimum mean coverage for output (default: 0 for region/window, 1 for base)\n"
          "         -C, --max-coverage=MAXCOVERAGE\n"
          "                    maximum mean coverage for output\n"
          "         -q, --min-base-quality=QUAL\n"
          "                    don't count bases with lower base quality\n"
          "         --combined\n"
          "                    output combined statistics for all samples\n"
          "         -a, --annotate\n"
          "                    add additional column of y/n instead of\n"
          "                    skipping records not satisfying the criteria\n"
          "         -m, --fix-mate-overlaps\n"
          "                    detect overlaps of mate reads and handle them on per-base basis\n"
          "base subcommand options:\n"
          "         -L, --regions=FILENAME|REGION\n"
          "                    list or regions of interest or a single region in form chr:beg-end (optional)\n"
          "         -z, --report-zero-coverage (DEPRECATED, use --min-coverage=0 instead)\n"
          "                    don't skip zero coverage bases\n"
          "region subcommand options:\n"
          "         -L, --regions=FILENAME|REGION\n"
          "                    list or regions of interest or a single region in form chr:beg-end (required)\n"
          "         -T, --cov-threshold=COVTHRESHOLD\n"
          "                    multiple thresholds can be provided,\n"
          "                    for each one an extra column will be added,\n"
          "                    the percentage of bases in the region\n"
          "                    where coverage is more than this value\n"
          "window subcommand options:\n"
          "         -w, --window-size=WINDOWSIZE\n"
          "                    breadth of the window, in bp (required)\n"
          "         --overlap=OVERLAP\n"
          "                    overlap of successive windows, in bp (default is 0)\n"
          "         -T, --cov-threshold=COVTHRESHOLD\n"
          "                    same meaning as in 'region' subcommand\n",
          stderr);
}

bool parse_args(int argc, char** argv, Options* o, std::string* err) {
    o->mode = argv[1];
    if (o->mode == "base") o->min_cov = 1;  // depth.d:1113-1114
    struct Spec { const char* lng; char sht; int kind; };
    static const Spec specs[] = {
        {"filter", 'F', 1}, {"output-filename", 'o', 1}, {"nthreads", 't', 1}, {"min-coverage", 'c', 1},
        {"max-coverage", 'C', 1}, {"min-base-quality", 'q', 1}, {"annotate", 'a', 0}, {"combined", 0, 0},
        {"fix-mate-overlaps", 'm', 0}, {"regions", 'L', 1}, {"report-zero-coverage", 'z', 0},
        {"cov-threshold", 'T', 1}, {"window-size", 'w', 1}, {"overlap", 0, 1}, {"gpus", 0, 1}};
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        const Spec* sp = nullptr;
        std::string val;
        bool have_val = false;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            size_t eq = a.find('=');
            std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            for (auto& s : specs) if (name == s.lng) sp = &s;
            if (eq != std::string::npos) { val = a.substr(eq + 1); have_val = true; }
        } else if (a.size() >= 2 && a[0] == '-' && a[1] != '-') {
            for (auto& s : specs) if (s.sht && a[1] == s.sht) sp = &s;
            if (sp && a.size() > 2) { val = a.substr(a[2] == '=' ? 3 : 2); have_val = true; }
        }
        if (!sp) { o->bams.push_back(a); continue; }
        if (sp->kind == 1 && !have_val) {
            if (i + 1 >= argc) { *err = "Missing value for argument " + a + "."; return false; }
            val = argv[++i];
        }
        std::string n = sp->lng;
        if (n == "filter") { o->filter = val; o->has_filter = true; }
        else if (n == "output-filename") o->output_fn = val;
        else if (n == "nthreads") o->n_threads = atoi(val.c_str());
        else if (n == "min-coverage") o->min_cov = atof(val.c_str());
        else if (n == "max-coverage") o->max_cov = atof(val.c_str());
        else if (n == "min-base-quality") o->min_bq = atoi(val.c_str());
        else if (n == "annotate") o->annotate = true;
        else if (n == "combined") o->combined = true;
        else if (n == "fix-mate-overlaps") o->fix_mate = true;
        else if (n == "regions") { o->regions = val; o->has_regions = true; }
        else if (n == "report-zero-coverage") o->report_zero = true;
        else if (n == "cov-threshold") o->thresholds.push_back((uint32_t)strtoul(val.c_str(), nullptr, 10));
        else if (n == "window-size") o->window = strtoull(val.c_str(), nullptr, 10);
        else if (n == "overlap") o->overlap = strtoull(val.c_str(), nullptr, 10);
        else if (n == "gpus") o->gpus = atoi(val.c_str());
    }
    if (o->mode == "window") o->has_regions = false;  // -L is not parsed in window mode (depth.d:1139)
    return true;
}

constexpr size_t kMaxCliThresholds = 64;

struct Out {
    FILE* fp = stdout;
    std::string buf;
    void put(const char* s, size_t n) {
        buf.append(s, n);
        if (buf.size() > (4u << 20)) flush();
    }
    void put(const std::string& s) { put(s.data(), s.size()); }
    void flush() {
        if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), fp);
        buf.clear();
    }
};

inline char* u64toa(uint64_t v, char* end) {  // writes backwards, returns start
    do { *--end = (char)('0' + v % 10); v /= 10; } while (v);
    return end;
}

struct Fail { std::string msg; };
void check(sbx_ctx* c, int rc) { if (rc != SBX_OK) throw Fail{sbx_last_error(c)}; }

bool region_less(const sbx_region& a, const sbx_region& b) {
    if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
    if (a.start != b.start) return a.start < b.start;
    return a.end < b.end;
}

// ---------------------------------------------------------------------------------------------
// depth base: PerBasePrinter (depth.d:402-607) driven from the device's dense counter tiles.
// A "column" exists at every position spanned by >= 1 admitted read (covered[] from the device).
// ---------------------------------------------------------------------------------------------
class BasePrinter {
public:
    BasePrinter(sbx_ctx* c, const Options& o, Out& out, const std::vector<std::string>& samples)
        : c_(c), o_(o), out_(out), samples_(samples) {
        sbx_header_info hi;
        sbx_header(c, &hi);
        n_ref_ = hi.n_ref;
        S_ = o.combined ? 1u : (uint32_t)samples.size();
    }
    void set_bed(const std::vector<sbx_region>& bed) { bed_ = bed; raw_ = bed; bed_provided_ = true; cur_head_ = 0; raw_head_ = 0; }
    void header() {
        std::string h = "REF\tPOS\tCOV\tA\tC\tG\tT\tDEL\tREFSKIP";
        if (!o_.combined) h += "\tSAMPLE";
        if (o_.annotate) h += "\tFLAG";
        h += "\n";
        out_.put(h);
    }
    // Device-formatted output (K6, sbx_format_base_rows): without -L the text is a pure function of the
    // position -- a column's rows, or with min_cov == 0 all-zero rows for every position of every contig
    // (what push/close/writeEmptyColumns add up to) -- and with -L and min_cov > 0 it is the same restricted
    // to the merged regions (outputRequired, depth.d:558-565).  -L with min_cov == 0 keeps the stateful
    // host emulation below (raw BED consumption quirks of writeEmptyColumns).
    bool device_format_applies() const {
        if (getenv("SBX_HOST_FORMAT")) return false;
        if (o_.min_cov < 0) return false;
        return !bed_provided_ || o_.min_cov > 0;
    }
    void run_device(int r0, int r1) {
        // the library formats on the device and hands the text over piece by piece (pinned buffers, the next piece is
        // formatted and copied while this one is written)
        auto range = [&](uint32_t r, uint64_t b, uint64_t e) {
            out_.flush();
            check(c_, sbx_stream_base_rows(c_, r, (uint32_t)b, (uint32_t)e, o_.min_cov, o_.max_cov, o_.annotate ? 1 : 0,
                                           [](void* u, const char* d, size_t n) -> int { return fwrite(d, 1, n, (FILE*)u) == n ? 0 : 1; },
                                           out_.fp));
        };
        if (bed_provided_) {      // merged, sorted regions
            for (auto& g : bed_)
                if ((int)g.ref_id >= r0 && (int)g.ref_id < r1) range(g.ref_id, g.start, g.end);
            return;
        }
        for (int r = r0; r < r1; ++r) {
            const uint64_t len = (uint64_t)sbx_ref_length(c_, r);
            if (o_.min_cov == 0) {
                // Every position of a contig with pileup columns has rows.  A contig WITHOUT columns is zero-filled
                // only before the first and after the last contig that has some: push() fills from the previous
                // column's contig straight to the current one and skips what lies between (depth.d:574-583), close()
                // fills everything after the last column (depth.d:593-606).
                uint64_t b0 = 0, e0 = 0;
                check(c_, sbx_next_active_range(c_, (uint32_t)r, 0, &b0, &e0));
                if (b0 == ~0ULL) {
                    if (!seen_columns_) range((uint32_t)r, 0, len);
                    else pending_empty_.push_back(r);
                    continue;
                }
                seen_columns_ = true;
                pending_empty_.clear();
                range((uint32_t)r, 0, len);
            }
            uint64_t from = o_.min_cov == 0 ? len : 0;
            for (;;) {       // otherwise only stretches with admitted reads can have rows
                uint64_t b, e;
                check(c_, sbx_next_active_range(c_, (uint32_t)r, from, &b, &e));
                if (b == ~0ULL) break;
                b = std::max(b, from);
                if (o_.min_cov == 0) host_columns(r, b, e);       // alignments hanging over the contig end: columns only
                else range((uint32_t)r, b, e);
                from = e;
            }
        }
    }
    // rows of [beg, end) of contig r from context `c` (a slice of the pipelined run: min_cov > 0, no -L)
    void run_slice(sbx_ctx* c, uint32_t r, uint64_t beg, uint64_t end) {
        uint64_t from = beg;
        for (;;) {
            uint64_t b, e;
            check(c, sbx_next_active_range(c, r, from, &b, &e));
            if (b == ~0ULL || b >= end) break;
            b = std::max(b, from);
            e = std::min(e, end);
            out_.flush();
            check(c, sbx_stream_base_rows(c, r, (uint32_t)b, (uint32_t)e, o_.min_cov, o_.max_cov, o_.annotate ? 1 : 0,
                                          [](void* u, const char* d, size_t n) -> int { return fwrite(d, 1, n, (FILE*)u) == n ? 0 : 1; },
                                          out_.fp));
            from = e;
        }
    }
    void run_device_empty(int r) {
        std::vector<char> text;
        const uint64_t len = (uint64_t)sbx_ref_length(c_, r), CH = 8u << 20;
        for (uint64_t p = 0; p < len; p += CH) {
            const uint64_t q = std::min(len, p + CH);
            size_t need = 0;
            text.resize((size_t)(q - p) * 40 * S_ + 64);
            int rc = sbx_format_base_rows(c_, (uint32_t)r, (uint32_t)p, (uint32_t)q, o_.min_cov, o_.max_cov, o_.annotate ? 1 : 0,
                                          text.data(), text.size(), &need);
            if (rc == SBX_ENOMEM && need > text.size()) {
                text.resize(need);
                rc = sbx_format_base_rows(c_, (uint32_t)r, (uint32_t)p, (uint32_t)q, o_.min_cov, o_.max_cov, o_.annotate ? 1 : 0,
                                          text.data(), text.size(), &need);
            }
            check(c_, rc);
            out_.flush();
            fwrite(text.data(), 1, need, out_.fp);
        }
    }
    void host_columns(int r, uint64_t b, uint64_t e) {
        std::vector<uint32_t> cnt((size_t)(e - b) * S_ * SBX_NCOUNTERS);
        std::vector<uint8_t> cov((size_t)(e - b));
        check(c_, sbx_depth_base_tile(c_, (uint32_t)r, (uint32_t)b, (uint32_t)e, cnt.data(), cov.data()));
        for (uint64_t x = b; x < e; ++x)
            if (cov[(size_t)(x - b)]) write_column(r, (int64_t)x, &cnt[(size_t)(x - b) * S_ * SBX_NCOUNTERS]);
        out_.flush();
    }
    // rows of contigs [r0, r1) (the batch the device has just processed); finish() after the last batch
    void run_refs(int r0, int r1) {
        if (device_format_applies()) { run_device(r0, r1); return; }
        std::vector<uint32_t> cnt;
        std::vector<uint8_t> cov;
        const uint64_t CH = 1u << 20;
        for (int r = r0; r < r1; ++r) {
            uint64_t from = 0;
            for (;;) {
                uint64_t b, e;
                check(c_, sbx_next_active_range(c_, (uint32_t)r, from, &b, &e));
                if (b == ~0ULL) break;
                for (uint64_t p = b; p < e; p += CH) {
                    uint64_t q = std::min(e, p + CH);
                    cnt.resize((size_t)(q - p) * S_ * SBX_NCOUNTERS);
                    cov.resize((size_t)(q - p));
                    check(c_, sbx_depth_base_tile(c_, (uint32_t)r, (uint32_t)p, (uint32_t)q, cnt.data(), cov.data()));
                    for (uint64_t x = p; x < q; ++x)
                        if (cov[(size_t)(x - p)]) push(r, (int64_t)x, &cnt[(size_t)(x - p) * S_ * SBX_NCOUNTERS]);
                }
                from = e;
            }
        }
    }
    void finish() {
        if (!device_format_applies()) { close(); return; }
        if (o_.min_cov == 0 && !bed_provided_)       // contigs without columns after the last one that had some
            for (int r : pending_empty_) run_device_empty(r);
    }

private:
    sbx_ctx* c_;
    const Options& o_;
    Out& out_;
    const std::vector<std::string>& samples_;
    int n_ref_ = 0;
    uint32_t S_ = 1;
    bool bed_provided_ = false;
    std::vector<sbx_region> bed_;   // NonOverlappingRegionStatsCollector view (depth.d:171-198)
    size_t cur_head_ = 0;
    std::vector<sbx_region> raw_;   // raw_bed, consumed by writeEmptyColumns (depth.d:464-486)
    size_t raw_head_ = 0;
    int prev_ref_ = -2;
    int64_t prev_pos_ = 0;
    std::vector<std::string> tails_;
    bool seen_columns_ = false;            // device-formatted -c 0 output: has any contig so far had a pileup column?
    std::vector<int> pending_empty_;       // ... contigs without columns seen since the last one that had some

    static bool fully_left_of(const sbx_region& g, uint32_t ref, uint32_t pos) { return g.ref_id < ref || (g.ref_id == ref && g.end <= pos); }
    static bool overlaps(const sbx_region& g, uint32_t ref, uint32_t pos) { return g.ref_id == ref && g.start <= pos && pos < g.end; }

    bool output_required(int ref, int64_t pos) {  // depth.d:558-565
        if (!bed_provided_) return true;
        while (cur_head_ < bed_.size() && fully_left_of(bed_[cur_head_], (uint32_t)ref, (uint32_t)pos)) ++cur_head_;
        return cur_head_ < bed_.size() && overlaps(bed_[cur_head_], (uint32_t)ref, (uint32_t)pos);
    }
    void init_tails() {  // depth.d:436-450
        if (!tails_.empty()) return;
        if (o_.combined) {
            tails_.push_back("\t0\t0\t0\t0\t0\t0\t0");
            if (o_.annotate) tails_[0] += (o_.min_cov > 0 ? "\tn" : "\ty");
        } else {
            for (auto& s : samples_) {
                tails_.push_back("\t0\t0\t0\t0\t0\t0\t0\t" + s);
                if (o_.annotate) tails_.back() += (o_.min_cov > 0 ? "\tn" : "\ty");
            }
        }
    }
    void emit_empty(const char* ref_name, size_t ref_len, long from, long to) {
        char num[24];
        for (long pos = from; pos < to; ++pos) {
            char* e = num + sizeof num;
            char* s = u64toa((uint64_t)pos, e);
            for (auto& t : tails_) {
                out_.put(ref_name, ref_len);
                out_.put("\t", 1);
                out_.put(s, (size_t)(e - s));
                out_.put(t);
                out_.put("\n", 1);
            }
        }
    }
    void write_empty(long ref_id, long start, long end) {  // depth.d:452-487
        if (o_.min_cov > 0 && !o_.annotate) return;
        const char* name = sbx_ref_name(c_, (int)ref_id);
        size_t nl = strlen(name);
        init_tails();
        if (!bed_provided_) { emit_empty(name, nl, start, end); return; }
        if (raw_head_ >= raw_.size() || raw_[raw_head_].ref_id > (uint32_t)ref_id) return;
        while (raw_head_ < raw_.size() && raw_[raw_head_].ref_id < (uint32_t)ref_id) ++raw_head_;
        while (raw_head_ < raw_.size() && raw_[raw_head_].ref_id == (uint32_t)ref_id) {
            sbx_region& f = raw_[raw_head_];
            if (fully_left_of(f, (uint32_t)ref_id, (uint32_t)start)) { ++raw_head_; continue; }
            long from = std::max<long>(start, f.start), to = std::min<long>(end, f.end);
            if (from >= to) break;
            emit_empty(name, nl, from, to);
            f.start = (uint32_t)to;
            if (f.start >= f.end) ++raw_head_;
        }
        bed_.assign(raw_.begin() + (long)raw_head_, raw_.end());   // collector rebuilt from what is left (depth.d:485)
        cur_head_ = 0;
    }
    void write_column(int ref, int64_t pos, const uint32_t* cnt) {  // depth.d:534-555
        const char* name = sbx_ref_name(c_, ref);
        size_t nl = strlen(name);
        char num[24];
        for (uint32_t s = 0; s < S_; ++s) {
            const uint32_t* v = cnt + (size_t)s * SBX_NCOUNTERS;
            uint64_t total = (uint64_t)v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6];
            bool ok = (double)total >= o_.min_cov && (double)total <= o_.max_cov;
            if (!ok && !o_.annotate) return;  // return, not continue (depth.d:540-541)
            out_.put(name, nl);
            auto num_field = [&](uint64_t x) {
                char* e = num + sizeof num;
                char* b = u64toa(x, e);
                out_.put("\t", 1);
                out_.put(b, (size_t)(e - b));
            };
            num_field((uint64_t)pos);
            num_field(total);
            num_field(v[0]); num_field(v[1]); num_field(v[2]); num_field(v[3]);
            num_field(v[5]); num_field(v[6]);
            if (!o_.combined) { out_.put("\t", 1); out_.put(samples_[s]); }
            if (o_.annotate) out_.put(ok ? "\ty" : "\tn", 2);
            out_.put("\n", 1);
        }
    }
    void push(int ref, int64_t pos, const uint32_t* cnt) {  // depth.d:567-591
        if (o_.min_cov > 0) {
            if (output_required(ref, pos)) write_column(ref, pos, cnt);
            return;
        }
        if (prev_ref_ == -2) {
            for (int id = 0; id < ref; ++id) write_empty(id, 0, (long)sbx_ref_length(c_, id));
            write_empty(ref, 0, (long)pos);
        } else if (prev_ref_ != ref) {
            write_empty(prev_ref_, (long)prev_pos_ + 1, (long)sbx_ref_length(c_, prev_ref_));
            write_empty(ref, 0, (long)pos);
        } else if (prev_pos_ != pos - 1) {
            write_empty(ref, (long)prev_pos_ + 1, (long)pos);
        }
        prev_ref_ = ref;
        prev_pos_ = pos;
        if (output_required(ref, pos)) write_column(ref, pos, cnt);
    }
    void close() {  // depth.d:593-606
        if (!(o_.min_cov == 0)) return;
        if (prev_ref_ == -2) {
            for (int id = 0; id < n_ref_; ++id) write_empty(id, 0, (long)sbx_ref_length(c_, id));
        } else {
            write_empty(prev_ref_, (long)prev_pos_ + 1, (long)sbx_ref_length(c_, prev_ref_));
            for (int id = prev_ref_ + 1; id < n_ref_; ++id) write_empty(id, 0, (long)sbx_ref_length(c_, id));
        }
    }
};

std::string fmt_g(float f) {  // D's write(float) == %g with 6 significant digits (depth.d:859-864)
    char b[64];
    snprintf(b, sizeof b, "%g", (double)f);
    return b;
}

void print_bed_header(Out& out, const Options& o, size_t n_before) {  // depth.d:643-659
    static const char* def[] = {"chrom", "chromStart", "chromEnd"};
    std::string h = "# ";
    for (size_t i = 0; i < std::min<size_t>(3, n_before); ++i) h += std::string(def[i]) + "\t";
    for (size_t k = 3; k < n_before; ++k) h += "F" + std::to_string(k) + "\t";
    h += "readCount\tmeanCoverage";
    for (auto t : o.thresholds) h += "\tpercentage" + std::to_string(t);
    if (!o.combined) h += "\tsampleName";
    if (o.annotate) h += "\tmeanCovWithinBounds";
    h += "\n";
    out.put(h);
}

// printRegionStats (depth.d:847-876)
void print_region_row(Out& out, const Options& o, const std::string& prefix, uint32_t length, const sbx_region_stats& st,
                      const uint32_t* cov, const std::string& sample) {
    float mean_cov = (float)st.n_bases / (float)length;
    bool ok = (double)mean_cov >= o.min_cov && (double)mean_cov <= o.max_cov;
    if (!ok && !o.annotate) return;
    std::string row = prefix;
    row += std::to_string(st.n_reads) + "\t" + fmt_g(mean_cov);
    for (size_t j = 0; j < o.thresholds.size(); ++j) {
        float pct = (float)cov[j] * 100 / (float)length;
        if (o.thresholds[j] == 0) pct = 100.0f;
        row += "\t" + fmt_g(pct);
    }
    if (!o.combined) row += "\t" + sample;
    if (o.annotate) row += ok ? "\ty" : "\tn";
    row += "\n";
    out.put(row);
}

// position of the first pileup column of the run (first admitted read), or false if there is none
bool first_column(sbx_ctx* c, int r0, int r1, int* ref_out, uint64_t* pos_out) {
    std::vector<uint32_t> cnt;
    std::vector<uint8_t> cov;
    uint32_t T = 0, S = 0;
    check(c, sbx_tile_info(c, &T, &S));
    for (int r = r0; r < r1; ++r) {
        uint64_t from = 0;
        for (;;) {
            uint64_t b, e;
            check(c, sbx_next_active_range(c, (uint32_t)r, from, &b, &e));
            if (b == ~0ULL) break;
            for (uint64_t p = b; p < e; p += 65536) {
                uint64_t q = std::min(e, p + 65536);
                cnt.resize((size_t)(q - p) * S * SBX_NCOUNTERS);
                cov.resize((size_t)(q - p));
                check(c, sbx_depth_base_tile(c, (uint32_t)r, (uint32_t)p, (uint32_t)q, nullptr, cov.data()));     // `covered` alone
                for (uint64_t x = p; x < q; ++x)
                    if (cov[(size_t)(x - p)]) { *ref_out = r; *pos_out = x; return true; }
            }
            from = e;
        }
    }
    return false;
}

// first / last pileup column of the resident run inside [beg, end) of contig r (a slice of a sharded job)
bool first_column_in(sbx_ctx* c, uint32_t r, uint64_t beg, uint64_t end, uint64_t* pos_out) {
    std::vector<uint8_t> cov;
    uint64_t from = beg;
    while (from < end) {
        uint64_t b, e;
        check(c, sbx_next_active_range(c, r, from, &b, &e));
        if (b == ~0ULL || b >= end) return false;
        b = std::max(b, from);
        e = std::min(e, end);
        for (uint64_t p = b; p < e; p += 65536) {
            const uint64_t q = std::min(e, p + 65536);
            cov.resize((size_t)(q - p));
            check(c, sbx_depth_base_tile(c, r, (uint32_t)p, (uint32_t)q, nullptr, cov.data()));
            for (uint64_t x = p; x < q; ++x)
                if (cov[(size_t)(x - p)]) { *pos_out = x; return true; }
        }
        from = e;
    }
    return false;
}
bool last_column_from(sbx_ctx* c, uint32_t r, uint64_t beg, uint64_t* pos_out) {
    std::vector<std::pair<uint64_t, uint64_t>> runs;
    for (uint64_t from = beg;;) {
        uint64_t b, e;
        check(c, sbx_next_active_range(c, r, from, &b, &e));
        if (b == ~0ULL) break;
        runs.push_back({std::max(b, from), e});
        from = e;
    }
    std::vector<uint8_t> cov;
    for (size_t i = runs.size(); i-- > 0;) {
        for (uint64_t q = runs[i].second; q > runs[i].first;) {
            const uint64_t p = q > runs[i].first + 65536 ? q - 65536 : runs[i].first;
            cov.resize((size_t)(q - p));
            check(c, sbx_depth_base_tile(c, r, (uint32_t)p, (uint32_t)q, nullptr, cov.data()));
            for (uint64_t x = q; x > p; --x)
                if (cov[(size_t)(x - 1 - p)]) { *pos_out = x - 1; return true; }
            q = p;
        }
    }
    return false;
}

// What a job sharded over several devices collected for the window printer (run_sharded): the statistics of every full window,
// of the windows behind a contig's end that alignments hanging over it finish or leave unfinished, the first column of the run
// and the last column of every contig -- everything PerWindowPrinter's rules below are stated in.
struct WindowData {
    std::vector<uint64_t> base, n_full;                   // per contig: index of its window 0 in st / cov, number of full windows
    std::vector<sbx_region_stats> st;                     // [window][S]
    std::vector<uint32_t> cov;                            // [window][S][max(1, n_thr)]
    std::vector<std::vector<sbx_region_stats>> extra_st;  // per contig: windows n_full ..
    std::vector<std::vector<uint32_t>> extra_cov;
    std::vector<char> has_cols;
    std::vector<uint64_t> firstcol, lastcol;
};

// PerWindowPrinter (depth.d:933-1077), fed one batch of contigs at a time.  Windows k = [k*step, k*step + w),
// step = w - overlap, live in a ring of n = ceil(w / step) slots in the reference; what it prints is, per window:
//   * n_reads / n_bases of the window as a region -- except in the FIRST ring of the run (windows 1 .. n-1 of contig 0
//     when the first pileup column lies on it): is_first_occurrence starts out false there (depth.d:1031-1032), so only
//     reads starting inside the window are counted;
//   * coverage thresholds over the columns in [cs, k*step + w), cs = (k - n)*step + w for k >= n: every column updates
//     all n slots of the ring (depth.d:215-226), including a slot whose window has not begun when w is not a multiple
//     of the step;
//   * all k with k*step + w <= length for a contig with columns, length / step all-zero windows for a read-less contig;
//     nothing for windows finished before the first column of the run (the sample list does not exist yet);
//   * the first read-less contig AFTER the last contig with columns continues that contig's window coordinates and
//     shows the statistics its unfinished windows held: close() does not reset the ring (depth.d:1070-1076).
struct WindowPrinter {
    sbx_ctx* c;
    const Options& o;
    Out& out;
    const std::vector<std::string>& samples;
    bool have_first = false;     // the first pileup column of the whole run has been seen
    int fref = 0;
    uint64_t fpos = 0;
    int last_cols_ref = -1;      // the last contig with columns so far, the number of windows it printed,
    uint64_t last_nl = 0;
    std::vector<sbx_region_stats> stale_st;      // and what its n unfinished windows hold
    std::vector<uint32_t> stale_cov;
    std::vector<int> pending_empty;              // read-less contigs seen since
    const WindowData* data = nullptr;            // a sharded job: the statistics were collected slice by slice; `c` answers for the header only

    bool has_columns(int r) {
        if (data) return data->has_cols[(size_t)r] != 0;
        uint64_t b0 = 0, e0 = 0;
        check(c, sbx_next_active_range(c, (uint32_t)r, 0, &b0, &e0));
        return b0 != ~0ULL;
    }
    bool first_column_of_run(int r0, int r1) {
        if (!data) return first_column(c, r0, r1, &fref, &fpos);
        for (int r = r0; r < r1; ++r)
            if (data->has_cols[(size_t)r]) { fref = r; fpos = data->firstcol[(size_t)r]; return true; }
        return false;
    }
    void collected_stats(int r, uint64_t k0, uint64_t k1, std::vector<sbx_region_stats>& st, std::vector<uint32_t>& cov) {
        const uint32_t s_n = S();
        const size_t cstride = std::max<size_t>(1, o.thresholds.size());
        const uint64_t nf = data->n_full[(size_t)r];
        const auto& xs = data->extra_st[(size_t)r];
        const auto& xc = data->extra_cov[(size_t)r];
        for (uint64_t k = k0; k < k1; ++k) {
            const sbx_region_stats* ps = nullptr;
            const uint32_t* pc = nullptr;
            if (k < nf) { ps = &data->st[(size_t)(data->base[(size_t)r] + k) * s_n]; pc = &data->cov[(size_t)(data->base[(size_t)r] + k) * s_n * cstride]; }
            else if ((k - nf + 1) * s_n <= xs.size()) { ps = &xs[(size_t)(k - nf) * s_n]; pc = &xc[(size_t)(k - nf) * s_n * cstride]; }
            if (!ps) continue;
            std::copy(ps, ps + s_n, st.begin() + (size_t)(k - k0) * s_n);
            std::copy(pc, pc + s_n * cstride, cov.begin() + (size_t)(k - k0) * s_n * cstride);
        }
    }

    uint32_t S() const { return o.combined ? 1u : (uint32_t)samples.size(); }
    uint64_t step() const { return (uint64_t)o.window - (uint64_t)o.overlap; }
    uint64_t ring() const { return ((uint64_t)o.window + step() - 1) / step(); }

    // statistics of windows [k0, k1) of contig r (st: [k][S], cov: [k][S][n_thr])
    void window_stats(int r, uint64_t k0, uint64_t k1, std::vector<sbx_region_stats>& st, std::vector<uint32_t>& cov) {
        const uint32_t s_n = S();
        const size_t n_thr = o.thresholds.size(), cstride = std::max<size_t>(1, n_thr);
        const uint64_t w = o.window, st_ = step(), n = ring();
        st.assign((size_t)(k1 - k0) * s_n, sbx_region_stats{0, 0});
        cov.assign((size_t)(k1 - k0) * s_n * cstride, 0);
        if (k1 <= k0) return;
        if (data) { collected_stats(r, k0, k1, st, cov); return; }
        const uint64_t len = (uint64_t)std::max<int64_t>(0, sbx_ref_length(c, r));
        if (o.overlap == 0 && k1 * w <= len) {      // full, disjoint windows: the engine's own window statistics
            check(c, sbx_depth_window_stats(c, (uint32_t)r, k0, k1 - k0, st.data(), cov.data()));
            return;
        }
        // the first ring of the run
        const uint64_t anom_from = (r == 0 && fref == 0) ? (fpos < w ? 0 : (fpos - w) / st_ + 1) : n;
        std::vector<sbx_region> reg, creg;
        std::vector<uint32_t> min_start;
        bool any_min = false, extended = false;
        for (uint64_t k = k0; k < k1; ++k) {
            reg.push_back({(uint32_t)r, (uint32_t)(k * st_), (uint32_t)(k * st_ + w)});
            const bool anom = k >= 1 && k >= anom_from && k < n;
            min_start.push_back(anom ? (uint32_t)(k * st_) : 0u);
            any_min |= anom;
            const uint64_t cs = k < n ? k * st_ : (k - n) * st_ + w;
            extended |= cs != k * st_;
            creg.push_back({(uint32_t)r, (uint32_t)cs, (uint32_t)(k * st_ + w)});
        }
        std::vector<uint8_t> seen(reg.size());
        std::vector<uint32_t> cov1(reg.size() * s_n * cstride);
        if (any_min && o.fix_mate)
            throw Fail{"--fix-mate-overlaps with --overlap > 0: the first pileup column lies in the first ring of windows of the first contig "
                       "(the reference counts only reads that start inside those windows, depth.d:1031-1032); not supported on the device path"};
        if (any_min) check(c, sbx_depth_region_stats_from(c, reg.data(), reg.size(), min_start.data(), st.data(), cov1.data(), seen.data()));
        else check(c, sbx_depth_region_stats(c, reg.data(), reg.size(), st.data(), cov1.data(), seen.data()));
        if (extended && n_thr) {
            std::vector<sbx_region_stats> st2(reg.size() * s_n);
            check(c, sbx_depth_region_stats(c, creg.data(), creg.size(), st2.data(), cov1.data(), seen.data()));
        }
        for (size_t i = 0; i < reg.size() * s_n; ++i)
            for (size_t t = 0; t < n_thr; ++t) cov[i * cstride + t] = cov1[i * n_thr + t];
    }
    void rows(const std::string& name, uint64_t start, const sbx_region_stats* st, const uint32_t* cov) {
        const std::string prefix = name + "\t" + std::to_string(start) + "\t" + std::to_string(start + o.window) + "\t";
        static const sbx_region_stats zero{0, 0};
        static const uint32_t zcov[kMaxCliThresholds] = {0};
        const size_t cstride = std::max<size_t>(1, o.thresholds.size());
        for (uint32_t s2 = 0; s2 < S(); ++s2)
            print_region_row(out, o, prefix, (uint32_t)o.window, st ? st[s2] : zero, cov ? cov + s2 * cstride : zcov, samples[s2]);
    }
    void zero_windows(int r) {       // printEmptyWindows (depth.d:1039-1044)
        const uint64_t cnt = (uint64_t)std::max<int64_t>(0, sbx_ref_length(c, r)) / step();
        const std::string name = sbx_ref_name(c, r);
        for (uint64_t k = 0; k < cnt; ++k) rows(name, k * step(), nullptr, nullptr);
    }
    // position of the last pileup column of contig r (it has one)
    uint64_t last_column(int r) {
        if (data) return data->lastcol[(size_t)r];
        uint64_t from = 0, lb = 0, le = 0;
        for (;;) {
            uint64_t b, e;
            check(c, sbx_next_active_range(c, (uint32_t)r, from, &b, &e));
            if (b == ~0ULL) break;
            lb = b; le = e; from = e;
        }
        uint32_t T = 0, Sn = 0;
        check(c, sbx_tile_info(c, &T, &Sn));
        std::vector<uint32_t> cnt;
        std::vector<uint8_t> cov;
        for (uint64_t q = le; q > lb;) {
            const uint64_t p = q > lb + 65536 ? q - 65536 : lb;
            cnt.resize((size_t)(q - p) * Sn * SBX_NCOUNTERS);
            cov.resize((size_t)(q - p));
            check(c, sbx_depth_base_tile(c, (uint32_t)r, (uint32_t)p, (uint32_t)q, nullptr, cov.data()));         // `covered` alone
            for (uint64_t x = q; x > p; --x) if (cov[(size_t)(x - 1 - p)]) return x - 1;
            q = p;
        }
        return 0;
    }
    void contig(int r) {
        const uint64_t len = (uint64_t)std::max<int64_t>(0, sbx_ref_length(c, r)), w = o.window;
        // windows are finished as the columns advance (push) and then up to the contig's length (close / contig change):
        // alignments hanging over the end of the contig can finish windows that end beyond it
        const uint64_t lastcol = last_column(r);
        const uint64_t nw = std::max<uint64_t>(len >= w ? (len - w) / step() + 1 : 0, lastcol >= w ? (lastcol - w) / step() + 1 : 0);
        const std::string name = sbx_ref_name(c, r);
        const size_t cstride = std::max<size_t>(1, o.thresholds.size());
        std::vector<sbx_region_stats> st;
        std::vector<uint32_t> cov;
        const uint64_t CH = 1u << 18;
        for (uint64_t k0 = 0; k0 < nw; k0 += CH) {
            const uint64_t k1 = std::min(nw, k0 + CH);
            window_stats(r, k0, k1, st, cov);
            for (uint64_t k = k0; k < k1; ++k) {
                if (r == fref && k * step() + w <= fpos) continue;       // finished before the first column of the run
                rows(name, k * step(), &st[(size_t)(k - k0) * S()], &cov[(size_t)(k - k0) * S() * cstride]);
            }
        }
        // what the ring still holds when this contig ends
        last_cols_ref = r;
        last_nl = nw;
        window_stats(r, nw, nw + ring(), stale_st, stale_cov);
    }
    void run_refs(int r0, int r1) {
        // --fix-mate-overlaps with overlapping windows: a window is the region [k step, k step + w) of the closed form (reduce.hip) as
        // long as (a) w is a multiple of the step -- otherwise a ring slot also collects per-COLUMN mate terms of the columns in front of
        // its window, which the closed form of a region does not know -- and (b) no window of the run's first ring is printed (window_stats
        // below: is_first_occurrence, depth.d:1031-1032, interacts with the mate status there).  Everything else is refused.
        if (o.overlap > 0 && o.fix_mate && o.window % step() != 0)
            throw Fail{"--fix-mate-overlaps with an --overlap whose step (window - overlap) does not divide the window is not supported on the device path"};
        if (!have_first) {
            if (!first_column_of_run(r0, r1)) return;   // no column yet: windows so far print nothing
            have_first = true;
        }
        for (int r = std::max(r0, fref); r < r1; ++r) {
            if (!has_columns(r)) { pending_empty.push_back(r); continue; }
            for (int e : pending_empty) zero_windows(e);      // read-less contigs between two with columns: push() resets first
            pending_empty.clear();
            contig(r);
        }
    }
    void finish() {
        if (!have_first) return;
        bool first = true;
        const size_t cstride = std::max<size_t>(1, o.thresholds.size());
        for (int e : pending_empty) {
            if (first && last_cols_ref >= 0) {
                const uint64_t cnt = (uint64_t)std::max<int64_t>(0, sbx_ref_length(c, e)) / step();
                const std::string name = sbx_ref_name(c, e);
                for (uint64_t i = 0; i < cnt; ++i) {
                    if (i < ring()) rows(name, (last_nl + i) * step(), &stale_st[(size_t)i * S()], &stale_cov[(size_t)i * S() * cstride]);
                    else rows(name, (last_nl + i) * step(), nullptr, nullptr);
                }
            } else zero_windows(e);
            first = false;
        }
    }
};

// PerBedRegionPrinter (depth.d:879-931): statistics are gathered batch by batch, rows are printed at the end
// in input order -- and not at all unless some column fell inside some region (the samples array is created
// lazily)
struct RegionPrinter {
    sbx_ctx* c;
    const Options& o;
    Out& out;
    const std::vector<std::string>& samples;
    const std::vector<sbx_region>& raw;
    const std::vector<std::string>& lines;
    std::vector<sbx_region_stats> st;
    std::vector<uint32_t> cov;
    std::vector<uint8_t> seen;

    void prepare() {
        const uint32_t S = o.combined ? 1u : (uint32_t)samples.size();
        const size_t n_thr = std::max<size_t>(1, o.thresholds.size());
        if (st.empty()) { st.assign(raw.size() * S, sbx_region_stats{0, 0}); cov.assign(raw.size() * S * n_thr, 0); seen.assign(raw.size(), 0); }
    }
    void run_refs(int r0, int r1) {
        prepare();
        std::vector<size_t> ids;
        for (size_t i = 0; i < raw.size(); ++i)
            if ((int)raw[i].ref_id >= r0 && (int)raw[i].ref_id < r1) ids.push_back(i);
        collect(c, ids);
    }
    // statistics of the raw regions `ids` from the run resident in context cx (a sharded job: the device that owns them; the rows
    // of different devices are disjoint, prepare() has been called before the threads started)
    void collect(sbx_ctx* cx, const std::vector<size_t>& ids) {
        const uint32_t S = o.combined ? 1u : (uint32_t)samples.size();
        const size_t n_thr = std::max<size_t>(1, o.thresholds.size());
        std::vector<sbx_region> sub;
        for (size_t i : ids) sub.push_back(raw[i]);
        if (sub.empty()) return;
        std::vector<sbx_region_stats> st2(sub.size() * S);
        std::vector<uint32_t> cov2(sub.size() * S * n_thr);
        std::vector<uint8_t> seen2(sub.size());
        check(cx, sbx_depth_region_stats(cx, sub.data(), sub.size(), st2.data(), cov2.data(), seen2.data()));
        const size_t nt = o.thresholds.size();
        for (size_t j = 0; j < ids.size(); ++j) {
            seen[ids[j]] = seen2[j];
            for (uint32_t s2 = 0; s2 < S; ++s2) {
                st[ids[j] * S + s2] = st2[j * S + s2];
                for (size_t t = 0; t < nt; ++t) cov[(ids[j] * S + s2) * n_thr + t] = cov2[(j * S + s2) * nt + t];
            }
        }
    }
    void finish() {
        const uint32_t S = o.combined ? 1u : (uint32_t)samples.size();
        const size_t n_thr = std::max<size_t>(1, o.thresholds.size());
        bool any = false;
        for (auto v : seen) any |= v != 0;
        if (!any) return;
        for (size_t id = 0; id < raw.size(); ++id) {
            std::string l = lines[id];
            while (!l.empty() && isspace((unsigned char)l.back())) l.pop_back();   // stripRight (depth.d:904)
            l += "\t";
            for (uint32_t s2 = 0; s2 < S; ++s2)
                print_region_row(out, o, l, raw[id].end - raw[id].start, st[id * S + s2], &cov[(id * S + s2) * n_thr], samples[s2]);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Several devices (`--gpus N`, SBX_DEVICES=0,1,...): ONE process, one context per device, each driven by its own thread.
// The job shards by POSITION (sbx_plan_shards): outputs of disjoint position ranges are disjoint, so nothing travels between
// the devices -- every context runs its slices (sbx_run_interval: only the BGZF blocks the BAI lists for them are uploaded
// and inflated) and hands over its share:
//   base    the text of its positions, formatted on the device and streamed piece by piece (sbx_stream_base_rows) -- with -o into its
//           own byte range of the file (pwrite at the offset the measured sizes of the slices before it add up to; the devices
//           write side by side), without -o in genome order through the one output stream, slices dealt round-robin so that
//           device k + 1 computes while device k prints;
//   region  the statistics of the BED regions whose first position it owns (a region is never split);
//   window  the statistics of the windows of its slices (cuts are multiples of the window size), the first / last columns,
//           and behind a contig's end the windows that alignments hanging over it finish or leave unfinished
// -- and the printers above print from what was collected, with the reference's rules.  Option sets whose output depends on
// the order of the whole stream (`window --overlap`, `base -L`, `base -c 0`, host formatting) run on one device, as before.
// The torch.distributed driver (python -m sambamba_amd.dist_depth) keeps the RCCL all-reduce form of the north star.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kBaiEnd = 1u << 29;       // the coordinate limit of the BAI's binning scheme
struct Sharded {
    const Options& o;
    Out& out;
    const std::vector<const char*>& paths;
    const sbx_filter& filt;
    int mode_id;
    std::vector<int> devices;
    sbx_ctx* ctx0;                               // the context opened by depth_main (on devices[0])
    const std::vector<std::string>& samples;
    std::vector<sbx_region> merged;              // -L (region mode)
    std::vector<sbx_ctx*> cx;
    std::mutex mu;
    std::condition_variable cv;
    std::string failure;
    std::vector<double> busy_run, busy_out;

    static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
    void fail(const std::string& m) {
        std::lock_guard<std::mutex> g(mu);
        if (failure.empty()) failure = m.empty() ? std::string("a device of the sharded run failed") : m;
        cv.notify_all();
    }
    template <class P> bool wait_for(P&& pred) {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return !failure.empty() || pred(); });
        return failure.empty();
    }
    // context of device k: depth_main's for k == 0, opened here (on the worker's thread, next to the others) otherwise
    sbx_ctx* context(size_t k) {
        if (k == 0) return ctx0;
        char e[512] = {0};
        sbx_ctx* c = sbx_open(paths.data(), (int)paths.size(), devices[k], e, sizeof e);
        if (!c) throw Fail{e};
        { std::lock_guard<std::mutex> g(mu); cx[k] = c; }
        check(c, sbx_set_filter(c, &filt));
        check(c, sbx_set_params(c, mode_id, (uint8_t)o.min_bq, o.fix_mate, o.combined, (uint32_t)o.window, (uint32_t)o.overlap,
                                o.thresholds.data(), (int)o.thresholds.size()));
        if (!merged.empty()) check(c, sbx_set_regions(c, merged.data(), merged.size()));
        return c;
    }
    // run every worker, join them all, rethrow the first failure
    template <class W> void run_workers(W&& work) {
        cx.assign(devices.size(), nullptr);
        cx[0] = ctx0;
        busy_run.assign(devices.size(), 0);
        busy_out.assign(devices.size(), 0);
        std::vector<std::thread> th;
        struct Join { std::vector<std::thread>& t; ~Join() { for (auto& x : t) if (x.joinable()) x.join(); } } join{th};
        for (size_t k = 0; k < devices.size(); ++k)
            th.emplace_back([&, k] {
                try { work(k, context(k)); }
                catch (const Fail& f) { fail(f.msg); }
                catch (const std::exception& e) { fail(e.what()); }
            });
        for (auto& x : th) x.join();
        if (!failure.empty()) throw Fail{failure};
    }
    std::vector<sbx_shard> plan(uint32_t align) {
        sbx_header_info hi;
        check(ctx0, sbx_header(ctx0, &hi));
        std::vector<int64_t> lens((size_t)hi.n_ref);
        for (int r = 0; r < hi.n_ref; ++r) lens[(size_t)r] = sbx_ref_length(ctx0, r);
        size_t n = 0;
        std::vector<sbx_shard> sh((size_t)hi.n_ref + devices.size() + 1);
        if (sbx_plan_shards(lens.data(), hi.n_ref, (int32_t)devices.size(), align, sh.data(), sh.size(), &n) != SBX_OK) throw Fail{"internal: shard plan"};
        sh.resize(n);
        return sh;
    }
    // sbx_run_interval over [beg - slack, end (+ slack)) with the slack --fix-mate-overlaps needs in region / window mode: a read that lies
    // past the overlap with its mate is counted differently from an unpaired one (status `past`, depth.d:717-845), so the mate must be in
    // the run even when it ends before the slice.  The slack starts at one linear-index window and is raised to the longest alignment
    // the run reports -- never silently too small.
    void run_with_mate_slack(sbx_ctx* c, uint32_t ref, uint64_t beg, uint64_t end, bool both_sides) {
        if (!o.fix_mate) { check(c, sbx_run_interval(c, ref, (uint32_t)beg, (uint32_t)end)); return; }
        uint64_t slack = 16384;
        for (int attempt = 0; attempt < 4; ++attempt) {
            const uint64_t lo = beg > slack ? beg - slack : 0, hi = both_sides ? std::min<uint64_t>(end + slack, 0x7FFFFFFFull) : end;
            check(c, sbx_run_interval(c, ref, (uint32_t)lo, (uint32_t)hi));
            sbx_run_stats st;
            check(c, sbx_last_run_stats(c, &st));
            if (st.max_alignment_span <= slack) return;
            slack = (st.max_alignment_span + 16383) / 16384 * 16384;
        }
        throw Fail{"--fix-mate-overlaps: the alignments of a slice span more than " + std::to_string(slack) + " positions; run on one device"};
    }

    // ---- base ----
    struct Slice { uint32_t ref; uint64_t beg, end, print_end; size_t owner; };
    // bytes of the text of a slice (the device's measuring pass; nothing is copied)
    uint64_t measure(sbx_ctx* c, const Slice& sl) {
        uint64_t total = 0, from = sl.beg;
        for (;;) {
            uint64_t b, e;
            check(c, sbx_next_active_range(c, sl.ref, from, &b, &e));
            if (b == ~0ULL || b >= sl.print_end) break;
            b = std::max(b, from);
            e = std::min(e, sl.print_end);
            size_t need = 0;
            const int rc = sbx_format_base_rows(c, sl.ref, (uint32_t)b, (uint32_t)e, o.min_cov, o.max_cov, o.annotate ? 1 : 0, nullptr, 0, &need);
            if (rc != SBX_OK && rc != SBX_ENOMEM) check(c, rc);
            total += need;
            from = e;
        }
        return total;
    }
    struct Sink { int fd; uint64_t off; FILE* fp; };
    static int sink_write(void* u, const char* d, size_t n) {
        Sink* k = (Sink*)u;
        if (k->fp) return fwrite(d, 1, n, k->fp) == n ? 0 : 1;
        while (n) {
            const ssize_t w = pwrite(k->fd, d, n, (off_t)k->off);
            if (w < 0) { if (errno == EINTR) continue; return 1; }
            d += w; n -= (size_t)w; k->off += (uint64_t)w;
        }
        return 0;
    }
    void stream(sbx_ctx* c, const Slice& sl, Sink* sink) {
        uint64_t from = sl.beg;
        for (;;) {
            uint64_t b, e;
            check(c, sbx_next_active_range(c, sl.ref, from, &b, &e));
            if (b == ~0ULL || b >= sl.print_end) break;
            b = std::max(b, from);
            e = std::min(e, sl.print_end);
            check(c, sbx_stream_base_rows(c, sl.ref, (uint32_t)b, (uint32_t)e, o.min_cov, o.max_cov, o.annotate ? 1 : 0, sink_write, sink));
            from = e;
        }
    }
    void base() {
        const size_t N = devices.size();
        const bool to_file = out.fp != stdout;
        sbx_header_info hi;
        check(ctx0, sbx_header(ctx0, &hi));
        uint64_t total = 0;
        for (int r = 0; r < hi.n_ref; ++r) total += (uint64_t)std::max<int64_t>(0, sbx_ref_length(ctx0, r));
        // slices: a device's share cut so that every slice still fills a device once (the lane-per-block Huffman kernel takes one
        // residency however few blocks it gets) and the buffers hold a fraction of the share
        uint64_t want = std::max<uint64_t>(total / (4 * N), 16u << 20);
        if (const char* e = getenv("SBX_SLICE_POSITIONS")) want = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));
        want = (want + 1023) / 1024 * 1024;
        std::vector<Slice> sl;
        for (const sbx_shard& sh : plan(1024)) {
            const uint64_t len = (uint64_t)sbx_ref_length(ctx0, (int)sh.ref_id);
            const uint64_t n = ((uint64_t)(sh.end - sh.beg) + want - 1) / want, step = (((uint64_t)(sh.end - sh.beg) + n - 1) / n + 1023) / 1024 * 1024;
            for (uint64_t b = sh.beg; b < sh.end; b += step) {
                const uint64_t e = std::min<uint64_t>(sh.end, b + step);
                sl.push_back({sh.ref_id, b, e, e == len ? 0xFFFFFFFFull : e, (size_t)sh.shard});     // (columns of alignments hanging over the contig end)
            }
        }
        // one output stream: deal the slices round-robin, so that the devices compute next to the one that prints
        if (!to_file) for (size_t g = 0; g < sl.size(); ++g) sl[g].owner = g % N;
        std::vector<uint64_t> size(sl.size(), 0);
        std::vector<char> measured(sl.size(), 0), written(sl.size(), 0);
        out.flush();
        fflush(out.fp);
        const uint64_t head = to_file ? (uint64_t)ftello(out.fp) : 0;
        const int fd = to_file ? fileno(out.fp) : -1;
        run_workers([&](size_t k, sbx_ctx* c) {
            for (size_t g = 0; g < sl.size(); ++g) {
                if (sl[g].owner != k) continue;
                double t0 = now();
                // (the last slice of a contig also takes the reads that START behind the contig's end, up to the index's coordinate limit)
                check(c, sbx_run_interval(c, sl[g].ref, (uint32_t)sl[g].beg, sl[g].print_end == 0xFFFFFFFFull ? kBaiEnd : (uint32_t)sl[g].end));
                busy_run[k] += now() - t0;
                Sink sink{fd, 0, to_file ? nullptr : out.fp};
                if (to_file) {
                    const uint64_t sz = measure(c, sl[g]);
                    { std::lock_guard<std::mutex> lk(mu); size[g] = sz; measured[g] = 1; cv.notify_all(); }
                    if (!wait_for([&] { for (size_t i = 0; i < g; ++i) if (!measured[i]) return false; return true; })) return;
                    sink.off = head;
                    for (size_t i = 0; i < g; ++i) sink.off += size[i];
                    t0 = now();
                    const uint64_t at = sink.off;
                    stream(c, sl[g], &sink);
                    if (sink.off - at != sz) throw Fail{"internal: measured " + std::to_string(sz) + " bytes of text, wrote " + std::to_string(sink.off - at)};
                } else {
                    if (!wait_for([&] { for (size_t i = 0; i < g; ++i) if (!written[i]) return false; return true; })) return;
                    t0 = now();
                    stream(c, sl[g], &sink);
                    fflush(out.fp);
                }
                busy_out[k] += now() - t0;
                std::lock_guard<std::mutex> lk(mu);
                written[g] = 1;
                cv.notify_all();
            }
        });
        if (to_file) {
            uint64_t all = head;
            for (uint64_t x : size) all += x;
            if (fseeko(out.fp, (off_t)all, SEEK_SET) != 0) throw Fail{"cannot seek in the output file"};
        }
    }

    // ---- region ----
    void region(RegionPrinter& rp) {
        const std::vector<sbx_shard> sh = plan(1024);
        sbx_header_info hi;
        check(ctx0, sbx_header(ctx0, &hi));
        // a region belongs to the device that owns its first position (regions starting at or beyond the end of their contig: the owner
        // of the contig's last position; regions of zero-length contigs: device 0 -- the one-device CLI prints a row for them as well)
        auto owner = [&](const sbx_region& g) -> size_t {
            const int64_t len = sbx_ref_length(ctx0, (int)g.ref_id);
            if (len <= 0) return 0;
            const uint64_t p = std::min<uint64_t>(g.start, (uint64_t)len - 1);
            for (const sbx_shard& x : sh)
                if (x.ref_id == g.ref_id && x.beg <= p && p < x.end) return x.shard;
            return 0;
        };
        std::vector<std::vector<size_t>> ids(devices.size());
        for (size_t i = 0; i < rp.raw.size(); ++i) ids[owner(rp.raw[i])].push_back(i);
        rp.prepare();
        run_workers([&](size_t k, sbx_ctx* c) {
            // reads are selected against ALL merged regions (a mate that reaches the pileup through a neighbour's region must still pair,
            // depth.d:717-758), but fetched only for the hull of the owned regions of a contig, widened by the mate slack on each side
            for (int r = 0; r < hi.n_ref; ++r) {
                std::vector<size_t> mine;
                uint64_t lo = ~0ULL, hi_ = 0;
                for (size_t i : ids[k])
                    if ((int)rp.raw[i].ref_id == r) { mine.push_back(i); lo = std::min<uint64_t>(lo, rp.raw[i].start); hi_ = std::max<uint64_t>(hi_, rp.raw[i].end); }
                if (mine.empty()) continue;
                if (hi_ <= lo) hi_ = lo + 1;
                double t0 = now();
                run_with_mate_slack(c, (uint32_t)r, lo, std::min<uint64_t>(hi_, 0x7FFFFFFFull), true);
                busy_run[k] += now() - t0;
                t0 = now();
                rp.collect(c, mine);
                busy_out[k] += now() - t0;
            }
        });
    }

    // ---- window (--overlap 0) ----
    void window(WindowPrinter& wp, WindowData& wd) {
        const uint64_t w = o.window;
        const std::vector<sbx_shard> sh = plan((uint32_t)w);
        sbx_header_info hi;
        check(ctx0, sbx_header(ctx0, &hi));
        const size_t n_ref = (size_t)hi.n_ref, S = wp.S(), cstride = std::max<size_t>(1, o.thresholds.size());
        wd.base.assign(n_ref + 1, 0); wd.n_full.assign(n_ref, 0);
        wd.extra_st.assign(n_ref, {}); wd.extra_cov.assign(n_ref, {});
        wd.has_cols.assign(n_ref, 0); wd.firstcol.assign(n_ref, ~0ULL); wd.lastcol.assign(n_ref, 0);
        uint64_t total = 0;
        for (size_t r = 0; r < n_ref; ++r) {
            wd.base[r] = total;
            wd.n_full[r] = (uint64_t)std::max<int64_t>(0, sbx_ref_length(ctx0, (int)r)) / w;
            total += wd.n_full[r];
        }
        wd.base[n_ref] = total;
        wd.st.assign((size_t)total * S, sbx_region_stats{0, 0});
        wd.cov.assign((size_t)total * S * cstride, 0);
        run_workers([&](size_t k, sbx_ctx* c) {
            WindowPrinter local{c, o, out, samples, false, 0, 0, -1, 0, {}, {}, {}};       // its window_stats() on this device's run
            for (const sbx_shard& x : sh) {
                if (x.shard != k) continue;
                const uint32_t r = x.ref_id;
                const uint64_t len = (uint64_t)sbx_ref_length(c, (int)r);
                double t0 = now();
                const bool last = x.end >= len;
                run_with_mate_slack(c, r, x.beg, last ? kBaiEnd : x.end, false);
                busy_run[k] += now() - t0;
                t0 = now();
                uint64_t fc = 0, lc = 0;
                const bool any = first_column_in(c, r, x.beg, last ? 0xFFFFFFFFull : x.end, &fc);
                if (any && last) last_column_from(c, r, x.beg, &lc);
                const uint64_t k0 = x.beg / w, k1 = last ? len / w : x.end / w;            // only full windows are printed
                const uint64_t CH = 1u << 18;
                std::vector<sbx_region_stats> st;
                std::vector<uint32_t> cv2;
                for (uint64_t a = k0; a < k1; a += CH) {
                    const uint64_t b = std::min(k1, a + CH);
                    st.assign((size_t)(b - a) * S, sbx_region_stats{0, 0});
                    cv2.assign((size_t)(b - a) * S * cstride, 0);
                    check(c, sbx_depth_window_stats(c, r, a, b - a, st.data(), cv2.data()));
                    std::copy(st.begin(), st.end(), wd.st.begin() + (size_t)(wd.base[r] + a) * S);
                    // (sbx_depth_window_stats packs the thresholds n_thr wide; the printer's rows are max(1, n_thr) wide)
                    const size_t nt = o.thresholds.size();
                    for (size_t i = 0; i < (size_t)(b - a) * S; ++i)
                        for (size_t t = 0; t < nt; ++t) wd.cov[((size_t)(wd.base[r] + a) * S + i) * cstride + t] = cv2[i * nt + t];
                }
                std::vector<sbx_region_stats> xs;
                std::vector<uint32_t> xc;
                if (any && last) {
                    // behind the contig's end: the windows that alignments hanging over it finish (depth.d:1057-1071 sees columns, not
                    // lengths), and the one the ring still holds when the contig ends
                    const uint64_t nw = std::max<uint64_t>(len >= w ? (len - w) / w + 1 : 0, lc >= w ? (lc - w) / w + 1 : 0);
                    const uint64_t nf = len / w;
                    local.window_stats((int)r, nf, nw + 1, xs, xc);
                }
                busy_out[k] += now() - t0;
                std::lock_guard<std::mutex> lk(mu);
                if (any) {
                    wd.has_cols[r] = 1;
                    wd.firstcol[r] = std::min(wd.firstcol[r], fc);
                    if (last) { wd.lastcol[r] = lc; wd.extra_st[r] = std::move(xs); wd.extra_cov[r] = std::move(xc); }
                }
            }
        });
        // a contig whose LAST slice holds no column but an earlier one does: its last column lies in an earlier slice, within the contig --
        // every window that can be finished is a full one; the printer asks for the last column only to find windows behind the end
        for (size_t r = 0; r < n_ref; ++r)
            if (wd.has_cols[r] && wd.extra_st[r].empty()) {
                wd.lastcol[r] = 0;
                wd.extra_st[r].assign(S, sbx_region_stats{0, 0});
                wd.extra_cov[r].assign(S * cstride, 0);
            }
    }
    void close_others() {
        for (size_t k = 1; k < cx.size(); ++k) if (cx[k]) sbx_close(cx[k]);
    }
    void report(double t_start) {
        std::string a;
        for (size_t k = 0; k < devices.size(); ++k) {
            char b[96];
            snprintf(b, sizeof b, " [device %d: run %.3f s, output %.3f s]", devices[k], busy_run[k], busy_out[k]);
            a += b;
        }
        fprintf(stderr, "[sbx-depth] sharded over %zu contexts:%s, total %.3f s since main\n", devices.size(), a.c_str(), now() - t_start);
    }
};

// (detached mode, see main) tells the waiting parent that the output is complete: everything buffered is written, the
// standard descriptors are closed -- a consumer on a pipe sees end-of-file now, not after the teardown -- and the status sent
int g_done_fd = -1;
void report_done(int rc) {
    if (g_done_fd < 0) return;
    fflush(nullptr);
    prctl(PR_SET_PDEATHSIG, 0);
    close(0); close(1); close(2);
    const unsigned char st = (unsigned char)rc;
    ssize_t n;
    do n = write(g_done_fd, &st, 1); while (n < 0 && errno == EINTR);
    close(g_done_fd);
    g_done_fd = -1;
}

int depth_main(int argc, char** argv) {
    if (argc < 3) { usage(); return 0; }
    std::string mode = argv[1];
    if (mode != "base" && mode != "region" && mode != "window") { usage(); return 0; }
    Options o;
    std::string perr;
    if (!parse_args(argc, argv, &o, &perr)) { fprintf(stderr, "sambamba-depth: %s\n", perr.c_str()); return 1; }
    if (o.mode == "region" && !o.has_regions) {
        fputs("BED file or a region must be provided in region mode\n", stderr);
        return 1;
    }
    Out out;
    sbx_ctx* ctx = nullptr;
    try {
        if (!o.output_fn.empty()) {
            out.fp = fopen(o.output_fn.c_str(), "w+");
            if (!out.fp) throw Fail{"Cannot open file `" + o.output_fn + "' in mode `w+' (No such file or directory)"};
        }
        if (o.mode == "base") {  // PerBasePrinter.init (depth.d:412-427)
            if (o.report_zero) o.min_cov = 0;
        }
        if (o.mode == "window") {
            if (!(o.window > 0)) throw Fail{"positive window size must be specified"};
            if (!(o.overlap < o.window)) throw Fail{"specified overlap is larger than window size"};
        }
        // The header line is printed by printer.init() before the BAM is opened (depth.d:1152),
        // except in region mode where it needs the first BED line (setBed, depth.d:912-923).
        std::vector<std::string> dummy_samples;
        sbx_filter filt;
        char ebuf[512] = {0};
        int rc = sbx_compile_filter(o.has_filter ? o.filter.c_str() : nullptr, &filt, ebuf, sizeof ebuf);
        if (rc != SBX_OK) throw Fail{ebuf};
        if (o.bams.empty()) throw Fail{"no input files"};
        std::vector<const char*> paths;
        for (auto& b : o.bams) paths.push_back(b.c_str());
        // base/window print their header before opening the file
        if (o.mode == "base") {
            std::string h = "REF\tPOS\tCOV\tA\tC\tG\tT\tDEL\tREFSKIP";
            if (!o.combined) h += "\tSAMPLE";
            if (o.annotate) h += "\tFLAG";
            h += "\n";
            out.put(h);
        } else if (o.mode == "window") {
            print_bed_header(out, o, 3);   // PerWindowPrinter.init (depth.d:1036)
        }
        const bool timing = getenv("SBX_TIMING") != nullptr;      // phase wall clock on stderr (profiles/, tools/cli_e2e.sh)
        auto now = [] { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
        const double t_start = now();
        // --gpus N / SBX_DEVICES=a,b,...: the devices of a sharded job (an ordinal may repeat: several contexts on one device -- how the
        // sharded path is tested on a one-GPU box)
        std::vector<int> devices;
        if (const char* e = getenv("SBX_DEVICES")) {
            for (const char* q = e; *q;) {
                char* end = nullptr;
                const long v = strtol(q, &end, 10);
                if (end == q || v < 0) throw Fail{std::string("SBX_DEVICES: a comma-separated list of device ordinals is expected, got '") + e + "'"};
                devices.push_back((int)v);
                q = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') throw Fail{std::string("SBX_DEVICES: a comma-separated list of device ordinals is expected, got '") + e + "'"};
            }
            if (o.gpus > 0 && (size_t)o.gpus < devices.size()) devices.resize((size_t)o.gpus);
        } else if (o.gpus > 1) {
            const int have = sbx_device_count();
            if (o.gpus > have) throw Fail{"--gpus " + std::to_string(o.gpus) + ": " + std::to_string(have) + " HIP device(s) visible"};
            for (int k = 0; k < o.gpus; ++k) devices.push_back(k);
        }
        ctx = sbx_open(paths.data(), (int)paths.size(), devices.empty() ? -1 : devices[0], ebuf, sizeof ebuf);
        if (!ctx) throw Fail{ebuf};
        const double t_open = now();
        double t_run = 0, t_print = 0;
        sbx_header_info hi;
        check(ctx, sbx_header(ctx, &hi));
        if (!hi.sorted_by_coordinate) throw Fail{"All files must be coordinate-sorted"};
        if (!hi.has_index) throw Fail{"All files must be indexed"};
        std::vector<std::string> samples;
        for (int s = 0; s < hi.n_samples; ++s) samples.push_back(sbx_sample_name(ctx, s));
        check(ctx, sbx_set_filter(ctx, &filt));
        int mode_id = o.mode == "base" ? SBX_MODE_BASE : o.mode == "region" ? SBX_MODE_REGION : SBX_MODE_WINDOW;
        check(ctx, sbx_set_params(ctx, mode_id, (uint8_t)o.min_bq, o.fix_mate, o.combined, (uint32_t)o.window,
                                  (uint32_t)o.overlap, o.thresholds.data(), (int)o.thresholds.size()));

        // -L (depth.d:1184-1212)
        std::vector<sbx_region> merged, raw;
        std::vector<std::string> raw_lines;
        if (o.has_regions) {
            // host-side header view for BED contig lookups
            BamHeaderInfo hv;
            for (int r = 0; r < hi.n_ref; ++r) hv.refs.push_back({sbx_ref_name(ctx, r), (int32_t)sbx_ref_length(ctx, r)});
            std::vector<BedInterval> ivs;
            std::vector<std::string> lines;
            std::vector<size_t> line_of;
            if (read_bed_file(o.regions, &ivs, &lines, &line_of)) {
                merged = bed_merged(ivs, hv);
                // raw list in file order; every kept region keeps its own input line (the reference
                // pairs them by index, which misaligns when a line is dropped -- SURVEY App. B-6)
                for (size_t i = 0; i < ivs.size(); ++i) {
                    int id = hv.find_ref(ivs[i].chr);
                    if (id < 0) continue;
                    raw.push_back({(uint32_t)id, (uint32_t)ivs[i].beg, (uint32_t)ivs[i].end});
                    raw_lines.push_back(lines[line_of[i]]);
                }
                if (o.mode == "region" && lines.empty()) throw Fail{"Attempting to fetch the front of an empty array of string"};
            } else {
                RegionString rs = parse_region_string(o.regions);
                int id = sbx_ref_id(ctx, rs.reference.c_str());
                if (id < 0) throw Fail{"couldn't open file " + o.regions + " or find reference " + rs.reference};
                sbx_region g{(uint32_t)id, rs.beg, rs.end};
                if (g.end == 0xFFFFFFFFu) g.end = (uint32_t)sbx_ref_length(ctx, id);
                merged.push_back(g);
                raw.push_back(g);
                raw_lines = {rs.reference + "\t" + std::to_string(g.start) + "\t" + std::to_string(g.end)};
            }
            if (merged.empty()) throw Fail{"Enforcement failed"};
            check(ctx, sbx_set_regions(ctx, merged.data(), merged.size()));
        }
        if (o.mode == "region") print_bed_header(out, o, split_ws(raw_lines.empty() ? std::string("a b c") : raw_lines[0]).size());
        // The device processes the file in batches of contigs sized to its free memory (one batch unless the
        // BAM is whole-genome sized); the printers are fed batch by batch, in contig order.
        uint64_t budget = 0;
        if (const char* e = getenv("SBX_BATCH_BYTES")) budget = strtoull(e, nullptr, 10);
        size_t n_batches = 0;
        check(ctx, sbx_plan_batches(ctx, budget, nullptr, 0, &n_batches));
        std::vector<sbx_batch> plan(n_batches);
        if (n_batches) check(ctx, sbx_plan_batches(ctx, budget, plan.data(), plan.size(), &n_batches));
        BasePrinter bp(ctx, o, out, samples);
        if (o.mode == "base" && o.has_regions) bp.set_bed(merged);
        // ---- several devices: the job sharded by position, one context per device (struct Sharded) ----
        if (devices.size() > 1) {
            bool can = false;
            const uint32_t S_eff = o.combined ? 1u : (uint32_t)samples.size();
            if (o.mode == "base") can = !o.has_regions && o.min_cov > 0 && bp.device_format_applies();
            else if (o.mode == "region") can = true;
            else {
                uint64_t total_win = 0;
                for (int r = 0; r < hi.n_ref; ++r) total_win += (uint64_t)std::max<int64_t>(0, sbx_ref_length(ctx, r)) / o.window;
                can = o.overlap == 0 && total_win * S_eff * (2 + std::max<size_t>(1, o.thresholds.size())) <= (1ull << 28);
            }
            if (!can) {
                fprintf(stderr, "[sbx-depth] --gpus: the output of this option set depends on the order of the whole stream (base -L, base -c 0, "
                                "window --overlap) or would not fit the host: running on one device\n");
            } else {
                Sharded sh{o, out, paths, filt, mode_id, devices, ctx, samples, o.mode == "region" ? merged : std::vector<sbx_region>{}, {}, {}, {}, {}, {}, {}};
                if (o.mode == "base") sh.base();
                else if (o.mode == "region") {
                    RegionPrinter rp{ctx, o, out, samples, raw, raw_lines, {}, {}, {}};
                    sh.region(rp);
                    rp.finish();
                } else {
                    WindowData wd;
                    WindowPrinter wp{ctx, o, out, samples, false, 0, 0, -1, 0, {}, {}, {}};
                    sh.window(wp, wd);
                    wp.data = &wd;
                    wp.run_refs(0, hi.n_ref);
                    wp.finish();
                }
                out.flush();
                if (out.fp != stdout) fclose(out.fp);
                if (timing) sh.report(t_start);
                if (!getenv("SBX_ORDERLY_EXIT")) { fflush(nullptr); report_done(0); _exit(0); }
                sh.close_others();
                sbx_close(ctx);
                return 0;
            }
        }
        // ---- `depth base` without -L and with -c > 0: the text is a pure function of the position, so the genome is cut into
        // slices that flow through three overlapping stages -- file -> device (sbx_prefetch_interval), the kernels
        // (sbx_run_interval), device -> text (sbx_stream_base_rows) -- on two contexts that alternate.  PCIe is full duplex:
        // the upload of slice k + 1 and the text of slice k - 1 travel while slice k is computed.
        if (o.mode == "base" && !o.has_regions && o.min_cov > 0 && bp.device_format_applies() && paths.size() == 1 &&
            !getenv("SBX_NO_PIPELINE") && (hi.compressed_bytes >= (256u << 20) || getenv("SBX_FORCE_PIPELINE"))) {
            // Contexts: TWO in the detached child (SBX_DETACH=1: upload, kernels and text of three different slices overlap; the exit of two
            // contexts is the child's business), ONE in the default one-process form (round 5): two contexts cost more at exit than their
            // overlap saves (0.86 s against 0.75 s for config 2 in round 3), but slices through ONE context still pay: the upload of slice
            // k + 1 travels while the text of slice k leaves -- the two PCIe directions -- and the buffers hold a quarter of the job, so
            // the process has a quarter of the device memory to give back when it ends (config 2: 0.66 -> see profiles/round5).
            // The slices are cut by positions, not by the planner's byte budget: when one does not fit (SBX_ENOMEM before any text was
            // written) the run falls back to the one-pass form, which goes through sbx_plan_batches.
            size_t n_ctx = g_done_fd >= 0 ? 2 : 1;
            if (const char* e = getenv("SBX_PIPELINE_CONTEXTS")) n_ctx = atoi(e) == 2 ? 2 : 1;
            struct Slice { uint32_t ref; uint64_t beg, end, print_end; };
            std::vector<Slice> sl;
            {
                uint64_t total = 0;
                for (int r = 0; r < hi.n_ref; ++r) total += (uint64_t)std::max<int64_t>(0, sbx_ref_length(ctx, r));
                // four slices of a chromosome-sized job: each slice still fills the device once (the lane-per-block Huffman kernel takes
                // one residency, ~16 ms, however few blocks it gets), and the text of the whole job -- what the pipeline is
                // bound by -- starts to flow after a quarter of the upload
                uint64_t want = std::max<uint64_t>(total / 4, 16u << 20);
                if (const char* e = getenv("SBX_SLICE_POSITIONS")) want = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));
                want = (want + 1023) / 1024 * 1024;
                for (int r = 0; r < hi.n_ref; ++r) {
                    const uint64_t len = (uint64_t)std::max<int64_t>(0, sbx_ref_length(ctx, r));
                    if (!len) continue;
                    const uint64_t n = (len + want - 1) / want, step = ((len + n - 1) / n + 1023) / 1024 * 1024;
                    for (uint64_t b = 0; b < len; b += step) {
                        const uint64_t e = std::min(len, b + step);
                        sl.push_back({(uint32_t)r, b, e, e == len ? 0xFFFFFFFFull : e});      // columns of alignments hanging over the contig end
                    }
                }
            }
            sbx_ctx* cx[2] = {ctx, nullptr};
            std::mutex mu;
            std::condition_variable cv;
            std::vector<int> uploaded(sl.size(), 0), computed(sl.size(), 0), printed(sl.size(), 0);
            std::string failure;
            int failure_code = SBX_OK;
            bool opened2 = false;
            double busy_up = 0, busy_run = 0, busy_print = 0, t_open2 = 0;       // seconds every stage was working (SBX_TIMING)
            std::vector<double> done_at(sl.size(), 0);
            auto fail = [&](const std::string& m, int code = SBX_EINVAL) {
                std::lock_guard<std::mutex> g(mu);
                if (failure.empty()) { failure = m.empty() ? std::string("pipeline stage failed") : m; failure_code = code; }
                cv.notify_all();
            };
            auto wait_for = [&](auto&& pred) { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return !failure.empty() || pred(); }); return failure.empty(); };
            auto mark = [&](std::vector<int>& v, size_t k) { std::lock_guard<std::mutex> g(mu); v[k] = 1; cv.notify_all(); };
            // (the stage threads start inside the scope of the guard that joins them: an exception while the second or third one is
            //  being created -- thread exhaustion -- must not destroy a running std::thread)
            std::thread opener, uploader, computer;
            // whatever leaves this scope -- an exception of any kind included -- first releases the stage threads, then joins them
            struct JoinAll {
                std::thread &a, &b, &c;
                decltype(fail)& stop;
                bool regular = false;
                ~JoinAll() {
                    if (!regular) stop("pipeline aborted", SBX_EINVAL);
                    if (a.joinable()) a.join();
                    if (b.joinable()) b.join();
                    if (c.joinable()) c.join();
                }
            };
            const double t0 = now();
            size_t n_printed = 0;
            bool all_printed = false;
            {
            JoinAll guard{opener, uploader, computer, fail};
            opener = std::thread([&] {       // the second context opens while the first slice is on its way
                char e2[512] = {0};
                const double to = now();
                const bool want2 = n_ctx == 2 && sl.size() > 1;
                sbx_ctx* c2 = want2 ? sbx_open(paths.data(), (int)paths.size(), -1, e2, sizeof e2) : nullptr;
                t_open2 = now() - to;
                if (want2 && !c2) { fail(e2); return; }
                if (c2 && (sbx_set_filter(c2, &filt) != SBX_OK ||
                           sbx_set_params(c2, mode_id, (uint8_t)o.min_bq, o.fix_mate, o.combined, (uint32_t)o.window, (uint32_t)o.overlap,
                                          o.thresholds.data(), (int)o.thresholds.size()) != SBX_OK)) { fail(sbx_last_error(c2)); sbx_close(c2); return; }
                std::lock_guard<std::mutex> g(mu);
                cx[1] = c2;
                opened2 = true;
                cv.notify_all();
            });
            uploader = std::thread([&] {
                for (size_t k = 0; k < sl.size(); ++k) {
                    if (!wait_for([&] { return k % n_ctx == 0 || opened2; })) return;
                    if (k >= n_ctx && !wait_for([&] { return computed[k - n_ctx] != 0; })) return;       // the context's compressed bytes are free again
                    sbx_ctx* c = cx[k % n_ctx];
                    const double tu = now();
                    const int rc = sbx_prefetch_interval(c, sl[k].ref, (uint32_t)sl[k].beg, (uint32_t)sl[k].end);
                    if (rc != SBX_OK) { fail(sbx_last_error(c), rc); return; }
                    busy_up += now() - tu;
                    mark(uploaded, k);
                }
            });
            computer = std::thread([&] {
                for (size_t k = 0; k < sl.size(); ++k) {
                    // (the run of slice k replaces the counters of slice k - n_ctx in its context: that text must have left)
                    if (!wait_for([&] { return uploaded[k] != 0 && (k < n_ctx || printed[k - n_ctx] != 0); })) return;
                    sbx_ctx* c = cx[k % n_ctx];
                    const double tr = now();
                    const int rc = sbx_run_interval(c, sl[k].ref, (uint32_t)sl[k].beg, (uint32_t)sl[k].end);
                    if (rc != SBX_OK) { fail(sbx_last_error(c), rc); return; }
                    busy_run += now() - tr;
                    mark(computed, k);
                }
            });
                for (size_t k = 0; k < sl.size(); ++k) {
                    if (!wait_for([&] { return computed[k] != 0; })) break;
                    const double tp = now();
                    try { bp.run_slice(cx[k % n_ctx], sl[k].ref, sl[k].beg, sl[k].print_end); }
                    catch (const Fail& f) { fail(f.msg); break; }
                    busy_print += now() - tp;
                    done_at[k] = now() - t0;
                    mark(printed, k);
                    ++n_printed;
                }
                std::lock_guard<std::mutex> g(mu);
                all_printed = n_printed == sl.size() && failure.empty();
                guard.regular = all_printed;       // (every stage is past its last wait: nothing to release)
            }
            if (!all_printed && failure_code == SBX_ENOMEM && n_printed == 0) {
                // nothing was written yet: the one-pass form below sizes its batches from the device's free memory
                if (cx[1]) sbx_close(cx[1]);
                if (timing) fprintf(stderr, "[sbx-depth] a slice of the pipeline does not fit the device next to the other context (%s): one pass instead\n", failure.c_str());
                goto one_pass;
            }
            if (!all_printed) { if (cx[1]) sbx_close(cx[1]); throw Fail{failure}; }
            out.flush();
            if (out.fp != stdout) fclose(out.fp);
            if (timing) {
                fprintf(stderr, "[sbx-depth] open %.3f s, %zu slices through upload / kernels / text on %zu context(s) in %.3f s (stages busy: upload %.3f, "
                                "kernels %.3f, text %.3f; second context opened in %.3f s), total %.3f s since main\n",
                        t_open - t_start, sl.size(), n_ctx, now() - t0, busy_up, busy_run, busy_print, t_open2, now() - t_start);
                std::string tl;
                for (double x : done_at) { char b[32]; snprintf(b, sizeof b, " %.3f", x); tl += b; }
                fprintf(stderr, "[sbx-depth] slices printed at%s s\n", tl.c_str());
            }
            // the process ends here: device memory, mappings and streams go with it (an orderly sbx_close of two contexts
            // frees tens of gigabytes buffer by buffer and costs 0.1 s that no caller is waiting for)
            if (!getenv("SBX_ORDERLY_EXIT")) { fflush(nullptr); report_done(0); _exit(0); }
            if (cx[1]) sbx_close(cx[1]);
            sbx_close(ctx);
            return 0;
        }
    one_pass:
        WindowPrinter wp{ctx, o, out, samples, false, 0, 0, -1, 0, {}, {}, {}};
        RegionPrinter rp{ctx, o, out, samples, raw, raw_lines, {}, {}, {}};
        for (auto& b : plan) {
            const double t0 = now();
            if (plan.size() == 1) check(ctx, sbx_run(ctx));
            else check(ctx, sbx_run_batch(ctx, b.first_ref, b.n_refs));
            const double t1 = now();
            const int r0 = (int)b.first_ref, r1 = (int)(b.first_ref + b.n_refs);
            // "Processing reference #N (name)" lines go to stderr in the reference (depth.d:1225-1229)
            if (o.mode == "region") rp.run_refs(r0, r1);
            else if (o.mode == "window") wp.run_refs(r0, r1);
            else bp.run_refs(r0, r1);
            t_run += t1 - t0;
            t_print += now() - t1;
            if (timing) {
                sbx_run_stats st;
                if (sbx_last_run_stats(ctx, &st) == SBX_OK)
                    fprintf(stderr, "[sbx-depth] batch refs [%d,%d): run %.3f s (h2d %.1f ms, device %.1f ms: inflate %.1f index %.1f accumulate %.1f), %llu records\n",
                            r0, r1, t1 - t0, st.ms_h2d, st.ms_total, st.ms_inflate, st.ms_index, st.ms_accumulate, (unsigned long long)st.n_records);
            }
        }
        if (o.mode == "region") rp.finish();
        else if (o.mode == "window") wp.finish();
        else if (o.mode == "base") bp.finish();
        out.flush();
        if (out.fp != stdout) fclose(out.fp);
        const double t_out = now();
        if (!getenv("SBX_ORDERLY_EXIT")) {      // (see the pipelined path: nobody waits for the frees)
            if (timing)
                fprintf(stderr, "[sbx-depth] open %.3f s, run %.3f s, print %.3f s, finish %.3f s, total %.3f s since main (exit without freeing)\n",
                        t_open - t_start, t_run, t_print, t_out - t_open - t_run - t_print, now() - t_start);
            fflush(nullptr);
            report_done(0);
            _exit(0);
        }
        sbx_close(ctx);
        if (timing)
            fprintf(stderr, "[sbx-depth] open %.3f s, run %.3f s, print %.3f s, finish %.3f s, close %.3f s, total %.3f s since main\n", t_open - t_start,
                    t_run, t_print, t_out - t_open - t_run - t_print, now() - t_out, now() - t_start);
        return 0;
    } catch (const Fail& f) {
        out.flush();
        fprintf(stderr, "sambamba-depth: %s\n", f.msg.c_str());
        if (ctx) sbx_close(ctx);
        return 1;
    } catch (const std::exception& e) {
        out.flush();
        fprintf(stderr, "sambamba-depth: %s\n", e.what());
        if (ctx) sbx_close(ctx);
        return 1;
    }
}

}  // namespace

// One process by default.  SBX_DETACH=1: the work runs in a child process and this one returns as soon as the child reports that the
// output is complete and its descriptors are closed.  What is left for the child then is the teardown of a HIP process with tens
// of gigabytes mapped -- ~0.3 s inside the driver (measured: the same for an orderly close and for _exit, profiles/round3) -- which
// nothing downstream depends on; but the orphan still holds the device while it goes, and the next command of a shell loop or a
// scheduler would start against its memory: a caller has to ask for that trade.
int main(int argc, char** argv) {
    int fd[2];
    const char* det = getenv("SBX_DETACH");
    if (!det || !*det || *det == '0' || getenv("SBX_NO_DETACH") || pipe(fd) != 0) return depth_main(argc, argv);
    const pid_t self = getpid();
    const pid_t pid = fork();                  // before anything touches HIP: a device context does not survive a fork
    if (pid < 0) { close(fd[0]); close(fd[1]); return depth_main(argc, argv); }
    if (pid == 0) {
        close(fd[0]);
        fcntl(fd[1], F_SETFD, FD_CLOEXEC);
        prctl(PR_SET_PDEATHSIG, SIGTERM);      // a caller that kills the command kills the work
        if (getppid() != self) _exit(1);       // (it died before the line above took effect)
        g_done_fd = fd[1];
        const int rc = depth_main(argc, argv);
        report_done(rc);
        _exit(rc);
    }
    close(fd[1]);
    unsigned char st = 0;
    ssize_t n;
    do n = read(fd[0], &st, 1); while (n < 0 && errno == EINTR);
    if (n == 1) _exit(st);
    int ws = 0;                                // the child ended without a report: its fate is the command's
    while (waitpid(pid, &ws, 0) < 0 && errno == EINTR) {}
    if (WIFSIGNALED(ws)) {                     // killed by a signal (SIGPIPE from a consumer that left, SIGKILL ...): die by the same one
        signal(WTERMSIG(ws), SIG_DFL);
        raise(WTERMSIG(ws));
        return 128 + WTERMSIG(ws);             // (a signal that cannot kill this process)
    }
    return WIFEXITED(ws) ? WEXITSTATUS(ws) : 1;
}
