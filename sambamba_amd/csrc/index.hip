// index.hip -- K2 `record_index`: find every BAM record in the inflated stream and describe it.
//
// Replaces, for the whole file at once, the serial loop of BamReadRange.readNext
// (BioD/bio/std/hts/bam/readrange.d:118-173: read int32 block_size, slice block_size bytes),
// the BamRead field accessors (read.d:907-1003), BamRead.basesCovered (read.d:255-262), the
// -F filter objects (sambamba/utils/common/filtering.d:86-214), the RG -> sample lookup of
// CustomBamRead (depth.d:240-250; read.d:1070-1087,1219-1230) and pileupColumns' zero-span
// filter (pileup.d:510).
//
// The record chain is inherently serial (each block_size tells where the next record starts),
// so it is cut at BGZF block granularity: one lane per BGZF block
//   1. guesses the first record start inside its block with a structural plausibility test
//      (ids/positions in range, lengths consistent with block_size, NUL-terminated name,
//      two further records chain correctly and are coordinate-sorted),
//   2. walks the chain to the end of its block -> (count, exit offset);
// then `chain_check` tests exit[b-1] == entry[b] for every block in parallel; if any block is
// inconsistent (rare: a decoy inside a record that looks like a record chain, or a block without any
// record start), `chain_repair` follows the chain serially from the first such block, keeping the
// pre-computed walk of every block whose guess turns out right.  By induction from the exactly
// known offset of the first record, a consistent chain IS the true chain -- the guess only buys
// parallelism, it can never change the result.  A scan of the counts gives every block its slot range in the
// descriptor array and `describe` walks once more, now writing one 32-byte RecDesc per record
// and the [lo,hi) record range of every position tile the record overlaps.
#include "common.hpp"
#include "kernels.hpp"
#include "regex_nfa.hpp"

#include <algorithm>

namespace sbx {

namespace {

constexpr int kWalkThreads = 64;

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) {
    uint16_t v;
    __builtin_memcpy(&v, p, 2);
    return v;
}

// CIGAR_TYPE table of cigar.d:116 -- bit0: consumes query, bit1: consumes reference (MIDNSHP=X)
constexpr uint32_t kCigarType = 0x3C1A7u;
__device__ __forceinline__ uint32_t cig_type(uint32_t raw) { return (kCigarType >> ((raw & 15u) * 2u)) & 3u; }

// Structural plausibility of a BAM record starting at offset o of the stream (fixed part only).
__device__ bool plausible_record(const uint8_t* U, uint64_t total, uint64_t o, const RefTable& refs, uint64_t* next,
                                 uint32_t* sort_key_ref, int32_t* sort_key_pos) {
    if (o + 36 > total) return false;
    const uint8_t* p = U + o;
    int64_t bs = (int32_t)ld32(p);
    int32_t ref = (int32_t)ld32(p + 4);
    if (ref < -1 || ref >= refs.n_ref) return false;
    int32_t pos = (int32_t)ld32(p + 8);
    if (pos < -1) return false;
    if (ref >= 0 && pos > refs.ref_len[ref]) return false;
    uint32_t bmn = ld32(p + 12);
    uint32_t l_name = bmn & 0xFFu;
    if (l_name < 1) return false;
    uint32_t fnc = ld32(p + 16);
    uint32_t n_cigar = fnc & 0xFFFFu;
    int32_t l_seq = (int32_t)ld32(p + 20);
    if (l_seq < 0) return false;
    int32_t nref = (int32_t)ld32(p + 24);
    if (nref < -1 || nref >= refs.n_ref) return false;
    int32_t npos = (int32_t)ld32(p + 28);
    if (npos < -1) return false;
    int64_t fixed = 32 + (int64_t)l_name + 4 * (int64_t)n_cigar + ((int64_t)l_seq + 1) / 2 + (int64_t)l_seq;
    if (bs < fixed || bs > (int64_t)(1 << 29)) return false;
    if (o + 4 + (uint64_t)bs > total) return false;
    // read name: printable, NUL only at the end
    const uint8_t* name = p + 36;
    if (name[l_name - 1] != 0) return false;
    if (l_name > 1 && (name[0] < 33 || name[0] > 126)) return false;
    if (n_cigar) {
        uint32_t op = ld32(name + l_name) & 15u;
        if (op > 8) return false;
    }
    *next = o + 4 + (uint64_t)bs;
    *sort_key_ref = (uint32_t)ref;
    *sort_key_pos = pos;
    return true;
}

// first record start in [from, limit) that passes the plausibility test 3 records deep
__device__ uint64_t guess_entry(const uint8_t* U, uint64_t total, uint64_t from, uint64_t limit, const RefTable& refs) {
    for (uint64_t o = from; o < limit; ++o) {
        uint64_t n1, n2, n3;
        uint32_t r1, r2, r3;
        int32_t p1, p2, p3;
        if (!plausible_record(U, total, o, refs, &n1, &r1, &p1)) continue;
        if (n1 == total) return o;
        if (!plausible_record(U, total, n1, refs, &n2, &r2, &p2)) continue;
        if (r2 < r1 || (r2 == r1 && p2 < p1)) continue;     // coordinate order (unmapped = 0xFFFFFFFF last)
        if (n2 == total) return o;
        if (!plausible_record(U, total, n2, refs, &n3, &r3, &p3)) continue;
        if (r3 < r2 || (r3 == r2 && p3 < p2)) continue;
        return o;
    }
    return kOffUnknown;
}

// walk the chain from `entry` until it leaves [.., block_end); returns exit offset or kOffInvalid
// ck (optional): offsets of the block's records number 64, 128 and 192 -- `describe` starts a lane at each of them, so
// that a block's ~230 records are described by four lanes with chains of 64 instead of one lane with a chain of 230
__device__ uint64_t walk_block(const uint8_t* U, uint64_t total, uint64_t entry, uint64_t block_end, uint32_t* count,
                               uint64_t* ck = nullptr) {
    uint64_t o = entry;
    uint32_t n = 0;
    if (ck) { ck[0] = kOffInvalid; ck[1] = kOffInvalid; ck[2] = kOffInvalid; }
    while (o < block_end) {
        if (ck && n && (n & 63u) == 0 && n <= 192u) ck[(n >> 6) - 1] = o;
        if (o + 4 > total) { *count = n; return kOffInvalid; }
        int64_t bs = (int32_t)ld32(U + o);
        if (bs < 32 || o + 4 + (uint64_t)bs > total) { *count = n; return kOffInvalid; }
        ++n;
        o += 4 + (uint64_t)bs;
    }
    *count = n;
    return o;
}

__global__ __launch_bounds__(kWalkThreads) void k_block_walk(const uint8_t* __restrict__ U, uint64_t total,
                                                              const uint64_t* __restrict__ out_off,
                                                              const uint32_t* __restrict__ isize, uint32_t n_blocks,
                                                              uint64_t first_record_off, RefTable refs,
                                                              uint64_t* __restrict__ entry, uint64_t* __restrict__ exit_,
                                                              uint32_t* __restrict__ count, uint64_t* __restrict__ ckpt) {
    uint32_t b = blockIdx.x * kWalkThreads + threadIdx.x;
    if (b >= n_blocks) return;
    uint64_t beg = out_off[b], end = beg + isize[b];
    if (end > total) end = total;     // the sub-stream of a -L run stops at a record boundary inside its last block
    uint64_t e;
    if (end <= first_record_off) {
        // block lies entirely inside the BAM header: the chain enters the next block at first_record_off
        entry[b] = first_record_off;
        exit_[b] = first_record_off;
        count[b] = 0;
        ckpt[3 * (size_t)b] = kOffInvalid; ckpt[3 * (size_t)b + 1] = kOffInvalid; ckpt[3 * (size_t)b + 2] = kOffInvalid;
        return;
    }
    if (beg <= first_record_off) e = first_record_off;       // exactly known
    else e = guess_entry(U, total, beg, end, refs);
    uint32_t n = 0;
    uint64_t x = kOffUnknown;
    if (e != kOffUnknown) x = walk_block(U, total, e, end, &n, ckpt + 3 * (size_t)b);
    else { ckpt[3 * (size_t)b] = kOffInvalid; ckpt[3 * (size_t)b + 1] = kOffInvalid; ckpt[3 * (size_t)b + 2] = kOffInvalid; }
    entry[b] = e;
    exit_[b] = x;
    count[b] = n;
}

// Parallel check of the guessed chain: entry[b] must equal exit[b-1] (exit of a block that contains
// no record start is its entry, see walk_block).  The lowest inconsistent block goes to *first_bad.
__global__ __launch_bounds__(kWalkThreads) void k_chain_check(const uint64_t* __restrict__ out_off,
                                                               const uint32_t* __restrict__ isize, uint32_t n_blocks,
                                                               uint64_t first_record_off, const uint64_t* __restrict__ entry,
                                                               const uint64_t* __restrict__ exit_, uint32_t* first_bad) {
    uint32_t b = blockIdx.x * kWalkThreads + threadIdx.x;
    if (b >= n_blocks) return;
    uint64_t end = out_off[b] + isize[b];
    if (end <= first_record_off) return;             // header-only block: exact by construction
    uint64_t want = (b == 0) ? first_record_off : exit_[b - 1];
    bool ok = want != kOffUnknown && want != kOffInvalid && entry[b] == want && exit_[b] != kOffUnknown && exit_[b] != kOffInvalid &&
              exit_[b] >= entry[b];
    if (!ok) atomicMin(first_bad, b);
}

// wave-uniform copy of lane i's 64-bit value (readlane works on 32-bit ints: cast each half to
// uint32_t before widening, or bit 31 of the low half sign-extends into the high half)
__device__ __forceinline__ uint64_t bcast64(uint64_t v, uint32_t i) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, i);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), i);
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// Serial repair from the first inconsistent block on (one wavefront; rare).  Everything before
// `from` is consistent with the exactly known first record offset, hence true; from there the chain
// is followed block by block: a block whose guessed entry equals the running offset keeps its
// pre-computed walk (O(1)), any other block is re-walked from the true entry.
__global__ __launch_bounds__(64) void k_chain_repair(const uint8_t* __restrict__ U, uint64_t total,
                                                      const uint64_t* __restrict__ out_off, const uint32_t* __restrict__ isize,
                                                      uint32_t n_blocks, uint64_t first_record_off, uint32_t from,
                                                      uint32_t stop_at_trusted, uint64_t* entry, uint64_t* exit_, uint32_t* count,
                                                      uint64_t* ckpt, uint32_t* n_rewalked) {
    const uint32_t lane = threadIdx.x;
    uint64_t cur = (from == 0) ? first_record_off : exit_[from - 1];
    uint32_t rewalked = 0;
    bool done = false;
    for (uint32_t b0 = from; b0 < n_blocks && !done; b0 += 64) {
        const uint32_t b = b0 + lane;
        uint64_t e = kOffUnknown, x = kOffUnknown, end = 0;
        uint32_t n = 0;
        if (b < n_blocks) { e = entry[b]; x = exit_[b]; n = count[b]; end = out_off[b] + isize[b]; if (end > total) end = total; }
        const uint32_t lim = n_blocks - b0 < 64 ? n_blocks - b0 : 64;
        bool dirty = false;
        for (uint32_t i = 0; i < lim; ++i) {
            const uint64_t e_i = bcast64(e, i), x_i = bcast64(x, i), end_i = bcast64(end, i);
            uint64_t ne, nx;
            uint32_t nn;
            if (cur == kOffInvalid) { ne = kOffInvalid; nx = kOffInvalid; nn = 0; }
            else if (end_i <= first_record_off) { ne = e_i; nx = x_i; nn = 0; }   // header-only block
            else if (e_i == cur && x_i != kOffUnknown) {
                // guess confirmed: everything from here to the next inconsistent block is already right
                if (stop_at_trusted && rewalked) { done = true; break; }
                ne = e_i; nx = x_i; nn = (uint32_t)__builtin_amdgcn_readlane((int)n, i);
            }
            else {
                uint32_t c = 0;
                nx = walk_block(U, total, cur, end_i, &c, ckpt + 3 * (size_t)(b0 + i));     // wave-uniform re-walk (every lane writes the same checkpoints)
                ne = cur;
                nn = c;
                ++rewalked;
            }
            if (lane == i && (ne != e || nx != x || nn != n)) { e = ne; x = nx; n = nn; dirty = true; }
            if (!(end_i <= first_record_off)) cur = nx;
        }
        if (dirty && b < n_blocks) { entry[b] = e; exit_[b] = x; count[b] = n; }
    }
    if (lane == 0) *n_rewalked += rewalked;
}

// ---- scan of the per-block counts (single workgroup, 3 phases; n_blocks is ~1e5..1e6) -----------
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void k_count_scan(const uint32_t* __restrict__ count, uint32_t n,
                                                              uint64_t* __restrict__ base) {
    __shared__ uint64_t part[kScanThreads];
    uint32_t t = threadIdx.x;
    uint32_t per = (n + kScanThreads - 1) / kScanThreads;
    uint32_t lo = t * per, hi = lo + per < n ? lo + per : n;
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += count[i];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (uint32_t d = 1; d < kScanThreads; d <<= 1) {
        uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = t ? part[t - 1] : 0;
    for (uint32_t i = lo; i < hi; ++i) {
        base[i] = run;
        run += count[i];
    }
    if (t == kScanThreads - 1) base[n] = part[kScanThreads - 1];
}

// ---- describe ---------------------------------------------------------------------------------
// first aux field with the given key (BamRead.opIndex, read.d:1070-1087; skipValue read.d:1219-1230):
// returns its type character and value pointer, 0 when absent or when the tag area is malformed before it
__device__ uint32_t find_tag(const uint8_t* t, const uint8_t* e, uint32_t key, const uint8_t** val) {
    while (t + 3 <= e) {
        const uint32_t k = (uint32_t)t[0] | ((uint32_t)t[1] << 8);
        const uint8_t ty = t[2];
        t += 3;
        const uint8_t* v = t;
        switch (ty) {
            case 'A': case 'c': case 'C': t += 1; break;
            case 's': case 'S': t += 2; break;
            case 'i': case 'I': case 'f': t += 4; break;
            case 'Z': case 'H': while (t < e && *t) ++t; ++t; break;
            case 'B': {
                if (t + 5 > e) return 0;
                const uint8_t sub = t[0];
                const uint32_t n = ld32(t + 1);
                const uint32_t w = (sub == 'c' || sub == 'C') ? 1u : (sub == 's' || sub == 'S') ? 2u : 4u;
                t += 5 + (uint64_t)n * w;
                break;
            }
            default: return 0;
        }
        if (t > e) return 0;
        if (k == key) { *val = v; return ty; }
    }
    return 0;
}

template <class T>
__device__ __forceinline__ bool cmp_op(uint32_t op, T x, T y) {
    switch (op) {
        case 0: return x > y;
        case 1: return x < y;
        case 2: return x >= y;
        case 3: return x <= y;
        case 4: return x == y;
        default: return x != y;
    }
}

// lexicographic comparison of two byte strings with a comparison operator (D's string comparison for ASCII)
__device__ bool cmp_str(uint32_t op, const uint8_t* a, uint32_t na, const char* b, uint32_t nb) {
    int c = 0;
    const uint32_t n = na < nb ? na : nb;
    for (uint32_t i = 0; i < n && c == 0; ++i) c = (int)a[i] - (int)(uint8_t)b[i];
    if (c == 0) c = na < nb ? -1 : na > nb ? 1 : 0;
    return cmp_op<int>(op, c, 0);
}

__device__ bool eval_filter(const DeviceFilter* f, const uint8_t* p /* at refID */, int32_t ref, int32_t pos, uint32_t bmn,
                            uint32_t fnc, int32_t l_seq, const uint8_t* tags, const uint8_t* tags_end) {
    // postfix program over a tiny bool stack (bit stack in a 64-bit word); the fields every record's
    // walk has loaded anyway come in registers
    uint64_t stack = 0;
    int sp = 0;
    uint32_t flag = fnc >> 16, mapq = (bmn >> 8) & 0xFF;
    for (int i = 0; i < f->n_ops; ++i) {
        const sbx_filter_op& op = f->ops[i];
        bool v = true;
        switch (op.kind) {
            case 0: v = (flag & op.mask) != 0; break;
            case 1: v = (flag & 1) && !(flag & 4) && !(flag & 8) && ref != (int32_t)ld32(p + 20); break;
            case 2: {
                if (op.field == 7) {      // avg_base_quality: float32 sum / length against the integer (filtering.d:189-191,206)
                    const uint32_t l_name = bmn & 0xFF, n_cigar = fnc & 0xFFFF;
                    const uint8_t* q = p + 32 + l_name + 4 * n_cigar + (((uint32_t)l_seq + 1) >> 1);
                    float sum = 0.0f;
                    for (int32_t k = 0; k < l_seq; ++k) sum += (float)q[k];
                    v = cmp_op<float>(op.cmp, sum / (float)l_seq, (float)op.value);
                    break;
                }
                int64_t x = 0;
                switch (op.field) {
                    case 0: x = ref; break;
                    case 1: x = pos; break;
                    case 2: x = mapq; break;
                    case 3: x = l_seq; break;
                    case 4: x = (int32_t)ld32(p + 20); break;
                    case 5: x = (int32_t)ld32(p + 24); break;
                    default: x = (int32_t)ld32(p + 28); break;
                }
                switch (op.cmp) {
                    case 0: v = x > op.value; break;
                    case 1: v = x < op.value; break;
                    case 2: v = x >= op.value; break;
                    case 3: v = x <= op.value; break;
                    case 4: v = x == op.value; break;
                    default: v = x != op.value; break;
                }
                break;
            }
            case 7: {     // IntegerTagFilter (filtering.d:233-252): integer or float tags only, anything else rejects
                const uint8_t* tv = nullptr;
                const uint32_t ty = find_tag(tags, tags_end, op.mask, &tv);
                int64_t iv = 0;
                bool is_int = true;
                switch (ty) {
                    case 'c': iv = (int8_t)tv[0]; break;
                    case 'C': iv = tv[0]; break;
                    case 's': iv = (int16_t)(tv[0] | (tv[1] << 8)); break;
                    case 'S': iv = (uint16_t)(tv[0] | (tv[1] << 8)); break;
                    case 'i': iv = (int32_t)ld32(tv); break;
                    case 'I': iv = (int64_t)ld32(tv); break;
                    default: is_int = false; break;
                }
                if (is_int) v = cmp_op<int64_t>(op.cmp, iv, op.value);
                else if (ty == 'f') v = cmp_op<float>(op.cmp, __uint_as_float(ld32(tv)), (float)op.value);
                else v = false;
                break;
            }
            case 9: {     // StringTagFilter (filtering.d:276-297): Z strings, A characters against one-character literals
                const uint8_t* tv = nullptr;
                const uint32_t ty = find_tag(tags, tags_end, op.mask, &tv);
                const char* lit = f->strings + (uint32_t)(op.value & 0xFFFFFFFF);
                const uint32_t nl = (uint32_t)(op.value >> 32);
                if (ty == 'Z' || ty == 'H') {     // Value.is_string: 'Z' or 'H' (tagvalue.d:426-427)
                    uint32_t n = 0;
                    while (tv + n < tags_end && tv[n]) ++n;
                    v = cmp_str(op.cmp, tv, n, lit, nl);
                } else if (ty == 'A') {
                    v = nl == 1 && cmp_op<int>(op.cmp, (int)tv[0], (int)(uint8_t)lit[0]);
                } else v = false;
                break;
            }
            case 10: {    // StringFieldFilter on read_name (filtering.d:264)
                const uint32_t l_name = bmn & 0xFF;
                v = cmp_str(op.cmp, p + 32, l_name ? l_name - 1 : 0, f->strings + (uint32_t)(op.value & 0xFFFFFFFF), (uint32_t)(op.value >> 32));
                break;
            }
            case 13: {    // sequence as text against the literal (StringFieldFilter, filtering.d:265)
                const uint32_t l_name = bmn & 0xFF, n_cigar = fnc & 0xFFFF;
                const uint8_t* sq = p + 32 + l_name + 4 * n_cigar;
                const char* lit = f->strings + (uint32_t)(op.value & 0xFFFFFFFF);
                const uint32_t nl = (uint32_t)(op.value >> 32), ns = l_seq > 0 ? (uint32_t)l_seq : 0u;
                int c = 0;
                for (uint32_t i = 0; i < ns && i < nl && c == 0; ++i) {
                    const uint32_t nib = (i & 1u) ? (sq[i >> 1] & 15u) : (sq[i >> 1] >> 4);
                    c = (int)(uint8_t)"=ACMGRSVTWYHKDBN"[nib] - (int)(uint8_t)lit[i];
                }
                if (c == 0) c = ns < nl ? -1 : ns > nl ? 1 : 0;
                v = cmp_op<int>(op.cmp, c, 0);
                break;
            }
            case 14: {    // cigarString(): decimal length + operation character per op, "" for no CIGAR (read.d:265-276)
                const uint32_t l_name = bmn & 0xFF, n_cigar = fnc & 0xFFFF;
                const uint8_t* cg = p + 32 + l_name;
                const char* lit = f->strings + (uint32_t)(op.value & 0xFFFFFFFF);
                const uint32_t nl = (uint32_t)(op.value >> 32);
                uint32_t k = 0;        // characters of the literal matched so far
                int c = 0;
                for (uint32_t i = 0; i < n_cigar && c == 0; ++i) {
                    const uint32_t raw = ld32(cg + 4 * i);
                    uint32_t len = raw >> 4, div = 1;
                    while (len / div >= 10) div *= 10;
                    for (; div && c == 0; div /= 10) {
                        const int ch = '0' + (int)((len / div) % 10);
                        if (k >= nl) c = 1; else c = ch - (int)(uint8_t)lit[k++];
                    }
                    if (c == 0) {
                        const int ch = (raw & 15u) < 9 ? (int)"MIDNSHP=X"[raw & 15u] : (int)'?';
                        if (k >= nl) c = 1; else c = ch - (int)(uint8_t)lit[k++];
                    }
                }
                if (c == 0 && k < nl) c = -1;
                v = cmp_op<int>(op.cmp, c, 0);
                break;
            }
            case 15: {    // RegexpFieldFilter / RegexpTagFilter (filtering.d:299-345): does the pattern match anywhere?
                const sbx_regex& re = f->regex[op.value & 1];
                const uint32_t l_name = bmn & 0xFF, n_cigar = fnc & 0xFFFF;
                if (op.field == 0) {
                    const uint8_t* nm = p + 32;
                    v = re_search(re, l_name ? l_name - 1 : 0, [&](uint32_t i) { return nm[i]; });
                } else if (op.field == 1) {
                    const uint8_t* sq = p + 32 + l_name + 4 * n_cigar;
                    v = re_search(re, l_seq > 0 ? (uint32_t)l_seq : 0u, [&](uint32_t i) {
                        const uint32_t nib = (i & 1u) ? (sq[i >> 1] & 15u) : (sq[i >> 1] >> 4);
                        return (uint8_t)"=ACMGRSVTWYHKDBN"[nib];
                    });
                } else if (op.field == 2) {
                    // cigarString(), generated character by character (the search reads positions in order)
                    const uint8_t* cg = p + 32 + l_name;
                    uint32_t total = 0;
                    for (uint32_t i = 0; i < n_cigar; ++i) { uint32_t len = ld32(cg + 4 * i) >> 4; do { ++total; len /= 10; } while (len); ++total; }
                    uint32_t oi = 0, div = 0;     // current op, divisor of its next digit (0: none left, the op character is next)
                    bool fresh = true;
                    v = re_search(re, total, [&](uint32_t) {
                        const uint32_t raw = ld32(cg + 4 * oi), len = raw >> 4;
                        if (fresh) { div = 1; while (len / div >= 10) div *= 10; fresh = false; }
                        if (div) { const uint8_t ch = (uint8_t)('0' + (len / div) % 10); div /= 10; return ch; }
                        ++oi; fresh = true;
                        return (uint8_t)((raw & 15u) < 9 ? "MIDNSHP=X"[raw & 15u] : '?');
                    });
                } else if (op.field == 3) {
                    const uint8_t* tv = nullptr;
                    const uint32_t tty = find_tag(tags, tags_end, op.mask, &tv);
                    if (tty == 'Z' || tty == 'H') {
                        uint32_t n = 0;
                        while (tv + n < tags_end && tv[n]) ++n;
                        v = re_search(re, n, [&](uint32_t i) { return tv[i]; });
                    } else v = false;
                } else v = false;
                break;
            }
            case 16: {    // a regular expression on ref_name / mate_ref_name: evaluated per reference on the host
                const int32_t id = op.field ? (int32_t)ld32(p + 20) : ref;
                v = id >= -1 && id < f->n_ref && f->ref_sets[(uint32_t)op.value + (uint32_t)(id + 1)] != 0;
                break;
            }
            case 12: v = false; break;
            case 8: {     // TagExistenceFilter (filtering.d:216-230)
                const uint8_t* tv = nullptr;
                const bool present = find_tag(tags, tags_end, op.mask, &tv) != 0;
                v = op.cmp == 5 ? present : !present;
                break;
            }
            case 3: { bool b2 = stack & 1; stack >>= 1; bool a2 = stack & 1; stack >>= 1; sp -= 2; v = a2 && b2; break; }
            case 4: { bool b2 = stack & 1; stack >>= 1; bool a2 = stack & 1; stack >>= 1; sp -= 2; v = a2 || b2; break; }
            case 5: { bool a2 = stack & 1; stack >>= 1; sp -= 1; v = !a2; break; }
            default: v = true; break;
        }
        stack = (stack << 1) | (v ? 1u : 0u);
        ++sp;
    }
    return sp > 0 ? (stack & 1) : true;
}

// RG:Z lookup: linear scan of the aux fields (read.d:1070-1087, skipValue read.d:1219-1230).
// returns sample id, 0 when the read has no RG tag, 0xFFFF when the id is not in the header.
__device__ uint32_t lookup_sample(const uint8_t* t, const uint8_t* e, const RgTable& rg) {
    while (t + 3 <= e) {
        uint8_t k0 = t[0], k1 = t[1], ty = t[2];
        t += 3;
        const uint8_t* v = t;
        switch (ty) {
            case 'A': case 'c': case 'C': t += 1; break;
            case 's': case 'S': t += 2; break;
            case 'i': case 'I': case 'f': t += 4; break;
            case 'Z': case 'H': while (t < e && *t) ++t; ++t; break;
            case 'B': {
                if (t + 5 > e) return 0;
                uint8_t sub = t[0];
                uint32_t n = ld32(t + 1);
                uint32_t w = (sub == 'c' || sub == 'C') ? 1u : (sub == 's' || sub == 'S') ? 2u : 4u;
                t += 5 + (uint64_t)n * w;
                break;
            }
            default: return 0;
        }
        if (k0 == 'R' && k1 == 'G') {
            if (ty != 'Z' && ty != 'H') return 0xFFFFu;
            uint32_t len = (uint32_t)((t - 1) - v);
            for (int g = 0; g < rg.n_rg; ++g) {
                const char* id = rg.ids + rg.id_off[g];
                uint32_t k = 0;
                while (k < len && id[k] && (uint8_t)id[k] == v[k]) ++k;
                if (k == len && id[k] == 0) return rg.sample_of[g];
            }
            return 0xFFFFu;
        }
    }
    return 0;
}

__global__ __launch_bounds__(kWalkThreads) void k_describe(const uint8_t* __restrict__ U, uint64_t total,
                                                            const uint64_t* __restrict__ out_off,
                                                            const uint32_t* __restrict__ isize, uint32_t n_blocks,
                                                            const uint64_t* __restrict__ entry,
                                                            const uint64_t* __restrict__ ckpt,
                                                            const uint64_t* __restrict__ base, RefTable refs,
                                                            const DeviceFilter* __restrict__ filt, RgTable rg,
                                                            uint32_t tile_pos, RecDesc* __restrict__ desc,
                                                            int32_t* __restrict__ rec_ref, uint64_t* __restrict__ name_hash,
                                                            uint32_t* tile_lo, uint32_t* tile_hi, IndexStats* stats) {
    // four lanes per BGZF block: lane `seg` describes records 64*seg .. 64*seg+63 of the block (the last one: the rest),
    // starting at the checkpoint the walk left for it
    const uint32_t gl = blockIdx.x * kWalkThreads + threadIdx.x;
    const uint32_t b = gl >> 2, seg = gl & 3u;
    if (b >= n_blocks) return;
    uint64_t end = out_off[b] + isize[b];
    if (end > total) end = total;
    uint64_t o = seg == 0 ? entry[b] : ckpt[3 * (size_t)b + seg - 1];
    uint64_t idx = base[b] + 64u * seg;
    uint32_t left = seg < 3 ? 64u : 0xFFFFFFFFu;
    unsigned long long n_rec = 0, n_adm = 0, n_bad = 0, n_urg = 0;
    // tile-range bookkeeping aggregated per lane: flush on tile change
    uint32_t cur_t0 = 0xFFFFFFFFu, cur_t1 = 0, cur_lo = 0, cur_hi = 0;
    auto flush = [&]() {
        if (cur_t0 == 0xFFFFFFFFu) return;
        for (uint32_t t = cur_t0; t <= cur_t1; ++t) {
            atomicMin(&tile_lo[t], cur_lo);
            atomicMax(&tile_hi[t], cur_hi);
        }
        cur_t0 = 0xFFFFFFFFu;
    };
    // The walk is a pointer chase (the next record's offset is in this record's first word), so the
    // next record's fixed fields are requested as soon as that word is known and travel while this
    // record's CIGAR / tags are being read: one memory round trip per record instead of two or three.
    struct Fixed { uint32_t w[6]; };     // block_size, refID, pos, bin_mq_nl, flag_nc, l_seq
    auto load_fixed = [&](uint64_t at) {
        Fixed f;
#pragma unroll
        for (int k = 0; k < 6; ++k) f.w[k] = ld32(U + at + 4 * k);
        return f;
    };
    bool have = o < end && o != kOffUnknown && o != kOffInvalid;
    Fixed fx = {{0, 0, 0, 0, 0, 0}};
    if (have) fx = load_fixed(o);
    while (have) {
        const uint8_t* p = U + o;
        int64_t bs = (int32_t)fx.w[0];
        const uint8_t* r = p + 4;
        int32_t ref = (int32_t)fx.w[1], pos = (int32_t)fx.w[2];
        uint32_t bmn = fx.w[3], fnc = fx.w[4];
        uint32_t l_name = bmn & 0xFF, mapq = (bmn >> 8) & 0xFF;
        uint32_t n_cigar = fnc & 0xFFFF, flag = fnc >> 16;
        int32_t l_seq = (int32_t)fx.w[5];
        const uint64_t o_next = o + 4 + (uint64_t)bs;
        --left;
        const bool have_next = bs >= 32 && o_next < end && left != 0;
        Fixed fn = {{0, 0, 0, 0, 0, 0}};
        if (have_next) fn = load_fixed(o_next);
        RecDesc d;
        d.rec_off = o;
        d.pos = pos;
        d.end = pos;
        d.l_seq = (uint32_t)l_seq;
        d.n_cigar = (uint16_t)n_cigar;
        d.l_name = (uint8_t)l_name;
        d.mapq = (uint8_t)mapq;
        d.flag = (uint16_t)flag;
        d.sample = 0;
        d.q_start = 0;
        d.kind = 0;
        d.pad = 0;
        ++n_rec;
        int64_t fixed = 32 + (int64_t)l_name + 4 * (int64_t)n_cigar + ((int64_t)(l_seq < 0 ? 0 : l_seq) + 1) / 2 + (l_seq < 0 ? 0 : l_seq);
        bool sane = l_seq >= 0 && bs >= fixed && ref >= -1 && ref < refs.n_ref;
        if (!sane) ++n_bad;
        bool admit = sane && !(flag & 0x4) && ref >= 0;                       // read.d:256, unmapped reads cover nothing
        if (admit) admit = eval_filter(filt, r, ref, pos, bmn, fnc, l_seq, r + fixed, r + bs);                              // filtering.d:36-38
        if (admit) {
            // basesCovered + shape of the CIGAR
            const uint8_t* cg = r + 32 + l_name;
            int64_t span = 0;
            uint32_t q_lead = 0;         // query bases before the first reference-consuming op
            uint32_t runs = 0;           // number of maximal runs of M/=/X
            bool in_run = false, other_ref = false, q_inside = false, seen_ref = false, zero_ref = false;
            for (uint32_t i = 0; i < n_cigar; ++i) {
                uint32_t op = ld32(cg + 4 * i);
                uint32_t ty = cig_type(op), len = op >> 4;
                // a zero-length reference-consuming op still occupies one pileup column in the reference
                // (PileupRead.incrementPosition tests offset >= length only after stepping, pileup.d:195-205):
                // such reads take the general path, which emulates it
                if (len == 0 && (ty & 2)) zero_ref = true;
                if (ty == 3) {
                    if (!in_run) { ++runs; in_run = true; }
                    span += len;
                    seen_ref = true;
                } else {
                    if (ty & 2) { other_ref = true; span += len; seen_ref = true; in_run = false; }
                    else if (ty & 1) {
                        if (!seen_ref) q_lead += len;
                        else { in_run = false; q_inside = true; }   // I or trailing S: ends the run
                    }
                    // H / P (ty == 0) neither end a run nor consume anything
                }
            }
            // q_inside is also set by a trailing soft clip, which is harmless for the fast path
            // as long as there is exactly one run and no D/N: positions map 1:1 onto the query.
            (void)q_inside;
            if (span <= 0 || span > 0x7FFFFFFF - (int64_t)pos) admit = false;   // pileup.d:510
            else {
                d.end = pos + (int32_t)span;
                if (runs == 1 && !other_ref && !zero_ref && q_lead <= 0xFFFF) { d.kind = 1; d.q_start = (uint16_t)q_lead; }
                else d.kind = 2;
            }
        }
        if (admit && refs.sel) {
            // the read must overlap one of the requested regions: pos < region.end && pos + span > region.start
            uint32_t lo = refs.sel_first[ref], hi = refs.sel_first[ref + 1];
            while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((int64_t)refs.sel[m].end > (int64_t)pos) hi = m; else lo = m + 1; }
            if (lo >= refs.sel_first[ref + 1] || (int64_t)refs.sel[lo].start >= (int64_t)d.end) { admit = false; d.kind = 0; d.end = d.pos; }
        }
        if (admit && rg.lookup) {
            const uint8_t* tags = r + fixed;
            uint32_t s = lookup_sample(tags, r + bs, rg);
            if (s == 0xFFFFu) { ++n_urg; s = 0; }
            d.sample = (uint16_t)s;
        }
        if (admit) {
            ++n_adm;
            uint32_t t0 = refs.tile_base[ref] + (uint32_t)pos / tile_pos;
            uint32_t t1 = refs.tile_base[ref] + (uint32_t)(d.end - 1) / tile_pos;
            uint32_t last_tile = refs.tile_base[ref + 1] - 1;      // clip alignments hanging over the contig end
            if (t1 > last_tile) t1 = last_tile;
            if (t0 > last_tile) { admit = false; d.kind = 0; d.end = d.pos; --n_adm; ++n_bad; }
            else if (t0 == cur_t0 && t1 == cur_t1) { cur_hi = (uint32_t)idx + 1; }
            else {
                flush();
                cur_t0 = t0; cur_t1 = t1; cur_lo = (uint32_t)idx; cur_hi = (uint32_t)idx + 1;
            }
        }
        desc[idx] = d;
        rec_ref[idx] = ref;
        if (name_hash) {   // FNV-1a over the read name without its NUL (CustomBamRead, depth.d:252-258)
            uint64_t h = 14695981039346656037ULL;
            const uint8_t* nm = r + 32;
            for (uint32_t k = 0; k + 1 < l_name; ++k) { h ^= nm[k]; h *= 1099511628211ULL; }
            name_hash[idx] = h;
        }
        ++idx;
        o = o_next;
        fx = fn;
        have = have_next;
    }
    flush();
    if (n_rec) atomicAdd(&stats->n_records, n_rec);
    if (n_adm) atomicAdd(&stats->n_admitted, n_adm);
    if (n_bad) atomicAdd(&stats->n_bad, n_bad);
    if (n_urg) atomicAdd(&stats->n_unknown_rg, n_urg);
}

// ---- active tile compaction (single workgroup ballot scan; n_tiles ~ 1e3..2e6) ---------------------
__global__ __launch_bounds__(kScanThreads) void k_tile_compact(const uint32_t* __restrict__ tile_lo,
                                                                const uint32_t* __restrict__ tile_hi, uint32_t n_tiles,
                                                                uint32_t* __restrict__ active, uint32_t* __restrict__ slot_of,
                                                                uint32_t* __restrict__ n_active) {
    // 1024 tiles per step, coalesced: rank inside the wave by ballot, wave totals through LDS
    __shared__ uint32_t wtot[kScanThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    uint32_t run = 0;                    // active tiles before this step (workgroup-uniform)
    for (uint32_t i0 = 0; i0 < n_tiles; i0 += kScanThreads) {
        const uint32_t i = i0 + t;
        const bool on = i < n_tiles && tile_hi[i] > tile_lo[i];
        const uint64_t m = __ballot(on);
        if (lane == 0) wtot[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
            const uint32_t v = wtot[w];
            before += w < wv ? v : 0u;
            all += v;
        }
        if (i < n_tiles) {
            if (on) {
                const uint32_t slot = run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                active[slot] = i;
                slot_of[i] = slot;
            } else {
                slot_of[i] = 0xFFFFFFFFu;
            }
        }
        run += all;
        __syncthreads();
    }
    if (t == 0) *n_active = run;
}

}  // namespace

namespace {
__global__ __launch_bounds__(1024) void k_max_u32(const uint32_t* __restrict__ in, uint64_t n, uint32_t* out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 1024) m = in[i] > m ? in[i] : m;
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_down(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
}  // namespace

void launch_max_u32(const uint32_t* d_in, uint64_t n, uint32_t* d_out, hipStream_t stream) {
    if (!n) return;
    uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (n + 1023) / 1024);
    hipLaunchKernelGGL(k_max_u32, dim3(grid), dim3(1024), 0, stream, d_in, n, d_out);
    SBX_HIP(hipGetLastError());
}

void launch_block_walk(const uint8_t* d_U, uint64_t total, const uint64_t* d_out_off, const uint32_t* d_isize,
                       uint32_t n_blocks, uint64_t first_record_off, RefTable refs, uint64_t* d_entry, uint64_t* d_exit,
                       uint32_t* d_count, uint64_t* d_ckpt, hipStream_t stream) {
    if (!n_blocks) return;
    dim3 grid((n_blocks + kWalkThreads - 1) / kWalkThreads), block(kWalkThreads);
    hipLaunchKernelGGL(k_block_walk, grid, block, 0, stream, d_U, total, d_out_off, d_isize, n_blocks, first_record_off,
                       refs, d_entry, d_exit, d_count, d_ckpt);
    SBX_HIP(hipGetLastError());
}

void launch_chain_check(const uint64_t* d_out_off, const uint32_t* d_isize, uint32_t n_blocks, uint64_t first_record_off,
                        const uint64_t* d_entry, const uint64_t* d_exit, uint32_t* d_first_bad, hipStream_t stream) {
    if (!n_blocks) return;
    dim3 grid((n_blocks + kWalkThreads - 1) / kWalkThreads), block(kWalkThreads);
    hipLaunchKernelGGL(k_chain_check, grid, block, 0, stream, d_out_off, d_isize, n_blocks, first_record_off, d_entry, d_exit,
                       d_first_bad);
    SBX_HIP(hipGetLastError());
}

void launch_chain_repair(const uint8_t* d_U, uint64_t total, const uint64_t* d_out_off, const uint32_t* d_isize,
                         uint32_t n_blocks, uint64_t first_record_off, uint32_t from, bool stop_at_trusted, uint64_t* d_entry,
                         uint64_t* d_exit, uint32_t* d_count, uint64_t* d_ckpt, uint32_t* d_n_rewalked, hipStream_t stream) {
    hipLaunchKernelGGL(k_chain_repair, dim3(1), dim3(64), 0, stream, d_U, total, d_out_off, d_isize, n_blocks,
                       first_record_off, from, stop_at_trusted ? 1u : 0u, d_entry, d_exit, d_count, d_ckpt, d_n_rewalked);
    SBX_HIP(hipGetLastError());
}

size_t count_scan_tmp_bytes(uint32_t) { return 0; }

void launch_count_scan(const uint32_t* d_count, uint32_t n_blocks, uint64_t* d_base, void*, size_t, hipStream_t stream) {
    hipLaunchKernelGGL(k_count_scan, dim3(1), dim3(kScanThreads), 0, stream, d_count, n_blocks, d_base);
    SBX_HIP(hipGetLastError());
}

void launch_describe(const uint8_t* d_U, uint64_t total, const uint64_t* d_out_off, const uint32_t* d_isize,
                     uint32_t n_blocks, const uint64_t* d_entry, const uint64_t* d_ckpt, const uint64_t* d_base, RefTable refs,
                     const DeviceFilter* d_filter, RgTable rg, uint32_t tile_pos, RecDesc* d_desc, int32_t* d_rec_ref,
                     uint64_t* d_name_hash, uint32_t* d_tile_lo, uint32_t* d_tile_hi, IndexStats* d_stats, hipStream_t stream) {
    if (!n_blocks) return;
    dim3 grid((uint32_t)(((uint64_t)n_blocks * 4 + kWalkThreads - 1) / kWalkThreads)), block(kWalkThreads);
    hipLaunchKernelGGL(k_describe, grid, block, 0, stream, d_U, total, d_out_off, d_isize, n_blocks, d_entry, d_ckpt, d_base,
                       refs, d_filter, rg, tile_pos, d_desc, d_rec_ref, d_name_hash, d_tile_lo, d_tile_hi, d_stats);
    SBX_HIP(hipGetLastError());
}

void launch_tile_compact(const uint32_t* d_tile_lo, const uint32_t* d_tile_hi, uint32_t n_tiles, uint32_t* d_active,
                         uint32_t* d_slot_of, uint32_t* d_n_active, hipStream_t stream) {
    hipLaunchKernelGGL(k_tile_compact, dim3(1), dim3(kScanThreads), 0, stream, d_tile_lo, d_tile_hi, n_tiles, d_active,
                       d_slot_of, d_n_active);
    SBX_HIP(hipGetLastError());
}

}  // namespace sbx
