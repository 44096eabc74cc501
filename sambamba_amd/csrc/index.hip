// index.hip -- K2 `record_index`: find every BAM record in the inflated stream and describe it.
//
// Replaces, for the whole work list at once, the serial loop of BamReadRange.readNext
// (BioD/bio/std/hts/bam/readrange.d:118-173: read int32 block_size, slice block_size bytes),
// the BamRead field accessors (read.d:907-1003), BamRead.basesCovered (read.d:255-262), the
// -F filter objects (sambamba/utils/common/filtering.d:86-214), the RG -> sample lookup of
// CustomBamRead (depth.d:240-250; read.d:1070-1087,1219-1230) and pileupColumns' zero-span
// filter (pileup.d:510).
//
// Three launches over the BGZF blocks of the work list:
//   1. `k_walk_blocks` -- the record chain (each block_size tells where the next record starts) is serial, so it is
//      cut at BGZF block granularity and walked by ONE LANE PER BLOCK: hundreds of thousands of pointer chases in flight
//      hide the latency of a dependent load per record.  A run of the work list starts at an exactly known record
//      boundary (first record of the file / a BAI chunk start); everywhere else the lane GUESSES the first record start
//      in its block with a structural plausibility test (ids / positions in range, lengths consistent with block_size,
//      NUL-terminated name, two further records chain and are coordinate-sorted).  The lane leaves the block-relative
//      offsets of its records (u16, written 16 bytes at a time) in the block's slice of K1's literal stream, which is
//      dead once the block is inflated -- no extra memory;
//   2. `k_check_scan` -- the guesses are VERIFIED, never trusted: the entry of block b must equal the exit of block b-1
//      (by induction from the exactly known start of the run a consistent chain IS the true chain; an inconsistency is
//      reported to the host, which repairs the chain serially -- `k_chain_repair`, rare -- and launches again with the
//      entries given); the same single workgroup scans the per-block record counts into slot ranges;
//   3. `k_describe_blocks` -- one wave per block, ONE LANE PER RECORD: fixed part, CIGAR span and shape, the -F program,
//      RG -> sample, name hash; writes the 32-byte RecDesc and marks the [lo,hi) record range of every position tile the
//      record overlaps (one atomic per tile change inside a wave).
// Bound: HBM latency / line traffic -- one 128-byte line per record in the walk and one or two in the describe pass
// (DESIGN.md section 4).
#include "common.hpp"
#include "kernels.hpp"
#include "regex_nfa.hpp"

#include <algorithm>

namespace sbx {

namespace {

constexpr uint64_t kStateMask = (1ull << 62) - 1;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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

// A read-only table behind a generic pointer, read at a wave-uniform index: through the constant address space the compiler makes it a
// SCALAR load (its own counter; nothing to do with the vector memory queue).  As a generic pointer the -F program was read with
// vector loads -- a dependent round trip to L2 per operation and record turn, queued behind whatever the wave had prefetched: that, not
// the record bytes, was what `describe` waited for (round 6).
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* as_constant(const T* p) {
    return (const __attribute__((address_space(4))) T*)(uintptr_t)p;
}

// operation i of a -F program, through the scalar unit (i is the same in every lane)
__device__ __forceinline__ sbx_filter_op filter_op(const DeviceFilter* f, int i) {
    // (four dwords: the scalar unit has no byte loads, and a byte field read on its own would be a vector load again)
    static_assert(sizeof(sbx_filter_op) == 16 && offsetof(sbx_filter_op, mask) == 4 && offsetof(sbx_filter_op, value) == 8, "layout of sbx_filter_op");
    const auto* w = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)&f->ops[i];
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    sbx_filter_op op;
    op.kind = (uint8_t)w0; op.field = (uint8_t)(w0 >> 8); op.cmp = (uint8_t)(w0 >> 16); op.pad = 0;
    op.mask = w1;
    op.value = (int64_t)(((uint64_t)w3 << 32) | w2);
    return op;
}

// CIGAR_TYPE table of cigar.d:116 -- bit0: consumes query, bit1: consumes reference (MIDNSHP=X)
constexpr uint32_t kCigarType = 0x3C1A7u;
__device__ __forceinline__ uint32_t cig_type(uint32_t raw) { return (kCigarType >> ((raw & 15u) * 2u)) & 3u; }

// Structural plausibility of a BAM record starting at offset o of the stream (fixed part only);
// `limit` = end of the run: no record may extend beyond it.
__device__ bool plausible_record(const uint8_t* U, uint64_t limit, uint64_t o, const RefTable& refs, uint64_t* next,
                                 uint32_t* sort_key_ref, int32_t* sort_key_pos) {
    if (o + 36 > limit) return false;
    const uint8_t* p = U + o;
    // (the fixed part in one go -- three loads in flight -- instead of a load per test: the candidates of a wavefront are 64 consecutive
    //  offsets, their 36-byte windows share one or two lines, and a chain of nine dependent round trips per candidate was what
    //  k_guess_entries spent its time on)
    uint32_t f[9];
    __builtin_memcpy(f, p, 36);
    int64_t bs = (int32_t)f[0];
    if (bs < 32 || bs > (int64_t)(1 << 29)) return false;
    int32_t ref = (int32_t)f[1];
    if (ref < -1 || ref >= refs.n_ref_own) return false;
    int32_t pos = (int32_t)f[2];
    if (pos < -1) return false;
    if (ref >= 0 && pos > refs.ref_len[refs.own_to_merged ? refs.own_to_merged[ref] : ref]) return false;
    uint32_t bmn = f[3];
    uint32_t l_name = bmn & 0xFFu;
    if (l_name < 1) return false;
    uint32_t fnc = f[4];
    uint32_t n_cigar = fnc & 0xFFFFu;
    int32_t l_seq = (int32_t)f[5];
    if (l_seq < 0) return false;
    int32_t nref = (int32_t)f[6];
    if (nref < -1 || nref >= refs.n_ref_own) return false;
    int32_t npos = (int32_t)f[7];
    if (npos < -1) return false;
    int64_t fixed = 32 + (int64_t)l_name + 4 * (int64_t)n_cigar + ((int64_t)l_seq + 1) / 2 + (int64_t)l_seq;
    if (bs < fixed) return false;
    if (o + 4 + (uint64_t)bs > limit) return false;
    // read name: printable, NUL only at the end
    // (eight characters per load, tested in registers: the true start's three records used to pay a round trip per character)
    const uint8_t* name = p + 36;
    const uint8_t last = name[l_name - 1];
    bool printable = true;
    for (uint32_t k = 0; k + 1 < l_name; k += 8) {
        uint64_t x;
        __builtin_memcpy(&x, name + k, 8);       // (may read a few bytes past the name: inside the record, or the slack behind the stream)
        const uint32_t n = l_name - 1 - k < 8u ? l_name - 1 - k : 8u;
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            const uint32_t ch = (uint32_t)(x >> (8 * i)) & 0xFFu;
            if (i < n && (ch < 33u || ch > 126u)) printable = false;
        }
        if (!printable) break;
    }
    if (last != 0 || !printable) return false;
    // CIGAR: known operations whose query-consuming lengths add up to l_seq (SAM 1.4; what every aligner and htslib
    // write).  A window that starts a byte or two off a true record passes every test above for a few percent of the
    // positions (the shifted fields stay in range) and, one time in a record length, chains into the true records
    // behind it -- a dozen decoys per chromosome-sized file without this test, none with it.
    if (n_cigar) {
        const uint8_t* cg = name + l_name;
        uint64_t qlen = 0;
        for (uint32_t k = 0; k < n_cigar; ++k) {
            const uint32_t op = ld32(cg + 4 * k);
            if ((op & 15u) > 8u) return false;
            if (cig_type(op) & 1u) qlen += op >> 4;
        }
        if (l_seq > 0 && qlen != (uint64_t)l_seq) return false;
    }
    *next = o + 4 + (uint64_t)bs;
    *sort_key_ref = (uint32_t)ref;
    *sort_key_pos = pos;
    return true;
}

// is `o` the start of a chain of three plausible, coordinate-sorted records (or of the last records of the run)?
__device__ bool plausible_chain(const uint8_t* W, uint64_t limit, uint64_t o, const RefTable& refs) {
    uint64_t n1, n2, n3;
    uint32_t r1, r2, r3;
    int32_t p1, p2, p3;
    if (!plausible_record(W, limit, o, refs, &n1, &r1, &p1)) return false;
    if (n1 == limit) return true;
    if (!plausible_record(W, limit, n1, refs, &n2, &r2, &p2)) return false;
    if (r2 < r1 || (r2 == r1 && p2 < p1)) return false;     // coordinate order (unmapped = 0xFFFFFFFF last)
    if (n2 == limit) return true;
    if (!plausible_record(W, limit, n2, refs, &n3, &r3, &p3)) return false;
    if (r3 < r2 || (r3 == r2 && p3 < p2)) return false;
    return true;
}

// walk the chain from `entry` through global memory until it leaves [.., block_end); returns the exit offset or
// kOffInvalid (repair path only)
__device__ uint64_t walk_block(const uint8_t* U, uint64_t limit, bool open_end, uint64_t entry, uint64_t block_end, uint32_t* count) {
    uint64_t o = entry;
    uint32_t n = 0;
    while (o < block_end) {
        if (o + 36 > limit) { *count = n; return open_end ? limit : kOffInvalid; }      // (open end: the record belongs to the next batch)
        int64_t bs = (int32_t)ld32(U + o);
        if (bs < 32) { *count = n; return kOffInvalid; }
        if (o + 4 + (uint64_t)bs > limit) { *count = n; return open_end ? limit : kOffInvalid; }
        ++n;
        o += 4 + (uint64_t)bs;
    }
    *count = n;
    return o;
}

// wave-uniform copy of lane i's 64-bit value (readlane works on 32-bit ints: cast each half to
// uint32_t before widening, or bit 31 of the low half sign-extends into the high half)
__device__ __forceinline__ uint64_t bcast64(uint64_t v, uint32_t i) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, i);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), i);
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// Serial repair from the first inconsistent block on (one wavefront; rare).  Everything before
// `from` is consistent with the exactly known start of its run, hence true; from there the chain
// is followed block by block: a block whose recorded entry equals the running offset keeps its
// recorded walk (O(1)), any other block is re-walked from the true entry.  Every run starts over at
// its known first record.
__global__ __launch_bounds__(64) void k_chain_repair(const uint8_t* __restrict__ U, const uint64_t* __restrict__ out_off,
                                                      const uint32_t* __restrict__ isize, const uint32_t* __restrict__ run_of,
                                                      const ChainRun* __restrict__ runs, uint32_t n_blocks, uint32_t from,
                                                      uint64_t* entry, uint64_t* exit_, uint32_t* count, uint32_t* n_rewalked) {
    const uint32_t lane = threadIdx.x;
    uint64_t cur = kOffInvalid;
    {
        const ChainRun r0 = runs[run_of[from]];
        cur = (from == r0.blk_first) ? r0.u_beg : exit_[from - 1];
    }
    uint32_t rewalked = 0;
    for (uint32_t b0 = from; b0 < n_blocks; b0 += 64) {
        const uint32_t b = b0 + lane;
        uint64_t e = kOffUnknown, x = kOffUnknown, end = 0, rbeg = 0, rend = 0;
        uint32_t n = 0, first = 0, open = 0;
        if (b < n_blocks) {
            e = entry[b]; x = exit_[b]; n = count[b];
            const ChainRun r = runs[run_of[b]];
            end = out_off[b] + isize[b];
            if (end > r.u_end) end = r.u_end;
            rbeg = r.u_beg; rend = r.u_end;
            first = r.blk_first == b ? 1u : 0u;
            open = r.open_end;
        }
        const uint32_t lim = n_blocks - b0 < 64 ? n_blocks - b0 : 64;
        bool dirty = false;
        for (uint32_t i = 0; i < lim; ++i) {
            const uint64_t e_i = bcast64(e, i), x_i = bcast64(x, i), end_i = bcast64(end, i), rend_i = bcast64(rend, i);
            if (__builtin_amdgcn_readlane((int)first, i)) cur = bcast64(rbeg, i);
            uint64_t ne, nx;
            uint32_t nn;
            if (cur == kOffInvalid || cur == kOffUnknown) { ne = kOffInvalid; nx = kOffInvalid; nn = 0; }
            else if (e_i == cur && x_i != kOffUnknown && x_i != kOffInvalid) {
                ne = e_i; nx = x_i; nn = (uint32_t)__builtin_amdgcn_readlane((int)n, i);     // recorded walk confirmed
            } else {
                uint32_t c = 0;
                nx = walk_block(U, rend_i, __builtin_amdgcn_readlane((int)open, i) != 0, cur, end_i, &c);     // wave-uniform re-walk
                ne = cur;
                nn = c;
                ++rewalked;
            }
            if (lane == i && (ne != e || nx != x || nn != n)) { e = ne; x = nx; n = nn; dirty = true; }
            cur = nx;
        }
        if (dirty && b < n_blocks) { entry[b] = e; exit_[b] = x; count[b] = n; }
    }
    if (lane == 0) *n_rewalked += rewalked;
}

// ---- scan of the per-block counts (single workgroup, 3 phases; n_blocks is ~1e5..1e6) -----------
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void k_count_scan(const uint32_t* __restrict__ count, uint32_t n,
                                                              uint64_t* __restrict__ base) {
    // tiles of 4 x kScanThreads counts, four consecutive ones per thread (one 16-byte load, coalesced); a shuffle scan inside the
    // wavefront, the sixteen wave totals through LDS, the running sum carried from tile to tile.  (Until round 6 every thread summed its
    // own stretch of n / 1024 counts, one strided load at a time, around a Hillis-Steele scan of the 1024 partials: 58 us for the 32 k
    // chunk lengths of a piece of K6's text, a sixth of what formatting the piece took.)
    __shared__ uint64_t wtot[kScanThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    uint64_t carry = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 4u * kScanThreads) {
        const uint32_t i = i0 + 4u * t;
        uint32_t v[4] = {0, 0, 0, 0};
        if (i + 4u <= n) {
            const uint4 q = *(const uint4*)(count + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            for (uint32_t k = 0; k < 4u; ++k) if (i + k < n) v[k] = count[i + k];
        }
        const uint64_t s = (uint64_t)v[0] + v[1] + v[2] + v[3];
        uint64_t incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = (uint64_t)__shfl_up((unsigned long long)incl, d, 64);
            if ((int)lane >= d) incl += o;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        uint64_t before = carry, tile = 0;
#pragma unroll
        for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
            const uint64_t x = wtot[w];
            if (w < wv) before += x;
            tile += x;
        }
        uint64_t run = before + incl - s;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (i + k < n) base[i + k] = run;
            run += v[k];
        }
        carry += tile;
        __syncthreads();
    }
    if (t == 0) base[n] = carry;
}

// ---- aux fields, the -F program, RG lookup -------------------------------------------------------
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

// A -F program of flag tests, integer fields and and / or / not only -- the default filter of `depth` (mapping_quality > 0 and not
// duplicate and not failed_quality_control, depth.d:1159) and most filters people write -- evaluated without the interpreter's string,
// tag and regular-expression machinery: the describe kernel built around it needs half the registers and runs at twice the occupancy
// (k_describe_blocks_simple below).  kernels.hpp: filter_op_is_simple says which operations it knows.
__device__ __forceinline__ bool eval_filter_simple(const DeviceFilter* f, const uint8_t* p /* at refID */, int32_t ref, int32_t pos, uint32_t bmn,
                                                   uint32_t fnc, int32_t l_seq) {
    uint64_t stack = 0;
    int sp = 0;
    const uint32_t flag = fnc >> 16, mapq = (bmn >> 8) & 0xFF;
    const auto* fc = as_constant(f);
    const int n_ops = fc->n_ops;
    for (int i = 0; i < n_ops; ++i) {
        const sbx_filter_op op = filter_op(f, i);
        bool v = true;
        switch (op.kind) {
            case 0: v = (flag & op.mask) != 0; break;
            case 1: v = (flag & 1) && !(flag & 4) && !(flag & 8) && ref != (int32_t)ld32(p + 20); break;
            case 2: {
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
                v = cmp_op<int64_t>(op.cmp, x, op.value);
                break;
            }
            case 12: v = false; break;
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

__device__ bool eval_filter(const DeviceFilter* f, const uint8_t* p /* at refID */, int32_t ref, int32_t pos, uint32_t bmn,
                            uint32_t fnc, int32_t l_seq, const uint8_t* tags, const uint8_t* tags_end) {
    // postfix program over a tiny bool stack (bit stack in a 64-bit word); the fields every record's
    // walk has loaded anyway come in registers
    uint64_t stack = 0;
    int sp = 0;
    uint32_t flag = fnc >> 16, mapq = (bmn >> 8) & 0xFF;
    const int n_ops = as_constant(f)->n_ops;
    for (int i = 0; i < n_ops; ++i) {
        const sbx_filter_op op = filter_op(f, i);
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


// ---- one record ---------------------------------------------------------------------------------
struct Described {
    RecDesc d;
    int32_t ref;
    uint64_t hash;
    uint32_t t0, t1;        // first / last position tile the alignment touches (admitted records only)
    bool admit, bad, urg;
};

// p: the record's block_size field.  kStaged: p is the lane's staging slot in LDS -- the first kStageHead bytes of the record (fixed
// part, name and CIGAR fit) -- and `tail_end` points behind the staged copy of the record's last kStageTail bytes (the tags fit);
// otherwise p is the record in global memory.  The rg table comes as the view the kernel prepared (LDS copy or the global arrays).
template <bool kStaged, bool kSimpleFilter = false>
__device__ __forceinline__ Described describe_record(const uint8_t* p, const uint8_t* tail_end, uint64_t o, const IndexArgs& a, const RgTable& rgv) {
    Described R;
    const int64_t bs = (int32_t)ld32(p);
    const uint8_t* r = p + 4;
    const int32_t ref_own = (int32_t)ld32(r), pos = (int32_t)ld32(r + 4);
    // (adjustTagsInRange, multireader.d:174-190: the id of the merged dictionary is what every later stage sees)
    const int32_t ref = (a.refs.own_to_merged && ref_own >= 0 && ref_own < a.refs.n_ref_own) ? a.refs.own_to_merged[ref_own] : ref_own;
    const uint32_t bmn = ld32(r + 8), fnc = ld32(r + 12);
    const uint32_t l_name = bmn & 0xFF, mapq = (bmn >> 8) & 0xFF;
    const uint32_t n_cigar = fnc & 0xFFFF, flag = fnc >> 16;
    const int32_t l_seq = (int32_t)ld32(r + 16);
    RecDesc& d = R.d;
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
    R.ref = ref;
    R.hash = 0;
    R.t0 = R.t1 = 0;
    R.urg = false;
    const int64_t fixed = 32 + (int64_t)l_name + 4 * (int64_t)n_cigar + ((int64_t)(l_seq < 0 ? 0 : l_seq) + 1) / 2 + (l_seq < 0 ? 0 : l_seq);
    const bool sane = l_seq >= 0 && bs >= fixed && ref_own >= -1 && ref_own < a.refs.n_ref_own;
    R.bad = !sane;
    bool admit = sane && !(flag & 0x4) && ref >= 0;                       // read.d:256, unmapped reads cover nothing
    const uint8_t* const tags_end = kStaged ? tail_end : r + bs;
    const uint8_t* const tags = kStaged ? tail_end - (bs - fixed) : r + fixed;
    if (admit) admit = kSimpleFilter ? eval_filter_simple(a.filt, r, ref, pos, bmn, fnc, l_seq)
                                     : eval_filter(a.filt, r, ref, pos, bmn, fnc, l_seq, tags, tags_end);      // filtering.d:36-38
    if (admit) {
        // basesCovered + shape of the CIGAR
        const uint8_t* cg = r + 32 + l_name;
        int64_t span = 0;
        uint32_t q_lead = 0;         // query bases before the first reference-consuming op
        uint32_t runs = 0;           // number of maximal runs of M/=/X
        bool in_run = false, other_ref = false, seen_ref = false, zero_ref = false;
        for (uint32_t i = 0; i < n_cigar; ++i) {
            const uint32_t op = ld32(cg + 4 * i);
            const uint32_t ty = cig_type(op), len = op >> 4;
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
                    else in_run = false;       // I or trailing S: ends the run (a trailing clip is harmless for the fast path)
                }
                // H / P (ty == 0) neither end a run nor consume anything
            }
        }
        if (span <= 0 || span > 0x7FFFFFFF - (int64_t)pos) admit = false;   // pileup.d:510
        else {
            d.end = pos + (int32_t)span;
            if (runs == 1 && !other_ref && !zero_ref && q_lead <= 0xFFFF) { d.kind = 1; d.q_start = (uint16_t)q_lead; }
            else d.kind = 2;
        }
    }
    if (admit && a.refs.sel) {
        // the read must overlap one of the requested regions: pos < region.end && pos + span > region.start
        uint32_t lo = a.refs.sel_first[ref], hi = a.refs.sel_first[ref + 1];
        while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if ((int64_t)a.refs.sel[m].end > (int64_t)pos) hi = m; else lo = m + 1; }
        if (lo >= a.refs.sel_first[ref + 1] || (int64_t)a.refs.sel[lo].start >= (int64_t)d.end) { admit = false; d.kind = 0; d.end = d.pos; }
    }
    if (admit && a.own_ref >= 0 && (ref != a.own_ref || (uint32_t)pos < a.own_beg || (uint32_t)pos >= a.own_end)) {
        admit = false; d.kind = 0; d.end = d.pos;       // owned by another shard (reads partitioned by start position)
    }
    // RG -> sample.  The reference builds a CustomBamRead -- and throws on a read group that is not in the header -- for
    // every read it iterates over, filtered or not (depth.d:240-250,1211-1214); with -L those are the reads the index
    // fetch returns, which the admitted ones stand for here.
    if (a.rg.lookup && sane && (admit || !a.refs.sel)) {
        const uint32_t s = lookup_sample(tags, tags_end, rgv);
        if (s == 0xFFFFu) R.urg = true;
        else d.sample = (uint16_t)s;
    }
    if (admit) {
        // (the records of a wavefront lie on one contig but for a handful of blocks: its tile base comes through the scalar unit -- a
        //  vector load here would queue behind the next turn's prefetched bytes and stall the parse for their whole latency)
        const int32_t ref_u = __builtin_amdgcn_readfirstlane(ref);
        const uint32_t tb_u = as_constant(a.refs.tile_base)[ref_u], tbn_u = as_constant(a.refs.tile_base)[ref_u + 1];
        uint32_t tb = tb_u, tbn = tbn_u;
        if (ref != ref_u) { tb = a.refs.tile_base[ref]; tbn = a.refs.tile_base[ref + 1]; }
        uint32_t t0 = tb + (uint32_t)pos / a.tile_pos;
        uint32_t t1 = tb + (uint32_t)(d.end - 1) / a.tile_pos;
        // An alignment may hang over the end of its contig -- the reference's pileup has no notion of a contig's length and makes a
        // column of every position a read covers (pileup.d:345-397) -- so a contig has spare tiles behind its last position
        // (RefTable::tile_base).  One that reaches beyond them is reported (IndexStats::over_tiles): the host enlarges the spare
        // region and repeats the pass; what this pass computes for the clipped record is never used.
        const uint32_t last_tile = tbn - 1;
        if (t1 > last_tile) {
            atomicMax(&a.stats->over_tiles, t1 - last_tile);
            t1 = last_tile;
            if (t0 > last_tile) t0 = last_tile;
        }
        R.t0 = t0; R.t1 = t1;
    }
    if (a.name_hash) {   // FNV-1a over the read name without its NUL (CustomBamRead, depth.d:252-258)
        uint64_t h = 14695981039346656037ULL;
        const uint8_t* nm = r + 32;
        for (uint32_t k = 0; k + 1 < l_name; ++k) { h ^= nm[k]; h *= 1099511628211ULL; }
        R.hash = h;
    }
    R.admit = admit;
    return R;
}

// ---- 1. walk ------------------------------------------------------------------------------------------
constexpr int kWalkThreads = 64;

// the block's record offsets (u16, relative to the block's first byte) live in its slice of the literal stream
__device__ __forceinline__ uint16_t* rec_list(uint8_t* scratch, uint64_t out_off_b, uint32_t b) {
    return (uint16_t*)(scratch + inflate_lit_offset(out_off_b, b));
}

// walk block b from entry E (known or guessed), leave its record offsets in the block's list, record entry / exit / count
__device__ void walk_block_from(const IndexArgs& a, uint32_t b, uint64_t E) {
    const uint64_t beg = a.out_off[b], blk_end = beg + a.isize[b];
    const ChainRun run = a.runs[a.run_of[b]];
    const uint64_t lo = beg > run.u_beg ? beg : run.u_beg, hi0 = blk_end < run.u_end ? blk_end : run.u_end;
    const uint64_t hi = hi0 > lo ? hi0 : lo;
    const bool last_of_run = b == run.blk_last, open_end = run.open_end != 0;
    uint32_t n = 0;
    uint64_t X = E;
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    u32x4v grp = {0, 0, 0, 0};                  // the last 8 offsets, oldest in the low half of .x
    uint16_t* list = rec_list(a.scratch, beg, b);
    if (E != kOffUnknown && E != kOffInvalid && E >= lo) {
        uint64_t o = E, straddler = kOffUnknown;
        bool okc = true;
        while (o < hi) {
            // a record that does not end inside the run: an error -- except in an open-ended run (ChainRun), where it belongs to the
            // next batch: its start is left in flags[8..9] and the chain leaves the run at its end
            if (o + 36 > run.u_end) { okc = open_end; straddler = o; break; }
            const int64_t bs = (int32_t)ld32(a.U + o);
            if (bs < 32) { okc = false; break; }
            if (o + 4 + (uint64_t)bs > run.u_end) { okc = open_end; straddler = o; break; }
            grp.x = (grp.x >> 16) | (grp.y << 16);
            grp.y = (grp.y >> 16) | (grp.z << 16);
            grp.z = (grp.z >> 16) | (grp.w << 16);
            grp.w = (grp.w >> 16) | ((uint32_t)(o - beg) << 16);
            ++n;
            if ((n & 7u) == 0) *(u32x4v*)(list + (n - 8)) = grp;
            o += 4 + (uint64_t)bs;
        }
        X = okc ? o : kOffInvalid;
        if (okc && straddler != kOffUnknown) {
            X = run.u_end;
            atomicMin((unsigned long long*)(a.flags + 8), (unsigned long long)straddler);
        }
        if (n & 7u) {
            for (uint32_t k = n & 7u; k < 8; ++k) {
                grp.x = (grp.x >> 16) | (grp.y << 16);
                grp.y = (grp.y >> 16) | (grp.z << 16);
                grp.z = (grp.z >> 16) | (grp.w << 16);
                grp.w = grp.w >> 16;
            }
            *(u32x4v*)(list + (n & ~7u)) = grp;
        }
    } else if (E != kOffUnknown && E != kOffInvalid) {
        X = kOffInvalid;        // an entry below the block: impossible for a true chain
    }
    a.entry[b] = E;
    a.exit_[b] = X;
    a.count[b] = n;
    const bool bad_block = E == kOffUnknown || E == kOffInvalid || X == kOffUnknown || X == kOffInvalid || X < E ||
                           (last_of_run && X != run.u_end);
    if (bad_block) atomicMin(a.flags + 0, b);
}

// The guess: the first offset of the block at which a chain of three plausible records starts.  ONE WAVE per block, one candidate offset
// per lane -- 64 consecutive bytes are one or two 128-byte lines for the whole wavefront, and nearly every candidate is rejected by its
// first load.  (Until round 6 the lane that walks the block scanned the candidates itself, one dependent load after the other, 64 lanes
// in 64 different blocks: ~45 M extra requests per chromosome -- as many as the walk itself makes -- and the walk took 2.4 ms where
// tools/calib5 says its 50 M scattered lines cost 1.0.)  Writes entry[b]; k_walk_blocks takes it from there.
constexpr int kGuessThreads = 256;
__global__ __launch_bounds__(kGuessThreads) void k_guess_entries(IndexArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * (kGuessThreads / 64) + (threadIdx.x >> 6);
    if (b >= a.n_blocks) return;
    const uint64_t beg = a.out_off[b], blk_end = beg + a.isize[b];
    const ChainRun run = a.runs[a.run_of[b]];
    const uint64_t lo = beg > run.u_beg ? beg : run.u_beg, hi0 = blk_end < run.u_end ? blk_end : run.u_end;
    const uint64_t hi = hi0 > lo ? hi0 : lo;
    uint64_t E = kOffUnknown;
    if (b == run.blk_first) E = run.u_beg;
    else {
        for (uint64_t base = lo; base < hi; base += 64) {
            const uint64_t o = base + lane;
            const bool ok = o < hi && plausible_chain(a.U, run.u_end, o, a.refs);
            const uint64_t m = __ballot(ok);
            if (m) { E = base + (uint64_t)__builtin_ctzll(m); break; }
        }
    }
    if (lane == 0) a.entry[b] = E;
}

__global__ __launch_bounds__(kWalkThreads) void k_walk_blocks(IndexArgs a) {
    const uint32_t b = blockIdx.x * kWalkThreads + threadIdx.x;
    if (b >= a.n_blocks) return;
    // (the entry: given by the caller after a repair, or guessed by k_guess_entries -- the first block of a run starts where the run does)
    const ChainRun run = a.runs[a.run_of[b]];
    walk_block_from(a, b, b == run.blk_first ? run.u_beg : a.entry_in[b]);
    if (a.inflate_status[b] != 0) atomicMin(a.flags + 1, b);
}

// Parallel repair of wrong guesses: every block that is not entered where its predecessor was left is walked again
// from there.  A wrong guess is isolated (its neighbours guessed right), so one round settles it; a block whose new
// exit disagrees with the next block's entry is caught by the next round.  *n_changed counts the blocks re-walked.
// A round reads exit_[b - 1] while other lanes of the same launch may be rewriting it (a block and its predecessor both being
// re-walked), so what a round sees is not deterministic and "nothing changed" is only a hint for the host loop: the chain this
// produces is NEVER accepted as it stands -- the engine always launches walk + k_check_scan again with the entries given
// (entry_in), and k_check_scan verifies exit[b - 1] == entry[b] for every block from the exactly known start of the run.  A
// chain that is still inconsistent then is reported as a corrupt file (engine.cpp: `entries_given` branch).
__global__ __launch_bounds__(kWalkThreads) void k_rewalk_mismatched(IndexArgs a, uint32_t* n_changed) {
    const uint32_t b = blockIdx.x * kWalkThreads + threadIdx.x;
    if (b >= a.n_blocks) return;
    const ChainRun run = a.runs[a.run_of[b]];
    if (b == run.blk_first) return;
    const uint64_t want = a.exit_[b - 1];
    if (want == kOffUnknown || want == kOffInvalid || want == a.entry[b]) return;
    walk_block_from(a, b, want);
    atomicAdd(n_changed, 1u);
}

// ---- 2. chain check + scan of the per-block counts (single workgroup; n_blocks is ~1e5..1e6) ------------------
// state[b] = 2 << 62 | records in blocks 0..b (the engine reads the last one); flags[0] = lowest inconsistent block,
// flags[2] != 0: the descriptor array is too small for the total
__global__ __launch_bounds__(kScanThreads) void k_check_scan(IndexArgs a) {
    // 4 x 1024 blocks per step, coalesced; the loads of all four sub-steps are issued before the first scan, so that the single
    // workgroup pays one memory round trip per 4096 blocks (it is the serial link between the walk and the describe launches);
    // inclusive scan inside the wave by shuffles, wave totals through LDS
    constexpr int kSub = 4;
    __shared__ uint64_t wtot[kSub][kScanThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6, n = a.n_blocks;
    uint64_t run = 0;                    // records in the blocks before this step (workgroup-uniform)
    uint32_t first_bad = 0xFFFFFFFFu;
    for (uint32_t i0 = 0; i0 < n; i0 += kSub * kScanThreads) {
        uint64_t v[kSub], incl[kSub];
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            const uint32_t i = i0 + (uint32_t)u * kScanThreads + t;
            v[u] = 0;
            if (i < n) {
                v[u] = a.count[i];
                const ChainRun r = a.runs[a.run_of[i]];
                if (i != r.blk_first && a.exit_[i - 1] != a.entry[i] && first_bad == 0xFFFFFFFFu) first_bad = i;
            }
        }
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            uint64_t x = v[u];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint64_t o = (uint64_t)__shfl_up((unsigned long long)x, d, 64);
                if ((int)lane >= d) x += o;
            }
            incl[u] = x;
            if (lane == 63) wtot[u][wv] = x;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            const uint32_t i = i0 + (uint32_t)u * kScanThreads + t;
            uint64_t before = 0, all = 0;
#pragma unroll
            for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
                const uint64_t x = wtot[u][w];
                before += w < wv ? x : 0;
                all += x;
            }
            if (i < n) a.state[i] = (2ull << 62) | (run + before + incl[u]);
            run += all;
        }
        __syncthreads();
    }
    if (first_bad != 0xFFFFFFFFu) atomicMin(a.flags + 0, first_bad);
    if (t == 0 && run > a.desc_cap) atomicOr(a.flags + 2, 1u);
}

// The same over SEVERAL workgroups in one launch (round 5: the single workgroup above is a serial link of 0.5 ms between the walk and the
// describe launches of a chromosome).  Workgroup g takes blocks [4096 g, 4096 g + 4096): chain check, local scan, then it PUBLISHES its total
// (`part[g]` = total | ready bit) and adds up the totals of the workgroups before it.  g is a TICKET taken on entry (scan_ticket), so
// the workgroups g waits for hold earlier tickets: they are running -- a workgroup takes its ticket when it starts -- and never wait for
// a later one; the look-back cannot deadlock however large the grid is (a whole genome has 3.1 M tiles = 757 workgroups, more than the
// device holds at once: the later ones simply start as the earlier ones finish).
constexpr uint32_t kScanPerWg = 4 * kScanThreads;
constexpr unsigned long long kScanReady = 1ull << 63;
constexpr uint32_t kScanMaxWgs = kScanPartWords - 2;      // (the last word of `part` is the ticket counter)

// the workgroup's place in the scan: a ticket, not blockIdx -- whoever holds ticket g knows that tickets 0 .. g - 1 have been taken by
// workgroups that are running (or done), whatever order the hardware started them in; the counter is the last word of `part`
__device__ __forceinline__ uint32_t scan_ticket(unsigned long long* part) {
    __shared__ uint32_t ticket;
    if (threadIdx.x == 0) ticket = (uint32_t)atomicAdd(part + (kScanPartWords - 1), 1ull);
    __syncthreads();
    return ticket;
}

__device__ __forceinline__ uint64_t lookback_sum(unsigned long long* part, uint32_t g, uint64_t* red /*[kScanThreads / 64]*/) {
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    uint64_t acc = 0;
    for (uint32_t w = t; w < g; w += kScanThreads) {
        unsigned long long x;
        do { x = __hip_atomic_load(part + w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (!(x & kScanReady));
        acc += x & ~kScanReady;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += (uint64_t)__shfl_down((unsigned long long)acc, d, 64);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    uint64_t all = 0;
#pragma unroll
    for (uint32_t w = 0; w < kScanThreads / 64; ++w) all += red[w];
    __syncthreads();
    return all;
}

__global__ __launch_bounds__(kScanThreads) void k_check_scan_mw(IndexArgs a, unsigned long long* part) {
    constexpr int kSub = 4;
    __shared__ uint64_t wtot[kSub][kScanThreads / 64];
    __shared__ uint64_t red[kScanThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6, n = a.n_blocks;
    const uint32_t g = scan_ticket(part);
    const uint32_t i0 = g * kScanPerWg;
    uint32_t first_bad = 0xFFFFFFFFu;
    uint64_t v[kSub], incl[kSub];
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
        const uint32_t i = i0 + (uint32_t)u * kScanThreads + t;
        v[u] = 0;
        if (i < n) {
            v[u] = a.count[i];
            const ChainRun r = a.runs[a.run_of[i]];
            if (i != r.blk_first && a.exit_[i - 1] != a.entry[i] && first_bad == 0xFFFFFFFFu) first_bad = i;
        }
    }
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
        uint64_t x = v[u];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = (uint64_t)__shfl_up((unsigned long long)x, d, 64);
            if ((int)lane >= d) x += o;
        }
        incl[u] = x;
        if (lane == 63) wtot[u][wv] = x;
    }
    __syncthreads();
    uint64_t before[kSub], total = 0;
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
        uint64_t b = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
            const uint64_t x = wtot[u][w];
            b += w < wv ? x : 0;
            all += x;
        }
        before[u] = total + b;
        total += all;
    }
    if (t == 0) __hip_atomic_store(part + g, (unsigned long long)total | kScanReady, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t run = lookback_sum(part, g, red);
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
        const uint32_t i = i0 + (uint32_t)u * kScanThreads + t;
        if (i < n) a.state[i] = (2ull << 62) | (run + before[u] + incl[u]);
    }
    if (first_bad != 0xFFFFFFFFu) atomicMin(a.flags + 0, first_bad);
    if (t == 0 && g == gridDim.x - 1 && run + total > a.desc_cap) atomicOr(a.flags + 2, 1u);       // (the last ticket: run + total = all records)
}

// ---- 3. describe: one wave per block, one lane per record ---------------------------------------------------
constexpr int kDescThreads = 256;

// (Round 6 measured the alternative to the per-lane slots below -- the wave copies the longest run of whole records that fits 8 KB to LDS
// with coalesced 16-byte loads (global_load_lds_dwordx4: no registers, every load of the piece in flight at once) and every lane parses its
// record from the piece: the whole stream once, coalesced, instead of head and tail of every record as scattered lines.  Correct (138 GPU
// tests, whole-share parity at config 2's size) and SLOWER: record_index 6.9 -> 8.2 ms on config 2 (8.9 with two 16-byte loads in flight
// through registers).  A piece holds 28 records, not 64: the parser's dependent LDS reads (name, CIGAR, the tag scan for RG:Z) cost a turn
// the same whatever number of lanes is busy, and there are 2.3 x the turns.  profiles/round6/call_k_describe_streamed_*.json.)
// Staging (round 5).  A lane used to follow its record through global memory field by field -- block_size, the fixed part, the CIGAR
// operations one by one, then the tag bytes one by one up to RG:Z and the read-group table byte by byte: about twenty DEPENDENT
// round trips per batch of 64 records; the kernel waits 79 % of its wave cycles (profiles/round4/pmc_sq_config2_full.csv), and the
// hypothesis was that it waits for that chain.  (It does not: see the note at the kernels below.)  Now the record offsets of the batch give every lane the start of its record AND of the
// next one, so the head (kStageHead bytes from the record's start) and the tail (the kStageTail bytes in front of the next record: the
// tags of a record without a long tag list) are fetched with seven independent 16-byte loads, copied to the lane's slot in LDS, and
// everything is parsed from there -- two round trips to global memory per batch: the offsets, the bytes.  The read-group table
// is copied to LDS once per workgroup.  A record that does not fit (long name or CIGAR, more than kStageTail bytes of tags, a
// filter that reads bases or qualities, a size that contradicts the chain) takes the old path through global memory.
constexpr uint32_t kStageHead = 64, kStageTail = 48, kStageSlot = kStageHead + kStageTail;
constexpr uint32_t kRgLdsIds = 256, kRgLdsMax = 16;
static_assert(kRgLdsIds <= (uint32_t)kDescThreads, "the read-group ids are copied to LDS one byte per thread");

template <bool kStage, bool kSimpleFilter = false>
__device__ __forceinline__ void describe_blocks_body(const IndexArgs& a) {
    __shared__ uint32_t tot[7];          // records, admitted, malformed, unknown read group of this workgroup's blocks; bytes K3 reads; longest span
    __shared__ __attribute__((aligned(16))) uint8_t stage[kDescThreads * kStageSlot];
    __shared__ __attribute__((aligned(4))) char rg_ids[kRgLdsIds];
    __shared__ uint32_t rg_off[kRgLdsMax];
    __shared__ uint16_t rg_sample[kRgLdsMax];
    if (threadIdx.x < 7) tot[threadIdx.x] = 0;
    RgTable rgv = a.rg;
    const bool rg_small = kStage && a.rg.lookup && a.rg.n_rg <= (int32_t)kRgLdsMax && a.rg.ids_bytes <= kRgLdsIds;
    if (rg_small) {
        if (threadIdx.x < a.rg.ids_bytes) rg_ids[threadIdx.x] = a.rg.ids[threadIdx.x];
        if (threadIdx.x < (uint32_t)a.rg.n_rg) { rg_off[threadIdx.x] = a.rg.id_off[threadIdx.x]; rg_sample[threadIdx.x] = a.rg.sample_of[threadIdx.x]; }
        rgv.ids = rg_ids; rgv.id_off = rg_off; rgv.sample_of = rg_sample;
    }
    // a filter that reads bases or qualities needs the body of the record: no staging
    bool filt_body = false;
    for (int k = 0; k < as_constant(a.filt)->n_ops; ++k) {
        const sbx_filter_op op = filter_op(a.filt, k);
        filt_body = filt_body || op.kind == 13 || (op.kind == 15 && op.field == 1) || (op.kind == 2 && op.field == 7);
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = blockIdx.x * (kDescThreads / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform for the compiler too)
    // (descriptor array too small: nothing is written, the host enlarges it and launches again)
    // (b is wave-uniform and none of these tables is written by this kernel: scalar loads)
    const uint32_t count = (b < a.n_blocks && !as_constant(a.flags)[2]) ? as_constant(a.count)[b] : 0u;
    uint32_t n_adm = 0, n_bad = 0, n_urg = 0, b_seq = 0, b_qual = 0, m_span = 0;
    if (count) {
        const uint64_t beg = as_constant(a.out_off)[b];
        const uint64_t base = (as_constant(a.state)[b] & kStateMask) - count;
        const uint16_t* list = rec_list(a.scratch, beg, b);
        const uint64_t chain_exit = as_constant(a.exit_)[b];        // where the record chain leaves the block: the end of its last record
        uint8_t* const slot = stage + threadIdx.x * kStageSlot;
        typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
        // The kernel around the simple evaluator keeps three turns in flight: the record offsets of turn k + 2, the head and tail bytes of
        // turn k + 1 (seven 16-byte loads per lane, issued right after turn k's bytes have gone to LDS) and the parse of turn k -- a turn
        // costs two dependent round trips to HBM otherwise (the offsets, then the bytes), and five waves per SIMD do not hide them.  The
        // kernel around the interpreter has no registers to hold a turn: it fetches, waits and parses turn after turn.
        struct Offs { uint32_t lo, hi; };        // block-relative start of the lane's record and of the one behind it
        struct Fetch { u32x4s h[4], t[3]; uint64_t o, o_next; bool live, can; };
        auto load_offs = [&](uint32_t i0) {
            Offs f{0u, 0u};
            const uint32_t i = i0 + lane;
            if (i < count) { f.lo = list[i]; f.hi = i + 1 < count ? (uint32_t)list[i + 1] : 0xFFFFFFFFu; }
            return f;
        };
        auto issue = [&](uint32_t i0, const Offs& f) {
            Fetch x;
            x.live = i0 + lane < count;
            x.o = beg + f.lo;
            x.o_next = f.hi != 0xFFFFFFFFu ? beg + f.hi : chain_exit;
            // (the head may reach up to 64 bytes past the stream: the allocation has that slack, IndexArgs::u_alloc)
            x.can = x.live && kStage && !filt_body && x.o_next >= x.o + 36 && x.o_next >= kStageTail && x.o_next <= a.u_alloc;
#pragma unroll
            for (int k = 0; k < 4; ++k) x.h[k] = u32x4s{0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 3; ++k) x.t[k] = u32x4s{0, 0, 0, 0};
            if (x.can) {
#pragma unroll
                for (int k = 0; k < 4; ++k) __builtin_memcpy(&x.h[k], a.U + x.o + 16 * k, 16);
#pragma unroll
                for (int k = 0; k < 3; ++k) __builtin_memcpy(&x.t[k], a.U + x.o_next - kStageTail + 16 * k, 16);
            }
            return x;
        };
        Offs offs{0u, 0u};
        Fetch cur;
        if (kSimpleFilter) {
            offs = load_offs(0);
            cur = issue(0, offs);
            offs = load_offs(64);
        }
        for (uint32_t i0 = 0; i0 < count; i0 += 64) {
            if (!kSimpleFilter) cur = issue(i0, load_offs(i0));
            const uint32_t i = i0 + lane;
            const bool live = cur.live;
            Described R;
            R.admit = false; R.bad = false; R.urg = false; R.t0 = R.t1 = 0;
            const uint64_t idx = base + i;
            const uint64_t o = cur.o, o_next = cur.o_next;
            bool staged = cur.can;
            if (staged) {
#pragma unroll
                for (int k = 0; k < 4; ++k) *(u32x4s*)(slot + 16 * k) = cur.h[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) *(u32x4s*)(slot + kStageHead + 16 * k) = cur.t[k];
                // does the record fit the slot, and does its size agree with the chain?
                const int64_t bs = (int32_t)cur.h[0].x;
                const uint32_t bmn = cur.h[0].w, fnc = cur.h[1].x;
                const int32_t l_seq = (int32_t)cur.h[1].y;
                const int64_t ln = bmn & 0xFFu, nc = fnc & 0xFFFFu;
                const int64_t fixed = 32 + ln + 4 * nc + ((int64_t)(l_seq < 0 ? 0 : l_seq) + 1) / 2 + (l_seq < 0 ? 0 : l_seq);
                staged = l_seq >= 0 && bs >= fixed && o + 4 + (uint64_t)bs == o_next && 36 + ln + 4 * nc <= (int64_t)kStageHead &&
                         bs - fixed <= (int64_t)kStageTail;
            }
            // the next turn's bytes leave now and arrive while this turn is parsed; the offsets of the turn behind it follow them
            if (kSimpleFilter && i0 + 64 < count) {
                cur = issue(i0 + 64, offs);
                offs = load_offs(i0 + 128);
            }
            if (live) {
                if (staged) R = describe_record<true, kSimpleFilter>(slot, slot + kStageSlot, o, a, rgv);
                else R = describe_record<false, kSimpleFilter>(a.U + o, nullptr, o, a, rgv);
                a.desc[idx] = R.d;
                a.rec_ref[idx] = R.ref;
                if (a.name_hash) a.name_hash[idx] = R.hash;
                n_adm += R.admit ? 1u : 0u;
                if (R.admit) {
                    b_seq += 4u * R.d.n_cigar + ((R.d.l_seq + 1u) >> 1); b_qual += R.d.l_seq;
                    const uint32_t sp = (uint32_t)(R.d.end - R.d.pos);
                    m_span = sp > m_span ? sp : m_span;
                }
                n_bad += R.bad ? 1u : 0u;
                n_urg += R.urg ? 1u : 0u;
            }
            // records of a wave have consecutive indices, so the lowest index of a tile is held by the first lane of a run
            // of equal tiles and the highest by the last one
            const bool adm = live && R.admit;
            const uint32_t t0 = adm ? R.t0 : 0xFFFFFFFFu;
            const uint32_t t0_prev = __shfl_up(t0, 1, 64), t0_next = __shfl_down(t0, 1, 64);
            if (adm) {
                if (lane == 0 || t0_prev != t0) atomicMin(&a.tile_lo[t0], (uint32_t)idx);
                if (lane == 63 || t0_next != t0) atomicMax(&a.tile_hi[t0], (uint32_t)idx + 1);
                for (uint32_t t = R.t0 + 1; t <= R.t1; ++t) {
                    atomicMin(&a.tile_lo[t], (uint32_t)idx);
                    atomicMax(&a.tile_hi[t], (uint32_t)idx + 1);
                }
            }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            n_adm += __shfl_xor(n_adm, d, 64);
            n_bad += __shfl_xor(n_bad, d, 64);
            n_urg += __shfl_xor(n_urg, d, 64);
            b_seq += __shfl_xor(b_seq, d, 64);
            b_qual += __shfl_xor(b_qual, d, 64);
            const uint32_t o = __shfl_xor(m_span, d, 64);
            m_span = o > m_span ? o : m_span;
        }
        if (lane == 0) {
            atomicAdd(&tot[0], count);
            if (n_adm) atomicAdd(&tot[1], n_adm);
            if (n_bad) atomicAdd(&tot[2], n_bad);
            if (n_urg) atomicAdd(&tot[3], n_urg);
            if (b_seq) atomicAdd(&tot[4], b_seq);
            if (b_qual) atomicAdd(&tot[5], b_qual);
            if (m_span) atomicMax(&tot[6], m_span);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // kIndexStatSlots accumulators (the host adds them up): tens of thousands of workgroups do not queue up on one cache line
        IndexStats* st = a.stats + (blockIdx.x & (kIndexStatSlots - 1));
        if (tot[0]) atomicAdd(&st->n_records, (unsigned long long)tot[0]);
        if (tot[1]) atomicAdd(&st->n_admitted, (unsigned long long)tot[1]);
        if (tot[2]) atomicAdd(&st->n_bad, (unsigned long long)tot[2]);
        if (tot[3]) atomicAdd(&st->n_unknown_rg, (unsigned long long)tot[3]);
        if (tot[4]) atomicAdd(&st->adm_seq_bytes, (unsigned long long)tot[4]);
        if (tot[5]) atomicAdd(&st->adm_qual_bytes, (unsigned long long)tot[5]);
        if (tot[6]) atomicMax(&st->max_span, tot[6]);
    }
}

// the staged form at 4 waves per SIMD (124 VGPRs; at 5 waves -- 96 VGPRs and a dozen spills -- it is 0.2 ms slower), and the unstaged
// body (SBX_K2_DESCRIBE=0).  What staging is worth, measured (profiles/round5/README.md section 2): against the unstaged body of THIS
// build 1.07 ms of config 2's record_index -- but against round 4's kernel under the profiler 4.52 -> 4.38 ms, 3 %: the unstaged body
// compiled into this template is slower than round 4's kernel was.  `describe` does not wait for its chain of dependent loads; it runs
// at the rate of scattered 128-byte lines a CU sustains (the walk, with a quarter of the occupancy, runs at the same rate per line).
__global__ __launch_bounds__(kDescThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_describe_blocks(IndexArgs a) { describe_blocks_body<true>(a); }
// the same with the simple filter evaluator (IndexArgs::simple_filter: the host looked at the program)
__global__ __launch_bounds__(kDescThreads) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_describe_blocks_simple(IndexArgs a) { describe_blocks_body<true, true>(a); }

// ---- active tile compaction (single workgroup ballot scan; n_tiles ~ 1e3..2e6) ---------------------
__global__ __launch_bounds__(kScanThreads) void k_tile_compact(const uint32_t* __restrict__ tile_lo,
                                                                const uint32_t* __restrict__ tile_hi, uint32_t n_tiles, uint32_t deep_thr,
                                                                uint32_t* __restrict__ active, uint32_t* __restrict__ slot_of,
                                                                uint32_t* __restrict__ n_active) {
    // 4 x 1024 tiles per step, coalesced, the loads of the four sub-steps in flight together: rank inside the wave by ballot,
    // wave totals through LDS
    constexpr int kSub = 4;
    __shared__ uint32_t wtot[kSub][kScanThreads / 64];
    __shared__ uint32_t deep_total;
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    if (t == 0) deep_total = 0;
    uint32_t n_deep = 0;                 // this thread's tiles with >= 2^16 records (K3 keeps 32-bit counters for them)
    uint32_t run = 0;                    // active tiles before this step (workgroup-uniform)
    for (uint32_t i0 = 0; i0 < n_tiles; i0 += kSub * kScanThreads) {
        uint32_t lo_i[kSub], hi_i[kSub];
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            const uint32_t i = i0 + (uint32_t)u * kScanThreads + t;
            lo_i[u] = i < n_tiles ? tile_lo[i] : 0u;
            hi_i[u] = i < n_tiles ? tile_hi[i] : 0u;
        }
        uint64_t m[kSub];
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            const bool on = hi_i[u] > lo_i[u];
            n_deep += on && hi_i[u] - lo_i[u] >= deep_thr ? 1u : 0u;
            m[u] = __ballot(on);
            if (lane == 0) wtot[u][wv] = (uint32_t)__popcll(m[u]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            const uint32_t i = i0 + (uint32_t)u * kScanThreads + t;
            uint32_t before = 0, all = 0;
#pragma unroll
            for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
                const uint32_t x = wtot[u][w];
                before += w < wv ? x : 0u;
                all += x;
            }
            if (i < n_tiles) {
                if (hi_i[u] > lo_i[u]) {
                    const uint32_t slot = run + before + (uint32_t)__popcll(m[u] & ((1ull << lane) - 1ull));
                    active[slot] = i;
                    slot_of[i] = slot;
                } else {
                    slot_of[i] = 0xFFFFFFFFu;
                }
            }
            run += all;
        }
        __syncthreads();
    }
    if (n_deep) atomicAdd(&deep_total, n_deep);
    __syncthreads();
    if (t == 0) { n_active[0] = run; n_active[1] = deep_total; }
}

// several workgroups, one launch (see k_check_scan_mw): workgroup g ranks the active tiles of [4096 g, 4096 g + 4096) behind the
// active tiles of the workgroups before it; n_active[0] / [1] (zeroed by the launcher) collect the totals
__global__ __launch_bounds__(kScanThreads) void k_tile_compact_mw(const uint32_t* __restrict__ tile_lo, const uint32_t* __restrict__ tile_hi,
                                                                   uint32_t n_tiles, uint32_t deep_thr, uint32_t* __restrict__ active,
                                                                   uint32_t* __restrict__ slot_of, uint32_t* __restrict__ n_active,
                                                                   unsigned long long* part) {
    constexpr int kSub = 4;
    __shared__ uint32_t wtot[kSub][kScanThreads / 64];
    __shared__ uint64_t red[kScanThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const uint32_t g = scan_ticket(part);
    const uint32_t i0 = g * kScanPerWg;
    uint32_t lo_i[kSub], hi_i[kSub], n_deep = 0;
    uint64_t m[kSub];
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
        const uint32_t i = i0 + (uint32_t)u * kScanThreads + t;
        lo_i[u] = i < n_tiles ? tile_lo[i] : 0u;
        hi_i[u] = i < n_tiles ? tile_hi[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
        const bool on = hi_i[u] > lo_i[u];
        n_deep += on && hi_i[u] - lo_i[u] >= deep_thr ? 1u : 0u;
        m[u] = __ballot(on);
        if (lane == 0) wtot[u][wv] = (uint32_t)__popcll(m[u]);
    }
    __syncthreads();
    uint32_t before[kSub], total = 0;
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
        uint32_t b = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
            const uint32_t x = wtot[u][w];
            b += w < wv ? x : 0u;
            all += x;
        }
        before[u] = total + b;
        total += all;
    }
    if (t == 0) __hip_atomic_store(part + g, (unsigned long long)total | kScanReady, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t run = (uint32_t)lookback_sum(part, g, red);
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
        const uint32_t i = i0 + (uint32_t)u * kScanThreads + t;
        if (i < n_tiles) {
            if (hi_i[u] > lo_i[u]) {
                const uint32_t slot = run + before[u] + (uint32_t)__popcll(m[u] & ((1ull << lane) - 1ull));
                active[slot] = i;
                slot_of[i] = slot;
            } else {
                slot_of[i] = 0xFFFFFFFFu;
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) n_deep += __shfl_down(n_deep, d, 64);
    if (lane == 0 && n_deep) atomicAdd(n_active + 1, n_deep);
    if (t == 0 && g == gridDim.x - 1) n_active[0] = run + total;
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

void launch_index_blocks(const IndexArgs& a, hipStream_t stream) {
    if (!a.n_blocks) return;
    IndexArgs w = a;
    if (!a.entry_in) {       // no entries given: guess them (into entry[], which the walk rewrites with the same values)
        const uint32_t per_g = kGuessThreads / 64;
        hipLaunchKernelGGL(k_guess_entries, dim3((a.n_blocks + per_g - 1) / per_g), dim3(kGuessThreads), 0, stream, a);
        SBX_HIP(hipGetLastError());
        w.entry_in = a.entry;
    }
    hipLaunchKernelGGL(k_walk_blocks, dim3((a.n_blocks + kWalkThreads - 1) / kWalkThreads), dim3(kWalkThreads), 0, stream, w);
    SBX_HIP(hipGetLastError());
    // (scan_part: kScanMaxWgs words the launcher owns -- IndexArgs::scan_part; SBX_K2_SCAN=0 keeps the single workgroup)
    static const bool mw = [] { const char* e = getenv("SBX_K2_SCAN"); return !e || atoi(e) != 0; }();
    const uint32_t n_wg = (a.n_blocks + kScanPerWg - 1) / kScanPerWg;
    if (mw && a.scan_part && n_wg > 1 && n_wg <= kScanMaxWgs) {
        SBX_HIP(hipMemsetAsync(a.scan_part, 0, (size_t)kScanPartWords * 8, stream));
        hipLaunchKernelGGL(k_check_scan_mw, dim3(n_wg), dim3(kScanThreads), 0, stream, a, a.scan_part);
    } else
        hipLaunchKernelGGL(k_check_scan, dim3(1), dim3(kScanThreads), 0, stream, a);
    SBX_HIP(hipGetLastError());
    const uint32_t per = kDescThreads / 64;
    const dim3 dgrid((a.n_blocks + per - 1) / per), dblock(kDescThreads);
    if (a.simple_filter) hipLaunchKernelGGL(k_describe_blocks_simple, dgrid, dblock, 0, stream, a);
    else hipLaunchKernelGGL(k_describe_blocks, dgrid, dblock, 0, stream, a);
    SBX_HIP(hipGetLastError());
}

void launch_rewalk_mismatched(const IndexArgs& a, uint32_t* d_n_changed, hipStream_t stream) {
    if (!a.n_blocks) return;
    hipLaunchKernelGGL(k_rewalk_mismatched, dim3((a.n_blocks + kWalkThreads - 1) / kWalkThreads), dim3(kWalkThreads), 0, stream, a, d_n_changed);
    SBX_HIP(hipGetLastError());
}

void launch_chain_repair(const uint8_t* d_U, const uint64_t* d_out_off, const uint32_t* d_isize, const uint32_t* d_run_of,
                         const ChainRun* d_runs, uint32_t n_blocks, uint32_t from, uint64_t* d_entry, uint64_t* d_exit,
                         uint32_t* d_count, uint32_t* d_n_rewalked, hipStream_t stream) {
    hipLaunchKernelGGL(k_chain_repair, dim3(1), dim3(64), 0, stream, d_U, d_out_off, d_isize, d_run_of, d_runs, n_blocks, from,
                       d_entry, d_exit, d_count, d_n_rewalked);
    SBX_HIP(hipGetLastError());
}

size_t count_scan_tmp_bytes(uint32_t) { return 0; }

void launch_count_scan(const uint32_t* d_count, uint32_t n_blocks, uint64_t* d_base, void*, size_t, hipStream_t stream) {
    hipLaunchKernelGGL(k_count_scan, dim3(1), dim3(kScanThreads), 0, stream, d_count, n_blocks, d_base);
    SBX_HIP(hipGetLastError());
}

void launch_tile_compact(const uint32_t* d_tile_lo, const uint32_t* d_tile_hi, uint32_t n_tiles, uint32_t deep_thr, uint32_t* d_active,
                         uint32_t* d_slot_of, uint32_t* d_n_active, hipStream_t stream, unsigned long long* d_scan_part) {
    static const bool mw = [] { const char* e = getenv("SBX_K2_SCAN"); return !e || atoi(e) != 0; }();
    const uint32_t n_wg = (n_tiles + kScanPerWg - 1) / kScanPerWg;
    if (mw && d_scan_part && n_wg > 1 && n_wg <= kScanMaxWgs) {
        SBX_HIP(hipMemsetAsync(d_scan_part, 0, (size_t)kScanPartWords * 8, stream));
        SBX_HIP(hipMemsetAsync(d_n_active, 0, 8, stream));
        hipLaunchKernelGGL(k_tile_compact_mw, dim3(n_wg), dim3(kScanThreads), 0, stream, d_tile_lo, d_tile_hi, n_tiles, deep_thr, d_active,
                           d_slot_of, d_n_active, d_scan_part);
    } else
        hipLaunchKernelGGL(k_tile_compact, dim3(1), dim3(kScanThreads), 0, stream, d_tile_lo, d_tile_hi, n_tiles, deep_thr, d_active,
                           d_slot_of, d_n_active);
    SBX_HIP(hipGetLastError());
}

}  // namespace sbx
