#!/usr/bin/env python3
"""Generates tools/calib2.hip: saturated issue cost (cycles per wave-instruction per SIMD) of the integer / LDS
instructions the inflate kernels could be built from, on gfx950.  Development tool (see calib.hip)."""
# {d} = destination / accumulator of stream i (also usable as a source), {b} = per-lane operand, {c} = second operand, {l} = LDS address
OPS = [
    ("v_add_u32", "v_add_u32 {d}, {d}, {b}"),
    ("v_sub_u32", "v_sub_u32 {d}, {d}, {b}"),
    ("v_and_b32", "v_and_b32 {d}, {d}, {b}"),
    ("v_or_b32", "v_or_b32 {d}, {d}, {b}"),
    ("v_xor_b32", "v_xor_b32 {d}, {d}, {b}"),
    ("v_mov_b32", "v_mov_b32 {d}, {b}"),
    ("v_bfrev_b32", "v_bfrev_b32 {d}, {d}"),
    ("v_ffbh_u32", "v_ffbh_u32 {d}, {d}"),
    ("v_lshrrev_b32 imm", "v_lshrrev_b32 {d}, 3, {d}"),
    ("v_lshrrev_b32 reg", "v_lshrrev_b32 {d}, {c}, {d}"),
    ("v_lshlrev_b32 reg", "v_lshlrev_b32 {d}, {c}, {d}"),
    ("v_min_u32", "v_min_u32 {d}, {d}, {b}"),
    ("v_max_u32", "v_max_u32 {d}, {d}, {b}"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {d}, {d}, {b}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {d}, {b}, {c}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {d}, {d}, 2, {b}"),
    ("v_add_lshl_u32", "v_add_lshl_u32 {d}, {d}, {b}, 2"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {d}, 2, {b}"),
    ("v_and_or_b32", "v_and_or_b32 {d}, {d}, {b}, {c}"),
    ("v_or3_b32", "v_or3_b32 {d}, {d}, {b}, {c}"),
    ("v_add3_u32", "v_add3_u32 {d}, {d}, {b}, {c}"),
    ("v_xad_u32", "v_xad_u32 {d}, {d}, {b}, {c}"),
    ("v_bfe_u32", "v_bfe_u32 {d}, {d}, {c}, 9"),
    ("v_bfi_b32", "v_bfi_b32 {d}, {b}, {d}, {c}"),
    ("v_alignbit_b32", "v_alignbit_b32 {d}, {d}, {b}, {c}"),
    ("v_alignbyte_b32", "v_alignbyte_b32 {d}, {d}, {b}, {c}"),
    ("v_perm_b32", "v_perm_b32 {d}, {d}, {b}, {c}"),
    ("v_sad_u32", "v_sad_u32 {d}, {d}, {b}, {c}"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {d}, {d}, {b}"),
    ("v_cmp_lt_u32 vcc", "v_cmp_lt_u32 vcc, {d}, {b}"),
    ("v_cmp_lt_u32 e64 sgpr", "v_cmp_lt_u32_e64 s[20:21], {d}, {b}"),
    ("v_cmp_lt_u16 vcc", "v_cmp_lt_u16 vcc, {d}, {b}"),
    ("v_cmp+v_addc (pair, count as 2)", "v_cmp_lt_u32 vcc, {b}, {c}\n v_addc_co_u32 {d}, vcc, 0, {d}, vcc"),
    ("v_cmp+v_cndmask vcc (pair)", "v_cmp_lt_u32 vcc, {b}, {c}\n v_cndmask_b32 {d}, {d}, {b}, vcc"),
    ("v_cndmask_b32 vcc (set before)", "v_cndmask_b32 {d}, {d}, {b}, vcc"),
    ("v_cndmask_b32 e64 sgpr", "v_cndmask_b32_e64 {d}, {d}, {b}, s[22:23]"),
    ("v_addc_co_u32", "v_addc_co_u32 {d}, vcc, {d}, {b}, vcc"),
    ("v_sub_co_u32", "v_sub_co_u32 {d}, vcc, {d}, {b}"),
    ("v_cmp_sdwa word1", "v_cmp_lt_u32_sdwa vcc, {d}, {b} src0_sel:DWORD src1_sel:WORD_1"),
    ("v_add_u32_sdwa", "v_add_u32_sdwa {d}, {d}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"),
    ("v_add_u32 dpp row_shr", "v_add_u32_dpp {d}, {d}, {b} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_pk_add_u16", "v_pk_add_u16 {d}, {d}, {b}"),
    ("v_pk_sub_i16", "v_pk_sub_i16 {d}, {d}, {b}"),
    ("v_pk_lshrrev_b16", "v_pk_lshrrev_b16 {d}, 15, {d} op_sel_hi:[0,1]"),
    ("v_pk_min_u16", "v_pk_min_u16 {d}, {d}, {b}"),
    ("v_pk_mad_u16", "v_pk_mad_u16 {d}, {d}, {b}, {c}"),
    ("v_dot2_u32_u16", "v_dot2_u32_u16 {d}, {b}, {c}, {d}"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 {d}, {b}, {c}, {d}"),
    ("v_dot8_u32_u4", "v_dot8_u32_u4 {d}, {b}, {c}, {d}"),
    ("v_msad_u8", "v_msad_u8 {d}, {d}, {b}, {c}"),
    ("v_lshrrev_b64", "v_lshrrev_b64 {q}, {c}, {q}"),
    ("v_lshlrev_b64", "v_lshlrev_b64 {q}, {c}, {q}"),
    ("v_lshl_add_u64", "v_lshl_add_u64 {q}, {q}, 0, {q}"),
    ("v_pk_mov_b32", "v_pk_mov_b32 {q}, {q}, {q} op_sel:[1,0]"),
    ("v_mov_b64", "v_mov_b64 {q}, {q}"),
    ("v_readlane_b32", "v_readlane_b32 s24, {d}, 5"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 s24, {d}"),
    ("s_and_b64 (salu only)", "s_and_b64 s[24:25], s[24:25], s[26:27]"),
    ("s_and_saveexec+s_or exec (pair)", "s_and_saveexec_b64 s[24:25], s[26:27]\n s_or_b64 exec, exec, s[24:25]"),
    ("s_nop 0", "s_nop 0"),
    ("ds_read_b32", "ds_read_b32 {d}, {l} offset:{o}"),
    ("ds_read_u8", "ds_read_u8 {d}, {l} offset:{o}"),
    ("ds_read_u16", "ds_read_u16 {d}, {l} offset:{o}"),
    ("ds_read2_b32", "ds_read2_b32 {q}, {l} offset0:{o4} offset1:{o41}"),
    ("ds_read_b64", "ds_read_b64 {q}, {l8} offset:{o8}"),
    ("ds_write_b32", "ds_write_b32 {l}, {d} offset:{o}"),
    ("ds_write_b16", "ds_write_b16 {l}, {d} offset:{o}"),
    ("ds_write_b8", "ds_write_b8 {l}, {d} offset:{o}"),
    ("ds_write_b64", "ds_write_b64 {l8}, {q} offset:{o8}"),
    ("ds_bpermute_b32", "ds_bpermute_b32 {d}, {l}, {d}"),
    ("ds_read_u8 stride 39 dwords (bank spread)", "ds_read_u8 {d}, {l39} offset:{os}"),
    ("ds_read_u16 stride 39 dwords", "ds_read_u16 {d}, {l39} offset:{os}"),
    ("ds_write_b8 stride 39 dwords", "ds_write_b8 {l39}, {d} offset:{os}"),
    ("ds_read_b128 stride 39 dwords", "ds_read_b128 v[100:103], {l39} offset:{os16}"),
    ("ds_read_b128 linear 16 B/lane", "ds_read_b128 v[100:103], {l16} offset:{o16}"),
    ("ds_read_b64 stride 39 dwords", "ds_read_b64 {q}, {l39} offset:{os8}"),
]
HEAD = r'''// GENERATED by tools/gen_calib2.py -- do not edit.  Saturated issue cost of single instructions on gfx950.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { unsigned long long cycles, rt; };
constexpr int kIters = 300;
#define PROLOGUE \
    __shared__ uint32_t lds[256 * 40]; \
    for (int i = threadIdx.x; i < 256 * 40; i += 256) lds[i] = (uint32_t)i; \
    __syncthreads(); \
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6; \
    uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; \
    uint32_t b0 = threadIdx.x * 3 + 1, c0 = 5 + (threadIdx.x & 7); \
    uint32_t la = lane * 4 + wv * 10240, la8 = lane * 8 + wv * 10240, la16 = lane * 16 + wv * 10240, la39 = lane * 156 + wv * 10240; \
    uint64_t q0 = threadIdx.x + 0x123456789ull, q1 = 77, q2 = 99, q3 = 1234567; \
    unsigned long long t0, r0, t1, r1; \
    asm volatile("s_mov_b64 s[20:21], 0\n s_mov_b64 s[22:23], 0x5555\n s_mov_b64 s[26:27], -1\n s_mov_b64 s[24:25], -1\n v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(b0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc"); \
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#define EPILOGUE \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory"); \
    if (lane == 0) { Rec r; r.cycles = t1 - t0; r.rt = r1 - r0; out[(blockIdx.x * 256 + threadIdx.x) / 64] = r; } \
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(q0 + q1 + q2 + q3);
#define OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) \
    : "v"(b0), "v"(c0), "v"(la), "v"(la8), "v"(la16), "v"(la39) \
    : "memory", "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "v100", "v101", "v102", "v103"
'''
def body(tmpl):
    lines = []
    n_inst = 0
    for i in range(8):
        s = tmpl.format(d="%%%d" % i, q="%%%d" % (8 + (i & 3)), b="%12", c="%13", l="%14", l8="%15", l16="%16", l39="%17",
                        o=256 * i, o4=32 * i, o41=32 * i + 1, o8=512 * i, o16=1024 * i, os=4 * (i & 3) + (i >> 2), os16=16 * (i & 1), os8=8 * (i & 3))
        lines.append(" " + s)
        n_inst += s.count("\n") + 1
    return "\\n".join(l.replace("\n", "\\n") for l in lines), n_inst
out = [HEAD]
table = []
for k, (name, tmpl) in enumerate(OPS):
    b, n = body(tmpl)
    out.append('__global__ __launch_bounds__(256) void k%d(Rec* out, uint32_t* sink, int iters) {\n PROLOGUE\n for (int it = 0; it < iters; ++it) {\n  asm volatile(".rept 8\\n%s\\n .endr\\n s_waitcnt lgkmcnt(0)\\n" OPERANDS);\n }\n EPILOGUE\n}\n' % (k, b))
    table.append('{"%s", k%d, %d}' % (name, k, 8 * n))
out.append('typedef void (*Kern)(Rec*, uint32_t*, int);\nstruct Test { const char* name; Kern k; int n; };\nstatic const Test tests[] = {\n ' + ",\n ".join(table) + "\n};\n")
out.append(r'''
int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    Rec* d_out; uint32_t* d_sink;
    CHECK(hipMalloc(&d_out, sizeof(Rec) * cus * 4 * 4));
    CHECK(hipMalloc(&d_sink, 4 * cus * 4 * 256));
    printf("%-44s %10s %10s %10s %8s\n", "instruction (8 independent streams)", "1 wave", "2 w/SIMD", "4 w/SIMD", "MHz@4");
    for (const Test& t : tests) {
        double res[3]; double f4 = 0;
        int wi = 0;
        for (int W : {1, 2, 4}) {
            const int blocks = cus * W;
            t.k<<<blocks, 256>>>(d_out, d_sink, 10);
            CHECK(hipDeviceSynchronize());
            t.k<<<blocks, 256>>>(d_out, d_sink, kIters);
            CHECK(hipDeviceSynchronize());
            std::vector<Rec> h(blocks * 4);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost));
            std::vector<double> cyc, mhz;
            for (const Rec& r : h) { cyc.push_back((double)r.cycles); if (r.rt) mhz.push_back((double)r.cycles / ((double)r.rt / 100.0)); }
            std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
            res[wi++] = cyc[cyc.size() / 2] / ((double)kIters * t.n) / W;
            f4 = mhz.empty() ? 0 : mhz[mhz.size() / 2];
        }
        printf("%-44s %10.2f %10.2f %10.2f %8.0f\n", t.name, res[0], res[1], res[2], f4);
    }
    return 0;
}
''')
open(__file__.replace("gen_calib2.py", "calib2.hip"), "w").write("\n".join(out))
