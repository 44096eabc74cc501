// tools/calib.hip -- issue-rate calibration of gfx950 for the integer instructions the inflate kernels are made of.
// (development tool, not part of the product: `hipcc --offload-arch=gfx950 -O2 -o calib calib.hip`)
//
// DESIGN's "VALU port busy" figures assumed one wave64 VALU instruction = 4 cycles of a SIMD's port,
// the hardware guide says 2.  This program measures it: W waves per SIMD each run the same unrolled block of one
// instruction kind (independent streams, or one dependent chain), every wave times itself with s_memtime, and the host
// prints cycles per wave-instruction as seen by ONE wave (latency-ish) and per SIMD (= wave cycles / W: the port's rate
// once enough waves are resident).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kReps = 64;      // instructions (or groups) per asm block
constexpr int kIters = 400;    // blocks per wave

struct Rec { unsigned long long cycles, rt; };

#define KERNEL(NAME, SETUP, BODY)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(Rec* out, uint32_t* sink, int iters) {                               \
        __shared__ uint32_t lds[256 * 16];                                                                          \
        for (int i = threadIdx.x; i < 256 * 16; i += 256) lds[i] = (uint32_t)(((i * 37) & 255) * 4);                \
        __syncthreads();                                                                                            \
        uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;                           \
        uint32_t b0 = threadIdx.x * 3 + 1, b1 = 0x00010001u, la = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096; \
        uint64_t q0 = threadIdx.x + 0x123456789ull;                                                                 \
        SETUP                                                                                                       \
        unsigned long long t0, r0, t1, r1;                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory"); \
        for (int it = 0; it < iters; ++it) {                                                                        \
            asm volatile(BODY                                                                                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0)  \
                         : "v"(b0), "v"(b1), "v"(la) : "memory", "vcc", "s20", "s21", "s22", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");                                             \
        }                                                                                                           \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory"); \
        if ((threadIdx.x & 63) == 0) {                                                                              \
            Rec r; r.cycles = t1 - t0; r.rt = r1 - r0;                                                              \
            out[(blockIdx.x * 256 + threadIdx.x) / 64] = r;                                                         \
        }                                                                                                           \
        sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)q0;                 \
    }

// operand map: %0..%7 = a0..a7, %8 = q0 (64-bit), %9 = b0, %10 = b1, %11 = la (LDS byte address)
KERNEL(k_add_indep, , ".rept 8\n v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n"
                      " v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n .endr\n")
KERNEL(k_add_dep, , ".rept 64\n v_add_u32 %0, %0, %9\n .endr\n")
KERNEL(k_pksub_indep, , ".rept 8\n v_pk_sub_i16 %0, %0, %9\n v_pk_sub_i16 %1, %1, %9\n v_pk_sub_i16 %2, %2, %9\n v_pk_sub_i16 %3, %3, %9\n"
                        " v_pk_sub_i16 %4, %4, %9\n v_pk_sub_i16 %5, %5, %9\n v_pk_sub_i16 %6, %6, %9\n v_pk_sub_i16 %7, %7, %9\n .endr\n")
KERNEL(k_dot2_indep, , ".rept 8\n v_dot2_u32_u16 %0, %9, %10, %0\n v_dot2_u32_u16 %1, %9, %10, %1\n v_dot2_u32_u16 %2, %9, %10, %2\n v_dot2_u32_u16 %3, %9, %10, %3\n"
                       " v_dot2_u32_u16 %4, %9, %10, %4\n v_dot2_u32_u16 %5, %9, %10, %5\n v_dot2_u32_u16 %6, %9, %10, %6\n v_dot2_u32_u16 %7, %9, %10, %7\n .endr\n")
KERNEL(k_dot2_dep, , ".rept 64\n v_dot2_u32_u16 %0, %9, %10, %0\n .endr\n")
KERNEL(k_perm_indep, , ".rept 8\n v_perm_b32 %0, %0, %9, %10\n v_perm_b32 %1, %1, %9, %10\n v_perm_b32 %2, %2, %9, %10\n v_perm_b32 %3, %3, %9, %10\n"
                       " v_perm_b32 %4, %4, %9, %10\n v_perm_b32 %5, %5, %9, %10\n v_perm_b32 %6, %6, %9, %10\n v_perm_b32 %7, %7, %9, %10\n .endr\n")
KERNEL(k_shr64_indep, , ".rept 16\n v_lshrrev_b64 %8, 1, %8\n v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n .endr\n")
KERNEL(k_shr64_only, , ".rept 64\n v_lshrrev_b64 %8, %10, %8\n .endr\n")
KERNEL(k_cndmask_indep, , ".rept 8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n"
                          " v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n .endr\n")
KERNEL(k_cmp_cnd, , ".rept 32\n v_cmp_gt_u32 vcc, %0, %9\n v_cndmask_b32 %1, %1, %9, vcc\n .endr\n")
// the canonical decode's triple: sub, shift, dot (chained through the accumulator as in decode_pairs)
KERNEL(k_decode_triple, , ".rept 21\n v_pk_sub_i16 %1, %9, %0\n v_pk_lshrrev_b16 %1, 15, %1 op_sel_hi:[0,1]\n v_dot2_u32_u16 %2, %1, %10, %2\n .endr\n v_add_u32 %0, %0, %2\n")
// LDS: dependent chain (address = value read), independent b32 reads, independent b128 reads
KERNEL(k_lds_dep, a0 = la;, ".rept 64\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %11\n v_and_b32 %0, 0x3ffc, %0\n .endr\n")
KERNEL(k_lds_b32_indep, , ".rept 8\n ds_read_b32 %0, %11\n ds_read_b32 %1, %11 offset:256\n ds_read_b32 %2, %11 offset:512\n ds_read_b32 %3, %11 offset:768\n"
                          " ds_read_b32 %4, %11 offset:1024\n ds_read_b32 %5, %11 offset:1280\n ds_read_b32 %6, %11 offset:1536\n ds_read_b32 %7, %11 offset:1792\n .endr\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_u16_indep, , ".rept 8\n ds_read_u16 %0, %11\n ds_read_u16 %1, %11 offset:256\n ds_read_u16 %2, %11 offset:512\n ds_read_u16 %3, %11 offset:768\n"
                          " ds_read_u16 %4, %11 offset:1024\n ds_read_u16 %5, %11 offset:1280\n ds_read_u16 %6, %11 offset:1536\n ds_read_u16 %7, %11 offset:1792\n .endr\n s_waitcnt lgkmcnt(0)\n")
// b128: lane addresses 16 bytes apart (a linear [row][lane] layout)
KERNEL(k_lds_b128_indep, la = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;,
       ".rept 16\n ds_read_b128 v[100:103], %11\n ds_read_b128 v[104:107], %11 offset:1024\n ds_read_b128 v[108:111], %11 offset:2048\n ds_read_b128 v[112:115], %11 offset:3072\n .endr\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_lds_w8_indep, , ".rept 8\n ds_write_b8 %11, %0\n ds_write_b8 %11, %1 offset:256\n ds_write_b8 %11, %2 offset:512\n ds_write_b8 %11, %3 offset:768\n"
                         " ds_write_b8 %11, %4 offset:1024\n ds_write_b8 %11, %5 offset:1280\n ds_write_b8 %11, %6 offset:1536\n ds_write_b8 %11, %7 offset:1792\n .endr\n s_waitcnt lgkmcnt(0)\n")
// a VALU stream with an LDS read every 8 instructions (does LDS issue steal VALU slots?)
KERNEL(k_mix_valu_lds, , ".rept 8\n ds_read_b32 %7, %11\n v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n"
                         " v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n .endr\n s_waitcnt lgkmcnt(0)\n")
// VALU with scalar instructions interleaved (SALU issues from the same wave: does it cost VALU slots?)
KERNEL(k_mix_valu_salu, , ".rept 32\n v_add_u32 %0, %0, %9\n s_and_b32 s20, s20, s21\n v_add_u32 %1, %1, %9\n s_or_b32 s22, s22, s21\n .endr\n")

typedef void (*Kern)(Rec*, uint32_t*, int);
struct Test { const char* name; Kern k; int valu_per_block; int other_per_block; };

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    Rec* d_out;
    uint32_t* d_sink;
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&d_out, sizeof(Rec) * max_blocks * 4));
    CHECK(hipMalloc(&d_sink, 4 * max_blocks * 256));
    std::vector<Test> tests = {
        {"v_add_u32 x8 independent", k_add_indep, 64, 0},
        {"v_add_u32 dependent chain", k_add_dep, 64, 0},
        {"v_pk_sub_i16 x8 independent", k_pksub_indep, 64, 0},
        {"v_dot2_u32_u16 x8 independent", k_dot2_indep, 64, 0},
        {"v_dot2_u32_u16 dependent chain", k_dot2_dep, 64, 0},
        {"v_perm_b32 x8 independent", k_perm_indep, 64, 0},
        {"v_lshrrev_b64 + 3 v_add", k_shr64_indep, 64, 0},
        {"v_lshrrev_b64 dependent", k_shr64_only, 64, 0},
        {"v_cndmask_b32 x8 independent", k_cndmask_indep, 64, 0},
        {"v_cmp + v_cndmask pairs", k_cmp_cnd, 64, 0},
        {"decode triple sub/shift/dot2 (63 + 1)", k_decode_triple, 64, 0},
        {"ds_read_b32 dependent (+2 valu)", k_lds_dep, 128, 64},
        {"ds_read_b32 independent", k_lds_b32_indep, 0, 64},
        {"ds_read_u16 independent", k_lds_u16_indep, 0, 64},
        {"ds_read_b128 independent (linear)", k_lds_b128_indep, 0, 64},
        {"ds_write_b8 independent", k_lds_w8_indep, 0, 64},
        {"7 v_add + 1 ds_read_b32", k_mix_valu_lds, 56, 8},
        {"v_add / s_and interleaved", k_mix_valu_salu, 64, 64},
    };
    printf("%-40s %3s %12s %12s %12s %10s\n", "test", "W", "cyc/inst/wave", "cyc/inst/SIMD", "wall cyc/inst", "MHz");
    for (const Test& t : tests) {
        for (int W : {1, 2, 4, 8}) {
            // 256-thread blocks: 4 waves, one per SIMD; W blocks per CU
            const int blocks = cus * W;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            t.k<<<blocks, 256>>>(d_out, d_sink, 20);   // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            t.k<<<blocks, 256>>>(d_out, d_sink, kIters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Rec> h(blocks * 4);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost));
            std::vector<double> cyc, mhz;
            for (const Rec& r : h) { cyc.push_back((double)r.cycles); if (r.rt) mhz.push_back((double)r.cycles / ((double)r.rt / 100.0)); }
            std::sort(cyc.begin(), cyc.end());
            std::sort(mhz.begin(), mhz.end());
            const double med = cyc[cyc.size() / 2];
            const double n_inst = (double)kIters * (t.valu_per_block + t.other_per_block);
            const double f = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
            printf("%-40s %3d %12.2f %12.2f %12.2f %10.0f\n", t.name, W, med / n_inst, med / n_inst / W,
                   f > 0 ? ms * 1e3 * f / n_inst / W : 0.0, f);
        }
    }
    return 0;
}
