// tools/calib4.hip -- cost of sparse token stores next to VALU work on gfx950, the design space of K1a's output path (development
// tool): per iteration one 16-byte input load (8 lanes), S stores of B bytes with L lanes active, 300 VALU instructions.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { unsigned long long cycles; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// kind 0: S separate streams, each store 16 bytes, stream s advances 16 bytes every `period` iterations (K1a today)
// kind 1: one stream, the S stores of an iteration are ADJACENT 16-byte pieces (a burst of 16 S bytes), advancing 16 S bytes per period
// kind 2: like 0 with 4-byte stores advancing 4 bytes
__global__ __launch_bounds__(64) void k_mix(Rec* out, uint8_t* buf, unsigned long long smask, int iters, int n_st, int kind, int period, uint32_t* sink) {
    const uint32_t lane = threadIdx.x;
    uint8_t* p = buf + ((size_t)blockIdx.x * 64 + lane) * 65536;
    u32x4 v = {lane, 1, 2, 3}, w = {5, 6, 7, 8};
    uint32_t a = lane, b = 3;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        unsigned long long save;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory");
        a += v.x;
        const uint32_t step = (uint32_t)(it / period);
        uint8_t* q = p + ((step * 16u) & 0x3FF0u);
        asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, exec, %3\n global_load_dwordx4 %0, %2, off\n s_mov_b64 exec, %1" : "+v"(v), "=&s"(save) : "v"(q), "s"(0x0101010101010101ull) : "memory");
        for (int s = 0; s < n_st; ++s) {
            uint8_t* qs;
            if (kind == 0) qs = p + 16384 + 12288 * s + ((step * 16u) & 0x2FF0u);
            else if (kind == 1) qs = p + 16384 + ((step * 16u * n_st + 16u * s) & 0x7FF0u);
            else qs = p + 16384 + 12288 * s + ((step * 4u) & 0x2FFCu);
            if (kind == 2) asm volatile("s_mov_b64 %0, exec\n s_and_b64 exec, exec, %3\n global_store_dword %1, %2, off\n s_mov_b64 exec, %0" : "=&s"(save) : "v"(qs), "v"(a), "s"(smask) : "memory");
            else asm volatile("s_mov_b64 %0, exec\n s_and_b64 exec, exec, %3\n global_store_dwordx4 %1, %2, off\n s_mov_b64 exec, %0" : "=&s"(save) : "v"(qs), "v"(w), "s"(smask) : "memory");
        }
        asm volatile(".rept 75\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) { Rec r; r.cycles = t1 - t0; out[blockIdx.x] = r; }
    sink[blockIdx.x * 64 + lane] = a + v.x;
}

int main() {
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    Rec* d_out; uint8_t* big; uint32_t* d_sink;
    CHECK(hipMalloc(&d_out, sizeof(Rec) * cus * 16));
    CHECK(hipMalloc(&d_sink, (size_t)cus * 16 * 64 * 4));
    CHECK(hipMalloc(&big, (size_t)cus * 15 * 64 * 65536));
    CHECK(hipMemset(big, 0, (size_t)cus * 15 * 64 * 65536));
    struct Cfg { const char* name; int n_st, kind, period; unsigned long long mask; };
    const unsigned long long m4 = 0x0001000100010001ull, m8 = 0x0101010101010101ull, m16 = 0x1111111111111111ull, m2 = 0x0000000100000001ull, m1 = 1ull;
    const Cfg cfgs[] = {
        {"no stores", 0, 0, 8, m8},
        {"2 streams x 16 B, 8 lanes (today)", 2, 0, 8, m8},
        {"2 streams x 16 B, 4 lanes", 2, 0, 8, m4},
        {"2 streams x 16 B, 2 lanes", 2, 0, 8, m2},
        {"2 streams x 16 B, 1 lane", 2, 0, 8, m1},
        {"1 stream x 16 B, 16 lanes", 1, 0, 8, m16},
        {"1 stream x 16 B, 8 lanes", 1, 0, 8, m8},
        {"1 stream x 16 B, 4 lanes", 1, 0, 8, m4},
        {"burst 2 x 16 B adjacent, 4 lanes", 2, 1, 8, m4},
        {"burst 4 x 16 B adjacent (64 B), 2 lanes", 4, 1, 8, m2},
        {"burst 4 x 16 B adjacent (64 B), 4 lanes", 4, 1, 8, m4},
        {"2 streams x 4 B, 16 lanes", 2, 2, 8, m16},
        {"2 streams x 4 B, 64 lanes", 2, 2, 8, ~0ull},
        {"2 streams x 16 B, 8 lanes, new line every iteration", 2, 0, 1, m8},
    };
    for (int wpc : {8, 10, 15}) {
        double base = 0;
        for (const Cfg& c : cfgs) {
            const int blocks = cus * wpc, iters = 2000;
            k_mix<<<blocks, 64>>>(d_out, big, c.mask, 50, c.n_st, c.kind, c.period, d_sink);
            CHECK(hipDeviceSynchronize());
            k_mix<<<blocks, 64>>>(d_out, big, c.mask, iters, c.n_st, c.kind, c.period, d_sink);
            CHECK(hipDeviceSynchronize());
            std::vector<Rec> h(blocks);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Rec) * blocks, hipMemcpyDeviceToHost));
            std::vector<double> cy;
            for (auto& r : h) cy.push_back((double)r.cycles);
            std::sort(cy.begin(), cy.end());
            const double med = cy[cy.size() / 2] / iters;
            if (c.n_st == 0) base = med;
            printf("waves/CU %2d  %-52s %8.1f cycles per iteration (+%.1f)\n", wpc, c.name, med, med - base);
        }
    }
    return 0;
}
