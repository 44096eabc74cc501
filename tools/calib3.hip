// tools/calib3.hip -- what a vector memory INSTRUCTION costs a CU on gfx950 when only a few of its lanes are active (development
// tool, see calib.hip): K1a issues one 16-byte load and two 16-byte stores per loop iteration with ~5 of 64 lanes active each.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Rec { unsigned long long cycles, rt; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// mode 0: store dwordx4, 1: load dwordx4, 2: store dword, 3: nothing (the VALU filler alone)
template <int kMode>
__global__ __launch_bounds__(64) void k(Rec* out, uint8_t* buf, unsigned long long mask, int iters, int filler, uint32_t* sink) {
    const uint32_t lane = threadIdx.x;
    uint8_t* p = buf + ((size_t)blockIdx.x * 64 + lane) * 256;     // every lane its own line
    u32x4 v = {lane, 1, 2, 3};
    uint32_t a = lane, b = 3;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        unsigned long long save;
        if (kMode == 0)
            asm volatile("s_mov_b64 %0, exec\n s_and_b64 exec, exec, %3\n global_store_dwordx4 %1, %2, off\n s_mov_b64 exec, %0" : "=&s"(save) : "v"(p), "v"(v), "s"(mask) : "memory");
        if (kMode == 1)
            asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, exec, %3\n global_load_dwordx4 %0, %2, off\n s_mov_b64 exec, %1" : "+v"(v), "=&s"(save) : "v"(p), "s"(mask) : "memory");
        if (kMode == 2)
            asm volatile("s_mov_b64 %0, exec\n s_and_b64 exec, exec, %3\n global_store_dword %1, %2, off\n s_mov_b64 exec, %0" : "=&s"(save) : "v"(p), "v"(a), "s"(mask) : "memory");
        for (int f = 0; f < filler; ++f) asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
        if ((it & 15) == 15) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    if (lane == 0) { Rec r; r.cycles = t1 - t0; r.rt = r1 - r0; out[blockIdx.x] = r; }
    sink[blockIdx.x * 64 + lane] = a + v.x;
}

// K1a-like iteration: wait for the load of the previous iteration, consume it, issue the next load and n_st stores, then VALU work
template <int kWait>
__global__ __launch_bounds__(64) void k_mix(Rec* out, uint8_t* buf, unsigned long long lmask, unsigned long long smask, int iters, int filler, int n_st, int advance, uint32_t* sink) {
    const uint32_t lane = threadIdx.x;
    uint8_t* p = buf + ((size_t)blockIdx.x * 64 + lane) * 65536;     // every lane its own 64 KiB region
    u32x4 v = {lane, 1, 2, 3}, w = {5, 6, 7, 8};
    uint32_t a = lane, b = 3, off = 0;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        unsigned long long save;
        if (kWait == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory");
        if (kWait == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(v) :: "memory");
        a += v.x;
        uint8_t* q = p + (off & 0xFFF0u);
        asm volatile("s_mov_b64 %1, exec\n s_and_b64 exec, exec, %3\n global_load_dwordx4 %0, %2, off\n s_mov_b64 exec, %1" : "+v"(v), "=&s"(save) : "v"(q), "s"(lmask) : "memory");
        for (int s = 0; s < n_st; ++s) {
            uint8_t* qs = q + 32768 + 8192 * s;
            asm volatile("s_mov_b64 %0, exec\n s_and_b64 exec, exec, %3\n global_store_dwordx4 %1, %2, off\n s_mov_b64 exec, %0" : "=&s"(save) : "v"(qs), "v"(w), "s"(smask) : "memory");
        }
        off += advance;
        for (int f = 0; f < filler; ++f) asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    if (lane == 0) { Rec r; r.cycles = t1 - t0; r.rt = r1 - r0; out[blockIdx.x] = r; }
    sink[blockIdx.x * 64 + lane] = a + v.x;
}

template <int kWait>
static void run_mix(Rec* d_out, uint8_t* d_buf, uint32_t* d_sink, int cus) {
    for (int wpc : {8, 15}) {
        for (int n_st : {0, 2}) {
            for (int advance : {0, 2}) {        // 0: always the same lines, 2: streaming like K1a (16 bytes every 8 iterations)
                const int blocks = cus * wpc, iters = 3000, filler = 75;      // 300 VALU per iteration
                k_mix<kWait><<<blocks, 64>>>(d_out, d_buf, 0x0101010101010101ull, 0x1010101010101010ull, 50, filler, n_st, advance, d_sink);
                CHECK(hipDeviceSynchronize());
                k_mix<kWait><<<blocks, 64>>>(d_out, d_buf, 0x0101010101010101ull, 0x1010101010101010ull, iters, filler, n_st, advance, d_sink);
                CHECK(hipDeviceSynchronize());
                std::vector<Rec> h(blocks);
                CHECK(hipMemcpy(h.data(), d_out, sizeof(Rec) * blocks, hipMemcpyDeviceToHost));
                std::vector<double> c;
                for (auto& r : h) c.push_back((double)r.cycles);
                std::sort(c.begin(), c.end());
                printf("mix wait vmcnt(%d) waves/CU %2d stores %d advance %d: %8.1f cycles per iteration per wave\n", kWait, wpc, n_st, advance, c[c.size() / 2] / iters);
            }
        }
    }
}

template <int kMode>
static void run(const char* name, Rec* d_out, uint8_t* d_buf, uint32_t* d_sink, int cus) {
    for (int wpc : {4, 8, 15}) {
        for (unsigned long long mask : {0x1ull, 0x0101010101010101ull, ~0ull}) {
            for (int filler : {0, 25}) {       // 25 x 4 = 100 VALU instructions between the memory instructions
                const int blocks = cus * wpc, iters = 2000;
                k<kMode><<<blocks, 64>>>(d_out, d_buf, mask, 50, filler, d_sink);
                CHECK(hipDeviceSynchronize());
                k<kMode><<<blocks, 64>>>(d_out, d_buf, mask, iters, filler, d_sink);
                CHECK(hipDeviceSynchronize());
                std::vector<Rec> h(blocks);
                CHECK(hipMemcpy(h.data(), d_out, sizeof(Rec) * blocks, hipMemcpyDeviceToHost));
                std::vector<double> c;
                for (auto& r : h) c.push_back((double)r.cycles);
                std::sort(c.begin(), c.end());
                const double med = c[c.size() / 2];
                printf("%-18s waves/CU %2d lanes %2d filler %3d VALU: %8.1f cycles per iteration per wave, %7.1f per CU per instruction\n", name, wpc,
                       __builtin_popcountll(mask), filler * 4, med / iters, med / iters / wpc);
            }
        }
    }
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    Rec* d_out; uint8_t* d_buf; uint32_t* d_sink;
    CHECK(hipMalloc(&d_out, sizeof(Rec) * cus * 16));
    CHECK(hipMalloc(&d_buf, (size_t)cus * 16 * 64 * 256));
    CHECK(hipMalloc(&d_sink, (size_t)cus * 16 * 64 * 4));
    CHECK(hipMemset(d_buf, 0, (size_t)cus * 16 * 64 * 256));
    {
        uint8_t* big;
        CHECK(hipMalloc(&big, (size_t)cus * 15 * 64 * 65536));
        run_mix<0>(d_out, big, d_sink, cus);
        run_mix<2>(d_out, big, d_sink, cus);
        CHECK(hipFree(big));
        return 0;
    }
    run<3>("no memory op", d_out, d_buf, d_sink, cus);
    run<0>("store dwordx4", d_out, d_buf, d_sink, cus);
    run<1>("load dwordx4", d_out, d_buf, d_sink, cus);
    run<2>("store dword", d_out, d_buf, d_sink, cus);
    return 0;
}
