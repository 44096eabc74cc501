// tools/calib5.hip -- the rate at which an MI355X serves SCATTERED 128-byte lines from HBM (development tool; the roof K2 `record_index`
// and the sequence fetches of K3 are held against).  One lane = one chain: it reads 4 (or 16) bytes of a line, then of
// another line `stride` bytes further on (the walk of a BGZF block: records ~283 bytes apart) or at a pseudo-random place of the lane's own
// 64 KB block -- `dep`: the next address depends on the loaded value (the record chain), `ind`: `inflight` independent loads per step
// (describe's head / tail fetches).  Lanes of a wavefront work on consecutive 64 KB blocks, as the kernels do.
//   hipcc --offload-arch=gfx950 -O3 -o calib5 calib5.hip && ./calib5 [GB of footprint, default 14]
// Prints one JSON line per configuration: G lines/s over the whole device and ns per line and CU.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every lane walks its 64 KB block in steps of `stride` bytes (+ what it loaded, which is zero: a true dependency the compiler cannot see through)
template <int kBytes>
__global__ __launch_bounds__(64) void k_chain(const uint8_t* __restrict__ buf, uint64_t n_blocks, uint32_t stride, uint32_t steps, uint32_t* sink) {
    const uint64_t b = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= n_blocks) return;
    const uint8_t* p = buf + b * 65536;
    uint32_t o = (uint32_t)(b * 37u) & 127u, acc = 0;
    for (uint32_t i = 0; i < steps; ++i) {
        uint32_t v;
        if (kBytes == 4) v = *(const uint32_t*)(p + (o & 0xFFFCu));
        else { const u32x4 q = *(const u32x4*)(p + (o & 0xFFF0u)); v = q.x | q.y | q.z | q.w; }
        acc += v;
        o += stride + v;
    }
    sink[b & 0xFFFFu] = acc;
}

// every lane issues `kInflight` independent loads of kBytes per step, each to another line of its block
template <int kBytes, int kInflight>
__global__ __launch_bounds__(256) void k_indep(const uint8_t* __restrict__ buf, uint64_t n_blocks, uint32_t stride, uint32_t steps, uint32_t* sink) {
    // one wavefront per 64 KB block, one lane per record of a turn (describe's shape): lane l reads around offset (turn * 64 + l) * stride
    const uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_blocks) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t* p = buf + b * 65536;
    uint32_t acc = 0;
    for (uint32_t t = 0; t < steps; ++t) {
        const uint32_t o = ((t * 64u + lane) * stride) & 0xFFFFu;
        uint32_t v[kInflight];
#pragma unroll
        for (int k = 0; k < kInflight; ++k) {
            const uint32_t a = (o + 16u * k) & (kBytes == 4 ? 0xFFFCu : 0xFFF0u);
            if (kBytes == 4) v[k] = *(const uint32_t*)(p + a);
            else { const u32x4 q = *(const u32x4*)(p + a); v[k] = q.x ^ q.w; }
        }
#pragma unroll
        for (int k = 0; k < kInflight; ++k) acc += v[k];
    }
    sink[(b * 64 + lane) & 0xFFFFu] = acc;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 14.0;
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const uint64_t n_blocks = (uint64_t)(gb * 1e9 / 65536);
    uint8_t* buf; uint32_t* sink;
    CHECK(hipMalloc(&buf, n_blocks * 65536 + 4096));
    CHECK(hipMemset(buf, 0, n_blocks * 65536 + 4096));
    CHECK(hipMalloc(&sink, 65536 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* what, auto&& launch, double lines) {
        launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("{\"what\": \"%s\", \"footprint_GB\": %.1f, \"blocks\": %llu, \"ms\": %.3f, \"G_lines_per_s\": %.2f, \"ns_per_line_and_CU\": %.2f, \"GB_per_s_at_128B\": %.0f}\n",
               what, gb, (unsigned long long)n_blocks, best, lines / best / 1e6, best * 1e6 * cus / lines, lines * 128 / best / 1e6);
        fflush(stdout);
    };
    const uint32_t steps = 230;         // records of a block
    const dim3 g1((unsigned)((n_blocks + 63) / 64)), g4((unsigned)((n_blocks + 3) / 4));
    run("walk shape: lane per block, dependent 4-byte loads 283 bytes apart", [&] { hipLaunchKernelGGL(k_chain<4>, g1, dim3(64), 0, 0, buf, n_blocks, 283u, steps, sink); }, (double)n_blocks * steps);
    run("lane per block, dependent 16-byte loads 283 bytes apart", [&] { hipLaunchKernelGGL(k_chain<16>, g1, dim3(64), 0, 0, buf, n_blocks, 283u, steps, sink); }, (double)n_blocks * steps);
    run("lane per block, dependent 4-byte loads 128 bytes apart (every line of the block once)", [&] { hipLaunchKernelGGL(k_chain<4>, g1, dim3(64), 0, 0, buf, n_blocks, 128u, 512u, sink); }, (double)n_blocks * 512);
    // describe's shape: 4 turns of 64 records; per record one load (1 line) / four 16-byte loads of 64 consecutive bytes (1.5 lines) / seven
    run("describe shape: wave per block, lane per record, 1 load of 4 bytes", [&] { hipLaunchKernelGGL((k_indep<4, 1>), g4, dim3(256), 0, 0, buf, n_blocks, 283u, 4u, sink); }, (double)n_blocks * 256);
    run("describe shape: 4 loads of 16 bytes (64 consecutive bytes: 1.5 lines)", [&] { hipLaunchKernelGGL((k_indep<16, 4>), g4, dim3(256), 0, 0, buf, n_blocks, 283u, 4u, sink); }, (double)n_blocks * 256 * 1.5);
    run("describe shape: 7 loads of 16 bytes (112 consecutive bytes: 1.875 lines)", [&] { hipLaunchKernelGGL((k_indep<16, 7>), g4, dim3(256), 0, 0, buf, n_blocks, 283u, 4u, sink); }, (double)n_blocks * 256 * 1.875);
    run("wave per block, every line of the block once (16-byte loads 128 bytes apart, 8 turns)", [&] { hipLaunchKernelGGL((k_indep<16, 1>), g4, dim3(256), 0, 0, buf, n_blocks, 128u, 8u, sink); }, (double)n_blocks * 512);
    return 0;
}
