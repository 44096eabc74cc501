// tools/pcie_bw.cpp -- host <-> device copy rates with pinned buffers placed on each NUMA node (development tool: decides whether
// the library should place its pinned staging / text buffers on the GPU's node).  hipcc -O2 -o pcie_bw pcie_bw.cpp
#include <hip/hip_runtime.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static long set_pref(int node) {      // MPOL_PREFERRED = 1, MPOL_DEFAULT = 0
    if (node < 0) return syscall(SYS_set_mempolicy, 0, nullptr, 0);
    unsigned long mask[16] = {0};
    mask[node / 64] = 1ul << (node % 64);
    return syscall(SYS_set_mempolicy, 1, mask, 1024);
}

int main() {
    int dev = 0;
    hipSetDevice(dev);
    char bus[64] = {0};
    hipDeviceGetPCIBusId(bus, sizeof bus, dev);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    for (char* p = path; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    int gpu_node = -1;
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &gpu_node) != 1) gpu_node = -1; fclose(f); }
    printf("GPU %s numa_node %d\n", bus, gpu_node);
    const size_t N = 1ull << 30;
    void* d = nullptr;
    hipMalloc(&d, N);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int node = -1; node < 4; ++node) {
        if (set_pref(node) != 0) { printf("node %d: set_mempolicy failed\n", node); continue; }
        void *h = nullptr, *h2 = nullptr;
        if (hipHostMalloc(&h, N, hipHostMallocDefault) != hipSuccess) { printf("node %d: hipHostMalloc failed\n", node); continue; }
        hipHostMalloc(&h2, N, hipHostMallocDefault);
        memset(h, 1, N); memset(h2, 2, N);
        set_pref(-1);
        auto run = [&](const char* what, auto&& f) {
            f();
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 4; ++k) f();
            hipDeviceSynchronize();
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("  node %2d %-22s %.1f GB/s\n", node, what, 4.0 * N / dt / 1e9);
        };
        void* d2 = nullptr;
        hipMalloc(&d2, N);
        run("H2D", [&] { hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s1); });
        run("D2H", [&] { hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s1); });
        run("H2D + D2H (sum of both)", [&] { hipMemcpyAsync(d, h, N / 2, hipMemcpyHostToDevice, s1); hipMemcpyAsync(h2, d2, N / 2, hipMemcpyDeviceToHost, s2); });
        hipFree(d2);
        hipHostFree(h); hipHostFree(h2);
    }
    return 0;
}
