// CPU check of host_io.hpp's BGZF header scan (tests/test_host_scan_cpu.py): the table built from pieces cut at hinted
// block starts must equal the serial one; wrong hints must fall back to the serial scan.  Test harness, not product.
#include <cstdio>
#include <cstdlib>

#include "../../sambamba_amd/csrc/host_io.hpp"

using namespace sbx;

static bool same(const BlockTable& a, const BlockTable& b) {
    return a.coffset == b.coffset && a.comp_off == b.comp_off && a.comp_len == b.comp_len && a.isize == b.isize && a.out_off == b.out_off;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FileMap f;
    f.open(argv[1]);
    setenv("SBX_SCAN_PARALLEL_MIN", "1000000000000", 1);
    const BlockTable serial = scan_bgzf(f.data, f.size);
    setenv("SBX_SCAN_PARALLEL_MIN", "0", 1);
    // 1. true block starts as hints (every third block)
    std::vector<uint64_t> hints;
    for (size_t i = 1; i < serial.size(); i += 3) hints.push_back(serial.coffset[i]);
    const BlockTable pieces = scan_bgzf(f.data, f.size, &hints);
    // 2. hints that are not block starts (a stale index): must fall back
    std::vector<uint64_t> stale;
    for (size_t i = 1; i < serial.size(); i += 5) stale.push_back(serial.coffset[i] + 7);
    const BlockTable fallback = scan_bgzf(f.data, f.size, &stale);
    // 3. a mixture
    std::vector<uint64_t> mixed = hints;
    if (serial.size() > 4) mixed.push_back(serial.coffset[serial.size() / 2] + 1);
    const BlockTable mixed_t = scan_bgzf(f.data, f.size, &mixed);
    printf("%zu blocks, %llu bytes: pieces %s, stale hints %s, mixed %s\n", serial.size(), (unsigned long long)serial.out_off.back(),
           same(serial, pieces) ? "ok" : "DIFFER", same(serial, fallback) ? "ok" : "DIFFER", same(serial, mixed_t) ? "ok" : "DIFFER");
    return same(serial, pieces) && same(serial, fallback) && same(serial, mixed_t) && serial.size() > 0 ? 0 : 1;
}
