// CPU driver of sambamba_amd/csrc/deflate_core.hpp (the code the device runs one lane per BGZF block): compresses a file into a
// BGZF stream on the host, so that tests can inflate it with zlib without a GPU.  Test infrastructure only.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../sambamba_amd/csrc/deflate_core.hpp"

// `deflate_host --lengths <max_len> f0 f1 ...`: the code lengths huffman_lengths gives the counts, one line
static int lengths_mode(int argc, char** argv) {
    const uint32_t max_len = (uint32_t)atoi(argv[2]);
    std::vector<uint16_t> freq;
    for (int k = 3; k < argc; ++k) freq.push_back((uint16_t)atoi(argv[k]));
    const uint32_t n = (uint32_t)freq.size();
    std::vector<uint8_t> len(n);
    std::vector<uint32_t> a(n);
    std::vector<uint16_t> sym(n), code(n);
    sbx::huffman_lengths(freq.data(), n, max_len, len.data(), a.data(), sym.data());
    sbx::canonical_codes(len.data(), n, max_len, code.data());
    for (uint32_t k = 0; k < n; ++k) printf("%u:%u%s", (unsigned)len[k], (unsigned)code[k], k + 1 < n ? " " : "\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 5 && std::string(argv[1]) == "--lengths") return lengths_mode(argc, argv);
    if (argc < 4) { fprintf(stderr, "usage: deflate_host <in> <out> <level>\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> in;
    uint8_t buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + k);
    fclose(f);
    const int level = atoi(argv[3]);
    uint32_t crc[256];
    for (uint32_t i = 0; i < 256; ++i) sbx::crc32_make_entry(crc, i);
    std::vector<uint16_t> table(1u << sbx::kHashBits);
    std::vector<uint8_t> slot(sbx::kBgzfSlot);
    std::vector<uint8_t> work_mem(sbx::kWorkBytes, 0xAB);        // (the device's slice is not zeroed either)
    sbx::DynWork* work = reinterpret_cast<sbx::DynWork*>(work_mem.data());
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 1;
    for (size_t off = 0; off < in.size(); off += sbx::kBgzfPayload) {
        const uint32_t n = (uint32_t)(in.size() - off < sbx::kBgzfPayload ? in.size() - off : sbx::kBgzfPayload);
        std::fill(table.begin(), table.end(), 0);
        const uint32_t len = sbx::bgzf_block(in.data() + off, n, level, slot.data(), table.data(), work, crc);
        fwrite(slot.data(), 1, len, o);
    }
    fclose(o);
    return 0;
}
