// CPU driver of sambamba_amd/csrc/deflate_core.hpp (the code the device runs one lane per BGZF block): compresses a file into a
// BGZF stream on the host, so that tests can inflate it with zlib without a GPU.  Test infrastructure only.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../sambamba_amd/csrc/deflate_core.hpp"

int main(int argc, char** argv) {
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
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 1;
    for (size_t off = 0; off < in.size(); off += sbx::kBgzfPayload) {
        const uint32_t n = (uint32_t)(in.size() - off < sbx::kBgzfPayload ? in.size() - off : sbx::kBgzfPayload);
        std::fill(table.begin(), table.end(), 0);
        const uint32_t len = sbx::bgzf_block(in.data() + off, n, level, slot.data(), table.data(), crc);
        fwrite(slot.data(), 1, len, o);
    }
    fclose(o);
    return 0;
}
