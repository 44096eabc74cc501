// bai_host -- test harness: the product's BAI builder (sambamba_amd/csrc/bai_writer.hpp, the restatement of IndexBuilder,
// BioD/bio/std/hts/bam/bai/indexing.d:52-346) fed from a host-side BAM reader (zlib), so that the bookkeeping can be checked
// on the CPU against the .bai files the reference's own test-suite ships.  Virtual offsets come from the same VoffCursor
// sbx_build_index applies to the device's record offsets (engine.cpp).
//   g++ -O2 -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ bai_host.cpp -lz
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../sambamba_amd/csrc/bai_writer.hpp"

static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t ld16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: bai_host in.bam out.bai\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    std::vector<uint8_t> file;
    for (uint8_t buf[1 << 16];;) { size_t n = fread(buf, 1, sizeof buf, f); if (!n) break; file.insert(file.end(), buf, buf + n); }
    fclose(f);
    std::vector<uint64_t> coff, ustart;
    std::vector<uint8_t> U;
    for (size_t p = 0; p + 18 <= file.size();) {
        const uint32_t bsize = ld16(&file[p + 16]) + 1u, xlen = ld16(&file[p + 10]);
        const uint32_t isize = ld32(&file[p + bsize - 4]);
        coff.push_back(p);
        ustart.push_back(U.size());
        const size_t at = U.size();
        U.resize(at + isize);
        z_stream z{};
        inflateInit2(&z, -15);
        z.next_in = &file[p + 12 + xlen];
        z.avail_in = bsize - 12 - xlen - 8;
        z.next_out = U.data() + at;
        z.avail_out = isize;
        const int rc = inflate(&z, Z_FINISH);
        inflateEnd(&z);
        if (rc != Z_STREAM_END && isize) { fprintf(stderr, "inflate failed at %zu\n", p); return 1; }
        p += bsize;
    }
    ustart.push_back(U.size());
    const uint64_t file_end = file.size();
    sbx::VoffCursor vc(coff.data(), ustart.data(), coff.size(), file_end);
    if (U.size() < 12 || memcmp(U.data(), "BAM\1", 4)) { fprintf(stderr, "not a BAM\n"); return 1; }
    uint64_t o = 8 + ld32(&U[4]);
    const int n_ref = (int)ld32(&U[o]);
    o += 4;
    for (int r = 0; r < n_ref; ++r) o += 8 + ld32(&U[o]);
    try {
        sbx::BaiBuilder bb(n_ref);
        while (o + 4 <= U.size()) {
            const uint32_t bs = ld32(&U[o]);
            const uint8_t* r = &U[o + 4];
            sbx::BaiRecord rec;
            rec.ref_id = (int32_t)ld32(r);
            rec.position = (int32_t)ld32(r + 4);
            const uint32_t l_name = r[8], n_cigar = ld16(r + 12);
            rec.bin = ld16(r + 10);
            rec.is_unmapped = (ld16(r + 14) & 4) != 0;
            int64_t span = 0;
            for (uint32_t k = 0; k < n_cigar; ++k) {
                const uint32_t op = ld32(r + 32 + l_name + 4 * k), ty = op & 15u;
                if (ty == 0 || ty == 2 || ty == 3 || ty == 7 || ty == 8) span += op >> 4;
            }
            rec.end_position = rec.position + (int32_t)span;
            rec.start_vo = vc.of_byte(o);
            rec.end_vo = vc.behind(o + 4 + bs);
            bb.put(rec);
            o += 4 + (uint64_t)bs;
        }
        const std::vector<uint8_t>& out = bb.finish();
        FILE* g = fopen(argv[2], "wb");
        fwrite(out.data(), 1, out.size(), g);
        fclose(g);
    } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
