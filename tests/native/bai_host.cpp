// bai_host -- test harness: the product's BAI builder (sambamba_amd/csrc/bai_writer.hpp, the restatement of IndexBuilder,
// BioD/bio/std/hts/bam/bai/indexing.d:52-346) fed from a host-side BAM reader (zlib), so that the bookkeeping can be checked
// on the CPU against the .bai files the reference's own test-suite ships.  Virtual offsets come from the same VoffCursor
// sbx_build_index applies to the device's record offsets (engine.cpp).
//   g++ -O2 -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ bai_host.cpp -lz
// `bai_host in.bam out.bai --parallel SEED`: the same file from bai_parallel.hpp -- the per-record step the device runs one lane
// per record, called here for the records of random batches in shuffled order, then bai_assemble.  Exit 3: the step called the
// input irregular (the engine then uses the serial builder).
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../sambamba_amd/csrc/bai_writer.hpp"
#include "../../sambamba_amd/csrc/bai_parallel.hpp"

#include <random>

static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t ld16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

// `bai_host --vo-selftest SEED`: the stateless virtual-offset searches of bai_parallel.hpp against the cursor of bai_writer.hpp on
// random block tables with empty blocks (EOF blocks in the middle of a concatenated file, several in a row) and random record
// boundaries, queried the way the serial loop queries the cursor (start of record i, end of record i = start of record i + 1)
static int vo_selftest(uint64_t seed) {
    std::mt19937_64 rng(seed);
    for (int trial = 0; trial < 2000; ++trial) {
        const size_t nb = 1 + rng() % 12;
        std::vector<uint64_t> coff(nb), ustart(nb + 1, 0);
        uint64_t c = 0;
        for (size_t b = 0; b < nb; ++b) {
            coff[b] = c;
            c += 28 + rng() % 500;
            const uint64_t isize = rng() % 3 == 0 ? 0 : 1 + rng() % 300;
            ustart[b + 1] = ustart[b] + isize;
        }
        const uint64_t file_end = c, total = ustart[nb];
        std::vector<uint64_t> cuts{0};
        for (uint64_t u = 0; u < total;) { u += 1 + rng() % 120; cuts.push_back(std::min(u, total)); }
        if (cuts.back() != total) cuts.push_back(total);
        sbx::VoffCursor vc(coff.data(), ustart.data(), nb, file_end);
        sbx::BaiArgs a{};
        a.coff = coff.data(); a.ustart = ustart.data(); a.n_blocks = (uint32_t)nb; a.file_end = file_end;
        for (size_t i = 0; i + 1 < cuts.size(); ++i) {
            if (cuts[i] == cuts[i + 1]) continue;
            const uint64_t s1 = vc.of_byte(cuts[i]), s2 = sbx::bai_vo_of(a, cuts[i]);
            const uint64_t e1 = vc.behind(cuts[i + 1]), e2 = sbx::bai_vo_behind(a, cuts[i + 1]);
            if (s1 != s2 || e1 != e2) {
                fprintf(stderr, "trial %d record %zu: start %llx / %llx, end %llx / %llx\n", trial, i, (unsigned long long)s1, (unsigned long long)s2,
                        (unsigned long long)e1, (unsigned long long)e2);
                return 1;
            }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 3 && std::string(argv[1]) == "--vo-selftest") return vo_selftest((uint64_t)atoll(argv[2]));
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
    if (argc >= 5 && std::string(argv[3]) == "--parallel") {
        std::mt19937_64 rng((uint64_t)atoll(argv[4]));
        std::vector<int32_t> ref_len;
        {
            uint64_t q = 8 + ld32(&U[4]) + 4;
            for (int r = 0; r < n_ref; ++r) { const uint32_t ln = ld32(&U[q]); ref_len.push_back((int32_t)ld32(&U[q + 4 + ln])); q += 8 + ln; }
        }
        // descriptors as index.hip's describe leaves them (rec_off relative to the batch's first inflated byte: here the file's)
        std::vector<sbx::RecDesc> desc;
        std::vector<int32_t> refs;
        uint64_t p = o;
        while (p + 4 <= U.size()) {
            const uint32_t bs = ld32(&U[p]);
            const uint8_t* r = &U[p + 4];
            sbx::RecDesc d{};
            d.rec_off = p;
            d.pos = (int32_t)ld32(r + 4);
            d.flag = ld16(r + 14);
            const uint32_t l_name = r[8], n_cigar = ld16(r + 12);
            int64_t span = 0;
            for (uint32_t k = 0; k < n_cigar; ++k) {
                const uint32_t op = ld32(r + 32 + l_name + 4 * k), ty = op & 15u;
                if (ty == 0 || ty == 2 || ty == 3 || ty == 7 || ty == 8) span += op >> 4;
            }
            const int32_t rid = (int32_t)ld32(r);
            const bool admitted = !(d.flag & 4) && rid >= 0 && span > 0;
            d.end = admitted ? d.pos + (int32_t)span : d.pos;
            desc.push_back(d);
            refs.push_back(rid);
            p += 4 + (uint64_t)bs;
        }
        sbx::BaiHostResults R;
        R.lin_off.assign((size_t)n_ref + 1, 0);
        for (int r = 0; r < n_ref; ++r) R.lin_off[(size_t)r + 1] = R.lin_off[(size_t)r] + sbx::bai_windows_for(ref_len[(size_t)r]);
        R.lin.assign(R.lin_off[(size_t)n_ref] + 1, ~0ull);
        R.lin_len.assign((size_t)n_ref + 1, 0);
        R.meta_end.assign((size_t)n_ref + 1, 0);
        R.n_mapped.assign((size_t)n_ref + 1, 0);
        R.n_unmapped.assign((size_t)n_ref + 1, 0);
        unsigned long long scalars[sbx::kBaiScalars] = {0};
        scalars[sbx::kBaiFirstVo] = ~0ull;
        sbx::BaiCarry carry{-1, 0, 0, 0, 0};
        for (uint64_t done = 0; done < desc.size();) {
            const uint64_t n = std::min<uint64_t>(desc.size() - done, 1 + rng() % (desc.size() < 50 ? 7 : desc.size() / 3 + 1));
            // a batch: its descriptors count from the batch's first byte, like a work list of the engine
            const uint64_t u_base = desc[done].rec_off;
            std::vector<sbx::RecDesc> bd(desc.begin() + done, desc.begin() + done + n);
            for (auto& d : bd) d.rec_off -= u_base;
            std::vector<sbx::BaiRun> runs(n + 1);
            scalars[sbx::kBaiNumRuns] = 0;
            sbx::BaiArgs a{};
            a.U = U.data() + u_base; a.desc = bd.data(); a.rec_ref = refs.data() + done; a.n = n; a.rec_base = done;
            a.u_base = u_base; a.u_next = done + n < desc.size() ? desc[done + n].rec_off : p;
            a.coff = coff.data(); a.ustart = ustart.data(); a.n_blocks = (uint32_t)coff.size(); a.file_end = file_end;
            a.carry = carry; a.n_ref = n_ref;
            a.lin = R.lin.data(); a.lin_off = R.lin_off.data(); a.lin_len = R.lin_len.data();
            a.meta_end = R.meta_end.data(); a.n_mapped = R.n_mapped.data(); a.n_unmapped = R.n_unmapped.data();
            a.scalars = scalars; a.runs = runs.data(); a.runs_cap = runs.size();
            std::vector<uint64_t> order(n);
            for (uint64_t k = 0; k < n; ++k) order[k] = k;
            std::shuffle(order.begin(), order.end(), rng);
            for (uint64_t k : order) sbx::bai_record_step(a, k);
            sbx::bai_carry_out(a, &carry);
            R.runs.insert(R.runs.end(), runs.begin(), runs.begin() + scalars[sbx::kBaiNumRuns]);
            done += n;
        }
        if (scalars[sbx::kBaiIrregular]) { fprintf(stderr, "irregular input\n"); return 3; }
        for (int k = 0; k < sbx::kBaiScalars; ++k) R.scalars[k] = scalars[k];
        R.last = carry;
        const std::vector<uint8_t> out = sbx::bai_assemble(n_ref, R);
        FILE* g = fopen(argv[2], "wb");
        fwrite(out.data(), 1, out.size(), g);
        fclose(g);
        return 0;
    }
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
