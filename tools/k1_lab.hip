// k1_lab -- development tool (not part of the product library): times the inflate kernels of csrc/inflate.hip on the BGZF blocks
// of a BAM, kernel by kernel with HIP events, and K1b with parts of its batch loop compiled out (lz77_resolve_body's kAblate bits:
// results of those launches are INVALID by construction; the point is what each part of the loop costs the kernel -- `ablate`) or
// with other window geometries (`geo`).  The default kernels run first and their output is compared with zlib's on the host, so a
// lab build that broke the real path says so.  (Round 6's A/Bs -- the copy primitive, 8-byte cooperative copies, one-step 32-byte
// copies -- were run from earlier versions of this file: profiles/round6/call_[a-f]_*.jsonl.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o k1_lab k1_lab.hip -lz && ./k1_lab file.bam [repeats]
//
// Prints one JSON line per measurement.
#include "../sambamba_amd/csrc/inflate.hip"

#include <zlib.h>

#include <functional>

#include "../sambamba_amd/csrc/host_io.hpp"

namespace sbx {
void require_device(int) {}

template <uint32_t kAblate>
__global__ __launch_bounds__(kResThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lab_k1b(SBX_LZ77_ARGS) {
    lz77_resolve_body<kHistDefault, kSpanDefault, true, true, kAblate>(SBX_LZ77_PASS);
}

// the A/B partner of the product body: a 17 .. 32-byte copy as two Short16 steps instead of a step and one 16-byte word
template <uint32_t kAblate>
__global__ __launch_bounds__(kResThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lab_k1b_t16(SBX_LZ77_ARGS) {
    lz77_resolve_body<kHistDefault, kSpanDefault, true, true, kAblate, false>(SBX_LZ77_PASS);
}

// window geometry variants of the product body
template <uint32_t kHist, uint32_t kSpan>
__global__ __launch_bounds__(kResThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lab_k1b_geo(SBX_LZ77_ARGS) {
    lz77_resolve_body<kHist, kSpan, true, true, 0>(SBX_LZ77_PASS);
}

struct Lab {
    InflateArgs a;
    hipStream_t stream = nullptr;
    int repeats = 5;
    template <class F>
    double time(F&& f) {
        EventTimer t;
        f();                                   // warm
        SBX_HIP(hipStreamSynchronize(stream));
        double best = 1e30, sum = 0;
        for (int r = 0; r < repeats; ++r) {
            t.start(stream);
            f();
            t.stop(stream);
            const double ms = t.ms();
            best = std::min(best, ms);
            sum += ms;
        }
        last_mean = sum / repeats;
        return best;
    }
    double last_mean = 0;
    std::function<void()> check;
    template <uint32_t kHist, uint32_t kSpan>
    void k1b_geo(bool verify = false) {
        const uint32_t per = kResThreads / 64;
        dim3 grid((a.n_blocks + per - 1) / per), block(kResThreads);
        const size_t lds = (size_t)per * (kHist + 1024u + kSpan + 16u) + 128 + (size_t)per * 256;
        const double ms = time([&] {
            hipLaunchKernelGGL((k_lab_k1b_geo<kHist, kSpan>), grid, block, lds, stream, a.lit, a.ent, a.nent, a.out_off, a.isize, a.n_blocks, a.block0, a.out, a.status);
        });
        printf("{\"kernel\": \"k1b\", \"hist\": %u, \"span\": %u, \"lds_per_wave\": %u, \"ms_best\": %.4f, \"ms_mean\": %.4f}\n", kHist, kSpan, kHist + 1024u + kSpan + 16u + 256u, ms, last_mean);
        if (verify) check();
        fflush(stdout);
    }
    void k1b_t16(bool verify = false) {
        const uint32_t per = kResThreads / 64;
        dim3 grid((a.n_blocks + per - 1) / per), block(kResThreads);
        const size_t lds = (size_t)per * (kHistDefault + 1024u + kSpanDefault + 16u) + 128 + (size_t)per * 256;
        const double ms = time([&] {
            hipLaunchKernelGGL((k_lab_k1b_t16<0>), grid, block, lds, stream, a.lit, a.ent, a.nent, a.out_off, a.isize, a.n_blocks, a.block0, a.out, a.status);
        });
        printf("{\"kernel\": \"k1b\", \"variant\": \"two Short16 steps\", \"ms_best\": %.4f, \"ms_mean\": %.4f}\n", ms, last_mean);
        if (verify) check();
        fflush(stdout);
    }
    template <uint32_t kAblate>
    void k1b(const char* what, bool verify = false) {
        const uint32_t per = kResThreads / 64;
        dim3 grid((a.n_blocks + per - 1) / per), block(kResThreads);
        const size_t lds = (size_t)per * (kHistDefault + 1024u + kSpanDefault + 16u) + 128 + (size_t)per * 256;
        const double ms = time([&] {
            hipLaunchKernelGGL((k_lab_k1b<kAblate>), grid, block, lds, stream, a.lit, a.ent, a.nent, a.out_off, a.isize, a.n_blocks, a.block0, a.out, a.status);
        });
        printf("{\"kernel\": \"k1b\", \"ablate\": %u, \"what\": \"%s\", \"ms_best\": %.4f, \"ms_mean\": %.4f}\n", kAblate, what, ms, last_mean);
        if (verify) check();
        fflush(stdout);
    }
};
}  // namespace sbx

int main(int argc, char** argv) {
    using namespace sbx;
    if (argc < 2) { fprintf(stderr, "usage: k1_lab file.bam [repeats]\n"); return 2; }
    try {
        FileMap f;
        f.open(argv[1]);
        BlockTable t = scan_bgzf(f.data, f.size);
        const uint32_t nb = (uint32_t)t.size();
        const uint64_t total = t.out_off.back();
        fprintf(stderr, "blocks %u, compressed %zu, inflated %llu\n", nb, f.size, (unsigned long long)total);
        DevBuf<uint8_t> d_in(f.size + 256), d_out(total + 4096), d_scr(inflate_scratch_bytes(nb)), d_lit(inflate_lit_bytes(total, nb));
        DevBuf<uint32_t> d_ent(inflate_ent_words(total, nb)), d_nent(nb), d_clen(nb), d_isz(nb), d_st(nb);
        DevBuf<uint64_t> d_coff(nb), d_ooff(nb);
        DevBuf<unsigned long long> d_tok(64);
        SBX_HIP(hipMemset(d_in.p + f.size, 0, 256));
        SBX_HIP(hipMemset(d_tok.p, 0, 64 * 8));
        SBX_HIP(hipMemcpy(d_in.p, f.data, f.size, hipMemcpyHostToDevice));
        SBX_HIP(hipMemcpy(d_coff.p, t.comp_off.data(), nb * 8ull, hipMemcpyHostToDevice));
        SBX_HIP(hipMemcpy(d_ooff.p, t.out_off.data(), nb * 8ull, hipMemcpyHostToDevice));
        SBX_HIP(hipMemcpy(d_clen.p, t.comp_len.data(), nb * 4ull, hipMemcpyHostToDevice));
        SBX_HIP(hipMemcpy(d_isz.p, t.isize.data(), nb * 4ull, hipMemcpyHostToDevice));
        Lab lab;
        SBX_HIP(hipStreamCreate(&lab.stream));
        if (argc > 2) lab.repeats = atoi(argv[2]);
        lab.a = InflateArgs{d_in.p, d_coff.p, d_clen.p, d_isz.p, d_ooff.p, d_out.p, nb, 0, d_scr.p, d_lit.p, d_ent.p, d_nent.p, d_st.p, d_tok.p};
        // the real path first: K1a, K1b, compared with zlib
        const double k1a = lab.time([&] { launch_k1a(lab.a, lab.stream); });
        printf("{\"kernel\": \"k1a\", \"ms_best\": %.4f, \"ms_mean\": %.4f}\n", k1a, lab.last_mean);
        lab.check = [&] {
            std::vector<uint32_t> st(nb);
            SBX_HIP(hipMemcpy(st.data(), d_st.p, nb * 4ull, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < nb; ++i) if (st[i]) throw Error(SBX_EFORMAT, "block " + std::to_string(i) + ": " + inflate_status_string(st[i]));
            // a sample of blocks against zlib (every 53rd)
            std::vector<uint8_t> got(65536), want(65536);
            uint32_t checked = 0;
            for (uint32_t i = 0; i < nb; i += 53) {
                if (!t.isize[i]) continue;
                SBX_HIP(hipMemcpy(got.data(), d_out.p + t.out_off[i], t.isize[i], hipMemcpyDeviceToHost));
                z_stream z; memset(&z, 0, sizeof z);
                inflateInit2(&z, -15);
                z.next_in = (Bytef*)(f.data + t.comp_off[i]); z.avail_in = t.comp_len[i];
                z.next_out = want.data(); z.avail_out = 65536;
                const int rc = inflate(&z, Z_FINISH);
                inflateEnd(&z);
                if (rc != Z_STREAM_END || z.total_out != t.isize[i] || memcmp(got.data(), want.data(), t.isize[i]))
                    throw Error(SBX_EFORMAT, "block " + std::to_string(i) + " differs from zlib");
                ++checked;
            }
            printf("{\"check\": \"zlib\", \"blocks\": %u, \"ok\": true}\n", checked);
            SBX_HIP(hipMemset(d_out.p, 0xA5, total));            // the next variant's bytes are its own
        };
        const double k1b = lab.time([&] { launch_k1b(lab.a, lab.stream); });
        printf("{\"kernel\": \"k1b_product\", \"ms_best\": %.4f, \"ms_mean\": %.4f}\n", k1b, lab.last_mean);
        lab.check();
        lab.k1b<0>("the product body", true);
        lab.k1b_t16(true);
        lab.k1b<0>("the product body, again");
        lab.k1b_t16();
        if (argc > 3 && std::string(argv[3]) == "geo") {
            lab.k1b_geo<1024, 1024>(true);
            lab.k1b_geo<1024, 1536>();
            lab.k1b_geo<1536, 1024>();
            lab.k1b_geo<2048, 1024>();
            lab.k1b_geo<2048, 1536>();
            lab.k1b_geo<2048, 2048>(true);
            lab.k1b_geo<3072, 1536>();
            lab.k1b_geo<4096, 1536>();
        }
        if (argc > 3 && std::string(argv[3]) == "ablate") {
            lab.k1b<1>("no literal copies (own + coop)");
            lab.k1b<2>("no far-match copies (own + coop)");
            lab.k1b<3>("no phase A");
            lab.k1b<64>("no cooperative copies in phase A");
            lab.k1b<16>("phase A: one own-lane step only");
            lab.k1b<4>("no phase B (near matches)");
            lab.k1b<8>("no phase C (write-out)");
            lab.k1b<3 | 4>("no phase A, no phase B");
            lab.k1b<3 | 4 | 8>("scan and loop only");
            lab.k1b<3 | 8>("phase B only");
        }
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "k1_lab: %s\n", e.what());
        return 1;
    }
}
