// deflate.hip -- BGZF compression on gfx950: the write side of the codec seam.
//
// Replaces bgzfCompress (BioD/bio/core/bgzf/compress.d:34-103) as BgzfOutputStream / BamWriter call it for every
// <= 0xFF00-byte piece of the uncompressed BAM stream (bgzf/outputstream.d, bam/writer.d:67-287): one deflate stream +
// CRC32 + ISIZE per block.  Like inflate, a BAM is hundreds of thousands of independent blocks, and LZ77 matching +
// entropy coding of ONE block is a serial chain (every match decides where the next token starts), so the mapping is
// again ONE LANE PER BGZF BLOCK -- 64 independent encoders per wavefront running sbx::bgzf_block (deflate_core.hpp:
// greedy matches through a 4 KiB hash table of last positions kept in a per-block slice of global scratch; by `level` stored, the
// fixed Huffman code in one pass, or a dynamic Huffman code -- symbol counts, code construction and a second pass over the
// input, its tables in a second 4 KiB slice per block).  Every block is written into a 64 KiB slot; a scan of the block lengths and a coalesced copy (one
// workgroup per block) then pack the slots into the BGZF stream.
// The index of the file written (or of any coordinate-sorted BAM) is built here too: k_bai_records, one lane per record.
// Bound: memory latency (every hash probe and window compare of a lane is its own cache line); it is a writer for
// harness-sized and production files alike, not a roofline kernel -- DESIGN.md reports its GB/s next to zlib's.
#include "common.hpp"
#include "deflate_core.hpp"
#include "kernels.hpp"
#include "bai_parallel.hpp"

namespace sbx {

namespace {

constexpr int kDefThreads = 64;

__global__ __launch_bounds__(kDefThreads) void k_bgzf_deflate(const uint8_t* __restrict__ in, uint64_t n_bytes, uint32_t n_blocks, int level,
                                                              uint8_t* __restrict__ slots, uint16_t* __restrict__ tables,
                                                              uint8_t* __restrict__ work, uint32_t* __restrict__ block_len) {
    __shared__ uint32_t crc_table[256];
    for (uint32_t k = threadIdx.x; k < 256; k += kDefThreads) crc32_make_entry(crc_table, k);
    __syncthreads();
    const uint32_t b = blockIdx.x * kDefThreads + threadIdx.x;
    if (b >= n_blocks) return;
    const uint64_t off = (uint64_t)b * kBgzfPayload;
    const uint32_t n = (uint32_t)(n_bytes - off < kBgzfPayload ? n_bytes - off : kBgzfPayload);
    block_len[b] = bgzf_block(in + off, n, level, slots + (size_t)b * kBgzfSlot, tables + ((size_t)b << kHashBits),
                              (DynWork*)(work + (size_t)b * kWorkBytes), crc_table);
}

// block b: block_len[b] bytes from its slot to out + offset[b]
__global__ __launch_bounds__(256) void k_pack_blocks(const uint8_t* __restrict__ slots, const uint32_t* __restrict__ block_len,
                                                     const uint64_t* __restrict__ offset, uint8_t* __restrict__ out) {
    const uint32_t b = blockIdx.x;
    const uint32_t n = block_len[b];
    const uint8_t* s = slots + (size_t)b * kBgzfSlot;
    uint8_t* d = out + offset[b];
    for (uint32_t i = threadIdx.x * 4u; i < n; i += 1024u) {
        if (i + 4 <= n) {
            uint32_t v;
            __builtin_memcpy(&v, s + i, 4);
            __builtin_memcpy(d + i, &v, 4);
        } else {
            for (uint32_t k = i; k < n; ++k) d[k] = s[k];
        }
    }
}

// the stored `bin` field of every record (bin_mq_nl >> 16, read.d:919-921): what the BAI groups chunks by
__global__ __launch_bounds__(256) void k_gather_bins(const uint8_t* __restrict__ U, const RecDesc* __restrict__ desc, uint64_t n,
                                                     uint16_t* __restrict__ bins) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint16_t v;
    __builtin_memcpy(&v, U + desc[i].rec_off + 4 + 10, 2);
    bins[i] = v;
}

// the index of a BAM, one lane per record (bai_parallel.hpp: what IndexBuilder's loop computes, as sums, minima, maxima and run heads)
__global__ __launch_bounds__(256) void k_bai_records(BaiArgs a) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) bai_record_step(a, i);
}
__global__ void k_bai_carry(BaiArgs a, BaiCarry* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) bai_carry_out(a, out);
}

}  // namespace

void launch_bai_records(const BaiArgs& a, BaiCarry* d_carry_out, hipStream_t stream) {
    if (a.n) {
        hipLaunchKernelGGL(k_bai_records, dim3((uint32_t)((a.n + 255) / 256)), dim3(256), 0, stream, a);
        SBX_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_bai_carry, dim3(1), dim3(64), 0, stream, a, d_carry_out);
    SBX_HIP(hipGetLastError());
}

size_t deflate_table_entries(uint32_t n_blocks) { return (size_t)n_blocks << kHashBits; }
size_t deflate_work_bytes(uint32_t n_blocks) { return (size_t)n_blocks * kWorkBytes; }

void launch_bgzf_deflate(const uint8_t* d_in, uint64_t n_bytes, uint32_t n_blocks, int level, uint8_t* d_slots, uint16_t* d_tables,
                         uint8_t* d_work, uint32_t* d_block_len, hipStream_t stream) {
    if (!n_blocks) return;
    SBX_HIP(hipMemsetAsync(d_tables, 0, deflate_table_entries(n_blocks) * 2, stream));
    hipLaunchKernelGGL(k_bgzf_deflate, dim3((n_blocks + kDefThreads - 1) / kDefThreads), dim3(kDefThreads), 0, stream, d_in, n_bytes,
                       n_blocks, level, d_slots, d_tables, d_work, d_block_len);
    SBX_HIP(hipGetLastError());
}

void launch_pack_blocks(const uint8_t* d_slots, const uint32_t* d_block_len, const uint64_t* d_offset, uint32_t n_blocks, uint8_t* d_out,
                        hipStream_t stream) {
    if (!n_blocks) return;
    hipLaunchKernelGGL(k_pack_blocks, dim3(n_blocks), dim3(256), 0, stream, d_slots, d_block_len, d_offset, d_out);
    SBX_HIP(hipGetLastError());
}

void launch_gather_bins(const uint8_t* d_U, const RecDesc* d_desc, uint64_t n_records, uint16_t* d_bins, hipStream_t stream) {
    if (!n_records) return;
    hipLaunchKernelGGL(k_gather_bins, dim3((uint32_t)((n_records + 255) / 256)), dim3(256), 0, stream, d_U, d_desc, n_records, d_bins);
    SBX_HIP(hipGetLastError());
}

}  // namespace sbx
