"""ctypes binding of include/sbx_depth.h (no torch types, plain pointers and sizes)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, "csrc", "libsbx_depth.so")
_CLI_PATH = os.path.join(HERE, "csrc", "sbx-depth")

SBX_MODE_BASE, SBX_MODE_REGION, SBX_MODE_WINDOW = 0, 1, 2
SBX_FILTER_MAX_OPS = 64
NCOUNTERS = 7


class SbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sbx error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class Region(C.Structure):
    _fields_ = [("ref_id", C.c_uint32), ("start", C.c_uint32), ("end", C.c_uint32)]


class HeaderInfo(C.Structure):
    _fields_ = [("n_ref", C.c_int32), ("n_samples", C.c_int32), ("n_read_groups", C.c_int32),
                ("sorted_by_coordinate", C.c_int32), ("has_index", C.c_int32), ("reserved", C.c_int32),
                ("n_bgzf_blocks", C.c_uint64), ("compressed_bytes", C.c_uint64), ("uncompressed_bytes", C.c_uint64)]


class RegionStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_bases", C.c_uint32)]


class FilterOp(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("field", C.c_uint8), ("cmp", C.c_uint8), ("pad", C.c_uint8),
                ("mask", C.c_uint32), ("value", C.c_int64)]


class RegexState(C.Structure):
    _fields_ = [("type", C.c_uint8), ("a", C.c_uint8), ("b", C.c_uint8), ("c", C.c_uint8)]


class Regex(C.Structure):
    _fields_ = [("n_states", C.c_uint8), ("n_classes", C.c_uint8), ("start", C.c_uint8), ("reserved", C.c_uint8),
                ("states", RegexState * 64), ("classes", (C.c_uint8 * 32) * 8)]


class Filter(C.Structure):
    _fields_ = [("n_ops", C.c_int32), ("reserved", C.c_int32), ("ops", FilterOp * SBX_FILTER_MAX_OPS),
                ("strings", C.c_char * 512), ("n_regex", C.c_int32), ("reserved2", C.c_int32), ("regex", Regex * 2)]


class RunStats(C.Structure):
    _fields_ = [("ms_inflate", C.c_double), ("ms_index", C.c_double), ("ms_accumulate", C.c_double),
                ("ms_reduce", C.c_double), ("ms_total", C.c_double), ("ms_h2d", C.c_double),
                ("n_records", C.c_uint64), ("n_admitted", C.c_uint64), ("n_bgzf_blocks", C.c_uint64),
                ("compressed_bytes", C.c_uint64), ("uncompressed_bytes", C.c_uint64), ("counter_bytes", C.c_uint64),
                ("covered_positions", C.c_uint64), ("launches_inflate", C.c_uint64), ("launches_index", C.c_uint64),
                ("launches_accumulate", C.c_uint64), ("ms_huffman", C.c_double), ("ms_lz77", C.c_double),
                ("n_malformed", C.c_uint64), ("n_runs", C.c_uint64), ("uploaded_bytes", C.c_uint64), ("accumulate_read_bytes", C.c_uint64),
                ("token_bytes", C.c_uint64), ("max_alignment_span", C.c_uint64), ("reserved2", C.c_uint64), ("reserved3", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/sbx_depth.h declares (checked by tests/test_abi.py)
ENOMEM = -8


class Batch(C.Structure):
    _fields_ = [("first_ref", C.c_uint32), ("n_refs", C.c_uint32), ("est_bytes", C.c_uint64)]

WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_char), C.c_size_t)

EXPORTS = [
    "sbx_abi_sizeof", "sbx_bgzf_compress", "sbx_write_bam", "sbx_build_index", "sbx_run_interval", "sbx_run_interval_owned", "sbx_prefetch_interval", "sbx_depth_base_tile_device", "sbx_parse_regions", "sbx_parsed_regions", "sbx_parsed_region_line", "sbx_inflate_blocks", "sbx_open", "sbx_close", "sbx_last_error", "sbx_header", "sbx_ref_name", "sbx_ref_length",
    "sbx_ref_id", "sbx_sample_name", "sbx_header_text", "sbx_compile_filter", "sbx_set_filter", "sbx_regex_search", "sbx_set_params",
    "sbx_set_regions", "sbx_run", "sbx_depth_base_tile", "sbx_depth_region_stats", "sbx_depth_region_stats_from",
    "sbx_depth_window_stats",
    "sbx_format_base_rows", "sbx_stream_base_rows", "sbx_plan_batches", "sbx_run_batch", "sbx_last_run_stats", "sbx_tile_info", "sbx_next_active_range", "sbx_preload",
    "sbx_device_count", "sbx_plan_shards", "sbx_format_base_rows_device",
]

_lib = None


def lib_path():
    return _LIB_PATH


def cli_path():
    return _CLI_PATH


def lib():
    """Load libsbx_depth.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError("libsbx_depth.so is missing at %s -- run `python -m sambamba_amd.build` "
                          "(the HIP library is the product; there is no CPU fallback)" % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.sbx_inflate_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_char_p, C.c_size_t]
    L.sbx_inflate_blocks.restype = C.c_int
    L.sbx_open.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.sbx_open.restype = C.c_void_p
    L.sbx_close.argtypes = [C.c_void_p]
    L.sbx_close.restype = None
    L.sbx_last_error.argtypes = [C.c_void_p]
    L.sbx_last_error.restype = C.c_char_p
    L.sbx_header.argtypes = [C.c_void_p, C.POINTER(HeaderInfo)]
    L.sbx_ref_name.argtypes = [C.c_void_p, C.c_int]
    L.sbx_ref_name.restype = C.c_char_p
    L.sbx_ref_length.argtypes = [C.c_void_p, C.c_int]
    L.sbx_ref_length.restype = C.c_int64
    L.sbx_ref_id.argtypes = [C.c_void_p, C.c_char_p]
    L.sbx_sample_name.argtypes = [C.c_void_p, C.c_int]
    L.sbx_sample_name.restype = C.c_char_p
    L.sbx_header_text.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.sbx_header_text.restype = C.c_char_p
    L.sbx_compile_filter.argtypes = [C.c_char_p, C.POINTER(Filter), C.c_char_p, C.c_size_t]
    L.sbx_set_filter.argtypes = [C.c_void_p, C.POINTER(Filter)]
    L.sbx_set_params.argtypes = [C.c_void_p, C.c_int, C.c_uint8, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
    L.sbx_set_regions.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.sbx_run.argtypes = [C.c_void_p]
    L.sbx_depth_base_tile.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.sbx_depth_region_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sbx_depth_window_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.sbx_format_base_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_int,
                                       C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sbx_stream_base_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_int, WRITE_FN, C.c_void_p]
    L.sbx_plan_batches.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sbx_run_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.sbx_last_run_stats.argtypes = [C.c_void_p, C.POINTER(RunStats)]
    L.sbx_tile_info.argtypes = [C.c_void_p, u32p, u32p]
    L.sbx_next_active_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, u64p, u64p]
    L.sbx_preload.argtypes = [C.c_void_p]
    L.sbx_format_base_rows_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_int, C.c_void_p,
                                              C.c_size_t, C.POINTER(C.c_size_t)]
    L.sbx_device_count.argtypes = []
    L.sbx_plan_shards.argtypes = [C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sbx_run_interval.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.sbx_bgzf_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.sbx_write_bam.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.sbx_build_index.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    L.sbx_prefetch_interval.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.sbx_run_interval_owned.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    L.sbx_depth_base_tile_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sbx_parse_regions.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.sbx_parsed_regions.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.sbx_parsed_region_line.argtypes = [C.c_void_p, C.c_size_t]
    L.sbx_parsed_region_line.restype = C.c_char_p
    L.sbx_abi_sizeof.argtypes = [C.c_char_p]
    L.sbx_abi_sizeof.restype = C.c_size_t
    L.sbx_depth_region_stats_from.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sbx_regex_search.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    # the structures above must have the layout the library was compiled with
    for name, ty in (("sbx_region", Region), ("sbx_header_info", HeaderInfo), ("sbx_region_stats", RegionStats),
                     ("sbx_filter_op", FilterOp), ("sbx_regex_state", RegexState), ("sbx_regex", Regex), ("sbx_filter", Filter),
                     ("sbx_run_stats", RunStats), ("sbx_batch", Batch)):
        if L.sbx_abi_sizeof(name.encode()) != C.sizeof(ty):
            raise ImportError("ctypes layout of %s (%d bytes) differs from libsbx_depth.so (%d bytes)" % (
                name, C.sizeof(ty), L.sbx_abi_sizeof(name.encode())))
    _lib = L
    return L


def inflate_blocks(comp, comp_off, comp_len, isize, out_off, out_size):
    """sbx_inflate_blocks over numpy arrays; returns the inflated bytes (numpy uint8)."""
    L = lib()
    comp = np.ascontiguousarray(comp, dtype=np.uint8)
    comp_off = np.ascontiguousarray(comp_off, dtype=np.uint64)
    comp_len = np.ascontiguousarray(comp_len, dtype=np.uint32)
    isize = np.ascontiguousarray(isize, dtype=np.uint32)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out = np.zeros(int(out_size), dtype=np.uint8)
    err = C.create_string_buffer(512)
    rc = L.sbx_inflate_blocks(comp.ctypes.data, comp_off.ctypes.data, comp_len.ctypes.data, isize.ctypes.data,
                              len(comp_len), out.ctypes.data, out_off.ctypes.data, err, 512)
    if rc != 0:
        raise SbxError(rc, err.value.decode())
    return out


def bgzf_compress(data, level=6, with_eof=True, device=-1):
    """sbx_bgzf_compress: the BGZF stream (bytes) of `data`, compressed on the device."""
    L = lib()
    src = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    cap = len(src) + len(src) // 2048 + 64 + 28 + 31
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    err = C.create_string_buffer(512)
    rc = L.sbx_bgzf_compress(src.ctypes.data if len(src) else None, len(src), int(level), int(with_eof), device, out.ctypes.data, cap, C.byref(n), err, 512)
    if rc != 0:
        raise SbxError(rc, err.value.decode())
    return out[:n.value].tobytes()


def write_bam(path, stream, level=6, with_index=True, device=-1):
    """sbx_write_bam: the uncompressed BAM byte stream `stream` as a BGZF file (+ .bai)."""
    L = lib()
    src = np.frombuffer(bytes(stream), dtype=np.uint8)
    err = C.create_string_buffer(512)
    rc = L.sbx_write_bam(path.encode(), src.ctypes.data, len(src), int(level), int(with_index), device, err, 512)
    if rc != 0:
        raise SbxError(rc, err.value.decode())


def build_index(bam_path, bai_path=None, device=-1):
    """sbx_build_index (`sambamba index`): writes bai_path (default: bam_path + ".bai")."""
    L = lib()
    err = C.create_string_buffer(512)
    rc = L.sbx_build_index(bam_path.encode(), (bai_path or bam_path + ".bai").encode(), device, err, 512)
    if rc != 0:
        raise SbxError(rc, err.value.decode())


def regex_search(pattern, text, options=""):
    """Host-side evaluation of `text =~ /pattern/options` with the NFA the device runs (sbx_regex_search)."""
    L = lib()
    L.sbx_regex_search.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    err = C.create_string_buffer(512)
    data = text if isinstance(text, bytes) else text.encode()
    rc = L.sbx_regex_search(pattern.encode(), options.encode(), data, len(data), err, 512)
    if rc < 0:
        raise SbxError(rc, err.value.decode())
    return bool(rc)


def compile_filter(query):
    f = Filter()
    err = C.create_string_buffer(512)
    rc = lib().sbx_compile_filter(query.encode() if query is not None else None, C.byref(f), err, 512)
    if rc != 0:
        raise SbxError(rc, err.value.decode())
    return f


class Depth:
    """Thin object wrapper over an sbx_ctx (one BAM or a list of BAMs, one device)."""

    def __init__(self, bam_path, device=-1):
        self._L = lib()
        paths = [bam_path] if isinstance(bam_path, str) else list(bam_path)
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        err = C.create_string_buffer(1024)
        self._ctx = self._L.sbx_open(arr, len(paths), device, err, 1024)
        if not self._ctx:
            raise SbxError(-1, err.value.decode())
        hi = HeaderInfo()
        self._check(self._L.sbx_header(self._ctx, C.byref(hi)))
        self.info = hi
        self.ref_names = [self._L.sbx_ref_name(self._ctx, i).decode() for i in range(hi.n_ref)]
        self.ref_lengths = [self._L.sbx_ref_length(self._ctx, i) for i in range(hi.n_ref)]
        self.sample_names = [self._L.sbx_sample_name(self._ctx, i).decode() for i in range(hi.n_samples)]
        self.n_samples_eff = hi.n_samples

    def _check(self, rc):
        if rc != 0:
            raise SbxError(rc, self._L.sbx_last_error(self._ctx).decode())

    def close(self):
        if self._ctx:
            self._L.sbx_close(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_filter(self, query):
        f = compile_filter(query)
        self._check(self._L.sbx_set_filter(self._ctx, C.byref(f)))

    def set_params(self, mode=SBX_MODE_BASE, min_bq=0, fix_mate_overlaps=False, combined=False, window=0, overlap=0,
                   thresholds=()):
        thr = np.asarray(list(thresholds), dtype=np.uint32)
        self._check(self._L.sbx_set_params(self._ctx, mode, min_bq, int(fix_mate_overlaps), int(combined), window, overlap,
                                           thr.ctypes.data if len(thr) else None, len(thr)))
        self.n_samples_eff = 1 if combined else self.info.n_samples

    def set_regions(self, regions):
        arr = (Region * len(regions))(*[Region(*r) for r in regions])
        self._check(self._L.sbx_set_regions(self._ctx, arr, len(regions)))

    def parse_regions(self, arg):
        """-L argument (BED file or region string) -> (merged [(ref, start, end)], raw [(ref, start, end)], raw input lines),
        parsed by the library exactly as the CLI parses it."""
        nm, nr = C.c_size_t(0), C.c_size_t(0)
        self._check(self._L.sbx_parse_regions(self._ctx, arg.encode(), C.byref(nm), C.byref(nr)))
        out = []
        for merged, n in ((1, nm.value), (0, nr.value)):
            arr = (Region * max(1, n))()
            self._check(self._L.sbx_parsed_regions(self._ctx, merged, arr, n))
            out.append([(arr[i].ref_id, arr[i].start, arr[i].end) for i in range(n)])
        lines = [self._L.sbx_parsed_region_line(self._ctx, i).decode() for i in range(nr.value)]
        return out[0], out[1], lines

    def preload(self):
        self._check(self._L.sbx_preload(self._ctx))

    def run(self):
        self._check(self._L.sbx_run(self._ctx))
        st = RunStats()
        self._check(self._L.sbx_last_run_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    def region_stats(self, regions, n_thresholds=0):
        """sbx_depth_region_stats: returns (n_reads[n][S], n_bases[n][S], cov[n][S][n_thr], seen[n])."""
        n, S = len(regions), self.n_samples_eff
        # a uint32 array [n][3] (ref_id, start, end) is the sbx_region layout itself: passed as is (200,000 BED lines cost
        # 0.2 s per call as a list of ctypes structures)
        arr = np.ascontiguousarray(regions, dtype=np.uint32).reshape(n, 3) if n else np.zeros((1, 3), dtype=np.uint32)
        st = np.zeros((n, S, 2), dtype=np.uint32)
        cov = np.zeros((n, S, max(1, n_thresholds)), dtype=np.uint32)
        seen = np.zeros(n, dtype=np.uint8)
        self._check(self._L.sbx_depth_region_stats(self._ctx, arr.ctypes.data, n, st.ctypes.data, cov.ctypes.data, seen.ctypes.data))
        return st[:, :, 0].copy(), st[:, :, 1].copy(), cov[:, :, :n_thresholds].copy(), seen

    def window_stats(self, ref_id, first, count, n_thresholds=0):
        S = self.n_samples_eff
        st = np.zeros((count, S, 2), dtype=np.uint32)
        cov = np.zeros((count, S, max(1, n_thresholds)), dtype=np.uint32)
        self._check(self._L.sbx_depth_window_stats(self._ctx, ref_id, first, count, st.ctypes.data, cov.ctypes.data))
        return st[:, :, 0].copy(), st[:, :, 1].copy(), cov[:, :, :n_thresholds].copy()

    def plan_batches(self, budget_bytes=0):
        """[(first_ref, n_refs, est_bytes)]: consecutive batches of contigs that fit the device (sbx_plan_batches)."""
        n = C.c_size_t(0)
        self._check(self._L.sbx_plan_batches(self._ctx, int(budget_bytes), None, 0, C.byref(n)))
        arr = (Batch * max(1, n.value))()
        self._check(self._L.sbx_plan_batches(self._ctx, int(budget_bytes), arr, n.value, C.byref(n)))
        return [(arr[i].first_ref, arr[i].n_refs, arr[i].est_bytes) for i in range(n.value)]

    def run_batch(self, first_ref, n_refs):
        """sbx_run restricted to the reads of contigs [first_ref, first_ref + n_refs)."""
        self._check(self._L.sbx_run_batch(self._ctx, int(first_ref), int(n_refs)))
        st = RunStats()
        self._check(self._L.sbx_last_run_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    def run_interval(self, ref_id, beg, end):
        """sbx_run restricted to the reads overlapping [beg, end) of ref_id (position sharding / streaming)."""
        self._check(self._L.sbx_run_interval(self._ctx, int(ref_id), int(beg), int(end)))
        st = RunStats()
        self._check(self._L.sbx_last_run_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    def run_interval_owned(self, ref_id, beg, end):
        """sbx_run_interval_owned: only the reads whose leftmost position lies in [beg, end), counted over all they cover."""
        self._check(self._L.sbx_run_interval_owned(self._ctx, int(ref_id), int(beg), int(end)))
        st = RunStats()
        self._check(self._L.sbx_last_run_stats(self._ctx, C.byref(st)))
        return st.as_dict()

    def base_counters_to_device(self, ref_id, beg, end, device_ptr):
        """sbx_depth_base_tile_device: counters of [beg, end) into device memory ((end - beg) * S * 7 uint32 at device_ptr)."""
        self._check(self._L.sbx_depth_base_tile_device(self._ctx, int(ref_id), int(beg), int(end), C.c_void_p(int(device_ptr))))

    def measure_base_rows(self, ref_id, beg, end, min_cov=1.0, max_cov=float("inf"), annotate=False):
        """Bytes of the text of [beg, end) (the device's measuring pass alone; nothing is copied)."""
        need = C.c_size_t(0)
        rc = self._L.sbx_format_base_rows(self._ctx, ref_id, beg, end, float(min_cov), float(max_cov), int(annotate), None, 0, C.byref(need))
        if rc != 0 and rc != ENOMEM:
            self._check(rc)
        return int(need.value)

    def format_base_rows_to_device(self, ref_id, beg, end, device_ptr, cap, min_cov=1.0, max_cov=float("inf"), annotate=False):
        """sbx_format_base_rows_device: the text of [beg, end) into device memory at device_ptr (cap bytes); returns its size."""
        need = C.c_size_t(0)
        self._check(self._L.sbx_format_base_rows_device(self._ctx, ref_id, beg, end, float(min_cov), float(max_cov), int(annotate),
                                                        C.c_void_p(int(device_ptr)) if device_ptr else None, int(cap), C.byref(need)))
        return int(need.value)

    def format_base_rows(self, ref_id, beg, end, min_cov=1.0, max_cov=float("inf"), annotate=False):
        """Text of `depth base` for [beg, end) of ref_id, formatted on the device (bytes)."""
        need = C.c_size_t(0)
        cap = max(1 << 16, (end - beg) * 40 * self.n_samples_eff)
        for _ in range(2):
            buf = C.create_string_buffer(cap)
            rc = self._L.sbx_format_base_rows(self._ctx, ref_id, beg, end, float(min_cov), float(max_cov), int(annotate),
                                              buf, cap, C.byref(need))
            if rc == ENOMEM and need.value > cap:
                cap = need.value
                continue
            self._check(rc)
            return buf.raw[:need.value]
        self._check(rc)

    def stream_base_rows(self, ref_id, beg, end, write, min_cov=1.0, max_cov=float("inf"), annotate=False):
        """sbx_stream_base_rows: `write(bytes)` is called with consecutive pieces of the text of [beg, end)."""
        def cb(_user, data, n):
            try:
                write(C.string_at(data, n))
                return 0
            except Exception:       # the library turns it into SBX_EIO
                return 1
        fn = WRITE_FN(cb)
        self._check(self._L.sbx_stream_base_rows(self._ctx, ref_id, beg, end, float(min_cov), float(max_cov), int(annotate), fn, None))

    def next_active_range(self, ref_id, start):
        """[beg, end) of the next stretch of tiles at or after `start` that hold admitted reads, or None (sbx_next_active_range)."""
        b, e = C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.sbx_next_active_range(self._ctx, ref_id, int(start), C.byref(b), C.byref(e)))
        if b.value == 0xFFFFFFFFFFFFFFFF:
            return None
        return int(b.value), int(e.value)

    def active_end(self, ref_id):
        """End of the last stretch of tiles of ref_id that hold admitted reads after the last run, or 0: beyond the contig's length
        when alignments hang over its end (the engine lays out as many spare tiles as they need)."""
        end, at = 0, 0
        while True:
            r = self.next_active_range(ref_id, at)
            if r is None:
                return end
            end = at = r[1]

    def covered(self, ref_id, beg, end):
        """One byte per position of [beg, end): non-zero iff a pileup column exists there (available after every kind of run)."""
        cov = np.zeros(end - beg, dtype=np.uint8)
        self._check(self._L.sbx_depth_base_tile(self._ctx, ref_id, beg, end, None, cov.ctypes.data))
        return cov

    def base_counters(self, ref_id, beg, end, with_covered=False):
        S = self.n_samples_eff
        out = np.zeros((end - beg, S, NCOUNTERS), dtype=np.uint32)
        cov = np.zeros(end - beg, dtype=np.uint8) if with_covered else None
        self._check(self._L.sbx_depth_base_tile(self._ctx, ref_id, beg, end, out.ctypes.data,
                                                cov.ctypes.data if with_covered else None))
        return (out, cov) if with_covered else out
