"""bench.py's accounting helpers (CPU): counter files are joined into the line only when they carry the stamp of the kernel sources
they were measured on; the instruction-issue roofline is plain arithmetic."""
import os
import sys

from tests.util import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_sources_hash_is_stable_and_sensitive(tmp_path, monkeypatch):
    h = bench.kernel_sources_hash()
    assert len(h) == 16 and h[0] == "c" and h == bench.kernel_sources_hash()
    # another kernel source list -> another stamp
    monkeypatch.setattr(bench, "KERNEL_SOURCES", bench.KERNEL_SOURCES[:-1])
    assert bench.kernel_sources_hash() != h


def test_a_reworded_comment_keeps_the_stamp_a_changed_statement_does_not(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    src = tmp_path / "sambamba_amd" / "csrc"
    src.mkdir(parents=True)
    for f in bench.KERNEL_SOURCES:
        (src / f).write_text("// %s\nint f() { return 1; }  /* one */\nconst char* s = \"// kept\";\n" % f)
    h = bench.kernel_sources_hash()
    (src / "index.hip").write_text("// another wording\n\nint f()   {\n  return 1; }\nconst char* s = \"// kept\"; // trailing\n")
    assert bench.kernel_sources_hash() == h
    (src / "index.hip").write_text("int f() { return 2; }\nconst char* s = \"// kept\";\n")
    assert bench.kernel_sources_hash() != h
    (src / "index.hip").write_text("int f() { return 1; }\nconst char* s = \"// changed\";\n")
    assert bench.kernel_sources_hash() != h


def test_unstamped_or_stale_counter_files_are_not_joined(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    src = tmp_path / "sambamba_amd" / "csrc"
    src.mkdir(parents=True)
    for f in bench.KERNEL_SOURCES:
        (src / f).write_text("// %s\nint marker = 1;\n" % f)
    prof = tmp_path / "profiles" / bench.PROFILE_ROUND
    prof.mkdir(parents=True)
    table, note = bench.pmc_table(2)
    assert table == {} and "no counter pass" in note
    stamp = bench.kernel_sources_hash()
    body = "kernel,counter,value,launches_summed\nk_lz77_resolve_exact,FETCH_SIZE,1000,all\nk_lz77_resolve_exact,WRITE_SIZE,500,all\n"
    (prof / "pmc_fetch_write_config2.csv").write_text("# sources deadbeefdeadbeef\n" + body)
    table, note = bench.pmc_table(2)
    assert table == {} and "stale" in note
    (prof / "pmc_fetch_write_config2.csv").write_text("# sources %s (stamp)\n" % stamp + body)
    table, note = bench.pmc_table(2)
    assert table["lz77_resolve"]["traffic"] == (2 * 1000 + 500) * 1024            # FETCH_SIZE doubled (gfx950), KiB units
    (src / "inflate.hip").write_text("int marker = 2;")                            # the kernel changes: the file is stale again
    assert bench.pmc_table(2)[0] == {}


def test_issue_roofline_arithmetic():
    r = bench.issue_roofline({"SQ_INSTS_VALU": 6.144e9, "SQ_INSTS_SALU": 1e9, "source": "x"}, 10.0)
    assert r["peak"] == round(bench.N_SIMD * bench.CLOCK_GHZ / 4.0, 2)             # G wave-instructions per second
    assert abs(r["frac"] - 1.0) < 1e-3 and r["all_wave_instructions"] == int(7.144e9)


def test_path_accounting_fields():
    """VERDICT r5 (next 6c): the fused-path figures are first-class fields; configs 3 / 4 leave the counters out of path_frac and
    keep the older definition beside it."""
    kern = {"huffman_decode": 20.0, "lz77_resolve": 20.0, "record_index": 5.0, "decode_accumulate": 5.0}        # 50 ms
    comp, cnt = 5e9, 7e9
    a = bench.path_accounting(comp, cnt, kern, {}, 2)
    assert a["path_GBps"] == 240.0 and a["path_frac"] == round(240.0 / bench.HBM_PEAK_GBS, 5) == a["path_frac_incl_counters"]
    assert a["traffic_total"] is None and a["path_traffic_over_algorithmic"] is None and a["path_algorithmic_bytes"] == int(12e9)
    pmc = {k: {"traffic": int(30e9)} for k in kern}
    b = bench.path_accounting(comp, cnt, kern, pmc, 2)
    assert b["traffic_total"] == int(120e9) and b["path_traffic_over_algorithmic"] == 10.0
    del pmc["record_index"]                        # one group without a stamped counter entry: no total
    assert bench.path_accounting(comp, cnt, kern, pmc, 2)["traffic_total"] is None
    c = bench.path_accounting(comp, cnt, kern, {}, 3)
    assert c["path_GBps"] == 100.0 and c["path_frac_incl_counters"] == a["path_frac"] and c["path_algorithmic_bytes"] == int(5e9)
    assert bench.path_accounting(comp, cnt, {k: 0.0 for k in kern}, {}, 2)["path_frac"] is None
