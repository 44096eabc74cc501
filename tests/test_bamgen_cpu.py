"""CPU check of the hand-made BAM writer used by the edge-case tests: the oracle must read what
tests/bamgen.py writes (so that a GPU-side mismatch can never be blamed on a malformed fixture)."""
import random

from tests import bamgen as bg
from tests.util import run_oracle


def test_bamgen_roundtrip_through_oracle(tmp_path):
    rng = random.Random(1)
    refs = [("c1", 3000)]
    raw = []
    pos = 5
    for i in range(50):
        raw.append(bg.make_record(0, pos, "10S30M2D10M", "ACGT" * 12 + "AC", 30, name="x%d" % i))
        pos += rng.randint(0, 20)
    p = str(tmp_path / "a.bam")
    info = bg.write_bam(p, refs, raw, block_size=500, levels=[0, 6])
    assert len(info["pieces"]) > 5
    out = run_oracle(["base", p]).decode().splitlines()
    assert out[0].startswith("REF\tPOS\tCOV")
    assert len(out) > 100
    # -L uses the BAI written by bamgen
    sub = run_oracle(["base", "-L", "c1:100-200", p]).decode().splitlines()
    assert all(100 - 1 <= int(l.split("\t")[1]) < 200 for l in sub[1:])
    assert sub[1:] == [l for l in out[1:] if 99 <= int(l.split("\t")[1]) < 200]
