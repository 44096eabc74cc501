"""The reference dictionary of several BAM files (sbx::merge_dictionaries, sambamba_amd/csrc/host_io.hpp) against a restatement
of what the reference computes: SamHeaderMerger.mergeSequenceDictionaries (BioD/bio/std/hts/utils/samheadermerger.d:127-177)
builds a graph of @SQ names -- an edge from every line to the next one of the same file -- and takes
DirectedGraph.topologicalSort (utils/graph.d:57-87: Kahn's algorithm, FIFO queue seeded in node order, successors in edge order,
repeated edges counted).  No GPU needed: the function is host code, compiled here with g++."""
import os
import random
import subprocess

import pytest

from tests.util import ROOT

SRC = os.path.join(ROOT, "tests", "native", "merge_dict_host.cpp")


@pytest.fixture(scope="module")
def merge(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("md") / "merge_dict_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, SRC])

    def run(dicts):
        out = subprocess.check_output([exe] + [",".join("%s:%d" % r for r in d) for d in dicts]).decode().splitlines()
        if out[0].startswith("error: "):
            return out[0][7:], None
        merged = [(x.split(":")[0], int(x.split(":")[1])) for x in out[0].split(",")] if out[0] else []
        maps = [[int(v) for v in line.split(",")] if line else [] for line in out[1:1 + len(dicts)]]
        return merged, maps
    return run


def restated(dicts):
    """graph.d / samheadermerger.d, line by line in Python"""
    nodes, index, edges, length = [], {}, [], {}

    def add(name):
        if name not in index:
            index[name] = len(nodes)
            nodes.append(name)
            edges.append([])
        return index[name]
    for d in dicts:
        prev = None
        for name, ln in d:
            if name in length and length[name] != ln:
                return "length"
            length.setdefault(name, ln)
            cur = add(name)
            if prev is not None:
                edges[prev].append(cur)
            prev = cur
    pred = [0] * len(nodes)
    for e in edges:
        for v in e:
            pred[v] += 1
    queue = [v for v in range(len(nodes)) if pred[v] == 0]
    out = []
    while queue:
        v = queue.pop(0)
        out.append(nodes[v])
        for w in edges[v]:
            pred[w] -= 1
            if pred[w] == 0:
                queue.append(w)
    if len(out) < len(nodes):
        return "cycle"
    return [(n, length[n]) for n in out]


def test_known_cases(merge):
    m, maps = merge([[("c1", 30000), ("cEmpty", 2000), ("c2", 9000)], [("c1", 30000), ("c2", 9000), ("cNew", 500)], [("cEmpty", 2000), ("c2", 9000)]])
    assert m == [("c1", 30000), ("cEmpty", 2000), ("c2", 9000), ("cNew", 500)] and maps == [[0, 1, 2], [0, 2, 3], [1, 2]]
    # the queue is FIFO: b (seen in the first file) before c, although c follows a directly in the second file
    m, maps = merge([[("a", 1), ("b", 2)], [("a", 1), ("c", 3)]])
    assert m == [("a", 1), ("b", 2), ("c", 3)] and maps == [[0, 1], [0, 2]]
    # a file whose first contig nobody else lists: it is a node without predecessor, queued in node order
    m, maps = merge([[("x", 5), ("y", 6)], [("w", 4), ("y", 6)]])
    assert m == [("x", 5), ("w", 4), ("y", 6)] and maps == [[0, 2], [1, 2]]
    err, _ = merge([[("a", 1), ("b", 2)], [("b", 2), ("a", 1)]])
    assert "NYI" in err
    err, _ = merge([[("a", 1), ("b", 2)], [("a", 3)]])
    assert err == "can't merge SAM headers: one of references with name a has length 1 while another one with the same name has length 3"


def test_random_dictionaries_against_the_restatement(merge):
    rng = random.Random(99)
    n_err = n_ok = 0
    for trial in range(150):
        universe = [("s%d" % k, rng.randrange(1, 10 ** 6)) for k in range(rng.randrange(1, 12))]
        dicts = []
        for _ in range(rng.randrange(1, 5)):
            d = [r for r in universe if rng.random() < 0.6] or [universe[0]]
            if rng.random() < 0.25:
                rng.shuffle(d)                       # orders that may contradict each other
            if rng.random() < 0.05:
                d[0] = (d[0][0], d[0][1] + 1)        # a length that may disagree
            dicts.append(d)
        want = restated(dicts)
        got, maps = merge(dicts)
        if want == "cycle":
            assert "NYI" in got; n_err += 1
        elif want == "length":
            assert got.startswith("can't merge SAM headers"); n_err += 1
        else:
            assert got == want
            for d, mp in zip(dicts, maps):
                assert [got[i] for i in mp] == d
            n_ok += 1
    assert n_ok > 50 and n_err > 5
