"""G7 (tests/golden/surface.json): the per-target surface of the REAL reference driven end to end --
target.compare_kmers -> resolve_sv -> get_summary -> write_results and runner.write_output
(sv_processor.py:212-234, 609-721, 731-866) -- against breakmer_amd.sv_processor.runner on the same inputs:
rows, summary lines and the BYTES of every output file the reference's writers produced.  Files whose record
order comes from CPython set iteration in the reference are compared with their records sorted (flag in the
fixture).  CPU: oracle-backed FakeEngine; the `gpu` variant runs the same comparison on the HIP engine."""
import json
import os
import sys

import pytest

from breakmer_amd import sv_processor as sp, synth

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_engine import FakeEngine  # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "surface.json")))


def _inputs(tmp_path, cases, k):
    bed, genes, data = [], ["header"], {}
    for gname, (chrom, start, end) in GOLD["genes"].items():
        genes.append("\t".join(["0", gname, chrom, "+", str(start), str(end)] + ["x"] * 6 + [gname]))
    for c in cases:
        r = synth.make_region(**c["gen"])
        assert r.name == c["name"]
        reads = r.read_strs()
        quals = ["".join(chr(33 + 20 + ((i * 7 + j) % 20)) for j in range(len(s))) for i, s in enumerate(reads)]
        sc = None if c["sc_mod"] is None else [s[:60] for i, s in enumerate(reads) if i % c["sc_mod"] == 0]
        bed.append("\t".join([r.chrom, str(r.start), str(r.end), r.name, "exon"]))
        data[r.name.upper()] = sp.RegionData(r.read_ids, reads, r.indel_only.tolist(), sc, r.window_str,
                                             [(p[0], p[1], p[2], p[3], synth.codes_to_str(p[4])) for p in r.partners], r.disc_reads, quals=quals)
    (tmp_path / "targets.bed").write_text("\n".join(bed) + "\n")
    (tmp_path / "genes.txt").write_text("\n".join(genes) + "\n")
    cfg = {"analysis_name": "g7k%d" % k, "targets_bed_file": str(tmp_path / "targets.bed"), "analysis_dir": str(tmp_path / "analysis"),
           "reference_data_dir": str(tmp_path / "ref"), "gene_annotation_file": str(tmp_path / "genes.txt"), "kmer_size": str(k),
           "keep_repeat_regions": True}
    return cfg, data


def _canon(kind, text):
    if kind == "sorted_lines":
        return "\n".join(sorted(text.split("\n")))
    if kind == "sorted_records":
        recs = text.split("\n")
        return "\n".join(sorted("\n".join(recs[i:i + 4]) for i in range(0, len(recs) - 3, 4)))
    if kind == "sorted_read_ids":
        ln = text.split("\n")
        ln[2] = ",".join(sorted(ln[2].split(",")))
        return "\n".join(ln)
    return text


def check_surface(tmp_path, engine_factory, native_calls):
    ks = sorted({c["k"] for c in GOLD["cases"]})
    nfiles = 0
    for k in ks:
        cases = [c for c in GOLD["cases"] if c["k"] == k]
        d = tmp_path / ("k%d_%s" % (k, "n" if native_calls else "p"))
        d.mkdir()
        cfg, data = _inputs(d, cases, k)
        run = sp.runner(cfg, region_data=data, engine_factory=engine_factory, native_calls=native_calls)
        rows = run.run()
        want_rows = [r for c in sorted(cases, key=lambda c: c["name"]) for r in c["rows"]]
        assert rows == want_rows, k
        for c in cases:
            t = run.targets[c["name"].upper()]
            assert run.summary[c["name"]] == c["summary"], c["tag"]
            assert len(t.kmers["clusters"]) == c["n_clusters"], c["tag"]
            for fn, (kind, want) in c["files"].items():
                if fn.startswith("@output/"):
                    path = d / "analysis" / "output" / c["name"] / fn[len("@output/"):]
                else:
                    path = d / "analysis" / "targets" / c["name"] / fn
                assert path.is_file(), (c["tag"], fn)
                assert _canon(kind, path.read_text()) == want, (c["tag"], fn)
                nfiles += 1
            if not c["rows"]:                                         # runner.run removes the output directory of a target without results (:201)
                assert not (d / "analysis" / "output" / c["name"]).exists(), c["tag"]
        for fn, want in GOLD["run_files"][str(k)].items():
            assert (d / "analysis" / "output" / fn).read_text() == want, fn
            nfiles += 1
    assert nfiles >= 50
    return nfiles


def test_surface_matches_reference(tmp_path):
    """oracle-backed engine, Python call tail (breakmer_amd/sv_caller.py)"""
    check_surface(tmp_path, lambda prm: FakeEngine(prm.get_kmer_size(), prm.get_sr_thresh('min')), native_calls=False)


@pytest.mark.gpu
@pytest.mark.parametrize("native_calls", [False, True], ids=["python_tail", "native_tail"])
def test_surface_matches_reference_gpu(tmp_path, native_calls):
    from breakmer_amd import hip_backend
    hip_backend.load_library()
    check_surface(tmp_path, None, native_calls)
