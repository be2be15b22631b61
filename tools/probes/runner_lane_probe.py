"""runner.run() over 8,192 synthetic targets as bench.py's runner_end_to_end times it, through the batch lane and the per-target way
    python tools/probes/runner_lane_probe.py"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: F401,E402  (as in bench.py: a large heap for the collector to walk)
from breakmer_amd import hip_backend as hb, synth, sv_processor as sp  # noqa: E402
import bench  # noqa: E402

regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(512)]
for rep in range(2):
    print(bench.time_runner(synth, regions, 31, cycles=16)["value"], "regions/s (bench.time_runner, lane as shipped)", flush=True)
packed = {id(r): hb.pack_reads(r.reads, r.read_lens) for r in regions}
d = tempfile.mkdtemp()
bed, genes, data = [], ["header"], {}
for c in range(16):
    for r in regions:
        name = r.name + ("C%d" % c if c else "")
        bed.append("\t".join([r.chrom, str(r.start), str(r.end), name, "exon"]))
        genes.append("\t".join(["0", name, "chr" + r.chrom, "+", str(r.start), str(r.end)] + ["x"] * 6 + [name]))
        data[name.upper()] = sp.RegionData(r.read_ids, None, None, None, r.window_str, [], r.disc_reads, read_codes=r.reads, read_lens=r.read_lens, read_packed=packed[id(r)])
open(os.path.join(d, "t.bed"), "w").write("\n".join(bed) + "\n")
open(os.path.join(d, "g.txt"), "w").write("\n".join(genes) + "\n")
cfg = {"analysis_name": "x", "targets_bed_file": os.path.join(d, "t.bed"), "gene_annotation_file": os.path.join(d, "g.txt"), "kmer_size": "31", "keep_repeat_regions": True, "batch_regions": 256}
ref = None
for way in (True, False, True, False):
    for depth in (2, 3):
        cfg["submit_depth"] = depth
        t0 = time.perf_counter(); run = sp.runner(cfg, region_data=data, batch_lane=way); rows = run.run(); dt = time.perf_counter() - t0
        if ref is None:
            ref = rows
        print("batch_lane %s submit_depth %d: %d targets %.3f s = %.1f k regions/s, %d rows, equal to the first run: %s" % (way, depth, len(data), dt, len(data) / dt / 1e3, len(rows), rows == ref), flush=True)
