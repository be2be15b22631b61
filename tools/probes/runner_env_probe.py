"""runner.run throughput under what bench.py has around it: argv[1] = plain | engines (six handles alive, a step run on each) | torch (torch.cuda initialised)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode.startswith("torch"):
    import torch
    if mode == "torch1": torch.set_num_threads(1)
    if mode != "torchimport": torch.cuda.init(); torch.zeros(1, device="cuda")
import bench
from breakmer_amd import hip_backend as hb, synth
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(512)]
keep = []
if mode == "engines":
    ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions[:256]]
    for i in range(6):
        e = hb.Engine(kmer_size=31); e.submit(ins); e.run(hb.BK_STAGE_ALL); keep.append(e)
import gc
if os.environ.get("BK_PROBE_GC") == "off": gc.disable()
if os.environ.get("BK_PROBE_GC") == "freeze": gc.freeze()
for rep in range(2):
    out = bench.time_runner(synth, regions, 31, cycles=16)
    print(mode, rep, out["value"], out["seconds"])
