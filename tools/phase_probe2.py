import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
hb.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libbk_stamps_probe")
names = ["P0 ref table", "P1 group reads", "P2 compact", "P3a count", "P3b insert", "P4 sort", "P5 postings"]
for label, kw in (("no SV", dict(sv_size=0)), ("del 200", dict())):
    regions = [synth.make_region(i, **kw) for i in range(256)]
    eng = hb.Engine(kmer_size=31)
    eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
    for it in range(2):
        eng.run(hb.BK_STAGE_KMER)
    st = [eng.stat(100 + i) for i in range(8)]
    print(label, "kmer kernel ms %.3f" % eng.kernel_ms(1), " ".join("%s=%.0f" % (names[i].split()[0], (st[i + 1] - st[i]) / 100.0) for i in range(7)), "U", eng.stat(4) // 256, "| phaseA %.0f slow %.0f nslow %d" % ((eng.stat(108) - st[3]) / 100.0, (eng.stat(109) - eng.stat(108)) / 100.0, eng.stat(110)))
