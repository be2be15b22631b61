import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
hb.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libbk_stamps_probe")
n = 256
regions = [synth.make_region(i) for i in range(n)]
eng = hb.Engine(kmer_size=31)
eng.submit([hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions])
for it in range(2):
    eng.run(hb.BK_STAGE_KMER)
st = [eng.stat(100 + i) for i in range(8)]
print("kmer kernel ms", eng.kernel_ms(1))
names = ["P0 ref table", "P1 group reads", "P2 compact", "P3a count", "P3b insert", "P4 sort", "P5 postings"]
for i in range(7):
    print("%-16s %8.1f us" % (names[i], (st[i + 1] - st[i]) / 100.0))

eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
eng.run(hb.BK_STAGE_KMER | hb.BK_STAGE_ASSEMBLE)
print("asm kernel ms", eng.kernel_ms(2))
print("rounds %d planned slots %d retired %d" % (eng.stat(106), eng.stat(104), eng.stat(105)))
acc = [eng.stat(100 + i) / 100.0 for i in range(4)]
print("asm region 0: outside check_read %.1f us | load_read %.1f | DP %.1f | decide+apply+bookkeeping %.1f" % tuple(acc))
