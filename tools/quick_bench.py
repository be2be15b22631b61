import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from breakmer_amd import hip_backend as hb, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t = time.time(); regions = [synth.make_region(i) for i in range(n)]; print("gen %.2fs" % (time.time() - t))
eng = hb.Engine(kmer_size=31)
t = time.time(); ins = [hb.RegionInput(r.reads, r.window, read_lens=r.read_lens) for r in regions]; print("to ascii %.2fs" % (time.time() - t))
t = time.time(); eng.submit(ins); print("submit %.2fs" % (time.time() - t))
for it in range(3):
    t = time.time(); eng.run(); dt = time.time() - t
    print("run %.1f ms wall | kernels total %.2f kmer %.2f asm %.2f ms | regions/s %.0f" % (dt * 1e3, eng.kernel_ms(0), eng.kernel_ms(1), eng.kernel_ms(2), n / dt))
print("cells %d calls %d contigs %d uniq %d mers %d" % (eng.stat(0), eng.stat(1), eng.stat(6), eng.stat(4), eng.stat(5)))
t = time.time(); c = [eng.contigs(i) for i in range(n)]; print("fetch contigs %.1f ms" % ((time.time() - t) * 1e3), sum(len(x) for x in c))
