"""Where the time of a step goes when every step submits its batch again (bench.py's value_with_submit): the library's own split of a
packed submit (stats 20, 21: row copies / H2D + waits), and the wall time of the driver's thread inside each call of the loop.
    python tools/probes/with_submit_probe.py [n_engines] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from breakmer_amd import hip_backend as hb, synth  # noqa: E402

n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
sub_th = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # host threads per submit (bk_config.submit_threads; 0: library default)
regions = [synth.make_region(i, depth=500, L=150, sv_type="del") for i in range(256)]
packs = [hb.pack_reads(r.reads, r.read_lens) for r in regions]
pins = [hb.RegionInput(None, r.window, packed=p) for r, p in zip(regions, packs)]
items = [(p, r.window_str.encode(), None) for r, p in zip(regions, packs)]
import bench  # noqa: E402  (the same call context as bench.py's loop)
ctx = bench.call_context_text(regions, bench.default_opts())

e = hb.Engine(kmer_size=31, submit_threads=sub_th)
for it in range(4):
    t = time.perf_counter(); e.submit(pins); dt = time.perf_counter() - t
    print("one handle, packed submit %d (waited for): %.2f ms wall; row copies %.2f ms, H2D + waits %.2f ms" % (it, dt * 1e3, e.stat(20) / 1e3, e.stat(21) / 1e3), flush=True)
    e.run(7)
e.close()

for how in ("submit", "submit_packed"):
    engs = [hb.Engine(kmer_size=31, submit_threads=sub_th) for _ in range(n_eng)]
    tm = {"sync": 0.0, "fetch": 0.0, "call": 0.0, "submit": 0.0, "ctx": 0.0, "run": 0.0}
    kms = [0.0] * 4

    def loop(k):
        n = len(engs); half = max(1, n // 2)
        state = ["idle"] * n
        done = started = t = 0
        while done < k:
            i = t % n; g = engs[i]
            if state[i] == "running":
                a0 = time.perf_counter(); g.sync(); a = time.perf_counter(); g.fetch(); b = time.perf_counter(); g.call_blob(); c = time.perf_counter()
                tm["sync"] += a - a0; tm["fetch"] += b - a; tm["call"] += c - b
                for q in range(4):
                    kms[q] += g.kernel_ms(q)
                done += 1; state[i] = "idle"
            if state[i] == "idle" and started < k:
                a = time.perf_counter()
                if how == "submit":
                    g.submit(pins, wait=False)
                else:
                    g.submit_packed(items, wait=False)
                tm["submit"] += time.perf_counter() - a
                state[i] = "submitted"; started += 1
            j = (t + half) % n; f = engs[j]
            if state[j] == "submitted":
                a = time.perf_counter(); f.set_call_context(ctx); b = time.perf_counter(); f.run(hb.BK_STAGE_ALL, sync=False); c = time.perf_counter()
                tm["ctx"] += b - a; tm["run"] += c - b
                state[j] = "running"
            t += 1
    loop(n_eng)
    for k_ in tm:
        tm[k_] = 0.0
    kms[:] = [0.0] * 4
    t0 = time.perf_counter(); loop(steps); dt = time.perf_counter() - t0
    print("[submit threads %d] " % sub_th + "%s, %d handles, %d steps: %.3f ms per step = %.1f k regions/s; driver thread per step: %s; last submit of handle 0: row copies %.2f ms, H2D + waits %.2f ms"
          % (how, n_eng, steps, dt / steps * 1e3, 256 * steps / dt / 1e3, ", ".join("%s %.3f" % (k_, v / steps * 1e3) for k_, v in tm.items()), engs[0].stat(20) / 1e3, engs[0].stat(21) / 1e3), flush=True)
    print("   kernels of a batch on the GPU's clock (first to last event / k-mer / assembler / realign): %s ms" % " / ".join("%.2f" % (v / steps) for v in kms), flush=True)
    for g in engs:
        g.close()
