"""Does the CLIP text tower finish sooner as two half-batch graphs on two streams than as one graph?  (GPU otherwise idle.)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch
from avlen_amd.harness import Workload
from avlen_amd import _lib as L, engine as E

wl = Workload(64, 4, spectrogram=(65, 26, 2), precision="bf16", pretraining=True, em_capacity=4, seed=5)
pol = wl.pi_l
eng = pol._engine()
tok = wl.dialog[0].contiguous()
print("tokens", tuple(tok.shape), "non-pad per env (mean):", float((tok != 0).sum(1).float().mean()))

def make(t):
    B = t.shape[0]
    out = torch.empty(B, 512, device="cuda")
    nb = L.lib.avlen_clip_text_workspace_bytes(C.byref(eng["clip"]), B)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def run():
        L.call("avlen_clip_text_fwd", C.byref(eng["clip"]), E.P(t), E.P(out), B, pol.prec, E.P(ws), nb, L.stream())
    return run, out, ws

def capture(run, stream):
    with torch.cuda.stream(stream):
        run(); run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            run()
    return g

s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
full, out_f, _w0 = make(tok)
gf = capture(full, s1)
splits = {}
for frac in (0.5, 0.6):
    k = int(64 * frac)
    ra, oa, _wa = make(tok[:k].contiguous()); rb, ob, _wb = make(tok[k:].contiguous())
    splits[frac] = (capture(ra, s1), capture(rb, s2), oa, ob, _wa, _wb)
k3 = (22, 43)
parts3 = [make(tok[a:b].contiguous()) for a, b in ((0, 22), (22, 43), (43, 64))]
g3 = [capture(p[0], s) for p, s in zip(parts3, (s1, s2, s3))]

def timed(fn, n=30):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def run_full():
    with torch.cuda.stream(s1):
        gf.replay()
    torch.cuda.current_stream().wait_stream(s1)

def run_split(frac):
    ga, gb = splits[frac][:2]
    def f():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ga.replay()
        with torch.cuda.stream(s2):
            gb.replay()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return f

def run3():
    cur = torch.cuda.current_stream()
    for s, g in zip((s1, s2, s3), g3):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            g.replay()
    for s in (s1, s2, s3):
        cur.wait_stream(s)

print("full batch, one graph: %.0f us" % timed(run_full))
for nb in (32, 16, 8):
    r_, o_, w_ = make(tok[:nb].contiguous())
    g_ = capture(r_, s1)
    def run_part(g_=g_):
        with torch.cuda.stream(s1):
            g_.replay()
        torch.cuda.current_stream().wait_stream(s1)
    print("first %d dialogs alone, one graph: %.0f us" % (nb, timed(run_part)))
for frac in splits:
    print("two graphs (%.0f%% / %.0f%%) on two streams: %.0f us" % (100 * frac, 100 - 100 * frac, timed(run_split(frac))))
print("three graphs on three streams: %.0f us" % timed(run3))
oa, ob = splits[0.5][2:4]
print("halves equal the full batch:", float((torch.cat([oa, ob]) - out_f).abs().max()))
