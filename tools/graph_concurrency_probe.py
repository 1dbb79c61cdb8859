"""Do parallel branches of a captured HIP graph actually overlap on this stack?"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import small_probe as sp

f1, f2 = sp.gemm16(64, 64, 8192), sp.gemm16(64, 64, 8192)
def serial():
    for _ in range(20): f1()
    for _ in range(20): f2()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def forked():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        for _ in range(20): f1()
    with torch.cuda.stream(s2):
        for _ in range(20): f2()
    cur.wait_stream(s1); cur.wait_stream(s2)

def timeit(fn, graph):
    fn(); torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        run = g.replay
    else:
        run = fn
    run(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 10 * 1e6

for name, fn in (("serial", serial), ("forked", forked)):
    for graph in (False, True):
        print(f"{name:7s} graph={graph}: {timeit(fn, graph):8.1f} us for 40 tiny-GEMM launches", flush=True)
