"""Do two HIP graphs (chains of small dependent kernels) replayed on two streams overlap on the GPU?  Times one chain, two chains
replayed on ONE stream, two chains on two streams, and the host time of a replay call."""
import time, torch
dev = "cuda"
N = 45
def chain(x):
    for _ in range(N):
        x = x * 1.0001 + 0.5
    return x
xs = [torch.zeros(64 * 1024, device=dev) for _ in range(2)]
graphs = []
cap = torch.cuda.Stream()
for x in xs:
    with torch.cuda.stream(cap):
        chain(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        y = chain(x)
    graphs.append(g)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    t_host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, t_host * 1e6
def one():
    with torch.cuda.stream(s1): graphs[0].replay()
def two_same():
    with torch.cuda.stream(s1): graphs[0].replay(); graphs[1].replay()
def two_streams():
    with torch.cuda.stream(s1): graphs[0].replay()
    with torch.cuda.stream(s2): graphs[1].replay()
for name, fn in (("one chain", one), ("two chains, one stream", two_same), ("two chains, two streams", two_streams)):
    tot, host = timed(fn)
    print(f"{name}: {tot:.1f} us per iteration (host time of the replay calls {host:.1f} us)")
def default_and_side():
    graphs[0].replay()                                   # current (default) stream
    with torch.cuda.stream(s2): graphs[1].replay()
def side_then_default():
    with torch.cuda.stream(s2): graphs[1].replay()
    graphs[0].replay()
for name, fn in (("two chains, default stream + side stream", default_and_side), ("two chains, side stream first then default", side_then_default)):
    tot, host = timed(fn)
    print(f"{name}: {tot:.1f} us per iteration (host time of the replay calls {host:.1f} us)")
