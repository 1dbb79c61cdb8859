import os, time, torch, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
from avlen_amd.harness import Workload
from avlen_amd import _lib as L
from avlen_amd import belief_predictor as BP
for name, kw, cus in (("no belief", dict(), 0), ("sync", dict(belief_predictor=True), 0), ("sync cls-first", dict(belief_predictor=True), 0),
                      ("sync", dict(belief_predictor=True), 0), ("async cus0", dict(belief_predictor=True, belief_async=True), 0), ("async cus32", dict(belief_predictor=True, belief_async=True), 32),
                      ("async cus16", dict(belief_predictor=True, belief_async=True), 16), ("async cus64", dict(belief_predictor=True, belief_async=True), 64)):
    BP._PRED_FIRST = "cls-first" not in name
    wl = Workload(64, 150, spectrogram=(65, 26, 2), **kw)
    L.lib.avlen_set_tower_x3_reserved_cus(cus)
    wl.cycle(); wl.cycle()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): wl.cycle()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(name, "%.1f env-steps/s" % (9600 / dt), "ms/cycle %.1f" % (dt * 1e3), "seq", (wl.seq.fast, wl.seq.slow) if wl.seq else None, flush=True)
    del wl
    L.lib.avlen_set_tower_x3_reserved_cus(0)
