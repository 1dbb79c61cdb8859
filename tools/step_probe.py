"""Wall-clock anatomy of one rollout step (N=64): time spent inside each policy call (each ends with the host-side sampling
sync) and the storage insert, with and without launch-ahead; plus pure GPU time of each policy's forward (events)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd.harness import Workload

def run(launch_ahead):
    wl = Workload(64, 150, launch_ahead=launch_ahead)
    for _ in range(20):
        wl.rollout_step()
    torch.cuda.synchronize()
    # monkeypatch timing wrappers
    acc = {}
    def wrap(obj, name, key):
        fn = getattr(obj, name)
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
            return r
        setattr(obj, name, w)
    wrap(wl.pi_q, "act_option", "pi_q.act_option"); wrap(wl.pi_g, "act", "pi_g.act"); wrap(wl.pi_l, "act_dialog", "pi_l.act_dialog")
    wrap(wl.rollouts, "insert", "storage.insert(enqueue)")
    if launch_ahead:
        wrap(wl.pi_q, "prefetch_act_option", "prefetch q"); wrap(wl.pi_g, "prefetch_act", "prefetch g"); wrap(wl.pi_l, "prefetch_act_dialog", "prefetch l")
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        wl.rollout_step()
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"launch_ahead={launch_ahead}: {tot/n*1e3:.3f} ms per step")
    for k, v in acc.items():
        print(f"   {k:28s} {v/n*1e6:8.1f} us")
    print(f"   {'other (python in step)':28s} {(tot - sum(acc.values()))/n*1e6:8.1f} us")
    # pure GPU time per policy forward: replay each graph alone
    for pol, name in ((wl.pi_q, "pi_q"), (wl.pi_g, "pi_g"), (wl.pi_l, "pi_l")):
        for key, g in pol._graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                g.graph.replay()
            e1.record(); torch.cuda.synchronize()
            print(f"   graph {name} {key[0]}/{key[1]}: {e0.elapsed_time(e1)/20*1e3:.0f} us GPU per replay")

if __name__ == "__main__" and len(sys.argv) == 1:
    run(False)
    run(True)


def host_profile():
    import cProfile, pstats
    wl = Workload(64, 150, launch_ahead=True)
    for _ in range(20):
        wl.rollout_step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(60):
        wl.rollout_step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "host":
    host_profile()
