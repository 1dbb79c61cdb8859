"""Probe: latency from a host trigger to the completion of a tiny kernel, (a) launched at the trigger (one recorded MULTICOPY through
avlen_cmds_run) and (b) enqueued BEFORE the trigger behind hipStreamWaitValue32 on a word of mapped pinned memory that the host
then writes.  Usage: python tools/waitvalue_probe.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avlen_amd import _lib as L

hip = C.CDLL("libamdhip64.so")
hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
hip.hipStreamWaitValue32.restype = C.c_int
flag = torch.zeros(16, dtype=torch.int32).pin_memory()
out = torch.zeros(16, dtype=torch.int64).pin_memory()
src = torch.arange(1, 4097, dtype=torch.int64, device="cuda")
st = torch.cuda.Stream()
raw = st.cuda_stream
fnp, onp = flag.numpy(), out.numpy()

def cmd_copy(k):
    srcs = (C.c_void_p * 1)(src.data_ptr() + 8 * k); dsts = (C.c_void_p * 1)(out.data_ptr()); sizes = (C.c_int64 * 1)(8)
    cmds = (L.Cmd * 1)()
    cmds[0].op, cmds[0].n = L.CMD_MULTICOPY, 1
    cmds[0].a, cmds[0].b, cmds[0].c, cmds[0].d = C.cast(srcs, C.c_void_p).value, C.cast(dsts, C.c_void_p).value, C.cast(sizes, C.c_void_p).value, raw
    return cmds, (srcs, dsts, sizes)

def wait_out(v, t0):
    while onp[0] != v:
        if time.perf_counter() - t0 > 1.0:
            return None
    return (time.perf_counter() - t0) * 1e6

torch.cuda.synchronize()
res_a, res_b = [], []
for k in range(1, 41):
    cmds, keep = cmd_copy(k)
    time.sleep(0.0005)
    t0 = time.perf_counter()
    L.call("avlen_cmds_run", cmds, 1)
    res_a.append(wait_out(k + 1, t0))
print("launch at the trigger: median %.1f us, min %.1f us" % (sorted(res_a)[len(res_a) // 2], min(res_a)))
for k in range(41, 81):
    cmds, keep = cmd_copy(k)
    rc = hip.hipStreamWaitValue32(raw, flag.data_ptr(), k, 0, 0xffffffff)           # 0 = hipStreamWaitValueGte
    if rc != 0:
        print("hipStreamWaitValue32 failed:", rc); break
    L.call("avlen_cmds_run", cmds, 1)
    time.sleep(0.0005)
    t0 = time.perf_counter()
    fnp[0] = k
    r = wait_out(k + 1, t0)
    if r is None:
        print("gated kernel never ran (value", k, ")"); break
    res_b.append(r)
if res_b:
    print("enqueued behind a wait-value, released by a host store: median %.1f us, min %.1f us" % (sorted(res_b)[len(res_b) // 2], min(res_b)))
torch.cuda.synchronize()
