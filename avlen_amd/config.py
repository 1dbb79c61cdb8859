"""Every switch of the package in one place.  Read ONCE at import from the environment (so a launcher can set them), documented here;
nothing else in `avlen_amd/` reads `os.environ`.  Decisions that were measured and settled are constants in the code, not knobs
(DESIGN.md lists what was tried and rejected).

    AVLEN_AUTO_AHEAD=0      no automatic launch-ahead of the followers after `share_encoders` (EncoderGroup.auto_launch); the
                            followers then run when they are called.  Default on.
    AVLEN_TEXT_CACHE=0      no per-row memo of the frozen CLIP text tower (every dialog runs the 12 blocks every call).  Default on.
    AVLEN_FOLD_TEXT=0       rollout text graph without text_projection folded into dialog_layer (A/B of the folding).  Default on.
    AVLEN_CLIP_STREAM=0     launch-per-GEMM text tower instead of the one-launch sequence-stationary tower (the reference kernels
                            the one-launch tower is tested against).  Default on.
    AVLEN_MAPPED_ACTIONS=0  sampled actions copied to the host by a copy launch instead of being stored into mapped pinned memory
                            by the heads kernel (host_actions then waits on an event instead of polling).  Default on.
    AVLEN_NATIVE_ALLREDUCE=1  DDPPO's gradient all-reduce through the library's own RCCL binding (avlen_grad_allreduce: one in-place
                            ncclAllReduce(avg) on the backward's stream, communicator bootstrapped over torch.distributed) instead of
                            torch.distributed.all_reduce + a scaling launch.  Default off: the torch path is the one exercised on
                            multi-GPU nodes so far.
    AVLEN_ROCTX=1           roctx ranges (rocprofv3 --marker-trace) around act* / dialog_ready / insert / update: the counterpart
                            of the reference trainer's pth_time / env_time bookkeeping (ppo_trainer.py:326-328, 726-734, 896).
                            Default off (a push / pop pair costs ~1 us per call).

The process's stream layout (caller's stream, the group's side stream, the storage's stream, the capture stream) is tuned for the
HIP runtime's default of FOUR hardware queues per process: GPU_MAX_HW_QUEUES other than 4 measured 3-25 % slower;
`check_hw_queues()` warns once when the variable is set to something else.
"""
import ctypes
import os
import warnings


def _flag(name, default):
    return os.environ.get(name, "1" if default else "0") != "0"


AUTO_AHEAD = _flag("AVLEN_AUTO_AHEAD", True)
TEXT_CACHE = _flag("AVLEN_TEXT_CACHE", True)
FOLD_TEXT = _flag("AVLEN_FOLD_TEXT", True)
CLIP_STREAM = _flag("AVLEN_CLIP_STREAM", True)
MAPPED_ACTIONS = _flag("AVLEN_MAPPED_ACTIONS", True)
ROCTX = _flag("AVLEN_ROCTX", False)
NATIVE_ALLREDUCE = _flag("AVLEN_NATIVE_ALLREDUCE", False)


def check_hw_queues():
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    if v is not None and v.strip() != "4":
        warnings.warn(f"avlen_amd: GPU_MAX_HW_QUEUES={v}: the rollout's stream layout is tuned for the runtime's default of 4 hardware "
                      "queues per process (2 / 3 / 5 / 6 / 8 queues measured 22 / 22 / 16 / 25 / 25 k env-steps/s against 29 k)",
                      RuntimeWarning)


# ---- roctx ranges ----------------------------------------------------------------------------------------------------------------
_roctx = None
if ROCTX:
    for _name in ("libroctx64.so", "librocprofiler-sdk-roctx.so"):
        try:
            _roctx = ctypes.CDLL(_name)
            _roctx.roctxRangePushA.argtypes = [ctypes.c_char_p]
            break
        except OSError:
            _roctx = None


class _Range:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name.encode()

    def __enter__(self):
        _roctx.roctxRangePushA(self.name)

    def __exit__(self, *exc):
        _roctx.roctxRangePop()
        return False


class _NoRange:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        return False


_NO = _NoRange()
_RANGES = {}


def trace_range(name):
    """`with trace_range("act_option"):` -- a roctx range when AVLEN_ROCTX=1, nothing otherwise."""
    if _roctx is None:
        return _NO
    r = _RANGES.get(name)
    if r is None:
        r = _RANGES[name] = _Range("avlen." + name)
    return r


def add_ranges(cls, names, prefix=""):
    """Wrap methods `names` of `cls` in roctx ranges -- only when AVLEN_ROCTX=1 (no wrapper, no overhead otherwise)."""
    if _roctx is None:
        return
    import functools
    for n in names:
        fn = getattr(cls, n)

        def w(*a, _fn=fn, _r=trace_range(prefix + n), **k):
            with _r:
                return _fn(*a, **k)
        setattr(cls, n, functools.wraps(fn)(w))
