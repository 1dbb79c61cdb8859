"""Reference import harness (TEST INFRASTRUCTURE — survey/build container only).

Loads the *unmodified* reference modules from /root/reference so that golden
vectors can be generated and the CPU restatement in ``oracle/restate.py`` can be
pinned against them.  Nothing here is imported by the product package and it
cannot run on the GPU box (``/root/reference`` is absent there).

The reference's hot-path modules import a number of non-numeric packages that
are not installed in this image (pynvml, torchsummary, habitat, gym, cv2, ...).
They are registered as inert stand-ins in ``sys.modules``;
every numeric code path executed is the reference's own.
"""
import sys, types, os, importlib.machinery
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("AVLEN_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "ss_baselines"))


class _Anything(types.ModuleType):
    """Module stand-in: any attribute is another inert stand-in / callable."""
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _Anything(self.__name__ + "." + name)
        setattr(self, name, v)
        return v

    def __call__(self, *a, **k):
        return None


class _StubFinder:
    """Meta-path finder: any submodule of a stubbed root resolves to an inert stand-in."""
    roots = set()

    @classmethod
    def find_spec(cls, fullname, path=None, target=None):
        if fullname.split(".")[0] in cls.roots:
            return importlib.machinery.ModuleSpec(fullname, cls, is_package=True)
        return None

    @staticmethod
    def create_module(spec):
        m = _Anything(spec.name)
        m.__path__ = []
        return m

    @staticmethod
    def exec_module(module):
        pass


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Return a namespace with the reference's classes."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present: " + REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)

    # bare packages with real __path__ so the trainer-importing __init__s are skipped
    for name, rel in [
        ("ss_baselines", "ss_baselines"),
        ("ss_baselines.common", "ss_baselines/common"),
        ("ss_baselines.savi", "ss_baselines/savi"),
        ("ss_baselines.savi.models", "ss_baselines/savi/models"),
        ("ss_baselines.savi.ppo", "ss_baselines/savi/ppo"),
        ("ss_baselines.av_nav", "ss_baselines/av_nav"),
        ("ss_baselines.av_nav.models", "ss_baselines/av_nav/models"),
    ]:
        _pkg(name, os.path.join(REF, rel))

    # inert stand-ins for non-numeric deps
    sys.meta_path.insert(0, _StubFinder)
    for name in ["pynvml", "pynvml.smi", "torchsummary", "clip", "cv2", "moviepy",
                 "moviepy.editor", "habitat", "habitat.utils", "habitat.utils.visualizations",
                 "habitat.utils.visualizations.utils", "habitat.core", "habitat.core.logging",
                 "habitat_sim", "habitat_sim.utils", "habitat_sim.utils.common",
                 "gym", "gym.spaces", "gym.spaces.dict_space", "gym.spaces.box",
                 "soundspaces", "soundspaces.tasks", "soundspaces.tasks.nav",
                 "ss_baselines.common.tensorboard_utils",
                 "ss_baselines.savi.models.dialog_encoder",
                 "quaternion", "skimage", "skimage.measure", "scipy.io", "librosa",
                 "matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image", "tqdm", "attr",
                 "networkx", "imageio", "seaborn", "mpl_toolkits", "mpl_toolkits.axes_grid1"]:
        if name in sys.modules:
            continue
        root = name.split(".")[0]
        if root == "ss_baselines":      # stub ONE submodule of a real package
            m = _Anything(name); m.__path__ = []
            sys.modules[name] = m
            continue
        if root != "soundspaces":       # the simulator package is always stubbed
            try:
                __import__(name)
                continue
            except Exception:
                pass
        _StubFinder.roots.add(root)
        __import__(name)

    nav = sys.modules["soundspaces.tasks.nav"]
    for cls, uuid in [("PoseSensor", "pose"), ("SpectrogramSensor", "spectrogram"),
                      ("LocationBelief", "location_belief"), ("CategoryBelief", "category_belief"),
                      ("Category", "category")]:
        setattr(nav, cls, type(cls, (), {"cls_uuid": uuid}))

    class Box:  # gym.spaces.Box look-alike (shape only)
        def __init__(self, low=0, high=1, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
    sys.modules["gym.spaces"].Box = Box
    sys.modules["gym.spaces.box"].Box = Box
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]

    # CLIP is absent: `clip.load` hands back a stub whose encode_text is the fixture embedding
    import fixtures as _fx

    class _StubClip(nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer = types.SimpleNamespace(width=512)

        def encode_text(self, tokens):
            return _fx.stub_text_embedding(tokens)
    sys.modules["clip"].load = lambda *a, **k: (_StubClip(), None)

    # torchvision's conv3x3/conv1x1 are two-line Conv2d factories (bias=False)
    if "torchvision" not in sys.modules:
        tv = _pkg("torchvision"); tvm = _pkg("torchvision.models")
        tvr = types.ModuleType("torchvision.models.resnet")
        tvr.conv3x3 = lambda i, o, stride=1, groups=1, dilation=1: nn.Conv2d(
            i, o, 3, stride=stride, padding=dilation, groups=groups, bias=False, dilation=dilation)
        tvr.conv1x1 = lambda i, o, stride=1: nn.Conv2d(i, o, 1, stride=stride, bias=False)
        sys.modules["torchvision.models.resnet"] = tvr
        tv.models = tvm; tvm.resnet = tvr

    ns = types.SimpleNamespace()
    from ss_baselines.savi.ppo import policy as ref_policy
    from ss_baselines.savi.ppo.ppo import PPO
    from ss_baselines.savi.models.rollout_storage import RolloutStorage, ExternalMemory
    from ss_baselines.savi.models.smt_state_encoder import SMTStateEncoder
    from ss_baselines.savi.models.dialog_state_encoder import DialogStateEncoder
    from ss_baselines.savi.models.audio_cnn import AudioCNN
    from ss_baselines.savi.models.visual_cnn import VisualCNN
    from ss_baselines.savi.models.smt_cnn import SMTCNN
    from ss_baselines.av_nav.models.rnn_state_encoder import RNNStateEncoder
    from ss_baselines.savi.models.belief_predictor import BeliefPredictor
    ns.BeliefPredictor = BeliefPredictor
    ns.policy = ref_policy
    ns.PPO = PPO
    ns.RolloutStorage = RolloutStorage
    ns.ExternalMemory = ExternalMemory
    ns.SMTStateEncoder = SMTStateEncoder
    ns.DialogStateEncoder = DialogStateEncoder
    ns.AudioCNN, ns.VisualCNN, ns.SMTCNN = AudioCNN, VisualCNN, SMTCNN
    ns.RNNStateEncoder = RNNStateEncoder
    ns.Box = Box
    _loaded = ns
    return ns


class ObsSpace:
    def __init__(self, spaces):
        self.spaces = spaces


class ActionSpace:  # rollout_storage.py:90 switches on this class *name*
    def __init__(self, n):
        self.n = n


def observation_space(spectrogram=(65, 26, 2), with_category=True):
    ns = load()
    sp = {
        "rgb": ns.Box(shape=(128, 128, 3)),
        "depth": ns.Box(shape=(128, 128, 1)),
        "spectrogram": ns.Box(shape=spectrogram),
        "category": ns.Box(shape=(21,)),
        "category_belief": ns.Box(shape=(21,)),
        "location_belief": ns.Box(shape=(2,)),
        "pose": ns.Box(shape=(4,)),
    }
    if not with_category:
        sp.pop("category")
    return ObsSpace(sp)


if __name__ == "__main__":
    ns = load()
    torch.manual_seed(0)
    pol = ns.policy.AudioNavOptionPolicy(observation_space(), ActionSpace(4), hidden_size=256, nhead=8,
                                         num_encoder_layers=1, num_decoder_layers=1, dropout=0.0,
                                         activation="relu", pretraining=True, use_belief_encoding=False,
                                         use_belief_as_goal=True, use_label_belief=True,
                                         use_location_belief=True, normalize_category_distribution=False,
                                         use_category_input=False, query_count_emb_size=32)
    print(sum(p.numel() for p in pol.parameters()))
