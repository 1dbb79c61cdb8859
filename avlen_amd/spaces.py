"""Minimal observation/action-space objects (shape carriers) for harnesses that run without habitat/gym.
The policies only read `.spaces[name].shape` and `.n`; the class NAME `ActionSpace` selects int64 actions in
the rollout storage exactly as in the reference (rollout_storage.py:90,102)."""


class Box:
    def __init__(self, shape, low=0.0, high=1.0, dtype="float32"):
        self.shape, self.low, self.high, self.dtype = tuple(shape), low, high, dtype


class ObsSpace:
    def __init__(self, spaces):
        self.spaces = dict(spaces)


class ActionSpace:
    def __init__(self, n):
        self.n = n


def savi_observation_space(spectrogram=(65, 26, 2), image=128, with_category=True):
    sp = {"rgb": Box((image, image, 3)), "depth": Box((image, image, 1)), "spectrogram": Box(spectrogram),
          "category": Box((21,)), "category_belief": Box((21,)), "location_belief": Box((2,)), "pose": Box((4,))}
    if not with_category:
        sp.pop("category")
    return ObsSpace(sp)


SMT_KW = dict(hidden_size=256, nhead=8, num_encoder_layers=1, num_decoder_layers=1, dropout=0.0, activation="relu",
              use_pretrained=False, pretrained_path="", use_belief_encoding=False, use_belief_as_goal=True,
              use_label_belief=True, use_location_belief=True, normalize_category_distribution=False)
