"""Parameter containers for the networks on the hot path.

These classes hold parameters ONLY (names, shapes and initialisers identical to the reference, built in
the reference's construction order so that the same ``torch.manual_seed`` yields the same initial
weights and checkpoints load key-for-key: SURVEY.md App. A).  None of them computes anything: the
arithmetic lives in libavlen_hip.so; ``forward`` raises.  torch.nn containers are used purely as named
parameter registries.

Reference: ss_baselines/savi/models/{audio_cnn,visual_cnn,smt_cnn,smt_resnet,smt_state_encoder,
dialog_state_encoder}.py, ss_baselines/av_nav/models/rnn_state_encoder.py, common/utils.py:61-72,
ss_baselines/savi/ppo/policy.py:279-297.
"""
import math
import torch
import torch.nn as nn


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container; compute runs in libavlen_hip.so")


def _no_fwd(m):
    """torch.nn leaf used as a registry: make an accidental eager call fail loudly."""
    def _raise(*a, **k):
        raise RuntimeError("avlen_amd never runs eager PyTorch layers; call the Policy API")
    m.forward = _raise
    return m


def conv_out(h, k, s, p=0):
    return (h + 2 * p - k) // s + 1


# ---------------------------------------------------------------------------------------------------------
class Cnn3Params(_Holder):
    """AudioCNN (audio_cnn.py:18-134) / VisualCNN (visual_cnn.py:19-143): `cnn.{0,2,4}` convs, `cnn.6` fc."""

    def __init__(self, in_ch, hw, geometry, output_size):
        super().__init__()
        h, w = hw
        chans = [in_ch, 32, 64, 64]
        layers = []
        self.geometry = list(geometry)
        for i, (k, s) in enumerate(geometry):
            layers.append(_no_fwd(nn.Conv2d(chans[i], chans[i + 1], kernel_size=k, stride=s)))
            if i < 2:
                layers.append(nn.Identity())          # ReLU slot (cnn.1 / cnn.3)
            h, w = conv_out(h, k, s), conv_out(w, k, s)
        layers.append(nn.Identity())                  # Flatten slot (cnn.5)
        layers.append(_no_fwd(nn.Linear(64 * h * w, output_size)))
        layers.append(nn.Identity())                  # ReLU slot (cnn.7)
        self.cnn = nn.Sequential(*layers)
        self.in_hw, self.out_hw, self.in_ch, self.output_size = tuple(hw), (h, w), in_ch, output_size
        for layer in self.cnn:                        # layer_init (audio_cnn.py:126-134): gain passed as `a`
            if isinstance(layer, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
                nn.init.constant_(layer.bias, val=0)


def audio_geometry(h, w):
    """audio_cnn.py:44-49."""
    return [(5, 2), (3, 2), (3, 1)] if (h < 30 or w < 30) else [(8, 4), (4, 2), (3, 1)]


VISUAL_GEOMETRY = [(8, 4), (4, 2), (3, 2)]            # visual_cnn.py:46-48


class _BasicBlock(_Holder):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = _no_fwd(nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False))
        self.bn1 = _no_fwd(nn.GroupNorm(16, planes))
        self.relu = nn.Identity()
        self.conv2 = _no_fwd(nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False))
        self.bn2 = _no_fwd(nn.GroupNorm(16, planes))
        self.downsample = downsample
        self.stride = stride


class ResNet18Params(_Holder):
    """CustomResNet (smt_resnet.py:56-149): GroupNorm(16), width/4, 64x64 input, fc 8192->64."""

    def __init__(self, num_input_channels):
        super().__init__()
        self.inplanes = 16
        self.conv1 = _no_fwd(nn.Conv2d(num_input_channels, 16, kernel_size=7, stride=1, padding=3, bias=False))
        self.bn1 = _no_fwd(nn.GroupNorm(16, 16))
        self.relu = nn.Identity()
        self.layer1 = self._make_layer(16, 1)
        self.layer2 = self._make_layer(32, 2)
        self.layer3 = self._make_layer(64, 2)
        self.layer4 = self._make_layer(128, 2)
        self.fc = _no_fwd(nn.Linear(128 * 8 * 8, 64))
        for m in self.modules():                      # smt_resnet.py:91-96
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.GroupNorm):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(_no_fwd(nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False)),
                                 _no_fwd(nn.GroupNorm(16, planes)))
        blocks = [_BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        blocks.append(_BasicBlock(planes, planes, 1, None))
        return nn.Sequential(*blocks)


class _TvBasicBlock(_Holder):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = _no_fwd(nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False))
        self.bn1 = _no_fwd(nn.BatchNorm2d(planes))
        self.relu = nn.Identity()
        self.conv2 = _no_fwd(nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False))
        self.bn2 = _no_fwd(nn.BatchNorm2d(planes))
        self.downsample = downsample
        self.stride = stride


class TvResNet18Params(_Holder):
    """Parameter container with torchvision.models.resnet18's module tree and state_dict keys (third party; used by
    belief_predictor.py:79-81 with conv1 replaced by Conv2d(2, 64, 7, 2, 3) and fc by Linear(512, n))."""

    def __init__(self, num_input_channels, num_classes):
        super().__init__()
        self.inplanes = 64
        self.conv1 = _no_fwd(nn.Conv2d(num_input_channels, 64, kernel_size=7, stride=2, padding=3, bias=False))
        self.bn1 = _no_fwd(nn.BatchNorm2d(64))
        self.relu = nn.Identity()
        self.maxpool = nn.Identity()
        self.layer1 = self._make_layer(64, 1)
        self.layer2 = self._make_layer(128, 2)
        self.layer3 = self._make_layer(256, 2)
        self.layer4 = self._make_layer(512, 2)
        self.avgpool = nn.Identity()
        self.fc = _no_fwd(nn.Linear(512, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(_no_fwd(nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False)),
                                 _no_fwd(nn.BatchNorm2d(planes)))
        blocks = [_TvBasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        blocks.append(_TvBasicBlock(planes, planes, 1, None))
        return nn.Sequential(*blocks)


class SMTCNNParams(_Holder):
    """SMTCNN (smt_cnn.py:32-76): rgb_encoder + depth_encoder, each 64-d."""

    def __init__(self, observation_space):
        super().__init__()
        self.feature_dims = 0
        if "rgb" in observation_space.spaces:
            self.rgb_encoder = ResNet18Params(observation_space.spaces["rgb"].shape[2])
            self.feature_dims += 64
        if "depth" in observation_space.spaces:
            self.depth_encoder = ResNet18Params(observation_space.spaces["depth"].shape[2])
            self.feature_dims += 64

        def weights_init(m):                          # layer_init (smt_cnn.py:67-76)
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, nn.init.calculate_gain("relu"))
                if m.bias is not None:
                    nn.init.constant_(m.bias, val=0)
        self.apply(weights_init)


class SMTStateEncoderParams(_Holder):
    """SMTStateEncoder (smt_state_encoder.py:27-96)."""

    def __init__(self, input_size, nhead=8, num_encoder_layers=1, num_decoder_layers=1, dim_feedforward=256,
                 dropout=0.1, activation="relu", pose_indices=None, pretraining=False, **_):
        super().__init__()
        assert num_encoder_layers == 1 and num_decoder_layers == 1 and activation == "relu", \
            "avlen_amd implements the 1-encoder/1-decoder ReLU configuration every AVLEN yaml uses"
        assert dropout == 0.0, "dropout must be 0 (all AVLEN yamls); the HIP path has no dropout"
        self._input_size, self._nhead, self._dim_feedforward = input_size, nhead, dim_feedforward
        self._pose_indices, self._pretraining = pose_indices, pretraining
        assert pose_indices is not None and pose_indices[1] - pose_indices[0] == 4
        self.pose_encoder = _no_fwd(nn.Linear(5, 16))
        fused = input_size + 16 - 4
        self.fusion_encoder = nn.Sequential(_no_fwd(nn.Linear(fused, dim_feedforward)), nn.Identity(),
                                            _no_fwd(nn.Linear(dim_feedforward, dim_feedforward)))
        self.transformer = _no_fwd(nn.Transformer(d_model=dim_feedforward, nhead=nhead, num_encoder_layers=1,
                                                  num_decoder_layers=1, dim_feedforward=dim_feedforward,
                                                  dropout=dropout, activation=activation))

    @property
    def hidden_state_size(self):
        return self._dim_feedforward

    @property
    def pose_indices(self):
        return self._pose_indices


class _PositionalEncoding(_Holder):
    def __init__(self, d_model, max_len):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


class DialogStateEncoderParams(_Holder):
    """DialogStateEncoder (dialog_state_encoder.py:42-99)."""

    def __init__(self, input_size, nhead=8, num_encoder_layers=1, num_decoder_layers=1, dim_feedforward=256,
                 dropout=0.1, activation="relu", pretraining=False, **_):
        super().__init__()
        assert dropout == 0.0 and activation == "relu"
        self._dim_feedforward, self._nhead = dim_feedforward, nhead
        self.fusion_encoder = nn.Sequential(_no_fwd(nn.Linear(input_size, dim_feedforward)), nn.Identity(),
                                            _no_fwd(nn.Linear(dim_feedforward, dim_feedforward)))
        self.dialog_transformer = _no_fwd(nn.Transformer(d_model=dim_feedforward, nhead=nhead, num_encoder_layers=1,
                                                         num_decoder_layers=1, dim_feedforward=dim_feedforward,
                                                         dropout=dropout, activation=activation))
        self.pos_encode = _PositionalEncoding(dim_feedforward, 100)

    @property
    def hidden_state_size(self):
        return self._dim_feedforward


class _ClipBlock(_Holder):
    def __init__(self, width, heads):
        super().__init__()
        self.attn = _no_fwd(nn.MultiheadAttention(width, heads))
        self.ln_1 = _no_fwd(nn.LayerNorm(width))
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", _no_fwd(nn.Linear(width, width * 4)))
        self.mlp.add_module("gelu", nn.Identity())
        self.mlp.add_module("c_proj", _no_fwd(nn.Linear(width * 4, width)))
        self.ln_2 = _no_fwd(nn.LayerNorm(width))


class _ClipTransformer(_Holder):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[_ClipBlock(width, heads) for _ in range(layers)])


class ClipTextParams(_Holder):
    """Text tower of OpenAI CLIP ViT-B/32 (public definition; call site policy.py:761,847-849).  Only the
    text side is instantiated: the reference loads the visual tower too but never calls it.
    Parameter names follow the CLIP package so `net.clip.*` checkpoint keys load (strict=False)."""

    def __init__(self, vocab_size=49408, context_length=77, width=512, heads=8, layers=12, embed_dim=512):
        super().__init__()
        self.context_length, self.vocab_size = context_length, vocab_size
        self.transformer = _ClipTransformer(width, layers, heads)
        self.token_embedding = _no_fwd(nn.Embedding(vocab_size, width))
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.ln_final = _no_fwd(nn.LayerNorm(width))
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim))
        self.heads = heads
        # CLIP.initialize_parameters
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        attn_std, fc_std = width ** -0.5, (2 * width) ** -0.5
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=width ** -0.5)


class RNNStateEncoderParams(_Holder):
    """RNNStateEncoder (rnn_state_encoder.py:11-47): 1-layer GRU, orthogonal weights, zero biases."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self._num_recurrent_layers = 1
        self.rnn = _no_fwd(nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=1))
        for name, param in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(param)
            elif "bias" in name:
                nn.init.constant_(param, 0)

    @property
    def num_recurrent_layers(self):
        return 1


class CategoricalNetParams(_Holder):
    def __init__(self, num_inputs, num_outputs):       # common/utils.py:61-68
        super().__init__()
        self.linear = _no_fwd(nn.Linear(num_inputs, num_outputs))
        nn.init.orthogonal_(self.linear.weight, gain=0.01)
        nn.init.constant_(self.linear.bias, 0)


class CriticHeadParams(_Holder):
    def __init__(self, input_size, n_out=1):           # policy.py:279-297
        super().__init__()
        self.fc = _no_fwd(nn.Linear(input_size, n_out))
        nn.init.orthogonal_(self.fc.weight)
        nn.init.constant_(self.fc.bias, 0)
