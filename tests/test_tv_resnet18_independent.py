"""The label classifier of BeliefPredictor is `torchvision.models.resnet18` (belief_predictor.py:79-81; un-vendored, absent here:
parity unpinned against torchvision itself).  The oracle's restatement (`oracle/restate.py:tv_resnet18`, the checker of
`avlen_belief_*` in tests/test_gpu_belief.py) is cross-checked here against an INDEPENDENT public implementation of the same
architecture, Hugging Face `transformers` `ResNetForImageClassification` (basic blocks, 64-128-256-512, 7x7/2 stem + 3x3/2 max
pool, eval-mode BatchNorm, global average pool, fc), with the SAME random weights mapped from torchvision's parameter names."""
import pytest
import torch

from oracle import restate as R

transformers = pytest.importorskip("transformers")


def _random_tv_resnet18(cin=2, classes=21, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    sd, p = {}, "classifier"

    def conv(name, co, ci, k):
        sd[name + ".weight"] = rn(co, ci, k, k, std=(ci * k * k) ** -0.5)

    def bn(name, c):
        sd[name + ".weight"] = 1 + rn(c, std=0.1); sd[name + ".bias"] = rn(c, std=0.1)
        sd[name + ".running_mean"] = rn(c, std=0.2); sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)

    conv(p + ".conv1", 64, cin, 7); bn(p + ".bn1", 64)
    ci = 64
    for li, co in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in (0, 1):
            q = f"{p}.layer{li}.{bi}"
            conv(q + ".conv1", co, ci if bi == 0 else co, 3); bn(q + ".bn1", co)
            conv(q + ".conv2", co, co, 3); bn(q + ".bn2", co)
            if bi == 0 and li > 1:
                conv(q + ".downsample.0", co, ci, 1); bn(q + ".downsample.1", co)
        ci = co
    sd[p + ".fc.weight"] = rn(classes, 512, std=512 ** -0.5); sd[p + ".fc.bias"] = rn(classes, std=0.1)
    return sd


def _to_hf(sd):
    p, out = "classifier", {}

    def cb(dst, conv, bn):
        out[dst + ".convolution.weight"] = sd[conv + ".weight"]
        for k in ("weight", "bias", "running_mean", "running_var"):
            out[dst + ".normalization." + k] = sd[bn + "." + k]

    cb("resnet.embedder.embedder", p + ".conv1", p + ".bn1")
    for li in (1, 2, 3, 4):
        for bi in (0, 1):
            q, h = f"{p}.layer{li}.{bi}", f"resnet.encoder.stages.{li - 1}.layers.{bi}"
            cb(h + ".layer.0", q + ".conv1", q + ".bn1")
            cb(h + ".layer.1", q + ".conv2", q + ".bn2")
            if q + ".downsample.0.weight" in sd:
                cb(h + ".shortcut", q + ".downsample.0", q + ".downsample.1")
    out["classifier.1.weight"], out["classifier.1.bias"] = sd[p + ".fc.weight"], sd[p + ".fc.bias"]
    return out


def test_oracle_tv_resnet18_equals_huggingface_resnet18():
    from transformers import ResNetConfig, ResNetForImageClassification
    sd = _random_tv_resnet18()
    cfg = ResNetConfig(num_channels=2, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2], layer_type="basic",
                       hidden_act="relu", downsample_in_first_stage=False, num_labels=21)
    hf = ResNetForImageClassification(cfg).eval()
    missing, unexpected = hf.load_state_dict(_to_hf(sd), strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    x = torch.randn(5, 2, 65, 26, generator=torch.Generator().manual_seed(1))           # the sensor's spectrogram, NCHW
    with torch.no_grad():
        ours = R.tv_resnet18(sd, "classifier", x)
        theirs = hf(pixel_values=x).logits
    assert ours.shape == theirs.shape == (5, 21)
    assert float((ours - theirs).abs().max()) < 2e-5 * max(1.0, float(theirs.abs().max())), float((ours - theirs).abs().max())
