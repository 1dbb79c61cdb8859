"""tower_grad_probe with the dialog fixture's rgb tower weights and the update_dialog test's images."""
import sys, os, json, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fixtures as fx, restate as R
from avlen_amd import _lib as L, engine as E, nets as N

specs = json.load(open(os.path.join(ROOT, "tests", "golden", "param_specs.json")))["dialog"]
pre = "net.visual_encoder.rgb_encoder."
sd_all = fx.state_dict_for({k: tuple(v) for k, v in specs.items()})
sd = {"t." + k[len(pre):]: v for k, v in sd_all.items() if k.startswith(pre)}
net = N.ResNet18Params(3)
net.load_state_dict({k[2:]: v for k, v in sd.items()})
net.cuda()
flat = E.FlatParams(net, ("",))
packed = E.Packed(flat.device)
view = E.resnet18_view(net, packed)
packed.refresh()
gview = E.resnet18_grad_view(view, flat)
imgs = torch.cat([fx.observations(f"dlgupd.obs{t}", 2)["rgb"] for t in range(3)], 0)       # (6,128,128,3)
B = imgs.shape[0]
x = R.resize_center_crop_64(imgs.permute(0, 3, 1, 2) / 255.0).permute(0, 2, 3, 1).contiguous().cuda()
dout = fx.sym("probe.d", (B, 64)).cuda()
nb = L.lib.avlen_resnet18_train_workspace_bytes(C.byref(view), B, 64, 64, 0)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
out = torch.empty(B, 64, device="cuda")
L.call("avlen_resnet18_train_fwd", C.byref(view), E.P(x), B, 64, 64, E.P(out), 64, 0, E.P(ws), nb, L.stream())
L.call("avlen_resnet18_train_bwd", C.byref(view), C.byref(gview), E.P(x), E.P(dout), 64, B, 64, 64, None, 0, E.P(ws), nb, L.stream())
torch.cuda.synchronize()
osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
xc = x.cpu().permute(0, 3, 1, 2).clone()
# hook the pre-ReLU outputs of layer4.0 to count near-threshold units
o = R.custom_resnet18(osd, "t", xc)
(o * dout.cpu()).sum().backward()
print("forward err", float((out.cpu() - o).abs().max() / o.abs().max()))
errs = sorted(((float((flat.grad_view(k[2:], v.shape).cpu().double() - v.grad.double()).norm() / (v.grad.double().norm() + 1e-30)), k)
               for k, v in osd.items()), reverse=True)
for e, k in errs[:6]:
    print(f"  {e:.3g} {k}")
for k in ("t.layer4.1.conv1.weight", "t.layer4.0.conv2.weight", "t.layer4.0.conv1.weight", "t.layer4.0.downsample.0.weight",
          "t.layer4.0.bn2.weight", "t.layer4.0.downsample.1.weight", "t.layer3.1.conv2.weight"):
    print(f"  {dict((b, a) for a, b in errs)[k]:.3g} {k}")
print("--- all, network order (last layer first)")
d = dict((b, a) for a, b in errs)
order = ["fc.weight"] + [f"layer{l}.{b}.{n}" for l in (4, 3, 2, 1) for b in (1, 0) for n in
                         ("bn2.weight", "conv2.weight", "bn1.weight", "conv1.weight", "downsample.1.weight", "downsample.0.weight")] + ["bn1.weight", "conv1.weight"]
for n in order:
    if "t." + n in d:
        print(f"  {d['t.' + n]:.3g} {n}")
