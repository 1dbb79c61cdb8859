"""Standalone gradient check of the GroupNorm ResNet-18 training path at the towers' 64x64 extent vs torch autograd on the oracle."""
import sys, os, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fixtures as fx, restate as R
from avlen_amd import _lib as L, engine as E, nets as N
from avlen_amd.spaces import savi_observation_space

Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
net = N.ResNet18Params(Cin)
spec = {k: tuple(v.shape) for k, v in net.state_dict().items()}
sd = fx.state_dict_for({"t." + k: v for k, v in spec.items()})
net.load_state_dict({k[2:]: v for k, v in sd.items()})
net.cuda()
flat = E.FlatParams(net, ("",))
packed = E.Packed(flat.device)
view = E.resnet18_view(net, packed)
packed.refresh()
gview = E.resnet18_grad_view(view, flat)
x = fx.uni("probe.x", (B, 64, 64, Cin)).cuda()
dout = fx.sym("probe.d", (B, 64)).cuda()
nb = L.lib.avlen_resnet18_train_workspace_bytes(C.byref(view), B, 64, 64, 0)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
out = torch.empty(B, 64, device="cuda")
L.call("avlen_resnet18_train_fwd", C.byref(view), E.P(x), B, 64, 64, E.P(out), 64, 0, E.P(ws), nb, L.stream())
dx = torch.empty_like(x)
L.call("avlen_resnet18_train_bwd", C.byref(view), C.byref(gview), E.P(x), E.P(dout), 64, B, 64, 64, E.P(dx), 0, E.P(ws), nb, L.stream())
torch.cuda.synchronize()
osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
xc = x.cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
o = R.custom_resnet18(osd, "t", xc)
(o * dout.cpu()).sum().backward()
print("forward err", float((out.cpu() - o).abs().max() / o.abs().max()))
worst = []
for k, v in osd.items():
    mine = flat.grad_view(k[2:], v.shape).cpu().double()
    err = float((mine - v.grad.double()).norm() / (v.grad.double().norm() + 1e-30))
    worst.append((err, k))
worst.sort(reverse=True)
print("worst param grads:", worst[:5])
print("dx err", float((dx.cpu().permute(0, 3, 1, 2) - xc.grad).norm() / xc.grad.norm()))
