"""Golden for `PPO.update_dialog` (dialog pre-training of pi_l, ss_baselines/savi/ppo/ppo.py:99-154) from the REFERENCE's own
PPO / RolloutStorage / AudioNavDialogPolicy (build container only):   python oracle/make_goldens_dialog.py
CLIP is absent: `encode_text` is the same stub embedding the pi_l goldens use (oracle/ref_harness.py), so everything except the
frozen text tower is the reference's arithmetic -- including the backward through both ResNet towers, the AudioCNN, the action
encoder, the SMT state encoder, dialog_layer and the dialog state encoder.  Stores outputs only."""
import json
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as fx          # noqa: E402
import ref_harness as rh       # noqa: E402
from make_goldens import save, build, OUT      # noqa: E402

torch.set_num_threads(8)
T, N = 3, 2


def fill(st, set_mem):
    """The stored dialog episode both sides see (written straight into the buffers)."""
    for t in range(T):
        o = fx.observations(f"dlgupd.obs{t}", N)
        for k in st.observations:
            st.observations[k][t].copy_(o[k])
        st.prev_actions[t].copy_(fx.ints(f"dlgupd.pa{t}", (N, 1), 4))
        st.all_dialog[t].copy_(fx.dialog_tokens(f"dlgupd.tok{t}", N))
        st.agent_step[t].copy_(fx.ints(f"dlgupd.as{t}", (N,), 3).float())
        st.o_actions[t].copy_(fx.ints(f"dlgupd.oa{t}", (N,), 3).float() + 1.0)        # classes 1..3 (class 0 has weight 0)
        st.o_masks[t].copy_(torch.tensor([1, 0] if t == 1 else [1, 1]))
    st.o_actions[0, 1] = 0.0                                                               # a weight-0 row
    st.em_vln_masks[:T].copy_(torch.from_numpy((fx.unit("dlgupd.mk", T * N * 3) < 0.7).astype("float32")).view(T, N, 3))
    set_mem(fx.memory("dlgupd.mem", 3, N, 276, 272), fx.sym("dlgupd.memd", (3, N, 256)))
    st.step = T


def main():
    ns = rh.load()
    pol, spec = build(ns, "dialog")
    agent = ns.PPO(pol, 0.2, 2, 2, 0.5, 0.05, lr=2.5e-4, eps=1e-5, max_grad_norm=0.2, use_normalized_advantage=False)
    st = ns.RolloutStorage(T, N, rh.observation_space(), rh.ActionSpace(4), 512, True, 3, 3, 3, 3, 3, 3, 276, 276, 308, 256,
                           num_recurrent_layers=-1, max_dialog_len=77, use_state_memory=True)

    def set_mem(mem, memd):
        st.em_vln.memory.copy_(mem.unsqueeze(1).expand_as(st.em_vln.memory))
        st.em_vln_dialog.memory.copy_(memd.unsqueeze(1).expand_as(st.em_vln_dialog.memory))
    fill(st, set_mem)
    sd0 = {k: v.clone() for k, v in pol.state_dict().items()}
    loss = agent.update_dialog(st)
    sd = pol.state_dict()
    keys = sorted(k for k in sd if sd[k].dtype == torch.float32 and not k.startswith("net.clip."))
    moved = [k for k in keys if not torch.equal(sd[k], sd0[k])]
    save("dialog_update", loss=loss.detach(), param_abs=np.array([float(sd[k].double().abs().sum()) for k in keys]),
         delta_abs=np.array([float((sd[k] - sd0[k]).double().abs().sum()) for k in keys]))
    with open(os.path.join(OUT, "dialog_update_keys.json"), "w") as f:
        json.dump({"keys": keys, "moved": moved}, f)
    print("moved tensors:", len(moved), "of", len(keys), " loss", float(loss))


if __name__ == "__main__":
    main()
