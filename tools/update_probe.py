"""Where does PPO.update spend its time? (host timers around synchronised phases)"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avlen_amd.harness import Workload

wl = Workload(64, 150, spectrogram=(257, 101, 2), precision="bf16", pretraining=True)
for _ in range(150):
    wl.rollout_step()
torch.cuda.synchronize()
t0 = time.perf_counter(); wl.update(); torch.cuda.synchronize(); print("update #1 (cold)", time.perf_counter() - t0)
for _ in range(150):
    wl.rollout_step()
torch.cuda.synchronize()
ro, ag, pol = wl.rollouts, wl.agent, wl.pi_q
s = ro.step
T = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
t = T()
last = {k: v[s] for k, v in ro.observations.items()}
nv = pol.get_value_option(last, ro.recurrent_hidden_states[s], ro.prev_actions[s], ro.masks[s], ro.external_memory_option[:, s],
                          ro.external_memory_masks[s], ro.query_state[s - 1], ro.last_query_info[s - 1])
ro.compute_returns(nv, True, 0.99, 0.95)
t1 = T(); print("get_value+gae", t1 - t)
adv = ag.get_advantages(ro).contiguous()
perm = torch.randperm(64)
env = perm[:32].cuda()
t2 = T()
b = ro.gather_minibatch(env, adv)
t3 = T(); print("gather_minibatch", t3 - t2)
feats, goal = pol.net.features(pol, b["obs"], b["prev_actions"], extra=b["query_state"])
t4 = T(); print("features (towers+audio, B=4800)", t4 - t3)
log = torch.zeros(1, 6, device="cuda")
ag._minibatch_step(ro, b, log[0])
t5 = T(); print("full minibatch step (incl. features again)", t5 - t4)
t6 = T(); out = ag.update(ro); t7 = T(); print("agent.update (4 minibatches)", t7 - t6)
