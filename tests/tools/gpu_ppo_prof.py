"""stage stamps of k_ppo_grad (a -DMM_PPO_PROF=1 build): MYOSIM_LIB=.../_variants/ppoprof/libmyosim_hip.so python tests/tools/gpu_ppo_prof.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from myosuite_amd import engine as E

od, ad, B, mb = 108, 39, 40960, int(os.environ.get("MB", 5120))
K = E.FusedPPO(od, ad, (64, 64, 64), (64, 64, 64), "sigmoid", max_minibatch=mb, learning_rate=3e-4, clipping_epsilon=0.3, entropy_cost=1e-2,
               value_cost=0.25, max_grad_norm=1.0)
P = K.param_count
dev = "cuda"
torch.manual_seed(0)
p = 0.1 * torch.randn(P, device=dev); g = torch.zeros(P, device=dev)
obs = torch.randn(B, od, device=dev); raw = torch.randn(B, ad, device=dev); lo = -40 + torch.randn(B, device=dev)
adv = torch.randn(B, device=dev); ret = torch.randn(B, device=dev); mean = torch.zeros(od, device=dev); std = torch.ones(od, device=dev)
idx = torch.argsort(torch.rand(B, device=dev))[:mb].contiguous()
nblk = 2 * ((mb + 15) // 16)  # upper bound (16-sample workgroups)
buf = torch.zeros(nblk, 64, dtype=torch.int64, device=dev)
for _ in range(3):
    K.grad(p, obs, mean, std, idx, raw, lo, adv, ret, g)
torch.cuda.synchronize()
E.lib().mm_ppo_debug_set_prof.argtypes = [C.c_void_p]
E.lib().mm_ppo_debug_set_prof(buf.data_ptr())
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); K.grad(p, obs, mean, std, idx, raw, lo, adv, ret, g); ev1.record()
torch.cuda.synchronize()
E.lib().mm_ppo_debug_set_prof(None)
print("grad + reduce launch pair: %.1f us" % (1e3 * ev0.elapsed_time(ev1)))
b = buf.cpu().numpy()
used = (b[:, 0] != 0).sum()
print("workgroups stamped", used)
names = {0: "start", 1: "desc", 2: "load_x", 3: "aux", 4: "forward", 5: "loss"}
for l in range(4): names[20 + l] = f"fwd L{l}"
for l in range(4): names[31 + 3 * l] = f"bwd L{l} dW"; names[32 + 3 * l] = f"bwd L{l} dX"
t0 = b[:used, 0].min()
for blk in (0, used // 2 - 1, used // 2, used - 1):
    st = sorted((int(b[blk, i]), i) for i in names if b[blk, i])
    print(f"workgroup {blk}: first stamp at +{st[0][0] - t0} cycles; stage durations (cycles):", ", ".join(f"{names[i]} {t - st[k - 1][0]}" for k, (t, i) in enumerate(st) if k))
    print("   total", st[-1][0] - st[0][0])
last = max(int(b[k, i]) for k in range(used) for i in names if b[k, i])
print("first start -> last stamp over all workgroups:", last - t0, "cycles; starts spread over", int(b[:used, 0].max() - t0))
