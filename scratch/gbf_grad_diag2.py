"""Isolate the pair-bias backward kernel: feed it the ORACLE's dL/d(bias) (fp32 autograd of the fp32 oracle, p = 0) on the mixed-length
batch of gbf_grad_diag.py trial 0 and compare its eight gradients with the oracle's; then the device's own G against the oracle's."""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
from oracle import mmdti_oracle as O
from g9util import tiny_cfg, product_model, load_fixture_weights, rel_l2
from mmdti_hip import ops
from mmdti_hip.functional import CELossFn
from gbf_grad_diag import cfg, trial_batch   # noqa

seed, want = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (11, 0)
B, nmax, trial = trial_batch(seed, want)
ocfg = cfg(False)
P = O.init_params(ocfg, seed=12, std=0.05)
batch, label = O.synth_batch(B, nmax, 20, ocfg, seed=1000 + trial, ragged=True)
N = batch["src_tokens"].shape[1]
H = 64
# oracle, fp32, with the bias as a retained intermediate
got = {}
orig = O.pair_bias
def pb(dist, et, Pq, bf16=False):
    o = orig(dist, et, Pq, bf16)
    o.retain_grad(); got["bias"] = o
    return o
O.pair_bias = pb
P0 = {k: v.clone().requires_grad_() for k, v in P.items()}
out = O.mm_forward(batch, P0, ocfg, net_target=label, bf16=False)
lo, _ = O.step_loss(out, label, "classification")
lo.backward()
O.pair_bias = orig
G = got["bias"].grad.view(B, H, N, N)                    # dL/d bias, fp32
print("oracle G: |G|", float(G.norm()), " max |row sum| / max |G|", float(G.sum(-1).abs().max() / G.abs().max()))
# device kernel on the oracle's G
dev = {k: v.cuda() for k, v in batch.items()}
ld = ops.pair_ld(N)
Gp = torch.zeros(B, H, N, ld); Gp[..., :N] = torch.nan_to_num(G)
gt = ops.pair_tile(Gp.cuda(), N, 0.0)
w1, w2 = P["gbf_proj.linear1.weight"].cuda().bfloat16(), P["gbf_proj.linear2.weight"].cuda().bfloat16()
names = ("gbf_proj.linear1.weight", "gbf_proj.linear1.bias", "gbf_proj.linear2.weight", "gbf_proj.linear2.bias", "gbf.mul.weight", "gbf.bias.weight", "gbf.means.weight", "gbf.stds.weight")
outg = [torch.zeros(P[n].numel(), device="cuda") for n in names]
ops.gbf_bias_bwd_full(gt, dev["src_distance"], dev["src_edge_type"], P["gbf.mul.weight"].cuda().view(-1), P["gbf.bias.weight"].cuda().view(-1),
                      P["gbf.means.weight"].cuda().view(-1), P["gbf.stds.weight"].cuda().view(-1), w1, P["gbf_proj.linear1.bias"].cuda(), w2, ld, *outg)
for n, g in zip(names, outg):
    print(f"  kernel on the oracle's G: {n:28s} vs fp32 oracle {rel_l2(g.cpu().view(P0[n].grad.shape), P0[n].grad):.2e}")
# the same with bf16-rounded weights in the oracle chain (what the kernel multiplies)
Pw = {k: (v.clone().bfloat16().float() if k in ("gbf_proj.linear1.weight", "gbf_proj.linear2.weight") else v.clone()).requires_grad_() for k, v in P.items() if k.startswith("gbf")}
bo = orig(batch["src_distance"], batch["src_edge_type"], Pw, False).view(B, H, N, N)
(bo * torch.nan_to_num(G)).sum().backward()
for n, g in zip(names, outg):
    print(f"  kernel on the oracle's G: {n:28s} vs fp32 chain with the kernel's bf16 weights {rel_l2(g.cpu().view(Pw[n].grad.shape), Pw[n].grad):.2e}")
# the device's own G
model = product_model(ocfg).cuda().eval()
load_fixture_weights(model, P)
from mmdti_hip.functional import PairBiasFn
cap = {}
real_bwd = PairBiasFn.backward
def bwd(ctx, g):
    st = ctx.st
    cap["g"] = (st.slot.g if st.slot is not None and st.slot.g is not None else g).detach().float().clone()
    return real_bwd(ctx, g)
PairBiasFn.backward = staticmethod(bwd)
logits, infonce, ct = model(**dev, return_infonce_loss=True, return_ct_loss=True, net_target=label.cuda())
(CELossFn.apply(logits, label.cuda()) + 0.1 * infonce + 0.1 * ct).backward()
Gd = ops.pair_untile(cap["g"], N).cpu()
Gm = torch.nan_to_num(G)
print("device G vs oracle G: rel L2", rel_l2(Gd, Gm), " device max |row sum| / max |G|", float(Gd.sum(-1).abs().max() / Gd.abs().max()))
pad = batch["src_tokens"].eq(0)
for b in range(B):
    nb = int((~pad[b]).sum())
    if nb < N:
        print(f"   molecule {b}: {nb} atoms; pad-query rows rel L2 {rel_l2(Gd[b, :, nb:, :nb], Gm[b, :, nb:, :nb]):.2e}, real-query rows {rel_l2(Gd[b, :, :nb, :nb], Gm[b, :, :nb, :nb]):.2e}; "
              f"|G| pad rows {float(Gm[b, :, nb:, :nb].norm()):.3e} real rows {float(Gm[b, :, :nb, :nb].norm()):.3e}")
