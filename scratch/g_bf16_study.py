"""What does keeping the pair-logit gradient G (dL/dS_l, chained through the 15 layers) in bf16 cost in gradient accuracy?
CPU: oracle with a backward hook that rounds the gradient flowing into each layer's bias input to bf16."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import mmdti_oracle as O
from g9util import refarch_cfg

class RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return x.view_as(x)
    @staticmethod
    def backward(ctx, g): return g.to(torch.bfloat16).to(torch.float32)

orig_layer = O.unimol_layer
def run(mode):
    def layer(x, bias, P, pre, cfg, training=False, bf16=False):
        if mode == "bf16G":
            bias = RoundGrad.apply(bias)
        return orig_layer(x, bias, P, pre, cfg, training, bf16)
    O.unimol_layer = layer
    cfg = refarch_cfg("classification", 600)
    P = {k: v.requires_grad_() for k, v in O.init_params(cfg, seed=92, std=0.02).items()}
    batch, label = O.synth_batch(4, 24, 32, cfg, seed=3, ragged=True)
    out = O.mm_forward(batch, P, cfg, net_target=label, training=True, bf16=(mode != "fp32"))
    loss, _ = O.step_loss(out, label, "classification")
    loss.backward()
    O.unimol_layer = orig_layer
    return {k: v.grad.clone() for k, v in P.items() if v.grad is not None}

g32, g16, g16G = run("fp32"), run("bf16"), run("bf16G")
def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
keys = [k for k in g32 if k.startswith(("encoder.", "gbf", "embed")) and "key.bias" not in k and "linear2.bias" not in k]
import statistics
for name, g in (("bf16 contract (fp32 G)", g16), ("bf16 contract + bf16 G", g16G)):
    errs = {k: rel(g[k], g32[k]) for k in keys if float(g32[k].abs().max()) > 0}
    worst = max(errs.items(), key=lambda t: t[1])
    print(f"{name:28s} vs fp32: median {statistics.median(errs.values()):.3e}  worst {worst[1]:.3e} ({worst[0]})")
errs = {k: rel(g16G[k], g16[k]) for k in keys if float(g16[k].abs().max()) > 0}
worst = max(errs.items(), key=lambda t: t[1])
print(f"bf16 G vs fp32 G (same contract): median {statistics.median(errs.values()):.3e}  worst {worst[1]:.3e} ({worst[0]})")
for k in ("gbf.means.weight", "gbf_proj.linear1.weight", "encoder.layers.0.self_attn.in_proj.weight", "encoder.layers.14.self_attn.in_proj.weight"):
    print(k, f"{rel(g16[k], g32[k]):.3e} -> {rel(g16G[k], g32[k]):.3e}")
