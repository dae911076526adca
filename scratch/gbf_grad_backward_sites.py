import os, sys, random, torch, math
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
from oracle import mmdti_oracle as O
from g9util import tiny_cfg, rel_l2
torch.set_num_threads(8)
def cfg():
    ocfg = tiny_cfg("classification", 40)
    kw = dict(emb_dropout=0.0, dropout=0.0, attn_dropout=0.0, pooler_dropout=0.0)
    ocfg.unimol = O.UniMolCfg(layers=2, dim=512, ffn=256, heads=64, K=128, vocab=31, **kw)
    kw2 = dict(hidden_dropout=0.0, attn_dropout=0.0)
    ocfg.cross, ocfg.roberta = O.CrossCfg(dim=512, heads=16, ffn=128, **kw2), O.RobertaCfg(layers=1, dim=512, heads=8, ffn=128, vocab=40, max_pos=40, **kw2)
    ocfg.infonce_dropout = 0.0
    return ocfg
rng = random.Random(11)
B = rng.choice([2, 3, 5, 8]); nmax = rng.choice([6, 14, 30, 46, 62, 78, 94, 110, 126, 142, 158, 190, 222, 256]); trial = 0
ocfg = cfg()
P = O.init_params(ocfg, seed=12, std=0.05)
batch, label = O.synth_batch(B, nmax, 20, ocfg, seed=1000 + trial, ragged=True)
rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
class GeluB(torch.autograd.Function):      # gelu with its derivative rounded to bf16 (the device's saved / recomputed gelu')
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x); return O.gelu(x)
    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        d = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        return g * rb(d)
mode = {}
orig = O.pair_bias
def pair_bias(dist, et, P, bf16=False):
    g = O.gaussian_layer(dist, et, P)
    if mode.get("dbasis"): g.register_hook(rb) if g.requires_grad else None
    u = O.linear(g, P["gbf_proj.linear1.weight"], P["gbf_proj.linear1.bias"], bf16, f16ok=False)
    if mode.get("du"): u.register_hook(rb)
    h = GeluB.apply(u) if mode.get("gelu") else O.gelu(u)
    o = O.linear(h, P["gbf_proj.linear2.weight"], P["gbf_proj.linear2.bias"], bf16, f16ok=False)
    if mode.get("G"): o.register_hook(rb)
    o = o.permute(0, 3, 1, 2).contiguous()
    return o.view(-1, o.size(-2), o.size(-1))
O.pair_bias = pair_bias
def grads(bf16):
    Pq = {k: v.clone().requires_grad_() for k, v in P.items()}
    out = O.mm_forward(batch, Pq, ocfg, net_target=label, bf16=bf16)
    l, _ = O.step_loss(out, label, "classification")
    l.backward()
    return {k: v.grad for k, v in Pq.items() if v.grad is not None}
ref = grads(False)
names = ("gbf.means.weight", "gbf.stds.weight", "gbf_proj.linear1.weight", "gbf_proj.linear1.bias", "gbf_proj.linear2.weight")
for tag, m in (("emulation", {}), ("+ G bf16", dict(G=1)), ("+ gelu' bf16", dict(gelu=1)), ("+ du bf16", dict(du=1)), ("+ dbasis bf16", dict(dbasis=1)), ("+ all four", dict(G=1, gelu=1, du=1, dbasis=1))):
    mode.clear(); mode.update(m)
    g = grads(True)
    print(f"{tag:18s}", "  ".join(f"{n.split('.')[-2][:9]}.{n.split('.')[-1][:4]} {rel_l2(g[n], ref[n]):.2e}" for n in names))
