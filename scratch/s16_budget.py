"""What would carrying the pair logits S in fp16 (site "s16") cost in accuracy?  CPU, oracle only, at 15L/512/64h.
`gain` scales the q/k projection weights so that the logits reach pretrained-like magnitudes."""
import sys, os, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mmdti_oracle as O
from g9util import refarch_cfg

def rel(a, b): return float((a - b).norm() / b.norm())

def run(B, atoms, tokens, seed, gain):
    cfg = refarch_cfg("classification", 600)
    P = O.init_params(cfg, seed=92, std=0.02)
    for k in P:
        if "encoder.layers" in k and "in_proj.weight" in k:
            P[k] = P[k].clone(); P[k][:1024] *= gain           # q and k rows
        if k.startswith("gbf_proj.linear2.weight"):
            P[k] = P[k] * gain
    batch, label = O.synth_batch(B, atoms, tokens, cfg, seed=seed, ragged=True)
    rows = {}
    with torch.no_grad():
        O.BF16_SITES = set()
        ref = O.mm_forward(batch, P, cfg, net_target=label, training=False, bf16=False)
        xo = torch.nn.functional.embedding(batch["src_tokens"], P["embed_tokens.weight"], padding_idx=0)
        bo = O.pair_bias(batch["src_distance"], batch["src_edge_type"], P, bf16=False)
        _, S = O.unimol_encoder(xo, bo, batch["src_tokens"].eq(0), P, cfg.unimol, bf16=False, with_aux=False)
        fin = torch.isfinite(S)
        rows["S_absmax"] = float(S[fin].abs().max()); rows["S_rms"] = float(S[fin].pow(2).mean().sqrt())
        for name, sites in (("contract", O.ALL_SITES), ("contract+s16", set(O.ALL_SITES) | {"s16"}), ("only_s16", {"s16"})):
            O.BF16_SITES = set(sites)
            o = O.mm_forward(batch, P, cfg, net_target=label, training=False, bf16=True)
            rows[name] = dict(enc=rel(o["enc"], ref["enc"]), pooled=rel(o["pooled"], ref["pooled"]), logits=rel(o["logits"], ref["logits"]),
                              infonce=abs(float(o["infonce"]) - float(ref["infonce"])) / float(ref["infonce"]),
                              ct=abs(float(o["ct"]) - float(ref["ct"])) / max(float(ref["ct"]), 1e-9))
        O.BF16_SITES = set(O.ALL_SITES)
    return rows

if __name__ == "__main__":
    out = {}
    for gain in (1.0, 8.0, 20.0):
        rows = run(6, 40, 48, 3, gain)
        out[f"gain{gain}"] = rows
        print(f"--- gain {gain}: |S| max {rows['S_absmax']:.2f} rms {rows['S_rms']:.3f}")
        for k, v in rows.items():
            if isinstance(v, dict):
                print(f"{k:14s} " + "  ".join(f"{n}={x:.2e}" for n, x in v.items()))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_s16_budget_cpu.json"), "w"), indent=1)
