"""Which bf16 rounding point accounts for the distance between the bf16 contract and fp32 at the reference depth?
CPU only: the oracle's emulate-bf16 mode with one site switched on / off at a time (VERDICT r01 'next' 1)."""
import sys, os, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mmdti_oracle as O
from g9util import refarch_cfg, T

def rel(a, b): return float((a - b).norm() / b.norm())

def run(B, atoms, tokens, seed):
    cfg = refarch_cfg("classification", 600)
    P = O.init_params(cfg, seed=92, std=0.02)
    batch, label = O.synth_batch(B, atoms, tokens, cfg, seed=seed, ragged=True)
    with torch.no_grad():
        O.BF16_SITES = set()
        ref = O.mm_forward(batch, P, cfg, net_target=label, training=False, bf16=False)
        rows = {}
        def one(name, sites):
            O.BF16_SITES = set(sites)
            o = O.mm_forward(batch, P, cfg, net_target=label, training=False, bf16=True)
            rows[name] = dict(enc=rel(o["enc"], ref["enc"]), bert=rel(o["bert"], ref["bert"]), pooled=rel(o["pooled"], ref["pooled"]),
                              infonce=abs(float(o["infonce"]) - float(ref["infonce"])) / float(ref["infonce"]),
                              ct=abs(float(o["ct"]) - float(ref["ct"])) / max(float(ref["ct"]), 1e-9))
        one("all", O.ALL_SITES)
        for s in sorted(O.ALL_SITES):
            one("only_" + s, {s})
        for s in sorted(O.ALL_SITES):
            one("all_but_" + s, O.ALL_SITES - {s})
        O.BF16_SITES = set(O.ALL_SITES)
    return rows

if __name__ == "__main__":
    out = {}
    for (B, a, t, seed) in ((4, 20, 24, 1), (8, 64, 96, 2)):
        rows = run(B, a, t, seed)
        out[f"B{B}_atoms{a}_tokens{t}"] = rows
        print(f"--- B={B} atoms<={a} tokens<={t}")
        for k, v in rows.items():
            print(f"{k:16s} " + "  ".join(f"{n}={x:.2e}" for n, x in v.items()))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_rounding_sites_cpu.json"), "w"), indent=1)
