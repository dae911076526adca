"""Would fp16 GEMM operands (the reference's own AMP dtype, tasks/trainer.py:181-182; same MFMA rate as bf16 on gfx950) bring the
embeddings inside the north star's 1e-3?  CPU only: the oracle's rounding-site emulation with the sites rounding to float16
instead of bfloat16 (VERDICT r02 'next' 3b).  Forward only, fp32 everywhere else.  -> profiles/r03_rounding_sites_fp16.json"""
import sys, os, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mmdti_oracle as O
from g9util import refarch_cfg

def rel(a, b): return float((a - b).norm() / b.norm())

def run(B, atoms, tokens, seed, std):
    cfg = refarch_cfg("classification", 600)
    P = O.init_params(cfg, seed=92, std=std)
    batch, label = O.synth_batch(B, atoms, tokens, cfg, seed=seed, ragged=True)
    rows = {}
    with torch.no_grad():
        O.BF16_SITES = set()
        ref = O.mm_forward(batch, P, cfg, net_target=label, training=False, bf16=False)
        def one(name, dtype, sites):
            O.ROUND_DTYPE, O.BF16_SITES = dtype, set(sites)
            o = O.mm_forward(batch, P, cfg, net_target=label, training=False, bf16=True)
            rows[name] = dict(enc=rel(o["enc"], ref["enc"]), bert=rel(o["bert"], ref["bert"]), logits=rel(o["logits"], ref["logits"]),
                              infonce=abs(float(o["infonce"]) - float(ref["infonce"])) / float(ref["infonce"]),
                              ct=abs(float(o["ct"]) - float(ref["ct"])) / max(float(ref["ct"]), 1e-9))
        gemm = {"w", "x", "qkv"}
        for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            one(f"{tag}_w_x_qkv", dt, gemm)                       # the sites VERDICT names: GEMM weights, GEMM inputs, stored q|k|v of tower 1
            one(f"{tag}_all_gemm_sites", dt, gemm | {"qkv2", "p"})  # + q, k, v and probabilities of towers 2 / fusion
            one(f"{tag}_all_sites", dt, O.ALL_SITES)              # + the fp16 pair logits (always fp16: "s16")
            for s in ("w", "x", "qkv"):
                one(f"{tag}_only_{s}", dt, {s})
        O.ROUND_DTYPE, O.BF16_SITES = torch.bfloat16, set(O.ALL_SITES)
    return rows

if __name__ == "__main__":
    out = {"note": "relative L2 (enc, bert, logits) / relative error (infonce, ct) of the oracle with the named sites rounding to the named "
                   "dtype against the pure-fp32 oracle; reference architecture 15L/512/64h + 6L RoBERTa + fusion, forward only, dropout off"}
    for (B, a, t, seed, std) in ((4, 20, 24, 1, 0.02), (8, 64, 96, 2, 0.02), (8, 64, 96, 2, 0.05)):
        rows = run(B, a, t, seed, std)
        out[f"B{B}_atoms{a}_tokens{t}_std{std}"] = rows
        print(f"--- B={B} atoms<={a} tokens<={t} weight std {std}")
        for k, v in rows.items():
            print(f"{k:24s} " + "  ".join(f"{n}={x:.2e}" for n, x in v.items()))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_rounding_sites_fp16.json"), "w"), indent=1)
