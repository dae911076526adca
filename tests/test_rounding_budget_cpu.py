"""Where does the distance between the bf16 contract and fp32 come from at the reference depth?  (VERDICT r01 'next' 1.)

CPU only: the oracle's emulation of the HIP path's rounding points, one site at a time, at 15L/512/64h + 6L RoBERTa on
a 4-molecule batch.  Result (full table: profiles/r02_rounding_sites_cpu.json, scratch/rounding_sites.py): rounding the
GEMM WEIGHTS to bf16 moves encoder_rep by ~2.8e-3 relative L2, rounding the GEMM INPUTS by ~2.7e-3, the stored q|k|v by
~1.0e-3; attention probabilities and projections are below 4e-4.  They add in quadrature to ~4e-3: the north star's 1e-3
on embeddings needs more than 8 mantissa bits on BOTH operands of every GEMM -- no single rounding point to fix.
Carrying tower 1's pair logits as fp16 (site "s16": the compact pair planes; profiles/r02_s16_budget_cpu.json) is a
hundred times below that budget: < 1e-4 on encoder_rep on its own, invisible in the total."""
import torch

from oracle import mmdti_oracle as O
from g9util import refarch_cfg


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def test_rounding_budget_at_reference_depth():
    cfg = refarch_cfg("classification", 600)
    P = O.init_params(cfg, seed=92, std=0.02)
    batch, label = O.synth_batch(4, 20, 24, cfg, seed=1, ragged=True)
    saved, was16 = set(O.BF16_SITES), O.FWD_F16
    try:
        O.set_forward_fp16(False)             # the attribution of the bf16 contract (every site bf16: MMDTI_FWD_FP16=0)
        with torch.no_grad():
            ref = O.mm_forward(batch, P, cfg, net_target=label, bf16=False)
            err = {}
            for name, sites in (("all", O.ALL_SITES), ("w", {"w"}), ("x", {"x"}), ("qkv", {"qkv"}), ("rest", {"p", "qkv2", "proj"}), ("s16", {"s16"}),
                                ("all_but_s16", O.ALL_SITES - {"s16"})):
                O.BF16_SITES = set(sites)
                o = O.mm_forward(batch, P, cfg, net_target=label, bf16=True)
                err[name] = (_rel(o["enc"], ref["enc"]), _rel(o["bert"], ref["bert"]))
            # the default contract since round 4: "w", "x", "qkv" round to fp16 instead (ops.FWD_F16) -- the same walk, all sites on
            O.BF16_SITES = set(O.ALL_SITES)
            O.set_forward_fp16(True)
            o16 = O.mm_forward(batch, P, cfg, net_target=label, bf16=True)
            err16 = (_rel(o16["enc"], ref["enc"]), _rel(o16["bert"], ref["bert"]))
    finally:
        O.BF16_SITES = saved
        O.set_forward_fp16(was16)
    enc = {k: v[0] for k, v in err.items()}
    assert err16[0] < 1e-3 and err16[1] < 1e-3, err16            # fp16 forward operands: inside the north star's 1e-3 on both towers
    assert 2e-3 < enc["all"] < 6e-3, enc                       # the bf16 contract itself sits 4x above 1e-3 ...
    assert enc["w"] > 1.5e-3 and enc["x"] > 1.5e-3, enc         # ... because of operand rounding on both sides of the GEMMs
    assert enc["qkv"] < 0.5 * enc["all"] and enc["rest"] < 1e-4, enc
    quad = (enc["w"] ** 2 + enc["x"] ** 2 + enc["qkv"] ** 2) ** 0.5
    assert abs(quad - enc["all"]) < 0.35 * enc["all"], (quad, enc)    # independent errors: they add in quadrature
    assert err["all"][1] < 4e-3, err                           # tower 2 (6 post-LN layers): ~2e-3
    assert enc["s16"] < 1.5e-4 and err["s16"][1] == 0.0, err    # fp16 pair logits alone: two orders below the operand rounding ...
    assert abs(enc["all"] - enc["all_but_s16"]) < 0.05 * enc["all"], enc     # ... and invisible next to it
