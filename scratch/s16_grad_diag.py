"""Gradient of the gbf parameters through tower 1 at the reference head count: compact (fp16 logits) vs fp32 pair planes vs
the oracle (with / without the s16 site)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
from oracle import mmdti_oracle as O
from g9util import tiny_cfg, product_model, load_fixture_weights, rel_l2, cosine
from mmdti_hip import ops
from mmdti_hip.functional import EmbeddingFn

def run(N, layers=2):
    ocfg = tiny_cfg("classification", 40)
    ocfg.unimol = O.UniMolCfg(layers=layers, dim=512, ffn=256, heads=64, K=128, vocab=31, emb_dropout=0.0, dropout=0.0, attn_dropout=0.0, pooler_dropout=0.0)
    ocfg.cross, ocfg.roberta = O.CrossCfg(dim=512, heads=16, ffn=128, hidden_dropout=0.0, attn_dropout=0.0), O.RobertaCfg(layers=1, dim=512, heads=8, ffn=128, vocab=40, max_pos=40, hidden_dropout=0.0, attn_dropout=0.0)
    P = {k: v.requires_grad_() for k, v in O.init_params(ocfg, seed=11, std=0.05).items()}
    batch, _ = O.synth_batch(2, N - 2, 12, ocfg, seed=N, ragged=False)
    dev = {k: v.cuda() for k, v in batch.items()}
    g = torch.randn(2, N, 512, generator=torch.Generator().manual_seed(1))
    res = {}
    for tag, compact, g16 in (("fp32", False, False), ("s16", True, False), ("s16g16", True, True)):
        ops.PAIR_COMPACT, ops.PAIR_G_BF16 = compact, g16
        model = product_model(ocfg).cuda().eval()
        load_fixture_weights(model, P)
        x = EmbeddingFn.apply(model.embed_tokens.weight, dev["src_tokens"], 0)
        bias = model.pair_bias(dev["src_distance"], dev["src_edge_type"])
        enc, s_last, _ = model.encoder.encode(x, bias, dev["src_tokens"].eq(0))
        (enc * g.cuda()).sum().backward()
        res[tag] = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        res[tag]["_enc"] = enc.detach().cpu()
    for tag, sites in (("oracle", O.ALL_SITES - {"s16"}), ("oracle_s16", O.ALL_SITES)):
        O.BF16_SITES = set(sites)
        for v in P.values(): v.grad = None
        xo = torch.nn.functional.embedding(batch["src_tokens"], P["embed_tokens.weight"], padding_idx=0)
        bo = O.pair_bias(batch["src_distance"], batch["src_edge_type"], P, bf16=True)
        eo, so = O.unimol_encoder(xo, bo, batch["src_tokens"].eq(0), P, ocfg.unimol, bf16=True, with_aux=False)
        (eo * g).sum().backward()
        res[tag] = {n: v.grad.clone() for n, v in P.items() if v.grad is not None}
        res[tag]["_enc"] = eo.detach()
    O.BF16_SITES = set(O.ALL_SITES)
    names = [n for n in res["fp32"] if n.startswith("gbf") or "layers.0.self_attn.in_proj" in n or n == "_enc"]
    print(f"--- N={N} layers={layers}")
    for n in names:
        a = res["fp32"][n]
        print(f"{n:34s} |fp32|={float(a.norm()):.3e}  " + "  ".join(f"{t}:{rel_l2(res[t][n], a):.2e}" for t in ("s16", "s16g16", "oracle", "oracle_s16")))

if __name__ == "__main__":
    run(130); run(209); run(40, layers=15)
