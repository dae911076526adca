"""The pair-bias table gradients on the batch shapes where two runs of the same step used to differ by up to 21 % (VERDICT r03 item 2:
scratch/ragged_stress.py seed 11 trial 61: B = 5, N = 159): (a) run-to-run repeatability of a training step (dropout on, reseeded),
(b) distance from the fp32 oracle's gradient with dropout off."""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
from oracle import mmdti_oracle as O
from g9util import tiny_cfg, product_model, load_fixture_weights, rel_l2
from mmdti_hip import collate
from mmdti_hip.runtime import dropout_state
from mmdti_hip.functional import CELossFn


def cfg(p):
    ocfg = tiny_cfg("classification", 40)
    kw = {} if p else dict(emb_dropout=0.0, dropout=0.0, attn_dropout=0.0, pooler_dropout=0.0)
    ocfg.unimol = O.UniMolCfg(layers=2, dim=512, ffn=256, heads=64, K=128, vocab=31, **kw)
    kw2 = {} if p else dict(hidden_dropout=0.0, attn_dropout=0.0)
    ocfg.cross, ocfg.roberta = O.CrossCfg(dim=512, heads=16, ffn=128, **kw2), O.RobertaCfg(layers=1, dim=512, heads=8, ffn=128, vocab=40, max_pos=40, **kw2)
    if not p:
        ocfg.infonce_dropout = 0.0
    return ocfg


def trial_batch(seed, want):
    rng = random.Random(seed)
    for trial in range(want + 1):
        B = rng.choice([2, 3, 5, 8])
        nmax = rng.choice([6, 14, 30, 46, 62, 78, 94, 110, 126, 142, 158, 190, 222, 256])
    return B, nmax, trial


def main():
  names = ("gbf.means.weight", "gbf.stds.weight", "gbf.mul.weight", "gbf.bias.weight", "gbf_proj.linear1.weight", "gbf_proj.linear1.bias", "gbf_proj.linear2.weight",
           "encoder.layers.0.self_attn.in_proj.weight", "embed_tokens.weight")
  for seed, want in ((11, 61), (11, 0), (3, 7)):
      B, nmax, trial = trial_batch(seed, want)
      # (a) repeatability, dropout on
      ocfg = cfg(True)
      P = O.init_params(ocfg, seed=12, std=0.05)
      model = product_model(ocfg).cuda().train()
      load_fixture_weights(model, P)
      batch, label = O.synth_batch(B, nmax, 20, ocfg, seed=1000 + trial, ragged=True)
      dev = {k: v.cuda() for k, v in batch.items()}
      counts = collate.atom_counts(batch["src_tokens"], 0)

      def step(m, **extra):
          dropout_state.reseed(77 + trial)
          m.zero_grad(set_to_none=True)
          logits, infonce, ct = m(**dev, **extra, return_infonce_loss=True, return_ct_loss=True, net_target=label.cuda())
          loss = CELossFn.apply(logits, label.cuda()) + 0.1 * infonce + 0.1 * ct
          loss.backward()
          torch.cuda.synchronize()
          return float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

      l1, g1 = step(model)
      l2, g2 = step(model)
      l3, g3 = step(model, atom_counts=counts)
      allrep = max(rel_l2(g2[n], g1[n]) for n in g1 if float(g1[n].abs().max()) > 0 and "linear2.bias" not in n and "pooler" not in n and "key.bias" not in n)
      print(f"seed {seed} trial {trial}: B={B} N={batch['src_tokens'].shape[1]} lens={counts.tolist()} loss {l1:.7f} {l2:.7f} ragged {l3:.7f}; worst dense-vs-dense over all parameters {allrep:.2e}")
      # (b) oracle distance, dropout off
      ocfg0 = cfg(False)
      model0 = product_model(ocfg0).cuda().eval()
      load_fixture_weights(model0, P)
      _, h = step(model0)
      P0 = {k: v.clone().requires_grad_() for k, v in P.items()}
      out = O.mm_forward(batch, P0, ocfg0, net_target=label, bf16=False)
      lo, _ = O.step_loss(out, label, "classification")
      lo.backward()
      Pe = {k: v.clone().requires_grad_() for k, v in P.items()}
      oute = O.mm_forward(batch, Pe, ocfg0, net_target=label, bf16=True)
      le, _ = O.step_loss(oute, label, "classification")
      le.backward()
      for n in names:
          print(f"   {n:45s} |g| {float(g1[n].norm()):.3e}  dense-vs-dense {rel_l2(g2[n], g1[n]):.2e}  ragged-vs-dense {rel_l2(g3[n], g1[n]):.2e}   "
                f"p=0: vs fp32 oracle {rel_l2(h[n], P0[n].grad):.2e}  vs 16-bit emulation {rel_l2(h[n], Pe[n].grad):.2e}  (emulation vs fp32 {rel_l2(Pe[n].grad, P0[n].grad):.2e})")


if __name__ == "__main__":
  main()
