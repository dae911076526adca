"""Which output of a 32-molecule training step differs between two identical runs (a 1-ulp difference in the loss was seen)?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
from oracle import mmdti_oracle as O
from g9util import tiny_cfg, product_model, load_fixture_weights, host_fields
from mmdti_hip.runtime import dropout_state
from mmdti_hip.trainer import FineTuner
ocfg = tiny_cfg("classification", 40)
ocfg.unimol = O.UniMolCfg(layers=3, dim=512, ffn=256, heads=64, K=128, vocab=31)
ocfg.cross, ocfg.roberta = O.CrossCfg(dim=512, heads=16, ffn=256), O.RobertaCfg(layers=2, dim=512, heads=8, ffn=256, vocab=40, max_pos=300)
P = O.init_params(ocfg, seed=12, std=0.05)
model = product_model(ocfg, dropout=True).cuda().train()
load_fixture_weights(model, P)
tuner = FineTuner(model, "classification", total_steps=1000)
for trial in (3, 9, 19):
    batch, label = O.synth_batch(32, [46, 49, 89][(3, 9, 19).index(trial)], 40, ocfg, seed=2000 + trial, ragged=True)
    dev = {k: v.cuda() for k, v in batch.items()}; dev.update(host_fields(batch))
    outs = []
    for rep in range(3):
        dropout_state.reseed(500 + trial)
        o = tuner.forward_backward(dev, label.cuda(), 0, False)
        torch.cuda.synchronize()
        outs.append((o.logits.clone(), o.task_loss.clone(), o.infonce_loss.clone(), o.ct_loss.clone(), o.loss.clone()))
    names = ("logits", "task", "infonce", "ct", "loss")
    print(trial, {n: all(torch.equal(outs[0][i], outs[r][i]) for r in (1, 2)) for i, n in enumerate(names)}, flush=True)
