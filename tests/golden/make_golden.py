"""Generate the golden vectors under tests/golden/ from the reference itself.

Run ONCE in the build container (where /root/reference is mounted read-only):

    python tests/golden/make_golden.py

It imports the pieces of the reference that are importable there
 -- models/infonce.py, models/contrastive.py, utils/util.py,
models/fds.py, models/mm_module.py -- plus the installed HuggingFace
RobertaModel (tower 2's arithmetic), feeds them small seeded inputs and stores
inputs + weights + outputs (+ gradients) as .npz.  Nothing from the reference
is copied: the fixtures are data only.  The GPU box has no /root/reference;
tests read only the .npz files.
"""
import importlib.util
import os
import sys
import types
import logging
import tempfile

import numpy as np
import torch

REF = os.environ.get("MMDTI_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, len(out), "arrays")


def grads_of(loss, *xs):
    gs = torch.autograd.grad(loss, xs, allow_unused=True)
    return [torch.zeros_like(x) if g is None else g for g, x in zip(gs, xs)]


# ---------------------------------------------------------------- G1 / G2
def g_infonce():
    m = load_by_path("ref_infonce", os.path.join(REF, "models/infonce.py"))
    for B in (2, 16):
        g = torch.Generator().manual_seed(100 + B)
        q = torch.randn(B, 50, generator=g, requires_grad=True)
        k = torch.randn(B, 50, generator=g, requires_grad=True)
        loss = m.info_nce(q, k, temperature=0.1)
        dq, dk = grads_of(loss, q, k)
        npz(f"g1_info_nce_B{B}", q=q, k=k, loss=loss, dq=dq, dk=dk)
    # The explicit-negatives branches (infonce.py:71-88) are unreachable in practice: the symmetric
    # F.cross_entropy(logits.T, labels) at :98 raises for any non-square logits.  Pinned as "raises" in the tests.
    # G2: the module at reduced width (projection width stays 50), unmasked mean incl. padded positions
    for mode in ("eval", "train_p0"):
        torch.manual_seed(11)
        mod = m.InfoNCE(64, 64)
        if mode == "eval":
            mod.eval()
        else:
            mod.train()
            mod.embed_dropout = 0.0
        g = torch.Generator().manual_seed(12)
        xq = torch.randn(5, 7, 64, generator=g, requires_grad=True)
        xk = torch.randn(5, 9, 64, generator=g, requires_grad=True)
        with torch.no_grad():
            xq[3:, 5:] = 0.0          # "padded" rows still enter the mean
        loss = mod(xq, xk)
        params = list(mod.parameters())
        gs = grads_of(loss, xq, xk, *params)
        arrays = dict(xq=xq, xk=xk, loss=loss, dxq=gs[0], dxk=gs[1])
        for (n, p), gp in zip(mod.named_parameters(), gs[2:]):
            arrays["w_" + n] = p
            arrays["g_" + n] = gp
        npz(f"g2_infonce_module_{mode}", **arrays)


# ---------------------------------------------------------------- G3
def g_contrastive():
    m = load_by_path("ref_contrastive", os.path.join(REF, "models/contrastive.py"))
    cases = {}
    for B in (8, 32):
        g = torch.Generator().manual_seed(200 + B)
        f = torch.randn(B, 64, generator=g)
        y = torch.randn(B, 1, generator=g) * 0.3
        yhat = y + 0.2 * torch.randn(B, 1, generator=g)
        wts = torch.rand(B, generator=g) + 0.5
        for w in (0.2, 1.0):
            for use_w in (False, True):
                fx = f.clone().requires_grad_(True)
                px = yhat.clone().requires_grad_(True)
                loss = m.CT_Regress(fx, y, px, weights=wts if use_w else None, w=w)
                df, dp = grads_of(loss, fx, px)
                cases[f"regress_B{B}_w{w}_uw{int(use_w)}"] = dict(f=f, y=y, yhat=yhat, wts=wts, w=w, use_w=use_w, loss=loss, df=df)
        # SupCon
        yc = (torch.rand(B, 1, generator=g) < 0.3).long()
        for use_w in (False, True):
            fx = f.clone().requires_grad_(True)
            if use_w:
                loss = m.CT_Single(fx, yc, None, weights=wts, w=0.2)
            else:
                loss = m.CT_Single(fx, yc, None, w=0.2)
            (df,) = grads_of(loss, fx)
            cases[f"single_B{B}_uw{int(use_w)}"] = dict(f=f, y=yc, wts=wts, use_w=use_w, loss=loss, df=df)
        # all-same label (no negatives -> zero loss), and one singleton class (denom 0 -> 1)
        fx = f.clone().requires_grad_(True)
        loss = m.CT_Single(fx, torch.ones(B, 1).long(), None)
        (df,) = grads_of(loss, fx)
        cases[f"single_same_B{B}"] = dict(f=f, y=torch.ones(B, 1).long(), loss=loss, df=df)
        y1 = torch.zeros(B, 1).long(); y1[0] = 1
        fx = f.clone().requires_grad_(True)
        loss = m.CT_Single(fx, y1, None)
        (df,) = grads_of(loss, fx)
        cases[f"single_singleton_B{B}"] = dict(f=f, y=y1, loss=loss, df=df)
        # regress with no negatives anywhere
        fx = f.clone().requires_grad_(True)
        loss = m.CT_Regress(fx, torch.zeros(B, 1), torch.zeros(B, 1), w=0.2)
        (df,) = grads_of(loss, fx)
        cases[f"regress_noneg_B{B}"] = dict(f=f, y=torch.zeros(B, 1), yhat=torch.zeros(B, 1), w=0.2, loss=loss, df=df)
        # multilabel
        ym = (torch.rand(B, 6, generator=g) < 0.4).long()
        for use_w in (False, True):
            fx = f.clone().requires_grad_(True)
            loss = m.CT_Multi(fx, ym, None, weights=wts if use_w else None)
            (df,) = grads_of(loss, fx)
            cases[f"multi_B{B}_uw{int(use_w)}"] = dict(f=f, y=ym, wts=wts, use_w=use_w, loss=loss, df=df)
    flat = {}
    for cn, d in cases.items():
        for k, v in d.items():
            flat[f"{cn}__{k}"] = v
    npz("g3_contrastive", **flat)


# ---------------------------------------------------------------- G4 / G7
def _stub_utils():
    util = load_by_path("ref_util", os.path.join(REF, "utils/util.py"))
    stub = types.ModuleType("utils")
    stub.calibrate_mean_var = util.calibrate_mean_var
    stub.logger = logging.getLogger("ref")
    sys.modules["utils"] = stub
    return util


def g_fds_and_pad():
    util = _stub_utils()
    # G7 pad helpers
    g = torch.Generator().manual_seed(5)
    toks = [torch.randint(1, 30, (n,), generator=g) for n in (5, 9, 3)]
    d2 = [torch.rand(n, n, generator=g) for n in (5, 9, 3)]
    co = [torch.rand(n, 3, generator=g) for n in (5, 9, 3)]
    arrays = {}
    for i, (t, d, c) in enumerate(zip(toks, d2, co)):
        arrays[f"tok{i}"], arrays[f"d{i}"], arrays[f"c{i}"] = t, d, c
    arrays["pad1d"] = util.pad_1d_tokens(toks, 0)
    arrays["pad2d"] = util.pad_2d(d2, 0.0)
    arrays["padc"] = util.pad_coords(co, 0.0)
    npz("g7_pad", **arrays)

    # calibrate_mean_var branches
    g = torch.Generator().manual_seed(6)
    x = torch.randn(6, 8, generator=g)
    m1, m2 = torch.randn(8, generator=g), torch.randn(8, generator=g)
    v1, v2 = torch.rand(8, generator=g) + 0.1, torch.rand(8, generator=g) * 30
    out_full = util.calibrate_mean_var(x.clone(), m1, v1, m2, v2)
    v1z = v1.clone(); v1z[[1, 4]] = 0.0
    out_part = util.calibrate_mean_var(x.clone(), m1, v1z, m2, v2)
    out_tiny = util.calibrate_mean_var(x.clone(), m1, torch.zeros(8), m2, v2)
    npz("g4_calibrate", x=x, m1=m1, v1=v1, m2=m2, v2=v2, v1z=v1z, out_full=out_full, out_part=out_part, out_tiny=out_tiny)

    # FDS module: needs `models` as a namespace (so models/__init__ -> unicore is not executed) and a CPU device.
    pkg = types.ModuleType("models"); pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    real_to = torch.Tensor.to

    def cpu_to(self, *a, **k):           # neutralise the hard-coded .to('cuda') (fds.py:84)
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            return self
        return real_to(self, *a, **k)

    torch.Tensor.to = cpu_to
    try:
        fds_mod = load_by_path("models.fds", os.path.join(REF, "models/fds.py"))
        import pandas as pd
        rng = np.random.default_rng(3)
        raw = np.concatenate([rng.normal(0.5, 2.0, size=400), [40.0]])      # one 3-sigma outlier
        for tag, kw in {
            "gauss51": dict(bucket_num=10, bucket_start=0, kernel="gaussian", ks=5, sigma=1, using_scale=True),
            "gauss52_bs2": dict(bucket_num=12, bucket_start=2, kernel="gaussian", ks=5, sigma=2, using_scale=False),
            "triang": dict(bucket_num=8, bucket_start=0, kernel="triang", ks=5, sigma=1, using_scale=True),
            "laplace": dict(bucket_num=8, bucket_start=0, kernel="laplace", ks=5, sigma=1, using_scale=True),
        }.items():
            with tempfile.TemporaryDirectory() as td:
                csv = os.path.join(td, "train.csv")
                pd.DataFrame({"TARGET": raw}).to_csv(csv, index=False)
                f = fds_mod.FDS(feature_dim=16, raw_data=csv, col_data="TARGET", device="cpu", **kw)
            g = torch.Generator().manual_seed(9)
            n = 300
            if kw["using_scale"]:
                lab = torch.randn(n, 1, generator=g) * 1.2
            else:
                lab = torch.randn(n, 1, generator=g) * 2.0 + 0.5
            lab[0] = float(f.min_value + f.bin_width * kw["bucket_num"])      # the max-label sample (bin == bucket_num)
            lab[1] = float(f.min_value - 1.0)                                  # below range
            feats0 = torch.randn(n, 16, generator=g) * (1 + lab.abs())
            feats0[:, 3] = 1.5                                                 # zero-variance column
            arrays = dict(raw=raw, labels=lab, feats0=feats0, min_value=f.min_value, bin_width=f.bin_width,
                          window=f.kernel_window, **{"cfg_" + k: (v if not isinstance(v, str) else np.array(v)) for k, v in kw.items()})
            arrays["label_bin"] = torch.Tensor([int((v - f.min_value) // f.bin_width) for v in lab[:, 0]])
            f.update_last_epoch_stats(0)
            f.update_running_stats(feats0.clone(), lab, 0)
            for k, v in f.state_dict().items():
                arrays["s0_" + k] = v.clone()
            # smoothing before any last-epoch stats exist: running_var_last=1, mean 0 -> identity-ish path
            f.update_last_epoch_stats(1)
            for k, v in f.state_dict().items():
                arrays["s1_" + k] = v.clone()
            xb = torch.randn(40, 16, generator=g)
            lb = lab[:40].clone()
            arrays["xb"] = xb
            arrays["smooth1"] = f.smooth(xb.clone(), lb, 1)
            arrays["smooth0"] = f.smooth(xb.clone(), lb, 0)                   # epoch < start_smooth -> unchanged
            feats1 = feats0 * 0.7 + 0.1
            f.update_running_stats(feats1.clone(), lab, 1)
            for k, v in f.state_dict().items():
                arrays["s2_" + k] = v.clone()
            f.update_last_epoch_stats(2)
            arrays["smooth2"] = f.smooth(xb.clone(), lb, 2)
            for k, v in f.state_dict().items():
                arrays["s3_" + k] = v.clone()
            npz(f"g4_fds_{tag}", **arrays)
    finally:
        torch.Tensor.to = real_to


# ---------------------------------------------------------------- G5
def g_cross():
    pkg = types.ModuleType("models"); pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    mm = load_by_path("models.mm_module", os.path.join(REF, "models/mm_module.py"))
    for tag, (D, H, FF) in {"d64h4": (64, 4, 128), "d128h4": (128, 4, 256)}.items():
        cfg = types.SimpleNamespace(hidden_size=D, num_attention_heads=H, intermediate_size=FF,
                                    attention_probs_dropout_prob=0.2, hidden_dropout_prob=0.3,
                                    hidden_act="gelu", layer_norm_eps=1e-12)
        torch.manual_seed(21)
        enc = mm.BertCrossEncoder(cfg, 1).eval()
        with torch.no_grad():
            for n, p in enc.named_parameters():
                if "LayerNorm" in n:
                    p.add_(0.1 * torch.randn_like(p))
                else:
                    p.copy_(0.05 * torch.randn_like(p))
        g = torch.Generator().manual_seed(22)
        B, L1, L2 = 3, 6, 9
        s1 = torch.randn(B, L1, D, generator=g, requires_grad=True)
        s2 = torch.randn(B, L2, D, generator=g, requires_grad=True)
        mask2 = torch.ones(B, L2); mask2[0, 6:] = 0; mask2[2, 3:] = 0
        ext = ((1.0 - mask2) * -10000.0).unsqueeze(1).unsqueeze(2)
        out = enc(s1, s2, ext)[-1]
        gout = torch.randn(out.shape, generator=g)
        loss = (out * gout).sum()
        params = list(enc.parameters())
        gs = grads_of(loss, s1, s2, *params)
        arrays = dict(s1=s1, s2=s2, mask2=mask2, out=out, gout=gout, ds1=gs[0], ds2=gs[1], heads=H)
        for (n, p), gp in zip(enc.named_parameters(), gs[2:]):
            arrays["w_" + n] = p
            if D <= 64:
                arrays["g_" + n] = gp
        npz(f"g5_cross_{tag}", **arrays)


# ---------------------------------------------------------------- G6
def g_roberta():
    os.environ["HF_HUB_OFFLINE"] = "1"
    from transformers import RobertaConfig, RobertaModel
    for attn_impl in ("eager", "sdpa"):
        cfg = RobertaConfig(vocab_size=40, hidden_size=32, num_hidden_layers=2, num_attention_heads=4,
                            intermediate_size=64, max_position_embeddings=24, type_vocab_size=1, pad_token_id=1,
                            layer_norm_eps=1e-12, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                            attn_implementation=attn_impl)
        torch.manual_seed(31)
        m = RobertaModel(cfg).eval()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "LayerNorm" in n:
                    p.add_(0.1 * torch.randn_like(p))
                elif p.dim() == 1:
                    p.copy_(0.05 * torch.randn_like(p))
        g = torch.Generator().manual_seed(32)
        B, L = 4, 11
        ids = torch.randint(4, 40, (B, L), generator=g)
        ids[:, 0] = 0
        lens = [11, 7, 3, 9]
        for b, n in enumerate(lens):
            ids[b, n - 1] = 2
            ids[b, n:] = 1
        ids[1, 4] = 1                       # a pad token in the MIDDLE of a sequence (position-id known answer)
        am = ids.ne(1).long()
        out = m(ids, am, return_dict=True)[0]
        gout = torch.randn(out.shape, generator=g)
        params = [p for n, p in m.named_parameters()]
        gs = torch.autograd.grad((out * gout).sum(), params, allow_unused=True)
        pos = m.embeddings.create_position_ids_from_input_ids(ids, 1)
        arrays = dict(input_ids=ids, attention_mask=am, out=out, gout=gout, position_ids=pos, heads=4)
        for (n, p), gp in zip(m.named_parameters(), gs):
            arrays["w_" + n] = p
            arrays["g_" + n] = torch.zeros_like(p) if gp is None else gp
            arrays["hasgrad_" + n] = np.array(gp is not None)
        npz(f"g6_roberta_{attn_impl}", **arrays)


# ---------------------------------------------------------------- G8
def g_gaussian():
    # models/mm_model.py cannot import (Uni-Core absent); the Gaussian basis is restated from :211-224 with torch in
    # float64 as an independent high-precision evaluation of the same closed form (NOT reference output).
    x = torch.tensor([0.0, 0.5, 1.5, 3.25, 9.0], dtype=torch.float64)
    mean = torch.tensor([1.0, 0.0, 2.5], dtype=torch.float64)
    std = torch.tensor([0.5, 1.0, 2.0], dtype=torch.float64)
    a = (2 * 3.14159) ** 0.5
    val = torch.exp(-0.5 * (((x[:, None] - mean) / std) ** 2)) / (a * std)
    npz("g8_gaussian", x=x, mean=mean, std=std, val=val, a=a)


if __name__ == "__main__":
    g_infonce()
    g_contrastive()
    g_fds_and_pad()
    g_cross()
    g_roberta()
    g_gaussian()
