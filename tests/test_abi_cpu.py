"""CPU-side checks of the C-ABI boundary and the host logic (no GPU, no compute launches)."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

from mmdti_hip import _abi


def test_library_exports_every_header_symbol():
    protos = _abi.parse_header()
    assert len(protos) >= 40
    dll = ctypes.CDLL(_abi.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/mmdti_hip.h but not exported"
    # and nothing mmdti_* is exported that the header does not declare (the header is the whole boundary)
    out = subprocess.run(["nm", "-D", "--defined-only", _abi.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("mmdti_")}
    assert exported == set(protos), exported ^ set(protos)


def test_header_has_no_torch_types_and_cites_reference():
    src = open(_abi.HEADER).read()
    assert "at::" not in src and "torch::" not in src and "#include <torch" not in src and "Tensor" not in src
    for cite in ("models/transformers.py", "models/infonce.py", "models/contrastive.py", "models/fds.py", "mm_model.py"):
        assert cite in src


def test_argument_validation_happens_before_any_launch():
    lib = _abi.lib()
    assert lib._dll.mmdti_abi_version() == 1
    with pytest.raises(_abi.MMDTIError, match="multiples of 8"):
        lib.mmdti_gemm_bf16(0, 16, 16, 16, 8, 8, 12, 12, 12, 8, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0)
    with pytest.raises(_abi.MMDTIError, match="exceeds"):
        lib.mmdti_pair_attn_fwd(0, 16, 16, 16, 16, 0, 1, 400, 8, 400, 0.35, 0.0, 0, 0, 0, 0, 0, 0, 0)
    with pytest.raises(_abi.MMDTIError, match="temperature"):
        lib.mmdti_infonce_dir(0, 16, 16, 4, 50, 0, 4, 0.0, 16, 16, 16, 16)


def test_product_path_has_no_cpu_fallback():
    from mmdti_hip import ops
    with pytest.raises(_abi.MMDTIError, match="no CPU fallback"):
        ops.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 1e-5)
    # nothing under the product package imports the oracle
    pkg = os.path.join(os.path.dirname(_abi.__file__))
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(root, f)).read().replace("the oracle", "").replace("oracle.mmdti_oracle`` with", ""), f


def test_lr_schedule_matches_hf():
    from transformers import get_linear_schedule_with_warmup
    from mmdti_hip.trainer import linear_warmup_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-4)
    sched = get_linear_schedule_with_warmup(opt, num_warmup_steps=3, num_training_steps=40)
    for step in range(40):
        assert opt.param_groups[0]["lr"] == pytest.approx(linear_warmup_lr(1e-4, step, 3, 40), rel=1e-12, abs=1e-18)
        opt.step(); sched.step()


def test_dictionary_and_model_state_dict_surface():
    from mmdti_hip.unicore_compat import Dictionary
    d = Dictionary.default_molecule()
    assert (d.pad(), d.bos(), d.eos(), d.unk()) == (0, 1, 2, 3) and len(d) == 30
    assert d.add_symbol("[MASK]", is_special=True) == 30 and d.index("Xx") == 3
    from mmdti_hip.models.mm_model import MM_Model
    from types import SimpleNamespace
    rcfg = SimpleNamespace(layers=1, dim=512, heads=8, ffn=64, vocab=40, max_pos=40, type_vocab=1, pad_idx=1, ln_eps=1e-12, hidden_dropout=0.1, attn_dropout=0.1)
    from mmdti_hip.models import mm_model as mm
    mol = mm.molecule_architecture(); mol.encoder_layers = 1
    m = MM_Model.from_configs(1, "regression", mol_args=mol, roberta_cfg=rcfg)
    keys = set(m.state_dict())
    assert "encoder.layers.0.self_attn.in_proj.weight" in keys and m.encoder.layers[0].self_attn.in_proj.weight.shape == (1536, 512)
    assert m.gbf.mul.weight.shape == (961, 1) and m.embed_tokens.weight.shape == (31, 512)
    with pytest.raises(ValueError):
        MM_Model.from_configs(3, "multiclass", mol_args=mol, roberta_cfg=rcfg)       # UnboundLocalError in the reference


def test_infonce_and_contrastive_argument_errors_on_cpu():
    from mmdti_hip.models.infonce import info_nce
    q = torch.randn(4, 50)
    with pytest.raises(ValueError):
        info_nce(q[0], q)
    with pytest.raises(ValueError):
        info_nce(q, q[:, :10])
    with pytest.raises(ValueError):
        info_nce(q, q, torch.randn(6, 50))
