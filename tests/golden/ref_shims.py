"""Build-container-only scaffolding that lets the reference's OWN files ``models/transformers.py``,
``models/mm_model.py`` and ``tasks/trainer.py`` import and run on CPU, so their outputs can be frozen as fixtures
(fixture G9).  Used by ``make_golden_g9.py`` only -- nothing on the product
path, no test and nothing on the GPU box imports this module.

What is REAL here: every line of the reference files named above (plus ``models/infonce.py``, ``models/contrastive.py``,
``models/fds.py``, ``models/mm_module.py``, ``utils/util.py``, ``utils/metrics.py``), the installed HuggingFace
``RobertaModel`` / ``AutoModel`` / ``AutoTokenizer`` and torch.

What is a STAND-IN (this build's code, not Uni-Core): the ``unicore`` package.  Uni-Core is an unpinned third-party
dependency absent from the reference tree and from this image, so ``TransformerEncoderLayer`` / ``LayerNorm`` below are
the oracle's restatement of its published pre-LN layer (``oracle.mmdti_oracle.unimol_layer``).  The fixtures therefore
pin the reference-owned WIRING (key-padding merge, bias chaining, x_norm before the final LN, tuple protocol, FDS
aliasing, collate, the trainer's step body) -- not Uni-Core's numerics, which stay "parity unpinned".

Other stubs are import plumbing only: ``utils`` (the reference's package ``__init__`` pulls in ``addict``, which is
absent; the stub re-exports the reference's own ``utils/util.py`` + ``utils/metrics.py`` symbols), ``config``
(re-exports the reference's ``config/model_config.py``), ``tasks.split`` (needs RDKit; ``Splitter`` is not used by
``fit_predict``), and neutralised hard-coded device moves (``.to('cuda')`` fds.py:84, ``.cuda()`` trainer.py:303-304).
"""
import importlib
import importlib.util
import logging
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import mmdti_oracle as O   # noqa: E402

REF = os.environ.get("MMDTI_REFERENCE", "/root/reference")


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------ unicore stand-in
class LayerNorm(nn.Module):
    """unicore.modules.LayerNorm stand-in: F.layer_norm semantics, eps 1e-5 (SURVEY 8c [UPSTREAM-RECALL])."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = tuple(normalized_shape)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(*self.normalized_shape))
        self.bias = nn.Parameter(torch.zeros(*self.normalized_shape))

    def forward(self, x):
        return O.layer_norm(x, self.weight, self.bias, self.eps)


class _SelfAttn(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout):
        super().__init__()
        self.in_proj = nn.Linear(embed_dim, 3 * embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.dropout = dropout


class TransformerEncoderLayer(nn.Module):
    """unicore.modules.TransformerEncoderLayer stand-in with Uni-Core's parameter names; the arithmetic is the oracle's
    ``unimol_layer`` (pre-LN, S returned after the bias add and before softmax)."""

    def __init__(self, embed_dim=768, ffn_embed_dim=3072, attention_heads=8, dropout=0.1, attention_dropout=0.1,
                 activation_dropout=0.0, activation_fn="gelu", post_ln=False):
        super().__init__()
        assert not post_ln and activation_fn == "gelu"
        self.embed_dim, self.attention_heads = embed_dim, attention_heads
        self.dropout, self.attention_dropout, self.activation_dropout = dropout, attention_dropout, activation_dropout
        self.self_attn = _SelfAttn(embed_dim, attention_heads, attention_dropout)
        self.self_attn_layer_norm = LayerNorm(embed_dim)
        self.fc1 = nn.Linear(embed_dim, ffn_embed_dim)
        self.fc2 = nn.Linear(ffn_embed_dim, embed_dim)
        self.final_layer_norm = LayerNorm(embed_dim)

    def forward(self, x, attn_bias=None, padding_mask=None, return_attn=False):
        assert return_attn and attn_bias is not None
        B, N, _ = x.shape
        H = self.attention_heads
        if padding_mask is not None:
            attn_bias = attn_bias.view(B, H, N, N).masked_fill(padding_mask.view(B, 1, 1, N).bool(), float("-inf")).view(B * H, N, N)
        cfg = O.UniMolCfg(heads=H, dropout=self.dropout, attn_dropout=self.attention_dropout, act_dropout=self.activation_dropout)
        P = {"l." + k: v for k, v in self.named_parameters()}
        return O.unimol_layer(x, attn_bias, P, "l.", cfg, training=self.training)


def init_bert_params(module):
    """unicore.modules.init_bert_params [UPSTREAM-RECALL]."""
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()


def get_activation_fn(name):
    return {"gelu": F.gelu, "tanh": torch.tanh, "relu": F.relu, "linear": lambda x: x}[name]


class Dictionary:
    """unicore.data.Dictionary [UPSTREAM-RECALL]: one symbol per line, index = line order."""

    def __init__(self):
        self.symbols, self.indices = [], {}

    def __len__(self):
        return len(self.symbols)

    def add_symbol(self, word, is_special=False):
        if word not in self.indices:
            self.indices[word] = len(self.symbols)
            self.symbols.append(word)
        return self.indices[word]

    def index(self, s):
        return self.indices.get(s, self.indices.get("[UNK]"))

    def pad(self):
        return self.index("[PAD]")

    def bos(self):
        return self.index("[CLS]")

    def eos(self):
        return self.index("[SEP]")

    def unk(self):
        return self.index("[UNK]")

    @classmethod
    def load(cls, path):
        d = cls()
        with open(path) as f:
            for line in f:
                w = line.rstrip().rsplit(" ", 1)[0]
                if w:
                    d.add_symbol(w)
        return d


MOL_SYMBOLS = ("[PAD] [CLS] [SEP] [UNK] C N O S H Cl F Br I Si P B Na K Al Ca Sn As Hg Fe Zn Cr Se Gd Au Li").split()


def install():
    """Register the stand-ins / stubs in sys.modules.  Returns the reference's utils/util.py module."""
    uc = types.ModuleType("unicore"); uc.__path__ = []
    m = types.ModuleType("unicore.modules")
    m.TransformerEncoderLayer, m.LayerNorm, m.init_bert_params = TransformerEncoderLayer, LayerNorm, init_bert_params
    u = types.ModuleType("unicore.utils"); u.get_activation_fn = get_activation_fn
    d = types.ModuleType("unicore.data"); d.Dictionary = Dictionary
    mo = types.ModuleType("unicore.models"); mo.BaseUnicoreModel = nn.Module
    for name, mod in (("unicore", uc), ("unicore.modules", m), ("unicore.utils", u), ("unicore.data", d), ("unicore.models", mo)):
        sys.modules[name] = mod
    # `utils`: the reference's own util.py / metrics.py behind a package stub (its __init__ needs addict)
    log = logging.getLogger("ref")
    utils = types.ModuleType("utils"); utils.__path__ = [os.path.join(REF, "utils")]
    sys.modules["utils"] = utils
    bl = types.ModuleType("utils.base_logger"); bl.logger = log     # the real one opens ./logs/*.log in the cwd
    sys.modules["utils.base_logger"] = bl
    util = importlib.import_module("utils.util")
    metrics = importlib.import_module("utils.metrics")
    utils.logger, utils.Metrics = log, metrics.Metrics
    for n in ("pad_1d_tokens", "pad_2d", "pad_coords", "calibrate_mean_var", "get_lds_kernel_window"):
        setattr(utils, n, getattr(util, n))
    cfg = types.ModuleType("config"); cfg.__path__ = [os.path.join(REF, "config")]
    sys.modules["config"] = cfg
    cfg.MODEL_CONFIG = importlib.import_module("config.model_config").MODEL_CONFIG
    # namespace packages so that models/__init__.py (-> nnmodel -> data -> rdkit) and tasks/__init__.py are not executed
    pkg = types.ModuleType("models"); pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg
    tasks = types.ModuleType("tasks"); tasks.__path__ = [os.path.join(REF, "tasks")]
    sys.modules["tasks"] = tasks
    split = types.ModuleType("tasks.split"); split.Splitter = object      # real file needs RDKit; unused by fit_predict
    sys.modules["tasks.split"] = split
    return util


class cpu_device_moves:
    """Neutralise the reference's hard-coded device moves (fds.py:84 ``.to('cuda')``, trainer.py:303-304 ``.cuda()``)."""

    def __enter__(self):
        self._to, self._cuda = torch.Tensor.to, torch.Tensor.cuda
        real_to = self._to

        def cpu_to(t, *a, **k):
            if a and isinstance(a[0], str) and a[0].startswith("cuda"):
                return t
            return real_to(t, *a, **k)

        torch.Tensor.to = cpu_to
        torch.Tensor.cuda = lambda t, *a, **k: t
        return self

    def __exit__(self, *exc):
        torch.Tensor.to, torch.Tensor.cuda = self._to, self._cuda


def make_tokenizer_json(chars="CNOSHFPIclnosBr()[]=#@+-123456789/\\"):
    """A local character-level SMILES tokenizer in HF `tokenizers` JSON form (ids: <s>=0 <pad>=1 </s>=2 <unk>=3)."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for c in chars:
        if c not in vocab:
            vocab[c] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split("", "isolated")
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    return tok.to_str(), len(vocab)


def fast_tokenizer(tok_json, max_len):
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast
    return PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_str(tok_json), bos_token="<s>", eos_token="</s>",
                                   pad_token="<pad>", unk_token="<unk>", model_max_length=max_len)
