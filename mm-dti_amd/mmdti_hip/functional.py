"""Module-level forward+backward of the fine-tune step, hand-scheduled over the C-ABI kernels.

Each ``torch.autograd.Function`` below is ONE node in the autograd graph for a whole reference module (15-layer
pair-bias encoder, RoBERTa tower, cross-modal block, InfoNCE head, ...).  Forward launches the kernels and keeps the
activations the backward needs; backward launches the gradient kernels in reverse order.  Weight gradients are
accumulated by the kernels straight into ``param.grad`` (views of the gradient arena), so autograd only carries the
activation gradients between modules.  torch is plumbing here (memory, streams, graph edges) -- no ATen math.

Precision contract (matches ``oracle.mmdti_oracle`` with ``bf16=True``): GEMM operands bf16, accumulation fp32, residual
stream / LayerNorm statistics / softmax / pair-bias chain S fp32, q|k|v and attention probabilities stored bf16.
"""
from __future__ import annotations

import os

import math
from types import SimpleNamespace
from typing import List, Optional

import torch

from . import ops
from .ops import BF16, F32
from .runtime import fused_views, wbf16, wfwd, gbuf, dropout_state, notify_grads_ready


# ------------------------------------------------------------------------------------------------- helpers
def _lin_bwd_params(dy_bf16, x_bf16, weight, bias, rows=None, bias_done=False):
    """dW += dy^T x ; db += colsum(dy)   (atomic accumulation into param.grad).  bias_done: the kernel that produced dy
    already accumulated its column sums (LayerNorm backward's bf16 copy).  Otherwise the bias gradient rides on the
    weight-gradient GEMM, which reads dy anyway (ops.linear_bwd_weight(db=...))."""
    gw = gbuf(weight)
    gb = gbuf(bias) if (bias is not None and not bias_done) else None
    if gw is not None:
        ops.linear_bwd_weight(dy_bf16, x_bf16, gw, rows=rows, db=gb if (gb is not None and gb.numel() == gw.shape[0]) else None)
        if gb is not None and gb.numel() == gw.shape[0]:
            gb = None
    if gb is not None:
        ops.colsum(dy_bf16, gb, cols=bias.numel())


def _lin_bwd_params_many(calls, raw_items=()):
    """calls: [(args, kwargs)] of _lin_bwd_params for Linears of ONE layer (same token rows): their weight gradients go out as a
    single grouped launch (ops.linear_bwd_weight_grouped); bias gradients ride on it, leftovers take the column-sum pass.
    raw_items: ready-made (dy, x, dw_buffer, db_buffer | None, rows | None) entries (fused query|key|value views)."""
    items, late = list(raw_items), []
    for args, kw in calls:
        dy, x, weight, bias = args[:4]
        rows, bias_done = kw.get("rows"), kw.get("bias_done", False)
        gw = gbuf(weight)
        gb = gbuf(bias) if (bias is not None and not bias_done) else None
        if gw is None:
            if gb is not None:
                late.append((dy, gb, bias.numel()))
            continue
        rides = gb is not None and gb.numel() == gw.shape[0]
        items.append((dy, x, gw, gb if rides else None, rows))
        if gb is not None and not rides:
            late.append((dy, gb, bias.numel()))
    if items:
        ops.linear_bwd_weight_grouped(items)
    for dy, gb, cols in late:
        ops.colsum(dy, gb, cols=cols)


# one C call per layer backward where the library has a sequencer for the variant at hand (csrc/layers.hip): the launch order,
# kernels and arguments of the op-by-op path below, minus ~140 us of Python per layer -- what paces a step of 16-32 molecules
LAYER_SEQ = os.environ.get("MMDTI_LAYER_SEQ", "1") != "0"
POOL_THEN_PROJECT = os.environ.get("MMDTI_INFONCE_POOL_FIRST", "1") != "0"      # InfoNCE head: pool the GELU outputs, then project


def _join_stream_after_backward():
    """A module backward that writes parameter gradients itself leaves autograd no leaf on its stream, so the engine would
    not join that stream at the end of backward(): if we are on a side stream, queue the join."""
    here = torch.cuda.current_stream()
    if here == torch.cuda.default_stream():
        return

    def _join(stream=here):
        cur = torch.cuda.current_stream()
        if cur != stream:
            cur.wait_stream(stream)

    torch.autograd.Variable._execution_engine.queue_callback(_join)


# Weight gradients held back to the end of the pair encoder's backward: the fused pair-bias backward that follows it runs
# alone on the chip and is latency-bound (waves parked 80 % of their cycles), so the weight-gradient GEMMs of the last few
# layers -- leaves of the backward graph -- are launched on their own stream right before it and run underneath it.
DEFER_WGRAD_LAYERS = int(os.environ.get("MMDTI_DEFER_WGRAD", "4"))
_wgrad_stream_obj = None
_wgrad_keep = []


def _launch_deferred_wgrads(deferred, layers):
    """deferred: argument tuples of _lin_bwd_params whose operands must stay alive until the stream is joined."""
    if not deferred:
        return
    main = torch.cuda.current_stream()
    ws = _wgrad_stream()
    ws.wait_stream(main)
    with torch.cuda.stream(ws):
        for group in deferred:                      # one list of _lin_bwd_params calls per layer
            _lin_bwd_params_many(group)
        for layer in layers:
            notify_grads_ready(layer.parameters())     # (recorded on this stream: the reducer's event sits behind the GEMMs)
    _wgrad_keep.append(deferred)

    def _join(stream=ws):
        torch.cuda.current_stream().wait_stream(stream)
        _wgrad_keep.clear()                             # operands may be recycled now: later work is ordered behind the join

    torch.autograd.Variable._execution_engine.queue_callback(_join)


def _wgrad_stream():
    global _wgrad_stream_obj
    if _wgrad_stream_obj is None:
        _wgrad_stream_obj = torch.cuda.Stream()
    return _wgrad_stream_obj


class _Sites:
    """Dropout site numbering inside one forward call."""

    def __init__(self):
        self.n = 0

    def next(self):
        self.n += 1
        return self.n


# ------------------------------------------------------------------------------------------------- Uni-Mol pair encoder
class PairEncoderFn(torch.autograd.Function):
    """TransformerEncoderWithPair.forward (models/transformers.py:96-183) minus the discarded aux outputs:
    emb-LN -> dropout -> zero padded rows -> L x [pre-LN pair-bias attention + FFN, S chained] -> final LN.

    inputs : emb [B,N,D] fp32, bias [B,H,N,ld] fp32 (ld >= N; standard layout is ld == N) or a tiled pair tensor (fp32, or the
             compact fp16 planes PairBiasFn produces on the hot path), padding_mask [B,N] bool|None
    outputs: x [B,N,D] fp32, S_last in the layout and dtype of bias (pre-softmax logits of the last layer, -inf at padded keys),
             x_pre [B,N,D] fp32 (stream before the final LN; only the discarded x_norm aux output reads it)
    pack (packing.PackedRows, compact tiled bias + key_tiles only): PACKED token rows -- emb / padding_mask / x / x_pre are
             [pack.M, D] / [pack.M]: every molecule's real tokens plus one representative pad row (all pad rows of a molecule
             are the same row at dropout 0); S_last then lacks the query rows past the representative one.
    """

    @staticmethod
    def forward(ctx, emb, bias, padding_mask, mod, training, key_tiles=None, pack=None, aux_grads=False):
        """aux_grads: S_last and x_pre are differentiable outputs (the reference's auxiliary returns -- attn, delta_pair_repr, x_norm,
        delta_pair_repr_norm, models/transformers.py:141-181 -- are functions of them); fp32 pair planes only.  MM_Model discards those
        returns (mm_model.py:559), so the hot path leaves it off and autograd never materialises a gradient for them."""
        if pack is not None:
            B, N, D = pack.B, pack.S, emb.shape[-1]
            if emb.dim() != 2 or emb.shape[0] != pack.M:
                raise ops.MMDTIError("PairEncoderFn: packed rows expect emb [pack.M, D]")
        else:
            B, N, D = emb.shape
        H = mod.attention_heads
        tiled = ops.pair_is_tiled(bias)        # [B,H,nt,nt,256] tile layout (see ops.pair_tile) or row-major [B,H,N,ld]
        compact = tiled and bias.dtype == torch.float16     # logits chain as fp16 (ops.PAIR_COMPACT, PairBiasFn)
        if not compact:
            key_tiles = None                   # (only the compact tiled kernels have a ragged form)
        if pack is not None and key_tiles is None:
            raise ops.MMDTIError("PairEncoderFn: packed rows need the compact tiled pair layout and key_tiles")
        if aux_grads and (compact or pack is not None):
            raise ops.MMDTIError("PairEncoderFn: differentiable auxiliary outputs need fp32 pair planes on padded rows")
        row_off = None if pack is None else pack.off
        # The gradient of a compact (fp16) bias is NOT an fp16 tensor (fp32, or bf16 on request), and autograd casts whatever a
        # backward returns to the dtype of the input.  PairBiasFn therefore hangs a slot on the bias it produces; the backward
        # below leaves the real gradient chain there and hands autograd a zero-storage placeholder.
        slot = getattr(bias, "_mmdti_grad_slot", None) if compact else None
        if compact and slot is None and bias.requires_grad:
            raise ops.MMDTIError("PairEncoderFn: an fp16 pair bias that needs a gradient must come from PairBiasFn")
        ld = ops.pair_ld(N) if tiled else bias.shape[-1]
        M = B * N if pack is None else pack.M
        p_emb = mod.emb_dropout if training else 0.0
        p_res = mod.dropout if training else 0.0
        p_att = mod.attention_dropout if training else 0.0
        seed = dropout_state.next_seed()
        sites = _Sites()
        st = SimpleNamespace(B=B, N=N, D=D, H=H, ld=ld, M=M, seed=seed, p_emb=p_emb, p_res=p_res, p_att=p_att, layers=[], kt=key_tiles, slot=slot,
                             row_off=row_off, packed=pack is not None)
        keep = any(ctx.needs_input_grad)      # inference (torch.no_grad / frozen inputs): nothing is kept for a backward --
                                              # the 15 per-layer logit tensors are freed as the stack advances
        emb = emb.contiguous()
        st.emb = emb
        st.pad = padding_mask
        st.site_emb = sites.next()
        eln = mod.emb_layer_norm
        x, _, st.emb_mean, st.emb_rstd = ops.layernorm_fwd(emb.view(M, D), eln.weight, eln.bias, eln.eps, want_f32=True, want_bf16=False,
                                                           row_zero=None if padding_mask is None else padding_mask.reshape(-1),
                                                           drop_p=p_emb, seed=seed, site=st.site_emb)
        scale = (D // H) ** -0.5
        s_prev = bias.contiguous()
        st.bias0 = s_prev if not mod.layers else None     # only needed to shape a zero gradient when there is no layer
        # Every Linear that closes a residual branch runs fused with the LayerNorm that reads its output (ops.linear_ln_fwd): out_proj
        # with this layer's final_layer_norm, fc2 with the NEXT layer's self_attn_layer_norm -- or, after the last layer, with the
        # encoder's final_layer_norm.  `nxt` carries that next LayerNorm's output into the next iteration.
        nlayers = len(mod.layers)
        nxt = None
        out = None
        # (fp16 forward operands: the library's sequence covers the compact pair planes -- the only layout with fp16 q | k | v kernels)
        seq = LAYER_SEQ and emb.is_cuda and (compact or not ops.FWD_F16) and not ops.kernel_timer.names and D == H * 8
        kp0 = ops._u8(padding_mask) if seq else None
        # small batches: ALL layers from one library call (see _unimol_stack_fwd); the per-layer loop below then has nothing left to do
        st.stack = None
        T = (_unimol_stack_tables(mod, M) if (seq and keep and STACK_SEQ and nlayers and M < STACK_MAX_ROWS and not aux_grads
                                                and mod.final_layer_norm is not None) else None)
        if T is not None:
            x, out, s_prev = _unimol_stack_fwd(st, T, mod, x, s_prev, kp0, key_tiles, pack is None, row_off, scale, sites, tiled)
        for li, layer in enumerate(mod.layers if T is None else ()):
            L = SimpleNamespace(x=x)
            ln1, ln2 = layer.self_attn_layer_norm, layer.final_layer_norm
            att = layer.self_attn
            if nxt is None:
                _, L.h1, L.m1, L.r1 = ops.layernorm_fwd(x, ln1.weight, ln1.bias, ln1.eps)
            else:
                L.h1, L.m1, L.r1 = nxt
            if seq:
                # the layer's six launches from ONE library call (csrc/layers.hip: the same kernels, arguments and order as below)
                L.site_att, L.site_o, L.site_f = sites.next(), sites.next(), sites.next()
                last = li == nlayers - 1
                nl = mod.layers[li + 1].self_attn_layer_norm if not last else mod.final_layer_norm
                x, ln_out, mn, rn = _unimol_layer_fwd_seq(st, layer, L, s_prev, kp0 if li == 0 else None, key_tiles, last and pack is None, row_off,
                                                          scale, nl, 0 if nl is None else (2 if last else 1))
                s_prev = L.s
                if not last:
                    nxt = (ln_out, mn, rn)
                elif nl is not None:
                    out, st.f_mean, st.f_rstd = ln_out, mn, rn
                if keep:
                    st.layers.append(L)
                continue
            L.qkv = ops.linear_fwd(L.h1, wfwd(att.in_proj.weight), att.in_proj.bias)
            L.site_att = sites.next()
            # (ragged batches: all-padding key tiles are skipped; the last layer writes them as -inf because its S is returned)
            L.s, L.o = ops.pair_attn_fwd(L.qkv, s_prev, padding_mask if li == 0 else None, B, N, H, ld, scale, p_att, seed, L.site_att,
                                         key_tiles=key_tiles, rag_store=(li == nlayers - 1) and pack is None, row_off=row_off)
            s_prev = L.s
            L.site_o = sites.next()
            L.x1, _, L.h2, L.m2, L.r2 = ops.linear_ln_fwd(L.o, wfwd(att.out_proj.weight), att.out_proj.bias, ln2.weight, ln2.bias, ln2.eps,
                                                          residual=x, drop_p=p_res, seed=seed, site=L.site_o)
            L.u = torch.empty(M, layer.fc1.weight.shape[0], device=emb.device, dtype=BF16)
            L.a = ops.linear_fwd(L.h2, wfwd(layer.fc1.weight), layer.fc1.bias, act=ops.ACT_GELU_FWD, aux_out=L.u)
            L.site_f = sites.next()
            if li + 1 < nlayers:
                nl = mod.layers[li + 1].self_attn_layer_norm
                x, _, h1n, m1n, r1n = ops.linear_ln_fwd(L.a, wfwd(layer.fc2.weight), layer.fc2.bias, nl.weight, nl.bias, nl.eps, residual=L.x1,
                                                        drop_p=p_res, seed=seed, site=L.site_f)
                nxt = (h1n, m1n, r1n)
            elif mod.final_layer_norm is not None:
                fl = mod.final_layer_norm
                x, out, _, st.f_mean, st.f_rstd = ops.linear_ln_fwd(L.a, wfwd(layer.fc2.weight), layer.fc2.bias, fl.weight, fl.bias, fl.eps,
                                                                    residual=L.x1, drop_p=p_res, seed=seed, site=L.site_f, want_f32=True, want_bf16=False)
            else:
                x = ops.linear_fwd(L.a, wfwd(layer.fc2.weight), layer.fc2.bias, residual=L.x1, out_dtype=F32, drop_p=p_res, seed=seed, site=L.site_f)
            if keep:
                st.layers.append(L)
        st.x_last = x
        if out is None:
            if mod.final_layer_norm is not None:
                fl = mod.final_layer_norm
                out, _, st.f_mean, st.f_rstd = ops.layernorm_fwd(x, fl.weight, fl.bias, fl.eps, want_f32=True, want_bf16=False)
            else:
                out = x
        if keep:
            ctx.st, ctx.mod = st, mod
        x_last = x.view(B, N, D) if pack is None else x
        if not aux_grads:
            ctx.mark_non_differentiable(s_prev, x_last)
        # (otherwise autograd hands the backward a freshly ZEROED tensor for each output nobody differentiates -- for S_last that is
        #  a 0.55 GB fill per step, found with scratch/fill_diag.py)
        ctx.set_materialize_grads(False)
        return (out.view(B, N, D) if pack is None else out), s_prev, x_last

    @staticmethod
    def backward(ctx, dout, ds_last, dx_pre):
        # ds_last / dx_pre: gradients of the auxiliary outputs (aux_grads; None otherwise -- gradients are not materialised)
        st, mod = ctx.st, ctx.mod
        if getattr(st, "consumed", False):     # (the saved activations are released layer by layer as the backward advances)
            raise ops.MMDTIError("PairEncoderFn: backward through the same forward twice (retain_graph) is not supported")
        st.consumed = True
        B, N, D, H, ld, M, seed = st.B, st.N, st.D, st.H, st.ld, st.M, st.seed
        scale = (D // H) ** -0.5
        if dout is None:                       # (nothing downstream used the encoder output)
            dout = torch.zeros(M, D, device=st.emb.device, dtype=F32)
        dout = dout.contiguous().view(M, D)
        if dx_pre is not None:
            dx_pre = dx_pre.contiguous().view(M, D).float()
        if mod.final_layer_norm is not None:
            fl = mod.final_layer_norm
            nxt = (st.p_res, st.layers[-1].site_f, gbuf(mod.layers[-1].fc2.bias)) if st.layers else None
            if st.stack is not None:
                nxt = (st.p_res, st.stack.site0 + 3 * (st.stack.T.nl - 1) + 2, gbuf(mod.layers[-1].fc2.bias))
            dx = ops.layernorm_bwd(dout, st.x_last, fl.weight, st.f_mean, st.f_rstd, gbuf(fl.weight), gbuf(fl.bias), dres=dx_pre, bf16_copy=nxt)
            dx, dx16 = dx if nxt is not None else (dx, None)
        else:
            dx, dx16 = (dout if dx_pre is None else dout + dx_pre), None
        G = None
        if ds_last is not None:
            # the chain starts from the gradient of the returned logits; padded keys carry none (the reference fills them in place,
            # transformers.py:122-135: no gradient passes a filled entry)
            G = ds_last.to(F32).contiguous().clone()
            if st.pad is not None and not ops.pair_is_tiled(G):
                G.masked_fill_(st.pad.view(B, 1, 1, N), 0.0)
            if G.shape[-1] > N and not ops.pair_is_tiled(G):
                G[..., N:] = 0.0
        if st.stack is not None:
            dx, G = _unimol_stack_bwd(st, dx, dx16, scale)
            notify_grads_ready(st.stack.T.params)
            st.stack = None
        below = [Lb.site_f for Lb in st.layers[:-1]]      # site of the FFN dropout of the layer UNDER each layer
        deferred, deferred_layers = [], []
        # (holding weight gradients back for the end pays where the pair-bias backward has work to hide -- large batches; at a few
        #  thousand rows every launch is latency-bound and the held layers would only miss the sequenced path below)
        n_defer = DEFER_WGRAD_LAYERS if (dout.is_cuda and M >= 8192) else 0
        seq_ok, seq_ws = _unimol_seq_workspace(st, mod) if (LAYER_SEQ and dout.is_cuda and st.layers) else (False, None)
        for li, layer, L in zip(range(len(st.layers) - 1, -1, -1), reversed(mod.layers), reversed(st.layers)):
            att, ln1, ln2 = layer.self_attn, layer.self_attn_layer_norm, layer.final_layer_norm
            hold = li < n_defer                          # this layer's weight gradients wait for the end (see _launch_deferred_wgrads)
            pending = []                                 # the layer's four weight gradients leave as ONE grouped launch

            def _wgrad(*args, **kw):
                pending.append((args, kw))
            if seq_ok and dx16 is not None and not hold:
                g_zero = G is None
                if g_zero:
                    G = (torch.empty if st.kt is None else torch.zeros)(L.s.shape, device=L.s.device, dtype=ops.pair_grad_dtype(L.s))
                dx, dx16 = _unimol_layer_bwd_seq(st, layer, L, dx, dx16, G, g_zero, scale, seq_ws,
                                                 None if li == 0 else (below[li - 1], gbuf(mod.layers[li - 1].fc2.bias)))
                L.__dict__.clear()
                notify_grads_ready(layer.parameters())
                continue
            # ---- FFN:  x2 = x1 + drop(fc2(gelu(fc1(LN2(x1)))))
            # (dx16 = bf16 dropout-backward copy of dx, written by the LayerNorm backward that produced dx)
            dy2 = dx16 if dx16 is not None else ops.cast_bf16(dx, st.p_res, seed, L.site_f)
            _wgrad(dy2, L.a, layer.fc2.weight, layer.fc2.bias, bias_done=dx16 is not None)
            du = ops.linear_bwd_input(dy2, wbf16(layer.fc2.weight), act=ops.ACT_GELU_DX, aux_in=L.u)
            _wgrad(du, L.h2, layer.fc1.weight, layer.fc1.bias)      # (the bias gradient rides on the weight-gradient GEMM)
            dh2 = ops.linear_bwd_input(du, wbf16(layer.fc1.weight))
            dx, dy1 = ops.layernorm_bwd(dh2, L.x1, ln2.weight, L.m2, L.r2, gbuf(ln2.weight), gbuf(ln2.bias), dres=dx,
                                        bf16_copy=(st.p_res, L.site_o, gbuf(att.out_proj.bias)))
            # ---- attention:  x1 = x + drop(out_proj(attn(LN1(x))))
            _wgrad(dy1, L.o, att.out_proj.weight, att.out_proj.bias, bias_done=True)
            do = ops.linear_bwd_input(dy1, wbf16(att.out_proj.weight))
            g_zero = G is None
            if g_zero:
                # (fp32, or bf16 under MMDTI_PAIR_G_BF16; skipped key tiles of G are never written)
                G = (torch.empty if st.kt is None else torch.zeros)(L.s.shape, device=L.s.device, dtype=ops.pair_grad_dtype(L.s))
            dqkv = ops.pair_attn_bwd(L.qkv, L.s, do, G, B, N, H, ld, scale, g_zero, st.p_att, seed, L.site_att, key_tiles=st.kt, row_off=st.row_off)
            _wgrad(dqkv, L.h1, att.in_proj.weight, att.in_proj.bias)
            dh1 = ops.linear_bwd_input(dqkv, wbf16(att.in_proj.weight))
            if li > 0:
                dx, dx16 = ops.layernorm_bwd(dh1, L.x, ln1.weight, L.m1, L.r1, gbuf(ln1.weight), gbuf(ln1.bias), dres=dx,
                                             bf16_copy=(st.p_res, below[li - 1], gbuf(mod.layers[li - 1].fc2.bias)))
            else:
                dx, dx16 = ops.layernorm_bwd(dh1, L.x, ln1.weight, L.m1, L.r1, gbuf(ln1.weight), gbuf(ln1.bias), dres=dx), None
            if hold:
                deferred.append(pending)
                deferred_layers.append(layer)
            else:
                _lin_bwd_params_many(pending)
            L.__dict__.clear()       # release this layer's activations (S_l is ~1 GB at the bench shape)
            if not hold:
                notify_grads_ready(layer.parameters())
        eln = mod.emb_layer_norm
        demb = ops.layernorm_bwd(dx, st.emb.view(M, D), eln.weight, st.emb_mean, st.emb_rstd, gbuf(eln.weight), gbuf(eln.bias),
                                 row_zero=None if st.pad is None else st.pad.reshape(-1), drop_p=st.p_emb, seed=seed, site=st.site_emb)
        notify_grads_ready(list(eln.parameters()) + ([] if mod.final_layer_norm is None else list(mod.final_layer_norm.parameters())))
        if G is None:
            G = torch.zeros_like(st.bias0)
        if st.slot is not None:
            st.slot.g = G                                                            # -> PairBiasFn.backward (see forward)
            G = torch.zeros((), device=G.device, dtype=torch.float16).expand(G.shape)
        _launch_deferred_wgrads(deferred, deferred_layers)
        _join_stream_after_backward()
        return (demb if st.packed else demb.view(B, N, D)), G, None, None, None, None, None, None


# ---- all layers of the tower behind one library call per direction (csrc/layers.hip: mmdti_unimol_stack_fwd / _bwd).  At the
# reference's batch size (16-32 molecules) the step is paced by the host: the per-layer calls below still cost ~90 us of Python per
# layer and direction (17 allocations, 60 marshalled arguments, shadow look-ups).  The stack calls take the parameters as pointer
# tables cached per model and keep the saved tensors at fixed offsets of one arena: one allocation and one call for 15 layers.
# Above MMDTI_STACK_MAX_ROWS rows the GPU sets the pace and the per-layer path (which frees each layer's activations as the
# backward advances and holds the last layers' weight gradients back) stays in charge.
STACK_SEQ = os.environ.get("MMDTI_STACK_SEQ", "1") != "0"
STACK_MAX_ROWS = int(os.environ.get("MMDTI_STACK_MAX_ROWS", "8192"))
# (weight gradients of the stack backward on a side stream, under the layer below: measured at 32 molecules -- 6.92 vs 6.95 ms, no
#  gain, the step is bound by the chip's throughput on small kernels, not by the length of one stream's chain -- so it stays opt-in)
STACK_SIDE_WGRAD = os.environ.get("MMDTI_STACK_SIDE_WGRAD", "0") != "0"
_stack_layouts = {}
_stack_sides = {}


def _ptr_table(ptrs):
    import ctypes
    arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
    return arr, ctypes.addressof(arr)


def _unimol_stack_tables(mod, M):
    """Pointer tables of the encoder's layers for the stack calls (cached on the parameter arena), or None when the stack calls do not
    cover this model: parameters outside one arena or frozen, layers of different shapes, dimensions the grouped weight-gradient kernels
    do not take."""
    layers = list(mod.layers)
    l0 = layers[0]
    arena = getattr(l0.fc1.weight, "_mmdti_arena", None)
    if arena is None or not ops.GROUPED_DW or M < ops.GROUPED_DW_MIN_ROWS:
        return None
    cache = arena.__dict__.setdefault("_stack_tables", {})
    key = (id(mod), ops.FWD_F16)
    T = cache.get(key)
    if T is None:
        D, F = l0.fc1.weight.shape[1], l0.fc1.weight.shape[0]
        eps = l0.final_layer_norm.eps

        def plist(l):
            a = l.self_attn
            return [a.in_proj.weight, a.in_proj.bias, a.out_proj.weight, a.out_proj.bias, l.final_layer_norm.weight, l.final_layer_norm.bias,
                    l.fc1.weight, l.fc1.bias, l.fc2.weight, l.fc2.bias, l.self_attn_layer_norm.weight, l.self_attn_layer_norm.bias]
        ok = D % 256 == 0 and F % 256 == 0
        for l in layers:
            ps = plist(l)
            ok = ok and all(q is not None and q.requires_grad and getattr(q, "_mmdti_arena", None) is arena for q in ps)
            ok = ok and tuple(l.fc1.weight.shape) == (F, D) and l.final_layer_norm.eps == eps and l.self_attn_layer_norm.eps == eps
            ok = ok and len(list(l.parameters())) == 12
        if not ok:
            cache[key] = False
            return None
        if ops.FWD_F16:
            arena._fresh16()
        sh16 = (arena.shadow16 if ops.FWD_F16 else arena.shadow).data_ptr()
        shb, gr, off = arena.shadow.data_ptr(), arena.grad.data_ptr(), arena.offsets
        fwd, bwd, grads, weights, params = [], [], [], [], []
        for l in layers:
            ps = plist(l)
            params += ps
            w_in, b_in, w_out, b_out, g2, bt2, w1, b1, w2, b2, g1, bt1 = ps
            for i, q in enumerate(ps):
                fwd.append(sh16 + 2 * off[id(q)] if i in (0, 2, 6, 8) else q.data_ptr())
            bwd += [shb + 2 * off[id(w2)], shb + 2 * off[id(w1)], shb + 2 * off[id(w_out)], shb + 2 * off[id(w_in)], g2.data_ptr(), g1.data_ptr()]
            grads += [gr + 4 * off[id(q)] for q in (w2, w1, w_out, w_in, b2, b1, b_out, b_in, g2, bt2, g1, bt1)]
            weights += [(q, id(q)) for q in (w_in, w_out, w1, w2)]
        T = cache[key] = SimpleNamespace(nl=len(layers), D=D, F=F, eps=eps, fwd=_ptr_table(fwd), bwd=_ptr_table(bwd), grads=_ptr_table(grads),
                                         weights=weights, params=params, arena=arena, f16=ops.FWD_F16, n_shadow16=arena.shadow16 is not None,
                                         probes=[(l.fc1.weight, l.fc1.weight.data_ptr(), gr + 4 * off[id(l.fc1.weight)]) for l in layers])
    if T is False:
        return None
    ver = arena._version
    for q, i in T.weights:                       # an in-place write since the last cast (load_state_dict on a bound model): re-cast
        if q._version != ver[i]:
            arena._fresh(q)
    if T.f16:
        arena._fresh16()
    for q, dptr, gptr in T.probes:               # storage or gradient views re-bound behind our back: the per-layer path looks them up
        g = q.grad
        if g is None or g.data_ptr() != gptr or q.data_ptr() != dptr:
            return None
    return T


def _unimol_stack_layout(M, D, F, s_bytes):
    key = (M, D, F, s_bytes)
    r = _stack_layouts.get(key)
    if r is None:
        import ctypes
        tiles = (D // 256) * (F // 256) * 2 + (D // 256) ** 2 * 4
        slab = ops.lib()._dll.mmdti_linear_dw_grouped_splits(tiles, M) * (2 * D * F + 4 * D * D) * 4
        out = (ctypes.c_longlong * 2)()
        ops.lib().mmdti_unimol_stack_layout(M, D, F, s_bytes, slab, ctypes.addressof(out))
        if len(_stack_layouts) > 4096:
            _stack_layouts.clear()
        r = _stack_layouts[key] = (int(out[0]), int(out[1]), slab)
    return r


def _stack_side():
    """(stream handle, address of the three events) for the side-stream weight gradients of a stack backward, or (0, 0)."""
    if not STACK_SIDE_WGRAD or torch.cuda.is_current_stream_capturing():
        return 0, 0
    main = torch.cuda.current_stream()
    key = (main.device.index, main.cuda_stream)
    ent = _stack_sides.get(key)
    if ent is None:
        side = torch.cuda.Stream(device=main.device)
        evs = [torch.cuda.Event() for _ in range(3)]
        for e in evs:
            e.record(main)                       # (creates the hipEvent_t behind the handle)
        ent = _stack_sides[key] = (side, evs, _ptr_table([e.cuda_event for e in evs]))
    return ent[0].cuda_stream, ent[2][1]


def _unimol_stack_fwd(st, T, mod, x, s_prev, key_pad, key_tiles, rag_store_last, row_off, scale, sites, tiled):
    """All layers' forward as ONE library call -> (x_last fp32, final LayerNorm output fp32, S of the last layer)."""
    M, D, F, B, N, H, ld = st.M, st.D, T.F, st.B, st.N, st.H, st.ld
    dev = x.device
    l0 = mod.layers[0]
    ln1, fl = l0.self_attn_layer_norm, mod.final_layer_norm
    _, h1, m1, r1 = ops.layernorm_fwd(x, ln1.weight, ln1.bias, ln1.eps)
    e = torch.empty
    s_last = torch.empty_like(s_prev) if tiled else e(B, H, N, ld, device=dev, dtype=F32)
    s_bytes = s_last.numel() * s_last.element_size()
    stride, ws_bytes, slab = _unimol_stack_layout(M, D, F, s_bytes)
    arena = e(stride * T.nl, device=dev, dtype=torch.uint8)
    x_last, out = e(M, D, device=dev, dtype=F32), e(M, D, device=dev, dtype=F32)
    st.f_mean, st.f_rstd = e(M, device=dev, dtype=F32), e(M, device=dev, dtype=F32)
    site0 = sites.n + 1
    sites.n += 3 * T.nl
    p = ops._p
    ops.lib().mmdti_unimol_stack_fwd(
        ops._stream(), T.nl, M, B, N, H, D, F, ld, float(scale), float(st.p_res), float(st.p_att), int(st.seed), site0, x.data_ptr(), h1.data_ptr(),
        s_prev.data_ptr(), p(key_pad), ops._pair_layout_s(s_prev, "pair_attn.bias"), p(key_tiles), int(rag_store_last), p(row_off), T.fwd[1],
        ops.ACT_GELU_FWD, float(T.eps), fl.weight.data_ptr(), fl.bias.data_ptr(), float(fl.eps), ops.GEMM_LN_MAX_K if ops.GEMM_LN else 0, arena.data_ptr(),
        arena.numel(), s_bytes, x_last.data_ptr(), s_last.data_ptr(), out.data_ptr(), st.f_mean.data_ptr(), st.f_rstd.data_ptr(),
        int(h1.dtype == torch.float16))
    st.stack = SimpleNamespace(T=T, arena=arena, s_bytes=s_bytes, ws_bytes=ws_bytes, slab=slab, site0=site0, x0=x, h1=h1, m1=m1, r1=r1, s_last=s_last)
    return x_last, out, s_last


def _unimol_stack_bwd(st, dx, dx16, scale):
    """All layers' backward as ONE library call -> (gradient of the stream entering layer 0, pair-gradient chain G)."""
    S = st.stack
    T = S.T
    M, D = st.M, st.D
    dev = dx.device
    G = (torch.empty if st.kt is None else torch.zeros)(S.s_last.shape, device=dev, dtype=ops.pair_grad_dtype(S.s_last))
    layout = ops._pair_layout_s(S.s_last, "pair_attn_bwd.s") | (4 if G.dtype == BF16 else 0)
    dx_final = torch.empty_like(dx)
    ws = torch.empty(S.ws_bytes, device=dev, dtype=torch.uint8)
    side, events = _stack_side()
    p = ops._p
    ops.lib().mmdti_unimol_stack_bwd(
        ops._stream(), T.nl, M, st.B, st.N, st.H, D, T.F, st.ld, float(scale), float(st.p_res), float(st.p_att), int(st.seed), S.site0, dx.data_ptr(),
        dx16.data_ptr(), dx_final.data_ptr(), S.x0.data_ptr(), S.h1.data_ptr(), S.m1.data_ptr(), S.r1.data_ptr(), S.s_last.data_ptr(), T.bwd[1],
        ops.ACT_GELU_DX, T.grads[1], G.data_ptr(), layout, 1, p(st.kt), p(st.row_off), S.arena.data_ptr(), S.arena.numel(), S.s_bytes, ws.data_ptr(),
        ws.numel(), S.slab, int(S.h1.dtype == torch.float16), side, events)
    return dx_final, G


def _unimol_layer_fwd_seq(st, layer, L, s_prev, key_pad, key_tiles, rag_store, row_off, scale, nl, next_mode):
    """One Uni-Mol layer's forward as ONE library call (csrc/layers.hip); fills L with the tensors the backward reads and returns
    (x_out fp32, LayerNorm output of x_out | None, its mean, its rstd)."""
    att, ln2 = layer.self_attn, layer.final_layer_norm
    M, D, F = st.M, st.D, layer.fc1.weight.shape[0]
    dev = L.x.device
    e = torch.empty
    a16 = L.h1.dtype                          # bf16, or fp16 (fp16 forward operands): the type of every forward GEMM input of the layer
    L.qkv, L.o = e(M, 3 * D, device=dev, dtype=a16), e(M, D, device=dev, dtype=a16)
    L.s = torch.empty_like(s_prev) if ops.pair_is_tiled(s_prev) else e(st.B, st.H, st.N, st.ld, device=dev, dtype=F32)
    L.x1, L.h2 = e(M, D, device=dev, dtype=F32), e(M, D, device=dev, dtype=a16)
    L.m2, L.r2 = e(M, device=dev, dtype=F32), e(M, device=dev, dtype=F32)
    L.u, L.a = e(M, F, device=dev, dtype=BF16), e(M, F, device=dev, dtype=a16)
    x_out = e(M, D, device=dev, dtype=F32)
    ln_out = mn = rn = None
    if next_mode:
        ln_out = e(M, D, device=dev, dtype=a16 if next_mode == 1 else F32)
        mn, rn = e(M, device=dev, dtype=F32), e(M, device=dev, dtype=F32)
    p = ops._p
    ops.lib().mmdti_unimol_layer_fwd(
        ops._stream(), M, st.B, st.N, st.H, D, F, st.ld, float(scale), float(st.p_res), float(st.p_att), int(st.seed), int(L.site_att), int(L.site_o),
        int(L.site_f), L.x.data_ptr(), L.h1.data_ptr(), s_prev.data_ptr(), p(key_pad), ops._pair_layout_s(s_prev, "pair_attn.bias"), p(key_tiles),
        int(rag_store), p(row_off), wfwd(att.in_proj.weight).data_ptr(), p(att.in_proj.bias), wfwd(att.out_proj.weight).data_ptr(), p(att.out_proj.bias),
        ln2.weight.data_ptr(), ln2.bias.data_ptr(), float(ln2.eps), wfwd(layer.fc1.weight).data_ptr(), p(layer.fc1.bias), ops.ACT_GELU_FWD,
        wfwd(layer.fc2.weight).data_ptr(), p(layer.fc2.bias), next_mode, p(nl.weight) if nl is not None else 0, p(nl.bias) if nl is not None else 0,
        float(nl.eps) if nl is not None else 0.0, ops.GEMM_LN_MAX_K if ops.GEMM_LN else 0, L.qkv.data_ptr(), L.s.data_ptr(), L.o.data_ptr(),
        L.x1.data_ptr(), L.h2.data_ptr(), L.m2.data_ptr(), L.r2.data_ptr(), L.u.data_ptr(), L.a.data_ptr(), x_out.data_ptr(), p(ln_out), p(mn), p(rn),
        int(a16 == torch.float16))
    return x_out, ln_out, mn, rn


def _unimol_seq_workspace(st, mod):
    """-> (usable, workspace) for mmdti_unimol_layer_bwd: the variant the library sequences is the hot one -- every parameter trainable, dimensions the grouped weight-gradient kernels take, no per-launch event
    timing requested.  The workspace holds one layer's temporaries and is shared by all layers of this backward."""
    lay = mod.layers[0]
    D, F, M = st.D, lay.fc1.weight.shape[0], st.M
    L0 = st.layers[0]
    ok = ((L0.h1.dtype != torch.float16 or L0.s.dtype == torch.float16)      # (fp16 operands: the sequencer covers the compact pair planes)
          and not ops.kernel_timer.names and ops.GROUPED_DW and D % 256 == 0 and F % 256 == 0 and M >= ops.GROUPED_DW_MIN_ROWS
          and all(p.requires_grad for p in mod.parameters()))
    if not ok:
        return False, None
    tiles = (D // 256) * (F // 256) * 2 + (D // 256) ** 2 * 4
    sk = ops.lib()._dll.mmdti_linear_dw_grouped_splits(tiles, M)
    nbytes = (M * F + 7 * M * D) * 2 + M * D * 4 + sk * (2 * D * F + 4 * D * D) * 4
    return True, torch.empty(nbytes + 256, device=st.emb.device, dtype=torch.uint8)


def _unimol_layer_bwd_seq(st, layer, L, dx, dx16, G, g_zero, scale, ws, below):
    """One Uni-Mol layer's backward as ONE library call (csrc/layers.hip): -> (dx, dx16) for the layer below (dx16 None at the
    lowest layer).  Same launches, arguments and order as the op-by-op body of PairEncoderFn.backward."""
    att, ln1, ln2 = layer.self_attn, layer.self_attn_layer_norm, layer.final_layer_norm
    M, D, F = st.M, st.D, layer.fc1.weight.shape[0]
    dx_out = torch.empty_like(dx)
    dx16_out = torch.empty(M, D, device=dx.device, dtype=BF16) if below is not None else None
    layout = ops._pair_layout_s(L.s, "pair_attn_bwd.s") | (4 if G.dtype == BF16 else 0)
    p = ops._p
    ops.lib().mmdti_unimol_layer_bwd(
        ops._stream(), M, st.B, st.N, st.H, D, F, st.ld, float(scale), float(st.p_res), float(st.p_att), int(st.seed),
        int(below[0]) if below is not None else 0, int(L.site_o), int(L.site_att), dx.data_ptr(), dx16.data_ptr(), dx_out.data_ptr(), p(dx16_out),
        p(below[1]) if below is not None else 0, L.a.data_ptr(), L.u.data_ptr(), ops.ACT_GELU_DX, L.h2.data_ptr(), L.x1.data_ptr(), L.m2.data_ptr(),
        L.r2.data_ptr(), L.o.data_ptr(), L.qkv.data_ptr(), L.s.data_ptr(), L.h1.data_ptr(), L.x.data_ptr(), L.m1.data_ptr(), L.r1.data_ptr(),
        wbf16(layer.fc2.weight).data_ptr(), wbf16(layer.fc1.weight).data_ptr(), wbf16(att.out_proj.weight).data_ptr(), wbf16(att.in_proj.weight).data_ptr(),
        ln2.weight.data_ptr(), ln1.weight.data_ptr(), gbuf(layer.fc2.weight).data_ptr(), gbuf(layer.fc1.weight).data_ptr(),
        gbuf(att.out_proj.weight).data_ptr(), gbuf(att.in_proj.weight).data_ptr(), gbuf(layer.fc1.bias).data_ptr(), gbuf(att.out_proj.bias).data_ptr(),
        gbuf(att.in_proj.bias).data_ptr(), gbuf(ln2.weight).data_ptr(), gbuf(ln2.bias).data_ptr(), gbuf(ln1.weight).data_ptr(), gbuf(ln1.bias).data_ptr(),
        G.data_ptr(), layout, int(g_zero), p(st.kt), p(st.row_off), (ws.data_ptr() + 255) // 256 * 256, ws.numel() - 256,
        int(L.h1.dtype == torch.float16))
    return dx_out, dx16_out


# ------------------------------------------------------------------------------------------------- Gaussian pair bias
class PairBiasFn(torch.autograd.Function):
    """gbf -> gbf_proj -> permute (models/mm_model.py:553-556): (dist [B,N,N] f32, edge_type [B,N,N] i64) ->
    bias [B,H,N,ld] fp32 -- or, on the fused hot path (K = F = 128, H = 64, N <= 272), the tiled pair layout, as fp16 unless
    MMDTI_PAIR_COMPACT=0.  The gradient chain of a compact bias is fp32 (or bf16), not fp16: PairEncoderFn.backward leaves it
    in the slot hung on the bias (``_mmdti_grad_slot``) and autograd only carries a placeholder; a gradient that reaches this
    function through autograd from any other consumer is added on top."""

    @staticmethod
    def forward(ctx, anchor, dist, edge_type, gbf, proj, ld, key_tiles_host=None, rows_host=None):
        # key_tiles_host ([B] ints on the HOST, ragged batches): real key tiles per molecule -- the fused compact path then
        # neither produces the bias of the all-padding key tiles nor visits them in the backward (PairEncoderFn skips them too)
        B, N, _ = dist.shape
        H = proj.linear2.weight.shape[0]
        dist, edge_type = dist.contiguous(), edge_type.contiguous()
        args = [gbf.mul.weight.view(-1), gbf.bias.weight.view(-1), gbf.means.weight.view(-1), gbf.stds.weight.view(-1)]
        fused = ops.gbf_bias_eligible(args[2].numel(), proj.linear1.weight.shape[0], H, ld)
        # the complete backward kernel recomputes basis / hidden from the inputs: the forward then saves nothing
        full = fused and ops.GBF_FULL_BWD and args[0].numel() <= ops.GBF_FULL_MAXE
        if edge_type.dtype != torch.int64 and not fused:
            edge_type = edge_type.long()      # (the unfused kernels read the reference's int64)
        if fused:
            # one kernel from distances to the [B,H,N,ld] bias (tiled pair layout whenever the MFMA pair-attention kernels can
            # take it: their loads become contiguous KiBs); without the complete backward kernel the three [P,128] intermediates
            # are kept for the backward
            keep = any(ctx.needs_input_grad) and not full  # inference: the kernel does not even write the intermediates
            tiled = ops.pair_tiled_ok(N)
            compact = tiled and ops.PAIR_COMPACT
            pre_f = pre_b = rb_f = rb_b = None
            if key_tiles_host is not None and compact and full:
                # (rows_host -- packed token rows: the bias of the query rows past a molecule's representative pad row is never read)
                if rows_host is not None:
                    pre_f, pre_b, rb_f, rb_b = ops.gbf_tile_prefixes(key_tiles_host, N, dist.device, rows_host)
                else:
                    pre_f, pre_b = ops.gbf_tile_prefixes(key_tiles_host, N, dist.device)
            out, saved = ops.gbf_bias_fwd(dist, edge_type, *args, wbf16(proj.linear1.weight), proj.linear1.bias,
                                          wbf16(proj.linear2.weight), proj.linear2.bias, ld, save=keep, tiled=tiled,
                                          save_grad=ops.GELU_SAVE_GRAD, compact=compact, tile_prefix=pre_f, row_blocks=rb_f)
            feat, u, h = saved if keep else (None, None, None)
            ugrad = ops.GELU_SAVE_GRAD
        else:
            feat = ops.gbf_features_fwd(dist, edge_type, *args)
            u = torch.empty(feat.shape[0], proj.linear1.weight.shape[0], device=dist.device, dtype=BF16)
            h = ops.linear_fwd(feat, wbf16(proj.linear1.weight), proj.linear1.bias, act=ops.ACT_GELU_FWD, aux_out=u)
            ugrad = ops.GELU_SAVE_GRAD
            o = ops.linear_fwd(h, wbf16(proj.linear2.weight), proj.linear2.bias, out_dtype=F32)
            out = ops.pair_permute_fwd(o, B, N, H, ld)
        ctx.st = SimpleNamespace(dist=dist, et=edge_type, feat=feat, u=u, h=h, B=B, N=N, H=H, ld=ld, fused=fused, ugrad=ugrad, full=full, slot=None,
                                 pre_b=pre_b if fused else None, rb_b=rb_b if fused else None)
        ctx.gbf, ctx.proj = gbf, proj
        if out.dtype == torch.float16:
            ctx.st.slot = out._mmdti_grad_slot = SimpleNamespace(g=None)
        return out

    @staticmethod
    def backward(ctx, g):
        st, gbf, proj = ctx.st, ctx.gbf, ctx.proj
        ps = [gbf.mul.weight, gbf.bias.weight, gbf.means.weight, gbf.stds.weight]
        if st.slot is not None:
            # compact bias: the gradient chain sits in the slot (PairEncoderFn.backward); what autograd delivered is a
            # zero-storage placeholder unless another consumer of the bias contributed a real fp16 gradient
            real = any(sd != 0 for sd in g.stride())
            chain, st.slot.g = st.slot.g, None
            if chain is None:
                g = g.float()
            elif real:
                g = chain.float().add_(g) if chain.dtype != F32 else chain.add_(g)
            else:
                g = chain
        if st.full:
            # ONE kernel: recompute, both dX products, GELU', the Gaussian backward and all four weight / bias gradients
            l1, l2 = proj.linear1, proj.linear2
            allp = [l1.weight, l1.bias, l2.weight, l2.bias] + ps
            grads = [gbuf(p) if p.requires_grad else torch.zeros_like(p) for p in allp]
            ops.gbf_bias_bwd_full(g.contiguous(), st.dist, st.et, *[p.view(-1) for p in ps], wbf16(l1.weight), l1.bias, wbf16(l2.weight),
                                  st.ld, *[gr.view(-1) for gr in grads], tile_prefix=st.pre_b, row_blocks=st.rb_b)
        elif st.fused and ps[0].numel() <= 4096:
            # one pass over G: re-layout, both dX products, GELU' and the Gaussian backward; only the two weight-gradient
            # GEMMs (contraction over all pairs) and their column sums remain
            grads = [gbuf(p) if p.requires_grad else torch.zeros_like(p) for p in ps]
            do, du = ops.gbf_bias_bwd(g.contiguous(), st.dist, st.et, *[p.view(-1) for p in ps], wbf16(proj.linear1.weight),
                                      wbf16(proj.linear2.weight), st.u, st.ld, *[gr.view(-1) for gr in grads], u_is_grad=st.ugrad)
            _lin_bwd_params(do, st.h, proj.linear2.weight, proj.linear2.bias)
            _lin_bwd_params(du, st.feat, proj.linear1.weight, proj.linear1.bias)
        else:
            do = ops.pair_permute_bwd(g.contiguous(), st.B, st.N, st.H, st.ld)          # [P,H] bf16
            _lin_bwd_params(do, st.h, proj.linear2.weight, proj.linear2.bias)
            du = ops.linear_bwd_input(do, wbf16(proj.linear2.weight), act=ops.ACT_MUL_AUX if st.ugrad else ops.ACT_GELU_BWD, aux_in=st.u)
            _lin_bwd_params(du, st.feat, proj.linear1.weight, proj.linear1.bias)
            dfeat = ops.linear_bwd_input(du, wbf16(proj.linear1.weight))
            if any(p.requires_grad for p in ps):
                grads = [gbuf(p) if p.requires_grad else torch.zeros_like(p) for p in ps]
                ops.gbf_features_bwd(st.dist, st.et, *[p.view(-1) for p in ps], dfeat, *[gr.view(-1) for gr in grads])
        notify_grads_ready(list(gbf.parameters()) + list(proj.parameters()))
        _join_stream_after_backward()
        return None, None, None, None, None, None, None, None


class PairCompactFn(torch.autograd.Function):
    """Row-major fp32 pair bias [B,H,N,ld] -> the compact tiled planes (fp16, -inf in every pad slot, gradient slot attached) that
    the ragged / packed pair-attention kernels stream.  Layout glue (plain indexing) for configurations the fused pair-bias kernel
    is not built for (head / basis counts other than 64 / 128): the hot path gets this layout straight from PairBiasFn."""

    @staticmethod
    def forward(ctx, bias, N):
        out = ops.pair_tile(bias[..., :N].float().clamp(max=65504.0), N).to(torch.float16)
        ctx.N, ctx.ld = N, bias.shape[-1]
        ctx.slot = out._mmdti_grad_slot = SimpleNamespace(g=None)
        return out

    @staticmethod
    def backward(ctx, g):
        chain, ctx.slot.g = ctx.slot.g, None
        real = any(sd != 0 for sd in g.stride())
        if chain is None:
            chain = g.float()
        elif real:
            chain = chain.float() + g.float()
        d = ops.pair_untile(chain.float(), ctx.N)
        if ctx.ld != ctx.N:
            d = torch.nn.functional.pad(d, (0, ctx.ld - ctx.N))
        return d, None


class EmbeddingFn(torch.autograd.Function):
    """nn.Embedding with padding_idx (mm_model.py:439-441,552)."""

    @staticmethod
    def forward(ctx, weight, ids, padding_idx):
        ctx.ids, ctx.w, ctx.pad = ids.contiguous(), weight, -1 if padding_idx is None else padding_idx
        return ops.embedding_fwd(ctx.ids, weight)

    @staticmethod
    def backward(ctx, dout):
        g = gbuf(ctx.w)
        if g is not None:
            ops.embedding_bwd_gemm(ctx.ids, ops.cast_bf16(dout.contiguous()), g, ctx.pad)
        notify_grads_ready([ctx.w])
        _join_stream_after_backward()
        return None, None, None


# ------------------------------------------------------------------------------------------------- BERT-style layer
def _bert_layer_fwd(st, s1_32, s1_16, s2_16, key_add, W, heads, p_hid, p_att, eps, seed, sites, self_attn):
    """Post-LN BERT layer with query from s1 and key/value from s2 (HF RobertaLayer; BertCrossAttentionLayer
    mm_module.py:615-626).  s1_32: [B*Lq,D] fp32; s1_16/s2_16 bf16; key_add [B,Lk] fp32.  Returns (out32, out16).
    Packed sequences (st.vl, an ops.AttnVarlen): the row arrays hold st.Mq / st.Mk packed rows instead of B*Lq / B*Lk, the keys of
    a sequence are its real rows only (key_add is None), Lq / Lk are the longest sequences."""
    B, Lq, Lk, D = st.B, st.Lq, st.Lk, st.D
    Mq, vl = st.Mq, st.vl
    hd = D // heads
    ld = (Lk + 7) // 8 * 8
    L = SimpleNamespace(s1_16=s1_16, s2_16=s2_16, ld=ld, heads=heads, self_attn=self_attn, W=W, eps=eps)
    L.site_att = sites.next()
    L.key_add, L.fused = key_add, ops.attn_eligible(Lq, Lk, hd, D)
    # query/key/value as ONE GEMM when their parameters sit back to back in the arena (trainer._qkv_groups) and the fused
    # attention kernels (which take row-strided q/k/v) run: [3D, D] for self-attention, [2D, D] (key|value) otherwise
    L.fw = L.fb = None
    if L.fused and W.q_b is not None:
        plist_w, plist_b = ((W.q_w, W.k_w, W.v_w), (W.q_b, W.k_b, W.v_b)) if self_attn else ((W.k_w, W.v_w), (W.k_b, W.v_b))
        L.fw, L.fb = fused_views(plist_w), fused_views(plist_b)
        if L.fw is None or L.fb is None:
            L.fw = L.fb = None
    L.seq = False
    if (LAYER_SEQ and self_attn and L.fw is not None and s1_32.is_cuda and not ops.kernel_timer.names and D % 256 == 0
            and W.i_w.shape[0] % 256 == 0 and Mq >= ops.GROUPED_DW_MIN_ROWS and ops.GROUPED_DW):
        # the layer's six launches from ONE library call (csrc/layers.hip: the same kernels, arguments and order as below)
        return _bert_layer_fwd_seq(st, L, s1_32, s1_16, key_add, W, heads, p_hid, p_att, eps, seed, sites)
    if (LAYER_SEQ and not self_attn and L.fw is not None and L.fused and s1_32.is_cuda and not ops.kernel_timer.names and D % 8 == 0
            and W.i_w.shape[0] % 8 == 0 and W.q_b is not None):
        # the cross-attention layer's six launches from ONE library call (csrc/layers.hip: mmdti_bert_cross_layer_fwd)
        return _bert_cross_layer_fwd_seq(st, L, s1_32, s1_16, s2_16, key_add, W, heads, p_hid, p_att, eps, seed, sites)
    if L.fw is not None:
        if self_attn:
            # (L.fw[3]: the forward-GEMM shadow of the fused weights; q, k, v are stored bf16 in every mode -- the attention kernels'
            #  operand type)
            L.qkv = ops.linear_fwd(s1_16, L.fw[3], L.fb[1], out_dtype=BF16)
            L.q, L.k, L.v = L.qkv[:, :D], L.qkv[:, D:2 * D], L.qkv[:, 2 * D:]
        else:
            L.q = ops.linear_fwd(s1_16, wfwd(W.q_w), W.q_b, out_dtype=BF16)
            L.qkv = ops.linear_fwd(s2_16, L.fw[3], L.fb[1], out_dtype=BF16)
            L.k, L.v = L.qkv[:, :D], L.qkv[:, D:]
    else:
        L.q = ops.linear_fwd(s1_16, wfwd(W.q_w), W.q_b, out_dtype=BF16)
        L.k = ops.linear_fwd(s2_16, wfwd(W.k_w), W.k_b, out_dtype=BF16)
        L.v = ops.linear_fwd(s2_16, wfwd(W.v_w), W.v_b, out_dtype=BF16)
    if L.fused:
        # scores, softmax, dropout and context in one kernel: the [B,heads,Lq,Lk] tensor never reaches HBM
        L.ctx, L.stats = ops.attn_fwd(L.q, L.k, L.v, key_add, B, heads, Lq, Lk, 1.0 / math.sqrt(hd), p_att, seed, L.site_att, vl=vl)
    else:
        if vl is not None:
            raise ops.MMDTIError("packed sequences need the fused attention kernels (head_dim 32 / 64, at most 256 tokens)")
        S = torch.empty(B, heads, Lq, ld, device=s1_32.device, dtype=F32)
        ops.gemm(L.q, L.k, M=Lq, N=Lk, K=hd, lda=D, ldb=D, out=S, ldc=ld, batch=(B, heads), sA=(Lq * D, hd), sB=(Lk * D, hd),
                 sC=(heads * Lq * ld, Lq * ld), alpha=1.0 / math.sqrt(hd))
        L.p, L.pd = ops.softmax_fwd(S, key_add, B, heads, Lq, Lk, ld, p_att, seed, L.site_att)
        del S
        L.ctx = torch.empty(B * Lq, D, device=s1_32.device, dtype=ops.act16())
        ops.gemm(L.pd, L.v, M=Lq, N=hd, K=Lk, lda=ld, ldb=D, transB=True, out=L.ctx, ldc=D, batch=(B, heads),
                 sA=(heads * Lq * ld, Lq * ld), sB=(Lk * D, hd), sC=(Lq * D, hd))
    L.site_o = sites.next()
    # (each closing Linear of a residual branch runs fused with the post-LN LayerNorm behind it: ops.linear_ln_fwd)
    L.y, L.a32, L.a16, L.am, L.ar = ops.linear_ln_fwd(L.ctx, wfwd(W.o_w), W.o_b, W.ln1_w, W.ln1_b, eps, residual=s1_32, drop_p=p_hid, seed=seed,
                                                      site=L.site_o, want_f32=True, want_bf16=True)
    L.u = torch.empty(Mq, W.i_w.shape[0], device=s1_32.device, dtype=BF16)
    L.i = ops.linear_fwd(L.a16, wfwd(W.i_w), W.i_b, act=ops.ACT_GELU_FWD, aux_out=L.u)
    L.site_f = sites.next()
    L.z, out32, out16, L.zm, L.zr = ops.linear_ln_fwd(L.i, wfwd(W.o2_w), W.o2_b, W.ln2_w, W.ln2_b, eps, residual=L.a32, drop_p=p_hid, seed=seed,
                                                      site=L.site_f, want_f32=True, want_bf16=True)
    L.p_hid, L.p_att = p_hid, p_att
    return L, out32, out16


def _bert_layer_bwd(st, L, dout, seed):
    """-> (ds1 fp32 [B*Lq,D], ds2 fp32 [B*Lk,D] or None when self_attn (then ds1 holds the sum))."""
    B, Lq, Lk, D = st.B, st.Lq, st.Lk, st.D
    Mq, Mk, vl = st.Mq, st.Mk, st.vl
    W, heads, ld = L.W, L.heads, L.ld
    hd = D // heads
    if getattr(L, "seq", False) is True and not ops.kernel_timer.names and all(gbuf(p) is not None for p in (W.o_w, W.o_b, W.i_w, W.i_b, W.o2_w, W.o2_b, W.ln1_w, W.ln1_b,
                                                                                                        W.ln2_w, W.ln2_b)):
        return _bert_layer_bwd_seq(st, L, dout, seed), None
    if getattr(L, "seq", False) == "cross" and not ops.kernel_timer.names and all(gbuf(p) is not None for p in (W.o_b, W.o2_b, W.ln1_w, W.ln1_b, W.ln2_w, W.ln2_b)):
        return _bert_cross_layer_bwd_seq(st, L, dout, seed)
    pend, raw = [], []                     # the layer's weight gradients leave as one grouped launch (see _lin_bwd_params_many)
    dz, dzb = ops.layernorm_bwd(dout, L.z, W.ln2_w, L.zm, L.zr, gbuf(W.ln2_w), gbuf(W.ln2_b), bf16_copy=(L.p_hid, L.site_f, gbuf(W.o2_b)))
    pend.append(((dzb, L.i, W.o2_w, W.o2_b), dict(bias_done=True)))
    du = ops.linear_bwd_input(dzb, wbf16(W.o2_w), act=ops.ACT_GELU_DX, aux_in=L.u)
    pend.append(((du, L.a16, W.i_w, W.i_b), {}))
    da = ops.linear_bwd_input(du, wbf16(W.i_w))
    # a32 = LN1(y) feeds the FFN AND the residual add of z: both gradients go through LN1's backward
    dy, dyb = ops.layernorm_bwd(da, L.y, W.ln1_w, L.am, L.ar, gbuf(W.ln1_w), gbuf(W.ln1_b), dy_add=dz, bf16_copy=(L.p_hid, L.site_o, gbuf(W.o_b)))
    pend.append(((dyb, L.ctx, W.o_w, W.o_b), dict(bias_done=True)))
    dctx = ops.linear_bwd_input(dyb, wbf16(W.o_w))
    dev = dout.device
    if L.fw is not None:
        # fused q|k|v (or k|v) projection: the attention backward writes straight into one [rows, 3D | 2D] gradient, which
        # then feeds ONE weight-gradient GEMM, one column sum and ONE input-gradient GEMM
        dqkv = torch.empty_like(L.qkv)
        if L.self_attn:
            outv = (dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
        else:
            outv = (torch.empty(Mq, D, device=dev, dtype=BF16), dqkv[:, :D], dqkv[:, D:])
        dq, dk, dv = ops.attn_bwd(L.q, L.k, L.v, L.key_add, dctx, L.stats, B, heads, Lq, Lk, 1.0 / math.sqrt(hd), L.p_att, seed, L.site_att,
                                  out=outv, vl=vl)
        raw.append((dqkv, L.s1_16 if L.self_attn else L.s2_16, L.fw[2], L.fb[2].view(-1), None))
        ds1 = dy
        if L.self_attn:
            ops.gemm(dqkv, L.fw[0], M=Mq, N=D, K=3 * D, lda=3 * D, ldb=D, transB=True, out=ds1, ldc=D, beta=1.0)
            _lin_bwd_params_many(pend, raw)
            return ds1, None
        pend.append(((dq, L.s1_16, W.q_w, W.q_b), {}))
        ops.gemm(dq, wbf16(W.q_w), M=Mq, N=D, K=D, lda=D, ldb=D, transB=True, out=ds1, ldc=D, beta=1.0)
        ds2 = ops.gemm(dqkv, L.fw[0], M=Mk, N=D, K=2 * D, lda=2 * D, ldb=D, transB=True, out_dtype=F32)
        _lin_bwd_params_many(pend, raw)
        return ds1, ds2
    if L.fused:
        dq, dk, dv = ops.attn_bwd(L.q, L.k, L.v, L.key_add, dctx, L.stats, B, heads, Lq, Lk, 1.0 / math.sqrt(hd), L.p_att, seed, L.site_att, vl=vl)
    else:
        dP = torch.empty(B, heads, Lq, ld, device=dev, dtype=F32)
        ops.gemm(dctx, L.v, M=Lq, N=Lk, K=hd, lda=D, ldb=D, out=dP, ldc=ld, batch=(B, heads), sA=(Lq * D, hd), sB=(Lk * D, hd),
                 sC=(heads * Lq * ld, Lq * ld))
        dv = torch.empty(B * Lk, D, device=dev, dtype=BF16)
        ops.gemm(L.pd, dctx, M=Lk, N=hd, K=Lq, lda=ld, ldb=D, transA=True, transB=True, out=dv, ldc=D, batch=(B, heads),
                 sA=(heads * Lq * ld, Lq * ld), sB=(Lq * D, hd), sC=(Lk * D, hd))
        dS = ops.softmax_bwd(L.p, dP, B, heads, Lq, Lk, ld, 1.0 / math.sqrt(hd), L.p_att, seed, L.site_att)
        del dP
        dq = torch.empty(B * Lq, D, device=dev, dtype=BF16)
        ops.gemm(dS, L.k, M=Lq, N=hd, K=Lk, lda=ld, ldb=D, transB=True, out=dq, ldc=D, batch=(B, heads),
                 sA=(heads * Lq * ld, Lq * ld), sB=(Lk * D, hd), sC=(Lq * D, hd))
        dk = torch.empty(B * Lk, D, device=dev, dtype=BF16)
        ops.gemm(dS, L.q, M=Lk, N=hd, K=Lq, lda=ld, ldb=D, transA=True, transB=True, out=dk, ldc=D, batch=(B, heads),
                 sA=(heads * Lq * ld, Lq * ld), sB=(Lq * D, hd), sC=(Lk * D, hd))
    pend.append(((dq, L.s1_16, W.q_w, W.q_b), {}))
    pend.append(((dk, L.s2_16, W.k_w, W.k_b), {}))
    pend.append(((dv, L.s2_16, W.v_w, W.v_b), {}))
    _lin_bwd_params_many(pend, raw)
    # ds1 = dy (residual) + dq.Wq ; ds2 = dk.Wk + dv.Wv     (fp32, accumulated by the GEMM's beta=1 epilogue)
    ds1 = dy
    ops.gemm(dq, wbf16(W.q_w), M=Mq, N=D, K=D, lda=D, ldb=D, transB=True, out=ds1, ldc=D, beta=1.0)
    if L.self_attn:
        ops.gemm(dk, wbf16(W.k_w), M=Mk, N=D, K=D, lda=D, ldb=D, transB=True, out=ds1, ldc=D, beta=1.0)
        ops.gemm(dv, wbf16(W.v_w), M=Mk, N=D, K=D, lda=D, ldb=D, transB=True, out=ds1, ldc=D, beta=1.0)
        return ds1, None
    ds2 = ops.gemm(dk, wbf16(W.k_w), M=Mk, N=D, K=D, lda=D, ldb=D, transB=True, out_dtype=F32)
    ops.gemm(dv, wbf16(W.v_w), M=Mk, N=D, K=D, lda=D, ldb=D, transB=True, out=ds2, ldc=D, beta=1.0)
    return ds1, ds2


def _bert_cross_layer_fwd_seq(st, L, s1_32, s1_16, s2_16, key_add, W, heads, p_hid, p_att, eps, seed, sites):
    """_bert_layer_fwd's cross-attention variant (queries from s1, fused key | value projection of s2, fused attention) as ONE library call."""
    B, Lq, Lk, D, Mq, Mk, vl = st.B, st.Lq, st.Lk, st.D, st.Mq, st.Mk, st.vl
    F = W.i_w.shape[0]
    dev = s1_32.device
    e = torch.empty
    L.site_o, L.site_f = sites.next(), sites.next()
    a16 = s1_16.dtype
    L.q, L.qkv, L.ctx = e(Mq, D, device=dev, dtype=BF16), e(Mk, 2 * D, device=dev, dtype=BF16), e(Mq, D, device=dev, dtype=a16)
    L.k, L.v = L.qkv[:, :D], L.qkv[:, D:]
    L.stats = e((B, heads, Lq, 2) if vl is None else (heads, Mq, 2), device=dev, dtype=F32)
    L.y, L.a32, L.a16 = e(Mq, D, device=dev, dtype=F32), e(Mq, D, device=dev, dtype=F32), e(Mq, D, device=dev, dtype=a16)
    L.am, L.ar, L.zm, L.zr = (e(Mq, device=dev, dtype=F32) for _ in range(4))
    L.u, L.i = e(Mq, F, device=dev, dtype=BF16), e(Mq, F, device=dev, dtype=a16)
    L.z, out32, out16 = e(Mq, D, device=dev, dtype=F32), e(Mq, D, device=dev, dtype=F32), e(Mq, D, device=dev, dtype=a16)
    L.seq, L.p_hid, L.p_att = "cross", p_hid, p_att
    p = ops._p
    ops.lib().mmdti_bert_cross_layer_fwd(
        ops._stream(), Mq, Mk, B, Lq, Lk, heads, D, F, float(1.0 / math.sqrt(D // heads)), float(p_hid), float(p_att), int(seed), int(L.site_att),
        int(L.site_o), int(L.site_f), s1_32.data_ptr(), s1_16.data_ptr(), s2_16.data_ptr(), p(key_add), *(ops._NO_VARLEN if vl is None else vl.args()),
        wfwd(W.q_w).data_ptr(), p(W.q_b), L.fw[3].data_ptr(), L.fb[1].data_ptr(), wfwd(W.o_w).data_ptr(), p(W.o_b), W.ln1_w.data_ptr(), W.ln1_b.data_ptr(),
        wfwd(W.i_w).data_ptr(), p(W.i_b), ops.ACT_GELU_FWD, wfwd(W.o2_w).data_ptr(), p(W.o2_b), W.ln2_w.data_ptr(), W.ln2_b.data_ptr(), float(eps),
        ops.GEMM_LN_MAX_K if ops.GEMM_LN else 0, L.q.data_ptr(), L.qkv.data_ptr(), L.ctx.data_ptr(), L.stats.data_ptr(), L.y.data_ptr(), L.a32.data_ptr(),
        L.a16.data_ptr(), L.am.data_ptr(), L.ar.data_ptr(), L.u.data_ptr(), L.i.data_ptr(), L.z.data_ptr(), out32.data_ptr(), out16.data_ptr(),
        L.zm.data_ptr(), L.zr.data_ptr(), int(a16 == torch.float16))
    return L, out32, out16


def _bert_cross_layer_bwd_seq(st, L, dout, seed):
    """_bert_layer_bwd of a layer that went through _bert_cross_layer_fwd_seq: its ten launches up to the weight gradients as ONE library
    call, then the weight gradients exactly as the op-by-op path launches them -> (ds1, ds2) fp32."""
    B, Lq, Lk, D, Mq, Mk, vl = st.B, st.Lq, st.Lk, st.D, st.Mq, st.Mk, st.vl
    W, heads = L.W, L.heads
    F = W.i_w.shape[0]
    dev = dout.device
    dout = dout.contiguous()
    e = torch.empty
    ds1, ds2 = e(Mq, D, device=dev, dtype=F32), e(Mk, D, device=dev, dtype=F32)
    dzb, du, dyb = e(Mq, D, device=dev, dtype=BF16), e(Mq, F, device=dev, dtype=BF16), e(Mq, D, device=dev, dtype=BF16)
    dq, dqkv = e(Mq, D, device=dev, dtype=BF16), torch.empty_like(L.qkv)
    nrow = heads * Mq if vl is not None else B * heads * Lq
    ws = e(4 * Mq * D + 4 * Mq * D + (nrow * 4 + 15) // 16 * 16 + 256, device=dev, dtype=torch.uint8)
    p = ops._p
    ops.lib().mmdti_bert_cross_layer_bwd(
        ops._stream(), Mq, Mk, B, Lq, Lk, heads, D, F, float(1.0 / math.sqrt(D // heads)), float(L.p_hid), float(L.p_att), int(seed), int(L.site_att),
        int(L.site_o), int(L.site_f), dout.data_ptr(), ds1.data_ptr(), ds2.data_ptr(), p(L.key_add), *(ops._NO_VARLEN if vl is None else vl.args()),
        L.q.data_ptr(), L.qkv.data_ptr(), L.stats.data_ptr(), L.y.data_ptr(), L.am.data_ptr(), L.ar.data_ptr(), L.u.data_ptr(), ops.ACT_GELU_DX,
        L.z.data_ptr(), L.zm.data_ptr(), L.zr.data_ptr(), wbf16(W.q_w).data_ptr(), L.fw[0].data_ptr(), wbf16(W.o_w).data_ptr(), wbf16(W.i_w).data_ptr(),
        wbf16(W.o2_w).data_ptr(), W.ln1_w.data_ptr(), W.ln2_w.data_ptr(), gbuf(W.o_b).data_ptr(), gbuf(W.o2_b).data_ptr(), gbuf(W.ln1_w).data_ptr(),
        gbuf(W.ln1_b).data_ptr(), gbuf(W.ln2_w).data_ptr(), gbuf(W.ln2_b).data_ptr(), dzb.data_ptr(), du.data_ptr(), dyb.data_ptr(), dq.data_ptr(),
        dqkv.data_ptr(), (ws.data_ptr() + 255) // 256 * 256, ws.numel() - 256)
    # the weight gradients, as the op-by-op path hands them over (same items, same order: _lin_bwd_params_many groups them by row count)
    pend = [((dzb, L.i, W.o2_w, W.o2_b), dict(bias_done=True)), ((du, L.a16, W.i_w, W.i_b), {}), ((dyb, L.ctx, W.o_w, W.o_b), dict(bias_done=True)),
            ((dq, L.s1_16, W.q_w, W.q_b), {})]
    _lin_bwd_params_many(pend, [(dqkv, L.s2_16, L.fw[2], L.fb[2].view(-1), None)])
    return ds1, ds2


def _bert_layer_fwd_seq(st, L, s1_32, s1_16, key_add, W, heads, p_hid, p_att, eps, seed, sites):
    """_bert_layer_fwd's self-attention / fused-projection / fused-attention variant as ONE library call (csrc/layers.hip)."""
    B, Lq, D, Mq, vl = st.B, st.Lq, st.D, st.Mq, st.vl
    F = W.i_w.shape[0]
    dev = s1_32.device
    e = torch.empty
    L.site_o, L.site_f = sites.next(), sites.next()          # (L.site_att was drawn by the caller: the same numbering as the op-by-op path)
    a16 = s1_16.dtype                         # bf16, or fp16 (fp16 forward operands); q | k | v and the saved gelu' stay bf16 in every mode
    L.qkv, L.ctx = e(Mq, 3 * D, device=dev, dtype=BF16), e(Mq, D, device=dev, dtype=a16)
    L.q, L.k, L.v = L.qkv[:, :D], L.qkv[:, D:2 * D], L.qkv[:, 2 * D:]
    L.stats = e((B, heads, Lq, 2) if vl is None else (heads, Mq, 2), device=dev, dtype=F32)
    L.y, L.a32, L.a16 = e(Mq, D, device=dev, dtype=F32), e(Mq, D, device=dev, dtype=F32), e(Mq, D, device=dev, dtype=a16)
    L.am, L.ar, L.zm, L.zr = (e(Mq, device=dev, dtype=F32) for _ in range(4))
    L.u, L.i = e(Mq, F, device=dev, dtype=BF16), e(Mq, F, device=dev, dtype=a16)
    L.z, out32, out16 = e(Mq, D, device=dev, dtype=F32), e(Mq, D, device=dev, dtype=F32), e(Mq, D, device=dev, dtype=a16)
    L.fused, L.seq, L.p_hid, L.p_att = True, True, p_hid, p_att
    p = ops._p
    ops.lib().mmdti_bert_layer_fwd(
        ops._stream(), Mq, B, Lq, heads, D, F, float(1.0 / math.sqrt(D // heads)), float(p_hid), float(p_att), int(seed), int(L.site_att), int(L.site_o),
        int(L.site_f), s1_32.data_ptr(), s1_16.data_ptr(), p(key_add), *(ops._NO_VARLEN if vl is None else vl.args()), L.fw[3].data_ptr(),
        L.fb[1].data_ptr(), wfwd(W.o_w).data_ptr(), p(W.o_b), W.ln1_w.data_ptr(), W.ln1_b.data_ptr(), wfwd(W.i_w).data_ptr(), p(W.i_b), ops.ACT_GELU_FWD,
        wfwd(W.o2_w).data_ptr(), p(W.o2_b), W.ln2_w.data_ptr(), W.ln2_b.data_ptr(), float(eps), ops.GEMM_LN_MAX_K if ops.GEMM_LN else 0,
        L.qkv.data_ptr(), L.ctx.data_ptr(), L.stats.data_ptr(), L.y.data_ptr(), L.a32.data_ptr(), L.a16.data_ptr(), L.am.data_ptr(), L.ar.data_ptr(),
        L.u.data_ptr(), L.i.data_ptr(), L.z.data_ptr(), out32.data_ptr(), out16.data_ptr(), L.zm.data_ptr(), L.zr.data_ptr(),
        int(a16 == torch.float16))
    return L, out32, out16


def _bert_layer_bwd_seq(st, L, dout, seed):
    """_bert_layer_bwd of a layer that went through _bert_layer_fwd_seq, as ONE library call -> ds1 (fp32)."""
    B, Lq, D, Mq, vl = st.B, st.Lq, st.D, st.Mq, st.vl
    W, heads = L.W, L.heads
    F = W.i_w.shape[0]
    dout = dout.contiguous()
    ds1 = torch.empty(Mq, D, device=dout.device, dtype=F32)
    tiles = 3 * (D // 256) ** 2 + 2 * (D // 256) * (F // 256) + (D // 256) ** 2
    sk = ops.lib()._dll.mmdti_linear_dw_grouped_splits(tiles, Mq)
    nrow = heads * Mq if vl is not None else B * heads * Lq
    nbytes = (Mq * F + 7 * Mq * D) * 2 + Mq * D * 4 + (nrow * 4 + 15) // 16 * 16 + sk * (4 * D * D + 2 * D * F) * 4
    ws = torch.empty(nbytes + 256, device=dout.device, dtype=torch.uint8)
    p = ops._p
    ops.lib().mmdti_bert_layer_bwd(
        ops._stream(), Mq, B, Lq, heads, D, F, float(1.0 / math.sqrt(D // heads)), float(L.p_hid), float(L.p_att), int(seed), int(L.site_att), int(L.site_o),
        int(L.site_f), dout.data_ptr(), ds1.data_ptr(), L.s1_16.data_ptr(), p(L.key_add), *(ops._NO_VARLEN if vl is None else vl.args()),
        L.qkv.data_ptr(), L.ctx.data_ptr(), L.stats.data_ptr(), L.y.data_ptr(), L.a16.data_ptr(), L.am.data_ptr(), L.ar.data_ptr(), L.u.data_ptr(),
        ops.ACT_GELU_DX, L.i.data_ptr(), L.z.data_ptr(), L.zm.data_ptr(), L.zr.data_ptr(), L.fw[0].data_ptr(), wbf16(W.o_w).data_ptr(),
        wbf16(W.i_w).data_ptr(), wbf16(W.o2_w).data_ptr(), W.ln1_w.data_ptr(), W.ln2_w.data_ptr(), L.fw[2].data_ptr(), L.fw[2].stride(0),
        L.fb[2].view(-1).data_ptr(), gbuf(W.o_w).data_ptr(), gbuf(W.o_b).data_ptr(), gbuf(W.i_w).data_ptr(), gbuf(W.i_b).data_ptr(),
        gbuf(W.o2_w).data_ptr(), gbuf(W.o2_b).data_ptr(), gbuf(W.ln1_w).data_ptr(), gbuf(W.ln1_b).data_ptr(), gbuf(W.ln2_w).data_ptr(),
        gbuf(W.ln2_b).data_ptr(), (ws.data_ptr() + 255) // 256 * 256, ws.numel() - 256, int(L.s1_16.dtype == torch.float16))
    return ds1


def _bert_stack_tables(mod, st):
    """Pointer tables of tower 2's layers for mmdti_bert_stack_fwd / _bwd (cached on the parameter arena), or None when the stack
    calls do not cover this model: parameters outside one arena or frozen, q | k | v not back to back, shapes the fused attention /
    grouped weight-gradient kernels do not take."""
    layers = list(mod.layers)
    W0 = bert_weights(layers[0])
    arena = getattr(W0.i_w, "_mmdti_arena", None)
    heads = mod.cfg.heads
    D, F = W0.i_w.shape[1], W0.i_w.shape[0]
    if (arena is None or not ops.GROUPED_DW or st.Mq < ops.GROUPED_DW_MIN_ROWS or D % heads or not ops.attn_eligible(st.Lq, st.Lk, D // heads, D)):
        return None
    cache = arena.__dict__.setdefault("_stack_tables", {})
    key = (id(mod), ops.FWD_F16, "bert")
    T = cache.get(key)
    if T is None:
        ok = D % 256 == 0 and F % 256 == 0
        Ws = [bert_weights(l) for l in layers]
        off = arena.offsets

        def plist(W):
            return [W.q_w, W.k_w, W.v_w, W.q_b, W.k_b, W.v_b, W.o_w, W.o_b, W.ln1_w, W.ln1_b, W.i_w, W.i_b, W.o2_w, W.o2_b, W.ln2_w, W.ln2_b]
        for l, W in zip(layers, Ws):
            ps = plist(W)
            ok = ok and all(q is not None and q.requires_grad and getattr(q, "_mmdti_arena", None) is arena for q in ps)
            ok = ok and tuple(W.i_w.shape) == (F, D) and len(list(l.parameters())) == 16
            if ok:      # q | k | v (weights and biases) back to back in the arena: one [3D, D] matrix, one [3D] bias
                ok = (off[id(W.k_w)] == off[id(W.q_w)] + W.q_w.numel() and off[id(W.v_w)] == off[id(W.k_w)] + W.k_w.numel()
                      and off[id(W.k_b)] == off[id(W.q_b)] + W.q_b.numel() and off[id(W.v_b)] == off[id(W.k_b)] + W.k_b.numel())
        if not ok:
            cache[key] = False
            return None
        if ops.FWD_F16:
            arena._fresh16()
        sh16 = (arena.shadow16 if ops.FWD_F16 else arena.shadow).data_ptr()
        shb, gr = arena.shadow.data_ptr(), arena.grad.data_ptr()
        fwd, bwd, grads, weights, params, probes = [], [], [], [], [], []
        for W in Ws:
            params += plist(W)
            w16 = lambda q: sh16 + 2 * off[id(q)]
            wb = lambda q: shb + 2 * off[id(q)]
            g = lambda q: gr + 4 * off[id(q)]
            fwd += [w16(W.q_w), W.q_b.data_ptr(), w16(W.o_w), W.o_b.data_ptr(), W.ln1_w.data_ptr(), W.ln1_b.data_ptr(), w16(W.i_w), W.i_b.data_ptr(),
                    w16(W.o2_w), W.o2_b.data_ptr(), W.ln2_w.data_ptr(), W.ln2_b.data_ptr()]
            bwd += [wb(W.q_w), wb(W.o_w), wb(W.i_w), wb(W.o2_w), W.ln1_w.data_ptr(), W.ln2_w.data_ptr()]
            grads += [g(W.q_w), g(W.q_b), g(W.o_w), g(W.o_b), g(W.i_w), g(W.i_b), g(W.o2_w), g(W.o2_b), g(W.ln1_w), g(W.ln1_b), g(W.ln2_w), g(W.ln2_b)]
            weights += [(q, id(q)) for q in (W.q_w, W.k_w, W.v_w, W.o_w, W.i_w, W.o2_w)]
            probes.append((W.i_w, W.i_w.data_ptr(), g(W.i_w)))
        T = cache[key] = SimpleNamespace(nl=len(layers), D=D, F=F, heads=heads, fwd=_ptr_table(fwd), bwd=_ptr_table(bwd), grads=_ptr_table(grads),
                                         weights=weights, params=params, f16=ops.FWD_F16, probes=probes)
    if T is False:
        return None
    ver = arena._version
    for q, i in T.weights:
        if q._version != ver[i]:
            arena._fresh(q)
    if T.f16:
        arena._fresh16()
    for q, dptr, gptr in T.probes:
        g = q.grad
        if g is None or g.data_ptr() != gptr or q.data_ptr() != dptr:
            return None
    return T


_bert_layouts = {}


def _bert_stack_layout(st, T):
    heads, D, F = T.heads, T.D, T.F
    nrow = heads * st.Mq if st.vl is not None else st.B * heads * st.Lq
    key = (st.Mq, D, F, nrow)
    r = _bert_layouts.get(key)
    if r is None:
        import ctypes
        tiles = 3 * (D // 256) ** 2 + 2 * (D // 256) * (F // 256) + (D // 256) ** 2
        slab = ops.lib()._dll.mmdti_linear_dw_grouped_splits(tiles, st.Mq) * (4 * D * D + 2 * D * F) * 4
        out = (ctypes.c_longlong * 2)()
        ops.lib().mmdti_bert_stack_layout(st.Mq, D, F, nrow * 8, nrow, slab, ctypes.addressof(out))
        if len(_bert_layouts) > 4096:
            _bert_layouts.clear()
        r = _bert_layouts[key] = (int(out[0]), int(out[1]), slab, nrow * 8)
    return r


def _bert_stack_fwd(st, T, x32, x16, key_add, cfg, p_hid, p_att, seed, sites):
    """All layers of tower 2 as ONE library call -> the tower's fp32 output [Mq, D]."""
    stride, ws_bytes, slab, stats_bytes = _bert_stack_layout(st, T)
    dev = x32.device
    arena = torch.empty(stride * T.nl, device=dev, dtype=torch.uint8)
    out32 = torch.empty(st.Mq, T.D, device=dev, dtype=F32)
    site0 = sites.n + 1
    sites.n += 3 * T.nl
    vl = st.vl
    ops.lib().mmdti_bert_stack_fwd(
        ops._stream(), T.nl, st.Mq, st.B, st.Lq, T.heads, T.D, T.F, float(1.0 / math.sqrt(T.D // T.heads)), float(p_hid), float(p_att), int(seed), site0,
        x32.data_ptr(), x16.data_ptr(), ops._p(key_add), *(ops._NO_VARLEN if vl is None else vl.args()), T.fwd[1], ops.ACT_GELU_FWD, float(cfg.ln_eps),
        ops.GEMM_LN_MAX_K if ops.GEMM_LN else 0, arena.data_ptr(), arena.numel(), stats_bytes, out32.data_ptr(), int(x16.dtype == torch.float16))
    st.stack = SimpleNamespace(T=T, arena=arena, ws_bytes=ws_bytes, slab=slab, stats_bytes=stats_bytes, site0=site0, x16=x16, key_add=key_add, p_hid=p_hid,
                               p_att=p_att)
    return out32


def _bert_stack_bwd(st, dout):
    """All layers' backward as ONE library call -> the gradient of the embeddings' LayerNorm output (fp32 [Mq, D])."""
    S = st.stack
    T = S.T
    dev = dout.device
    ds1 = torch.empty(st.Mq, T.D, device=dev, dtype=F32)
    ws = torch.empty(S.ws_bytes, device=dev, dtype=torch.uint8)
    vl = st.vl
    ops.lib().mmdti_bert_stack_bwd(
        ops._stream(), T.nl, st.Mq, st.B, st.Lq, T.heads, T.D, T.F, float(1.0 / math.sqrt(T.D // T.heads)), float(S.p_hid), float(S.p_att), int(st.seed),
        S.site0, dout.data_ptr(), ds1.data_ptr(), S.x16.data_ptr(), ops._p(S.key_add), *(ops._NO_VARLEN if vl is None else vl.args()), T.bwd[1],
        ops.ACT_GELU_DX, T.grads[1], T.D, S.arena.data_ptr(), S.arena.numel(), S.stats_bytes, ws.data_ptr(), ws.numel(), S.slab,
        int(S.x16.dtype == torch.float16))
    return ds1


def bert_weights(layer) -> SimpleNamespace:
    """Parameter view of an HF RobertaLayer / mm_module BertCrossAttentionLayer (identical sub-module names)."""
    a, o = layer.attention.self, layer.attention.output
    return SimpleNamespace(q_w=a.query.weight, q_b=a.query.bias, k_w=a.key.weight, k_b=a.key.bias, v_w=a.value.weight, v_b=a.value.bias,
                           o_w=o.dense.weight, o_b=o.dense.bias, ln1_w=o.LayerNorm.weight, ln1_b=o.LayerNorm.bias,
                           i_w=layer.intermediate.dense.weight, i_b=layer.intermediate.dense.bias,
                           o2_w=layer.output.dense.weight, o2_b=layer.output.dense.bias,
                           ln2_w=layer.output.LayerNorm.weight, ln2_b=layer.output.LayerNorm.bias)


class RobertaEncoderFn(torch.autograd.Function):
    """self.bert(input_ids, attention_mask)[0]  (mm_model.py:562): embeddings + L post-LN layers.
    `mod` exposes: word/position/token_type embedding weights, emb LayerNorm, layers, cfg (heads, eps, dropouts, pad)."""

    @staticmethod
    def forward(ctx, anchor, input_ids, attention_mask, mod, training, pack=None):
        """pack (packing.PackedRows over the right-padded input_ids): the tower runs on the packed rows -- every sequence's real
        tokens plus ONE representative pad row (pad word embedding + position padding_idx: all masked slots are that row) -- and
        returns [pack.M, D]; masked keys are left out of the attention instead of being added finfo.min (probability 0 either way)."""
        B, Lq = input_ids.shape
        cfg = mod.cfg
        D = mod.word.shape[1]
        seed = dropout_state.next_seed()
        sites = _Sites()
        p_hid = cfg.hidden_dropout if training else 0.0
        p_att = cfg.attn_dropout if training else 0.0
        ids = input_ids.contiguous()
        pos = ops.roberta_position_ids(ids, cfg.pad_idx)
        vl, Mq = None, B * Lq
        if pack is not None:
            ids, pos = ids.view(-1)[pack.gather], pos.view(-1)[pack.gather]      # (integer plumbing: the packed rows' ids)
            vl, Mq, Lq = ops.AttnVarlen(pack, pack), pack.M, pack.max_rows
        e = ops.embedding_fwd3(ids, mod.word, pos, mod.position, mod.token_type)      # word + position + token type 0, one pass
        zeros = None
        st = SimpleNamespace(B=B, Lq=Lq, Lk=Lq if vl is None else vl.Lk, D=D, seed=seed, ids=ids, pos=pos, zeros=zeros, e=e, layers=[], p_hid=p_hid, Mq=Mq, Mk=Mq, vl=vl)
        keep = any(ctx.needs_input_grad)      # inference: no per-layer activations are kept
        st.site_emb = sites.next()
        x32, x16, st.em, st.er = ops.layernorm_fwd(e.view(Mq, D), mod.emb_ln_w, mod.emb_ln_b, cfg.ln_eps, want_f32=True, want_bf16=True,
                                                   drop_p=p_hid, seed=seed, site=st.site_emb)
        key_add = ((1.0 - attention_mask.to(F32)) * torch.finfo(torch.float32).min).contiguous() if vl is None else None
        # small batches: ALL layers from one library call (see _bert_stack_fwd); the per-layer loop then has nothing left to do
        st.stack = None
        T = (_bert_stack_tables(mod, st) if (LAYER_SEQ and STACK_SEQ and keep and x32.is_cuda and Mq < STACK_MAX_ROWS and len(mod.layers)
                                             and not ops.kernel_timer.names) else None)
        if T is not None:
            x32 = _bert_stack_fwd(st, T, x32, x16, key_add, cfg, p_hid, p_att, seed, sites)
        for layer in (mod.layers if T is None else ()):
            L, x32, x16 = _bert_layer_fwd(st, x32, x16, x16, key_add, bert_weights(layer), cfg.heads, p_hid, p_att, cfg.ln_eps, seed, sites, True)
            if keep:
                st.layers.append(L)
        if keep:
            ctx.st, ctx.mod = st, mod
        return x32.view(B, Lq, D) if vl is None else x32

    @staticmethod
    def backward(ctx, dout):
        st, mod = ctx.st, ctx.mod
        if getattr(st, "consumed", False):
            raise ops.MMDTIError("RobertaEncoderFn: backward through the same forward twice (retain_graph) is not supported")
        st.consumed = True
        B, Lq, D = st.B, st.Lq, st.D
        dx = dout.contiguous().view(st.Mq, D)
        if st.stack is not None:
            dx = _bert_stack_bwd(st, dx)
            notify_grads_ready(st.stack.T.params)
            st.stack = None
        for layer, L in zip(reversed(list(mod.layers)), reversed(st.layers)):
            dx, _ = _bert_layer_bwd(st, L, dx, st.seed)
            L.__dict__.clear()
            notify_grads_ready(layer.parameters())
        de = ops.layernorm_bwd(dx, st.e.view(st.Mq, D), mod.emb_ln_w, st.em, st.er, gbuf(mod.emb_ln_w), gbuf(mod.emb_ln_b),
                               drop_p=st.p_hid, seed=st.seed, site=st.site_emb)
        cfg = mod.cfg
        de16 = ops.cast_bf16(de)
        for ids, w, pad in ((st.ids, mod.word, cfg.pad_idx), (st.pos, mod.position, cfg.pad_idx), (None, mod.token_type, -1)):
            g = gbuf(w)
            if g is not None:
                if ids is None:        # token type 0 everywhere: a one-row table is a column sum (ids unused), else explicit zeros
                    ids = st.ids if w.shape[0] == 1 else torch.zeros_like(st.ids)
                ops.embedding_bwd_gemm(ids, de16, g, pad)
        notify_grads_ready(mod.embeddings.parameters())
        _join_stream_after_backward()   # (MM_Model runs this tower on a side stream)
        return None, None, None, None, None, None


class CrossLayerFn(torch.autograd.Function):
    """One BertCrossEncoder layer (mm_module.py:663-677, :615-626): s1 attends to s2 under an additive key mask."""

    @staticmethod
    def forward(ctx, s1, s2, key_add, layer, cfg, training, packs=None):
        """packs = (PackedRows of s1, PackedRows of s2): s1 [M1, D] / s2 [M2, D] are packed rows, key_add is None -- the keys of a
        sequence are the REAL rows of s2 (the reference adds -10000 to padded keys, mm_model.py:392-393: probability exactly 0)."""
        seed = dropout_state.next_seed()
        sites = _Sites()
        p_hid = cfg.hidden_dropout if training else 0.0
        p_att = cfg.attn_dropout if training else 0.0
        if packs is not None:
            vl = ops.AttnVarlen(packs[0], packs[1])
            D = s1.shape[-1]
            st = SimpleNamespace(B=packs[0].B, Lq=vl.Lq, Lk=vl.Lk, D=D, seed=seed, Mq=vl.q_rows, Mk=vl.k_rows, vl=vl)
        else:
            B, Lq, D = s1.shape
            Lk = s2.shape[1]
            st = SimpleNamespace(B=B, Lq=Lq, Lk=Lk, D=D, seed=seed, Mq=B * Lq, Mk=B * Lk, vl=None)
        s1c = s1.contiguous().view(st.Mq, D)
        s1_16 = ops.cast_act16(s1c)
        s2_16 = ops.cast_act16(s2.contiguous().view(st.Mk, D))
        L, out32, _ = _bert_layer_fwd(st, s1c, s1_16, s2_16, None if key_add is None else key_add.contiguous(), bert_weights(layer), cfg.heads,
                                      p_hid, p_att, cfg.ln_eps, seed, sites, False)
        if any(ctx.needs_input_grad):
            ctx.st, ctx.L, ctx.layer = st, L, layer
        ctx.shapes = (s1.shape, s2.shape)
        return out32.view(s1.shape)

    @staticmethod
    def backward(ctx, dout):
        st, L = ctx.st, ctx.L
        ds1, ds2 = _bert_layer_bwd(st, L, dout.contiguous().view(st.Mq, st.D), st.seed)
        L.__dict__.clear()
        notify_grads_ready(ctx.layer.parameters())
        return ds1.view(ctx.shapes[0]), ds2.view(ctx.shapes[1]), None, None, None, None, None


class DropoutFn(torch.autograd.Function):
    """F.dropout on fp32 activations (mm_model.py:390-391, 79, 82)."""

    @staticmethod
    def forward(ctx, x, p, training):
        if not training or p == 0.0:
            ctx.p = 0.0
            return x
        ctx.p, ctx.seed = p, dropout_state.next_seed()
        return ops.dropout_f32(x.contiguous(), p, ctx.seed, 1)

    @staticmethod
    def backward(ctx, d):
        if ctx.p == 0.0:
            return d, None, None
        return ops.dropout_f32(d.contiguous(), ctx.p, ctx.seed, 1), None, None


# ------------------------------------------------------------------------------------------------- InfoNCE
def _pad8(n):
    return (n + 7) // 8 * 8


class InfoNCEFn(torch.autograd.Function):
    """InfoNCE.forward (models/infonce.py:23-38) + info_nce (:42-98) with implicit negatives.

    Under data parallelism `gather` is a callable [B_loc,2*d] -> [B_glob,2*d] (all-gather in rank order) and
    `reduce_scatter` its adjoint; anchors of this rank are rows [row0,row0+B_loc) of the global batch and the value
    returned is this rank's share of the global loss (the sum over ranks equals the single-process loss)."""

    @staticmethod
    def forward(ctx, query, positive, mod, training, gather, reduce_scatter, row0, packs=None):
        """packs = (PackedRows of query, PackedRows of positive): [M, D] packed rows; the unmasked mean over the padded positions
        (infonce.py:32-33) weights each representative pad row by the number of padded positions it stands for."""
        if packs is not None:
            B, D = packs[0].B, query.shape[-1]
            Nq, Np = packs[0].M, packs[1].M           # (rows of the whole side, not per sequence)
            if not (POOL_THEN_PROJECT and mod.info_proj_query[0].weight.shape[0] % 8 == 0 and mod.info_proj_positive[0].weight.shape[0] % 8 == 0):
                raise ops.MMDTIError("InfoNCEFn: packed rows need the pool-then-project form (hidden width % 8 == 0)")
        else:
            B, Nq, D = query.shape
            Np = positive.shape[1]
        d = mod.d_l
        ldp = _pad8(d)
        seed = dropout_state.next_seed()
        p = mod.embed_dropout if training else 0.0
        st = SimpleNamespace(B=B, Nq=Nq, Np=Np, D=D, d=d, ldp=ldp, seed=seed, p=p)
        dev = query.device

        def proj(x, seq, n, dropout_p, site, pack=None):
            L = SimpleNamespace()
            rows = B * n if pack is None else pack.M
            L.x16 = ops.cast_act16(x.contiguous().view(rows, D), dropout_p, seed, site)
            L.u = torch.empty(rows, seq[0].weight.shape[0], device=dev, dtype=BF16)
            # (h is pooled, not multiplied: bf16 in every mode)
            h = ops.linear_fwd(L.x16, wfwd(seq[0].weight), seq[0].bias, act=ops.ACT_GELU_FWD, aux_out=L.u, out_dtype=BF16)
            Hd = h.shape[1]
            if pack is not None:
                L.hbar = ops.seq_mean_packed_fwd(h, pack, Hd, Hd)
                L.mean = ops.linear_f32_fwd(L.hbar, seq[2].weight.detach(), seq[2].bias.detach())
                L.h = None
                return L
            if POOL_THEN_PROJECT and Hd % 8 == 0:
                # mean_t(W2 h_t + b2) == W2 mean_t(h_t) + b2: pool the GELU outputs (fp32), then ONE [B, Hd] x [d, Hd] fp32 linear
                # -- instead of a 50-wide bf16 GEMM over every token, its three backward GEMMs and a bf16 round of the
                # per-token projections.  (The unmasked mean over all positions is the reference's, infonce.py:32-33.)
                L.hbar = ops.seq_mean_fwd(h, B, n, Hd, Hd)
                L.mean = ops.linear_f32_fwd(L.hbar, seq[2].weight.detach(), seq[2].bias.detach())
                L.h = None
                return L
            L.h = h
            pr = torch.zeros(B * n, ldp, device=dev, dtype=BF16)
            ops.gemm(L.h, wbf16(seq[2].weight), M=B * n, N=d, K=L.h.shape[1], lda=L.h.shape[1], ldb=seq[2].weight.shape[1], out=pr, ldc=ldp,
                     bias=seq[2].bias)
            L.mean = ops.seq_mean_fwd(pr, B, n, d, ldp)
            return L

        st.packs = packs
        st.Lq = proj(query, mod.info_proj_query, Nq, p, 1, None if packs is None else packs[0])
        st.Lp = proj(positive, mod.info_proj_positive, Np, 0.0, 2, None if packs is None else packs[1])
        both = torch.cat((st.Lq.mean, st.Lp.mean), dim=1)            # [B, 2d] (tiny; one message under DDP)
        both_all = gather(both) if gather is not None else both
        Bg = both_all.shape[0]
        qh, st.qinv = ops.l2norm_fwd(both_all[:, :d])
        kh, st.kinv = ops.l2norm_fwd(both_all[:, d:])
        loss = torch.zeros(1, device=dev, dtype=F32)
        dqh, dkh = torch.zeros_like(qh), torch.zeros_like(kh)
        T = mod.temperature
        ops.infonce_dir(qh, kh, row0, B, T, loss, dqh, dkh)
        ops.infonce_dir(kh, qh, row0, B, T, loss, dkh, dqh)
        st.qh, st.kh, st.dqh, st.dkh, st.Bg, st.row0 = qh, kh, dqh, dkh, Bg, row0
        ctx.st, ctx.mod, ctx.rs = st, mod, reduce_scatter
        return (loss / (2.0 * Bg)).view(())

    @staticmethod
    def backward(ctx, dloss):
        st, mod = ctx.st, ctx.mod
        B, d, ldp = st.B, st.d, st.ldp
        dq = ops.l2norm_bwd(st.dqh, st.qh, st.qinv)
        dk = ops.l2norm_bwd(st.dkh, st.kh, st.kinv)
        dboth = torch.cat((dq, dk), dim=1)
        if ctx.rs is not None:
            dboth = ctx.rs(dboth)                                   # sum over ranks, keep own rows
        dboth = (dboth * dloss).contiguous()

        def proj_bwd(L, dmean, seq, n, dropout_p, site, want_dx, pack=None):
            if L.h is None:                                                                  # pooled first (see forward)
                gw, gb = gbuf(seq[2].weight), gbuf(seq[2].bias)
                dhbar = ops.linear_f32_bwd(L.hbar, seq[2].weight.detach(), None, dmean.contiguous(), gw, gb)
                Hd = L.hbar.shape[1]
                if pack is not None:
                    du = ops.seq_mean_packed_bwd(dhbar, pack, Hd, Hd, aux=L.u, aux_mode=1 if ops.GELU_SAVE_GRAD else 2)
                else:
                    du = ops.seq_mean_bwd(dhbar, B, n, Hd, Hd, aux=L.u, aux_mode=1 if ops.GELU_SAVE_GRAD else 2)
                _lin_bwd_params(du, L.x16, seq[0].weight, seq[0].bias)
                if not want_dx:
                    return None
                dx = ops.linear_bwd_input(du, wbf16(seq[0].weight), out_dtype=F32)
                if dropout_p > 0:
                    dx = ops.dropout_f32(dx, dropout_p, st.seed, site)
                return dx
            dpr = ops.seq_mean_bwd(dmean.contiguous(), B, n, d, ldp)                       # [B*n, ldp] bf16, pad cols 0
            gw = gbuf(seq[2].weight)
            if gw is not None:
                ops.linear_bwd_weight(dpr, L.h, gw)
            gb = gbuf(seq[2].bias)
            if gb is not None:
                ops.colsum(dpr, gb, cols=d)
            du = ops.gemm(dpr, wbf16(seq[2].weight), M=B * n, N=L.h.shape[1], K=d, lda=ldp, ldb=seq[2].weight.shape[1], transB=True,
                          act=ops.ACT_GELU_DX, aux_in=L.u)
            _lin_bwd_params(du, L.x16, seq[0].weight, seq[0].bias)
            if not want_dx:
                return None
            dx = ops.linear_bwd_input(du, wbf16(seq[0].weight), out_dtype=F32)
            if dropout_p > 0:
                dx = ops.dropout_f32(dx, dropout_p, st.seed, site)
            return dx

        pk = st.packs
        dxq = proj_bwd(st.Lq, dboth[:, :d], mod.info_proj_query, st.Nq, st.p, 1, ctx.needs_input_grad[0], None if pk is None else pk[0])
        dxp = proj_bwd(st.Lp, dboth[:, d:], mod.info_proj_positive, st.Np, 0.0, 2, ctx.needs_input_grad[1], None if pk is None else pk[1])
        notify_grads_ready(mod.parameters())
        if pk is not None:
            return dxq, dxp, None, None, None, None, None, None
        return (None if dxq is None else dxq.view(B, st.Nq, st.D), None if dxp is None else dxp.view(B, st.Np, st.D), None, None, None, None, None, None)


class InfoNCELossFn(torch.autograd.Function):
    """info_nce(query, positive_key) on already-pooled [B,d] embeddings (models/infonce.py:42-98, implicit negatives)."""

    @staticmethod
    def forward(ctx, q, k, temperature):
        B, d = q.shape
        qh, qinv = ops.l2norm_fwd(q.contiguous())
        kh, kinv = ops.l2norm_fwd(k.contiguous())
        loss = torch.zeros(1, device=q.device, dtype=F32)
        dqh, dkh = torch.zeros_like(qh), torch.zeros_like(kh)
        ops.infonce_dir(qh, kh, 0, B, temperature, loss, dqh, dkh)
        ops.infonce_dir(kh, qh, 0, B, temperature, loss, dkh, dqh)
        ctx.sv = (qh, qinv, kh, kinv, dqh, dkh)
        return (loss / (2.0 * B)).view(())

    @staticmethod
    def backward(ctx, dl):
        qh, qinv, kh, kinv, dqh, dkh = ctx.sv
        return ops.l2norm_bwd(dqh, qh, qinv) * dl, ops.l2norm_bwd(dkh, kh, kinv) * dl, None


# ------------------------------------------------------------------------------------------------- ConR / SupCon
class CTLossFn(torch.autograd.Function):
    """CT_Regress / CT_Single / CT_Multi (models/contrastive.py).  Only `feature` receives a gradient (the masks built
    from labels / predictions are not differentiable in the reference either)."""

    @staticmethod
    def forward(ctx, feature, mode, labels_f, labels_i, pred, weights, w, t, e, coef):
        f = feature.reshape(feature.shape[0], -1).contiguous()
        fh, inv = ops.l2norm_fwd(f)
        loss, G = ops.ct_loss_fwd(mode, fh, labels_f=labels_f, labels_i=labels_i, pred=pred, weights=weights, w=w, t=t, e=e, coef=coef)
        ctx.sv, ctx.t, ctx.shape = (fh, inv, G), t, feature.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, dl):
        fh, inv, G = ctx.sv
        df = ops.l2norm_bwd(ops.ct_loss_bwd(fh, G, ctx.t), fh, inv)
        return (df * dl).view(ctx.shape), None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------- FDS smooth / pool / head / losses
class FDSSmoothFn(torch.autograd.Function):
    """FDS.smooth (models/fds.py:157-190).  Out-of-place; the module wrapper copies back to keep the reference's
    in-place aliasing visible to the caller."""

    @staticmethod
    def forward(ctx, x, bins, flags, bs, bn, m1, v1, m2, v2):
        y, sc = ops.fds_smooth(x.contiguous(), bins, flags, bs, bn, m1, v1, m2, v2, want_scale=True)
        ctx.sc = sc
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy * ctx.sc, None, None, None, None, None, None, None, None


class MaskedPoolFn(torch.autograd.Function):
    """mm_model.py:572-576: zero padded rows, concat, sum / (#atom tokens + #SMILES tokens)."""

    @staticmethod
    def forward(ctx, a, t, mask_a, mask_t, packs=None):
        if packs is not None:          # packed rows: a [M1, D], t [M2, D]; the masks are implied by the packings
            ctx.packs = packs
            return ops.masked_pool_packed_fwd(a.contiguous(), t.contiguous(), packs[0], packs[1])
        ctx.packs = None
        ma, mt = ops._u8(mask_a), ops._u8(mask_t)
        ctx.sv = (ma, mt, a.shape[1], t.shape[1])
        return ops.masked_pool_fwd(a.contiguous(), t.contiguous(), ma, mt)

    @staticmethod
    def backward(ctx, dp):
        if ctx.packs is not None:
            da, dt = ops.masked_pool_packed_bwd(dp.contiguous(), ctx.packs[0], ctx.packs[1])
            return da, dt, None, None, None
        ma, mt, Na, Nt = ctx.sv
        da, dt = ops.masked_pool_bwd(dp.contiguous(), ma, mt, Na, Nt)
        return da, dt, None, None, None


class LinearF32Fn(torch.autograd.Function):
    """Small fp32 Linear (+tanh) for the classification head (mm_model.py:44-84)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        y = ops.linear_f32_fwd(x, weight, bias, act)
        # y is this node's OUTPUT: it must go through save_for_backward -- kept as a plain attribute it closes a reference
        # cycle (node -> y -> grad_fn -> node) that keeps the whole step's graph alive until the cyclic GC runs
        ctx.save_for_backward(y)
        ctx.sv = (x, weight, bias, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, act = ctx.sv
        (y,) = ctx.saved_tensors
        dx = ops.linear_f32_bwd(x, weight, y, dy.contiguous(), gbuf(weight), None if bias is None else gbuf(bias), act, want_dx=ctx.needs_input_grad[0])
        notify_grads_ready([weight] if bias is None else [weight, bias])
        return dx, None, None, None


class MSELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        loss, d = ops.mse_loss(pred.contiguous(), target.contiguous().to(F32))
        ctx.d = d
        return loss.view(())

    @staticmethod
    def backward(ctx, dl):
        return (ctx.d * dl).view_as(ctx.d), None


class CELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        loss, d = ops.ce_loss(logits.contiguous(), target.contiguous().view(-1).long())
        ctx.d = d
        return loss.view(())

    @staticmethod
    def backward(ctx, dl):
        return ctx.d * dl, None


class BCELogitsLossFn(torch.autograd.Function):
    """nn.BCEWithLogitsLoss() on [B, C] logits (models/nnmodel.py:28-29, multilabel_classification 'bce'); the reference hands the
    targets over as int64 (tasks/trainer.py:119), the kernel reads them as fp32."""

    @staticmethod
    def forward(ctx, logits, target):
        loss, d = ops.bce_logits_loss(logits.contiguous(), target.contiguous().to(F32))
        ctx.d = d
        return loss.view(())

    @staticmethod
    def backward(ctx, dl):
        return ctx.d * dl, None
