"""Tensor-level wrappers over the C ABI (include/mmdti_hip.h).

PyTorch is used here only as plumbing: device memory (torch.empty), the current HIP stream, and dtype/shape
bookkeeping.  Every function launches hand-written gfx950 kernels through ctypes; nothing falls back to ATen math.
"""
from __future__ import annotations

import os

import torch

from ._abi import lib, MMDTIError

BF16 = torch.bfloat16
F32 = torch.float32
F16 = torch.float16

ACT_NONE, ACT_GELU, ACT_GELU_BWD, ACT_TANH, ACT_GELU_G, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5
# forward GELU that saves gelu'(u) instead of u, backward = one multiply (MMDTI_GELU_SAVE_GRAD=0: save u, evaluate gelu' in the backward)
GELU_SAVE_GRAD = os.environ.get("MMDTI_GELU_SAVE_GRAD", "1") != "0"
# pair bias: the forward saves nothing and ONE kernel does the whole backward (0: the round-1 chain -- fused per-pair half,
# three saved [P,128] tensors and two weight-gradient GEMMs)
GBF_FULL_BWD = os.environ.get("MMDTI_GBF_FULL_BWD", "1") != "0"
ACT_GELU_FWD = ACT_GELU_G if GELU_SAVE_GRAD else ACT_GELU
ACT_GELU_DX = ACT_MUL_AUX if GELU_SAVE_GRAD else ACT_GELU_BWD
DT_F32, DT_BF16, DT_F32_ATOMIC, DT_F16, DT_AB_F16, DT_B_F16 = 0, 1, 2, 3, 16, 32

# fp16 forward operands (the DEFAULT since round 4; MMDTI_FWD_FP16=0 / set_forward_fp16(False) is the A/B switch back to bf16): every
# 16-bit tensor that feeds a FORWARD GEMM -- weights, LayerNorm / GELU / attention outputs, tower 1's q | k | v -- is fp16 instead of
# bf16 (the reference's own AMP dtype, tasks/trainer.py:181-182; v_mfma_f32_16x16x32_f16 runs at the bf16 rate).  Three more mantissa
# bits put encoder_rep / out_bert / logits within the north star's 1e-3 of the fp32 reference (6.0e-4 / 4.8e-4 / 7e-4 measured;
# profiles/r03_rounding_sites_fp16.json) where bf16 operands cannot (4.6e-3).  The backward keeps bf16 operands (gradients need bf16's
# range: the reference needs a GradScaler for the same reason); a saved fp16 activation is converted INSIDE the kernel that reads it --
# the weight-gradient GEMMs between LDS and the matrix pipe (MMDTI_DT_B_F16), the pair-attention backward on its way into LDS -- so the
# mode costs no pass over HBM.  Every fp16 store saturates at +-65504 (common.h f2h_sat).
FWD_F16 = os.environ.get("MMDTI_FWD_FP16", "1") != "0"


def set_forward_fp16(on: bool):
    global FWD_F16
    FWD_F16 = bool(on)


def act16():
    """dtype of the 16-bit activations that feed forward GEMMs"""
    return F16 if FWD_F16 else BF16


def _chk16(t, name, contiguous=True):
    if t.dtype not in (BF16, F16):
        raise MMDTIError(f"{name}: expected a bf16 (or, in the fp16 forward-operand mode, fp16) tensor, got {t.dtype}")
    return _chk(t, t.dtype, name, contiguous)
CT_REGRESS, CT_SINGLE, CT_MULTI = 0, 1, 2


_raw_stream, _raw_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _stream():
    # (torch.cuda.current_stream().cuda_stream costs ~9 us of Python per call -- a third of a small-batch step's host time)
    return _raw_stream(_raw_device())


def _p(t):
    return 0 if t is None else t.data_ptr()


def upload(t, device):
    """A small host tensor (per-batch descriptors: key-tile counts, tile prefixes, packed-row offsets) to the device WITHOUT stalling the
    host: through pinned memory (torch's caching host allocator) the copy is a true asynchronous enqueue; from pageable memory
    hipMemcpyAsync returns only once the copy has run -- i.e. after everything already queued on the stream -- which serialised the host
    with the previous step's kernels whenever a step was fed a fresh batch (found with bench.py's pipeline workload: 88 ms per step
    where the same step on a resident batch takes 31)."""
    if t.device.type != "cpu" or torch.device(device).type == "cpu":
        return t.to(device, non_blocking=True)
    return t.pin_memory().to(device, non_blocking=True)


def _chk(t, dtype, name, contiguous=True):
    if not t.is_cuda:
        raise MMDTIError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise MMDTIError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise MMDTIError(f"{name}: expected a contiguous tensor")
    return t


def _u8(mask):
    """bool mask -> uint8 view (no copy)."""
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return mask.contiguous().to(torch.uint8)


# --------------------------------------------------------------------------------------------- GEMM
def gemm(A, B, *, M, N, K, lda, ldb, out=None, ldc=None, transA=False, transB=False, batch=(1, 1),
         sA=(0, 0), sB=(0, 0), sC=(0, 0), splitk=1, alpha=1.0, beta=0.0, bias=None, residual=None, act=ACT_NONE,
         aux_in=None, aux_out=None, out_dtype=BF16, atomic=False, drop_p=0.0, seed=0, site=0, out_shape=None, colsum=None, arowsum=None,
         workspace=None):
    """C = epi(alpha * A.B^T); see mmdti_gemm_bf16.  A/B are bf16 tensors (any shape; lda/ldb given explicitly)."""
    _chk16(A, "gemm.A", contiguous=False)
    _chk16(B, "gemm.B", contiguous=False)
    # (one 16-bit type per product, except the weight gradient dW += dy^T.x of a saved fp16 activation x beside a bf16 dy: the kernel
    #  converts x between LDS and the matrix pipe -- MMDTI_DT_B_F16)
    b_cvt = A.dtype == BF16 and B.dtype == F16 and transA and transB and atomic and batch == (1, 1)
    if A.dtype != B.dtype and not b_cvt:
        raise MMDTIError(f"gemm: A is {A.dtype} but B is {B.dtype} (both operands of a product share one 16-bit type)")
    ldc = N if ldc is None else ldc
    if out is None:
        shape = out_shape if out_shape is not None else ((M, N) if batch == (1, 1) else (batch[0], batch[1], M, N))
        out = torch.empty(shape, device=A.device, dtype=out_dtype)
    c_dtype = DT_F32_ATOMIC if atomic else (DT_BF16 if out.dtype == BF16 else (DT_F16 if out.dtype == F16 else DT_F32))
    if b_cvt:
        c_dtype |= DT_B_F16
    elif A.dtype == F16:
        c_dtype |= DT_AB_F16
    if atomic and out.dtype != F32:
        raise MMDTIError("gemm: atomic accumulation needs an fp32 output")
    t0 = kernel_timer.begin("gemm")
    lib().mmdti_gemm_bf16(_stream(), A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, lda, ldb, ldc,
                          int(transA), int(transB), batch[0], batch[1], sA[0], sA[1], sB[0], sB[1], sC[0], sC[1],
                          splitk, float(alpha), float(beta), _p(bias), _p(residual), ldc if residual is None else residual.stride(-2),
                          act, _p(aux_in), _p(aux_out), N if (aux_in is None and aux_out is None) else (aux_in if aux_in is not None else aux_out).stride(-2),
                          c_dtype, float(drop_p), int(seed), int(site), _p(colsum), _p(arowsum), _p(workspace),
                          0 if workspace is None else workspace.numel() * workspace.element_size())
    kernel_timer.end("gemm", t0, 2.0 * M * N * K * batch[0] * batch[1],
                     tag=(M, N, K, int(transA), int(transB), batch[0] * batch[1], splitk, act, out.dtype == BF16, aux_in is not None or aux_out is not None,
                          residual is not None))
    return out


def linear_fwd(x, w, bias=None, *, act=ACT_NONE, residual=None, out_dtype=None, aux_out=None, drop_p=0.0, seed=0, site=0):
    """y[M,N] = epi(x[M,K] . w[N,K]^T + bias).  out_dtype None: the 16-bit type of x (bf16, or fp16 in the fp16 forward-operand
    mode) -- the output of a forward Linear that feeds the next one."""
    if out_dtype is None:
        out_dtype = x.dtype
    M, K = x.shape[0], x.shape[-1]
    N = w.shape[0]
    return gemm(x, w, M=M, N=N, K=K, lda=x.stride(0), ldb=w.stride(0), bias=bias, act=act, residual=residual,
                out_dtype=out_dtype, aux_out=aux_out, drop_p=drop_p, seed=seed, site=site)


GEMM_LN = os.environ.get("MMDTI_GEMM_LN", "1") != "0"       # 0: the Linear and the LayerNorm that follows it as two kernels
# Longest contraction the fused kernel takes.  Measured on MI355X (scratch/gemm_ln_bench.py, profiles/r03_gemm_ln_ab.json): at K = 512
# (out_proj, attention.output.dense) one kernel beats the two it replaces at every row count -- 50 vs 58 us at 33 280 rows, 135 vs 155
# at 65 536, 20 vs 28 at 1 600 --, at K = 2048 (fc2, output.dense) it loses -- 147 vs 127, 262 vs 220: with whole 512-column rows per
# workgroup only two workgroups fit a CU, and their K loop runs ~30 % behind the 128 x 128 kernel's four.
GEMM_LN_MAX_K = int(os.environ.get("MMDTI_GEMM_LN_MAX_K", "1024"))


def linear_ln_eligible(x, w, residual=None):
    return (GEMM_LN and w.shape[0] == 512 and x.shape[-1] % 64 == 0 and x.shape[-1] <= GEMM_LN_MAX_K and x.stride(-1) == 1 and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and x.shape[0] * x.stride(0) * 2 < 0x7fffffff
            and (residual is None or (residual.stride(-1) == 1 and residual.stride(0) % 4 == 0 and residual.data_ptr() % 16 == 0)))


def linear_ln_fwd(x, w, bias, gamma, beta, eps, *, residual=None, drop_p=0.0, seed=0, site=0, want_f32=False, want_bf16=True):
    """The Linear that closes a residual branch + the LayerNorm behind it (mmdti_gemm_ln_bf16):
    y = residual + dropout(x.w^T + bias) (fp32);  h = LN(y) -> (y, h32 | None, h16 | None, mean, rstd).  One kernel when the output
    is 512 wide (the reference architecture's width), else the GEMM and the LayerNorm kernels back to back -- same results up to
    the order of the row sums."""
    if not linear_ln_eligible(x, w, residual):
        y = linear_fwd(x, w, bias, residual=residual, out_dtype=F32, drop_p=drop_p, seed=seed, site=site)
        h32, h16, mean, rstd = layernorm_fwd(y, gamma, beta, eps, want_f32=want_f32, want_bf16=want_bf16)
        return y, h32, h16, mean, rstd
    _chk16(x, "linear_ln.x", contiguous=False); _chk16(w, "linear_ln.w", contiguous=False)
    if x.dtype != w.dtype:
        raise MMDTIError("linear_ln_fwd: x and w must share one 16-bit type")
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=F32)
    h32 = torch.empty(M, N, device=x.device, dtype=F32) if want_f32 else None
    h16 = torch.empty(M, N, device=x.device, dtype=act16()) if want_bf16 else None
    mean = torch.empty(M, device=x.device, dtype=F32)
    rstd = torch.empty(M, device=x.device, dtype=F32)
    t0 = kernel_timer.begin("gemm")
    lib().mmdti_gemm_ln_bf16(_stream(), x.data_ptr(), w.data_ptr(), _p(bias), _p(residual), M, N, K, x.stride(0), w.stride(0),
                             N if residual is None else residual.stride(0), float(drop_p), int(seed), int(site), y.data_ptr(), gamma.data_ptr(),
                             beta.data_ptr(), float(eps), _p(h32), _p(h16), mean.data_ptr(), rstd.data_ptr(),
                             (1 if x.dtype == F16 else 0) | (2 if (h16 is not None and h16.dtype == F16) else 0))
    kernel_timer.end("gemm", t0, 2.0 * M * N * K, tag=(M, N, K, 0, 0, 1, 1, "ln", want_bf16, want_f32, residual is not None))
    return y, h32, h16, mean, rstd


def linear_bwd_input(dy, w, *, act=ACT_NONE, aux_in=None, out_dtype=BF16, K_valid=None, colsum=None):
    """dx[M,K] = dy[M,N] . w[N,K]  (optionally * gelu'(aux_in)).  colsum: [K] fp32 buffer that receives += column sums of dx
    (the bias gradient of the Linear that produced this layer's input, when dx is that Linear's output gradient)."""
    M, N = dy.shape
    K = w.shape[1]
    return gemm(dy, w, M=M, N=K, K=(N if K_valid is None else K_valid), lda=dy.stride(0), ldb=w.stride(0), transB=True, act=act, aux_in=aux_in,
                out_dtype=out_dtype, colsum=colsum)


def _splitk_for(M, N, K):
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    ktiles = (K + 63) // 64
    # measured on MI355X (scratch/dw_sweep.py): about one workgroup per CU for few-tile outputs, 8 splits from 48 tiles up
    # -- every extra split is another atomic pass over the fp32 output
    sk = max(256 // tiles, min(8, max(1, 2048 // tiles)))
    return max(1, min(ktiles, sk))


DW_BIAS = os.environ.get("MMDTI_DW_BIAS", "1") != "0"
SPLITK_SLABS = os.environ.get("MMDTI_SPLITK_SLABS", "1") != "0"


def linear_bwd_weight(dy, x, dw, *, rows=None, db=None):
    """dw[N,K] += dy[M,N]^T . x[M,K]   (fp32 atomic accumulate into the gradient arena);  db[N] += column sums of dy (the
    bias gradient, taken inside the same pass over dy)."""
    M = dy.shape[0] if rows is None else rows
    N, K = dw.shape
    if x.dtype == F16 and (x.dim() != 2 or x.stride(1) != 1):
        x = x.contiguous().view(-1, x.shape[-1])        # (an fp16 x is converted inside the GEMM: see gemm())
    if db is not None:
        _chk(db, F32, "linear_bwd_weight.db")
        if not DW_BIAS:                                  # MMDTI_DW_BIAS=0: the separate column-sum pass (A/B switch)
            colsum(dy, db, cols=N)
            db = None
    # split-K scratch for the large-tile kernel (one fp32 partial per split, summed by a second pass: no atomics); from the
    # caching allocator, so it is tied to the launch stream like any other temporary
    ws = None
    if SPLITK_SLABS and N % 256 == 0 and K % 256 == 0 and M >= 4096:
        tiles = (N // 256) * (K // 256)
        ws = torch.empty(max(1, min(256 // tiles, M // 256)) * N * K, device=dy.device, dtype=F32)
    gemm(dy, x, M=N, N=K, K=M, lda=dy.stride(0), ldb=x.stride(0), transA=True, transB=True, out=dw, ldc=dw.stride(0),
         atomic=True, splitk=_splitk_for(N, K, M), arowsum=db, workspace=ws)
    return dw


GROUPED_DW = os.environ.get("MMDTI_GROUPED_DW", "1") != "0"


# (any row count: the kernel zero-fills the tail of the last 64-row K-tile)
GROUPED_DW_MIN_ROWS = int(os.environ.get("MMDTI_GROUPED_DW_MIN_ROWS", "128"))    # (64 x 64-tile kernel below 4096 rows: any count from 64 up)


def _dw_groupable(dy, x, dw, rows):
    return (dy.dtype == BF16 and x.dtype in (BF16, F16) and dw.dtype == F32 and dy.stride(-1) == 1 and x.stride(-1) == 1 and dw.stride(-1) == 1 and
            dw.shape[0] % 256 == 0 and dw.shape[1] % 256 == 0 and rows >= GROUPED_DW_MIN_ROWS and dy.stride(0) % 8 == 0 and
            x.stride(0) % 8 == 0 and dw.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and dw.data_ptr() % 16 == 0)


def linear_bwd_weight_grouped(items):
    """items: [(dy [rows, N_out] bf16, x [rows, N_in] bf16, dw [N_out, N_in] fp32, db [N_out] fp32 | None, rows | None)] -- the
    weight (and bias) gradients of several Linears over the same token rows in ONE launch (mmdti_linear_dw_grouped: the K
    split that fills the chip is shared by all of them, so the fp32 partial-sum traffic is a quarter of what the GEMMs need
    one at a time).  Items that do not fit the grouped kernel's shape rules run through linear_bwd_weight."""
    import ctypes
    # (an fp16 x -- a saved forward activation of the fp16 forward-operand mode -- is converted to bf16 inside the kernel, between LDS
    #  and the matrix pipe; a launch takes one operand type, so the groups are keyed by it as well)
    norm = [(dy, x, dw, db, dy.shape[0] if rows is None else rows) for dy, x, dw, db, rows in items]
    groups = {}
    for it in norm:
        if GROUPED_DW and _dw_groupable(it[0], it[1], it[2], it[4]):
            groups.setdefault((it[4], it[1].dtype), []).append(it)
        else:
            linear_bwd_weight(it[0], it[1], it[2], rows=it[4], db=it[3])
    for (rows, xdt), grp in groups.items():
        while grp:
            chunk, grp = grp[:8], grp[8:]
            if len(chunk) == 1:
                it = chunk[0]
                linear_bwd_weight(it[0], it[1], it[2], rows=rows, db=it[3])
                continue
            n = len(chunk)
            tiles = sum((it[2].shape[0] // 256) * (it[2].shape[1] // 256) for it in chunk)
            sk = lib()._dll.mmdti_linear_dw_grouped_splits(tiles, rows)
            ws = torch.empty(sk * sum(it[2].shape[0] * it[2].shape[1] for it in chunk), device=chunk[0][0].device, dtype=F32)
            vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
            t0 = kernel_timer.begin("gemm")
            lib().mmdti_linear_dw_grouped(_stream(), n, vp(*[it[0].data_ptr() for it in chunk]), vp(*[it[1].data_ptr() for it in chunk]),
                                          vp(*[it[2].data_ptr() for it in chunk]), vp(*[_p(it[3]) for it in chunk]),
                                          ip(*[it[2].shape[0] for it in chunk]), ip(*[it[2].shape[1] for it in chunk]),
                                          ip(*[it[0].stride(0) for it in chunk]), ip(*[it[1].stride(0) for it in chunk]),
                                          ip(*[it[2].stride(0) for it in chunk]), rows, ws.data_ptr(), ws.numel() * 4, int(xdt == F16))
            work = sum(2.0 * it[2].shape[0] * it[2].shape[1] * rows for it in chunk)
            kernel_timer.end("gemm", t0, work, tag=("grouped_dw", tuple((it[2].shape[0], it[2].shape[1]) for it in chunk), rows))


def colsum(x, out, cols=None):
    _chk(x, BF16, "colsum.x", contiguous=False)
    _chk(out, F32, "colsum.out")
    lib().mmdti_colsum_bf16(_stream(), x.data_ptr(), x.shape[0], x.shape[1] if cols is None else cols, x.stride(0), out.data_ptr())
    return out


# --------------------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, eps, *, want_f32=False, want_bf16=True, row_zero=None, drop_p=0.0, seed=0, site=0):
    _chk(x, F32, "layernorm.x")
    D = x.shape[-1]
    rows = x.numel() // D
    y32 = torch.empty_like(x) if want_f32 else None
    y16 = torch.empty(x.shape, device=x.device, dtype=act16()) if want_bf16 else None      # (feeds a forward GEMM)
    mean = torch.empty(rows, device=x.device, dtype=F32)
    rstd = torch.empty(rows, device=x.device, dtype=F32)
    rz = _u8(row_zero)
    t0 = kernel_timer.begin("ln_fwd")
    lib().mmdti_layernorm_fwd(_stream(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), rows, D, _p(y32), _p(y16),
                              mean.data_ptr(), rstd.data_ptr(), _p(rz), float(drop_p), int(seed), int(site),
                              int(y16 is not None and y16.dtype == F16))
    kernel_timer.end("ln_fwd", t0, float(rows) * D * (4 + (4 if want_f32 else 0) + (2 if want_bf16 else 0)))
    return y32, y16, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, *, dres=None, dy_add=None, row_zero=None, drop_p=0.0, seed=0, site=0,
                  bf16_copy=None):
    """-> dx (fp32); with bf16_copy=(p, site[, colsum_out]) also the bf16 dropout-backward copy of dx for the next GEMM (and
    its column sums accumulated into colsum_out): (dx, dx16)."""
    D = x.shape[-1]
    rows = x.numel() // D
    dx = torch.empty_like(x)
    rz = _u8(row_zero)
    dx16 = torch.empty(x.shape, device=x.device, dtype=BF16) if bf16_copy is not None else None
    p2, site2, csum = (tuple(bf16_copy) + (None,))[:3] if bf16_copy is not None else (0.0, 0, None)
    t0 = kernel_timer.begin("ln_bwd")
    lib().mmdti_layernorm_bwd(_stream(), dy.data_ptr(), DT_BF16 if dy.dtype == BF16 else DT_F32, _p(dy_add), x.data_ptr(), gamma.data_ptr(),
                              mean.data_ptr(), rstd.data_ptr(), rows, D, _p(dres), dx.data_ptr(), _p(dgamma), _p(dbeta), _p(rz),
                              float(drop_p), int(seed), int(site), _p(dx16), float(p2), int(site2), _p(csum))
    kernel_timer.end("ln_bwd", t0, float(rows) * D * (dy.element_size() + 4 + 4 + (4 if dres is not None else 0) + (4 if dy_add is not None else 0) +
                                                     (2 if bf16_copy is not None else 0)))
    return dx if bf16_copy is None else (dx, dx16)


# --------------------------------------------------------------------------------------------- small utilities
def cast_bf16(x, drop_p=0.0, seed=0, site=0):
    _chk(x, F32, "cast_bf16.x")
    y = torch.empty(x.shape, device=x.device, dtype=BF16)
    lib().mmdti_cast_f32_bf16(_stream(), x.data_ptr(), y.data_ptr(), x.numel(), float(drop_p), int(seed), int(site))
    return y


def cast_act16(x, drop_p=0.0, seed=0, site=0):
    """fp32 -> the 16-bit type of forward GEMM inputs (bf16; fp16 in the fp16 forward-operand mode), with optional dropout"""
    if not FWD_F16:
        return cast_bf16(x, drop_p, seed, site)
    _chk(x, F32, "cast_act16.x")
    y = torch.empty(x.shape, device=x.device, dtype=F16)
    lib().mmdti_cast_f32_f16(_stream(), x.data_ptr(), y.data_ptr(), x.numel(), float(drop_p), int(seed), int(site))
    return y


def to_bf16(x):
    """A saved forward activation as the bf16 operand of a BACKWARD GEMM: itself, or (fp16 forward-operand mode) its conversion.
    x: [rows, cols] with unit column stride (row-strided views allowed), cols % 8 == 0."""
    if x.dtype != F16:
        return x
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.contiguous().view(-1, x.shape[-1])
    y = torch.empty(x.shape, device=x.device, dtype=BF16)
    lib().mmdti_cast_f16_bf16(_stream(), x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), y.data_ptr())
    return y


def cast_f32(x):
    _chk(x, BF16, "cast_f32.x")
    y = torch.empty(x.shape, device=x.device, dtype=F32)
    lib().mmdti_cast_bf16_f32(_stream(), x.data_ptr(), y.data_ptr(), x.numel())
    return y


def dropout_f32(x, p, seed, site):
    _chk(x, F32, "dropout_f32.x")
    if p == 0.0:
        return x
    y = torch.empty_like(x)
    lib().mmdti_dropout_f32(_stream(), x.data_ptr(), y.data_ptr(), x.numel(), float(p), int(seed), int(site))
    return y


def axpy_(y, x, a=1.0):
    lib().mmdti_axpy_f32(_stream(), x.data_ptr(), y.data_ptr(), x.numel(), float(a))
    return y


def embedding_fwd(ids, table, out=None, accumulate=False):
    _chk(ids, torch.int64, "embedding.ids")
    _chk(table, F32, "embedding.table")
    V, D = table.shape
    if out is None:
        out = torch.empty(*ids.shape, D, device=table.device, dtype=F32)
    lib().mmdti_embedding_fwd(_stream(), ids.data_ptr(), table.data_ptr(), ids.numel(), D, V, out.data_ptr(), int(accumulate))
    return out


def embedding_fwd3(ids_a, table_a, ids_b, table_b, table_c, ids_c=None):
    """table_a[ids_a] + table_b[ids_b] + table_c[ids_c or 0] in one pass (RoBERTa word + position + token-type embeddings)."""
    _chk(ids_a, torch.int64, "embedding.ids_a"); _chk(ids_b, torch.int64, "embedding.ids_b")
    for t in (table_a, table_b, table_c):
        _chk(t, F32, "embedding.table")
    D = table_a.shape[1]
    out = torch.empty(*ids_a.shape, D, device=table_a.device, dtype=F32)
    lib().mmdti_embedding_fwd3(_stream(), ids_a.data_ptr(), table_a.data_ptr(), table_a.shape[0], ids_b.data_ptr(), table_b.data_ptr(),
                               table_b.shape[0], _p(ids_c), table_c.data_ptr(), table_c.shape[0], ids_a.numel(), D, out.data_ptr())
    return out


def embedding_bwd(ids, dout, dtable, padding_idx=-1):
    V, D = dtable.shape
    lib().mmdti_embedding_bwd(_stream(), ids.data_ptr(), dout.data_ptr(), ids.numel(), D, V, int(padding_idx), dtable.data_ptr())
    return dtable


def embedding_bwd_gemm(ids, dout_bf16, dtable, padding_idx=-1):
    """dtable[V,D] += onehot(ids)^T . dout  as one split-K MFMA GEMM (V == 1: a column sum)."""
    V, D = dtable.shape
    n = ids.numel()
    _chk(dout_bf16, BF16, "embedding_bwd_gemm.dout")
    if V == 1:
        if padding_idx != 0:
            colsum(dout_bf16.view(n, D), dtable.view(-1))
        return dtable
    ldv = (V + 7) // 8 * 8
    oh = torch.empty(n, ldv, device=dtable.device, dtype=BF16)
    lib().mmdti_onehot_bf16(_stream(), ids.data_ptr(), n, V, ldv, int(padding_idx), oh.data_ptr())
    gemm(oh, dout_bf16.view(n, D), M=V, N=D, K=n, lda=ldv, ldb=D, transA=True, transB=True, out=dtable, ldc=D, atomic=True,
         splitk=_splitk_for(V, D, n))
    return dtable


def roberta_position_ids(ids, pad_idx):
    _chk(ids, torch.int64, "position_ids.ids")
    out = torch.empty_like(ids)
    lib().mmdti_roberta_position_ids(_stream(), ids.data_ptr(), ids.shape[0], ids.shape[1], int(pad_idx), out.data_ptr())
    return out


# --------------------------------------------------------------------------------------------- Gaussian basis / pair layout
EDGE_DTYPES = (torch.int64, torch.int32, torch.int16)


def _chk_edge(edge_type):
    if edge_type.dtype not in EDGE_DTYPES or not edge_type.is_contiguous():
        raise TypeError(f"gbf.edge_type must be a contiguous int64 / int32 / int16 tensor, got {edge_type.dtype}")


def gbf_features_fwd(dist, edge_type, mul, bias, means, stds):
    _chk(dist, F32, "gbf.dist"); _chk(edge_type, torch.int64, "gbf.edge_type")
    P, K, E = dist.numel(), means.numel(), mul.numel()
    feat = torch.empty(P, K, device=dist.device, dtype=BF16)
    t0 = kernel_timer.begin("gbf_features_fwd")
    lib().mmdti_gbf_features_fwd(_stream(), dist.data_ptr(), edge_type.data_ptr(), mul.data_ptr(), bias.data_ptr(), means.data_ptr(),
                                 stds.data_ptr(), P, K, E, feat.data_ptr())
    kernel_timer.end("gbf_features_fwd", t0)
    return feat


def gbf_features_bwd(dist, edge_type, mul, bias, means, stds, dfeat, dmul, dbias, dmeans, dstds):
    P, K, E = dist.numel(), means.numel(), mul.numel()
    lib().mmdti_gbf_features_bwd(_stream(), dist.data_ptr(), edge_type.data_ptr(), mul.data_ptr(), bias.data_ptr(), means.data_ptr(),
                                 stds.data_ptr(), P, K, E, dfeat.data_ptr(), dmul.data_ptr(), dbias.data_ptr(), dmeans.data_ptr(),
                                 dstds.data_ptr())


def gbf_tile_prefixes(key_tiles_host, N, device, rows_host=None):
    """Ragged batches: the two [B+1] int32 device arrays (forward, complete backward) that tell the fused pair-bias kernels how
    many of each molecule's tiles to visit -- the 4x4 pair blocks of its first k_b key tiles, k_b = the count the ragged
    pair-attention kernels cover (pair_key_tiles_effective).  key_tiles_host: [B] ints on the HOST (no device sync).
    rows_host ([B] ints on the host: packed token rows per molecule, packing.PackedRows.rows_host): the blocks stop at the molecule's
    representative pad row as well -- then (fwd prefix, bwd prefix, fwd row blocks [B], bwd row blocks [B]) come back."""
    # (both kernels enumerate the 4x4 blocks that hold a real pair -- the tiled planes store nothing else --, column block slowest)
    nt, nb = pair_tiles(N), (N + 3) // 4
    ke = torch.tensor([pair_key_tiles_effective(int(k), nt) for k in key_tiles_host.tolist()], dtype=torch.int64)
    zero = torch.zeros(1, dtype=torch.int64)
    if rows_host is None:
        pre = torch.cat([zero, torch.cumsum(nb * torch.clamp(4 * ke, max=nb), 0)])
        both = upload(torch.stack([pre, pre]).to(torch.int32), device)
        return both[0], both[1]
    rbk = torch.clamp((torch.as_tensor(rows_host, device="cpu").to(torch.int64) + 3) // 4, max=nb)   # 4-row query blocks up to the representative pad row
    B = rbk.numel()
    pre = torch.cat([zero, torch.cumsum(rbk * torch.clamp(4 * ke, max=nb), 0)])
    flat = upload(torch.cat([pre, pre, rbk, rbk]).to(torch.int32), device)
    return flat[:B + 1], flat[B + 1:2 * B + 2], flat[2 * B + 2:3 * B + 2], flat[3 * B + 2:]


def gbf_bias_fwd(dist, edge_type, mul, bias, means, stds, w1, b1, w2, b2, ld, save=True, tiled=False, save_grad=False, compact=False,
                 tile_prefix=None, row_blocks=None):
    """Fused gbf + gbf_proj + permute -> (out [B,H,N,ld] fp32 -- or the tiled pair layout, fp32 or (compact) fp16 --,
    (feat, u, h) [P,128] bf16 or None)."""
    if compact and not tiled:
        raise MMDTIError("gbf_bias_fwd: compact pair planes exist in the tiled layout only")
    _chk(dist, F32, "gbf.dist"); _chk_edge(edge_type); _chk(w1, BF16, "gbf.w1"); _chk(w2, BF16, "gbf.w2")
    B, N, _ = dist.shape
    Hh, Fh = w2.shape
    K = w1.shape[1]
    out = pair_empty(B, Hh, N, dist.device, tiled, dtype=F16 if compact else F32) if tiled else torch.empty(B, Hh, N, ld, device=dist.device, dtype=F32)
    saved = tuple(torch.empty(B * N * N, 128, device=dist.device, dtype=BF16) for _ in range(3)) if save else None
    t0 = kernel_timer.begin("gbf_bias_fwd")
    lib().mmdti_gbf_bias_fwd(_stream(), dist.data_ptr(), edge_type.data_ptr(), edge_type.element_size(), mul.data_ptr(), bias.data_ptr(), means.data_ptr(), stds.data_ptr(),
                             w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), B, N, ld, K, Fh, Hh, mul.numel(), out.data_ptr(),
                             *([t.data_ptr() for t in saved] if save else [0, 0, 0]), int(tiled) | (2 if save_grad else 0) | (4 if compact else 0), _p(tile_prefix),
                             _p(row_blocks))
    # algorithmic bytes per atom pair: 4 (distance) + edge type in, 64 heads x 4 B (compact: 2 B) of bias out (+ 3 x 256 B kept for the backward)
    kept = _pair_kept if tile_prefix is not None else 1.0
    kernel_timer.end("gbf_bias_fwd", t0, kept * float(B * N * N) * (4 + edge_type.element_size() + Hh * out.element_size() + (3 * 256 if save else 0)))
    return out, saved


def gbf_bias_bwd(g, dist, edge_type, mul, bias, means, stds, w1, w2, u, ld, dmul, dbias, dmeans, dstds, u_is_grad=False):
    """Per-pair half of the fused pair-bias backward -> (do [P,64] bf16, du [P,128] bf16); Gaussian grads accumulated."""
    B, N, _ = dist.shape
    Hh, Fh = w2.shape
    P = B * N * N
    do = torch.empty(P, Hh, device=g.device, dtype=BF16)
    du = torch.empty(P, Fh, device=g.device, dtype=BF16)
    t0 = kernel_timer.begin("gbf_bias_bwd")
    _chk_edge(edge_type)
    lib().mmdti_gbf_bias_bwd(_stream(), g.data_ptr(), dist.data_ptr(), edge_type.data_ptr(), edge_type.element_size(), mul.data_ptr(), bias.data_ptr(), means.data_ptr(),
                             stds.data_ptr(), w1.data_ptr(), w2.data_ptr(), u.data_ptr(), B, N, ld, w1.shape[1], Fh, Hh, mul.numel(),
                             _pair_layout_g(g, "gbf_bias_bwd") | (2 if u_is_grad else 0), do.data_ptr(), du.data_ptr(), dmul.data_ptr(), dbias.data_ptr(), dmeans.data_ptr(),
                             dstds.data_ptr())
    # per pair: G 64 x 4 B + saved gelu' 256 B + distance / edge type in, do 128 B + du 256 B out
    kernel_timer.end("gbf_bias_bwd", t0, float(P) * (Hh * g.element_size() + 256 + 4 + edge_type.element_size() + 2 * Hh + 2 * Fh))
    return do, du


GBF_FULL_MAXE = 1536       # edge-type tables the complete backward kernel keeps in LDS (gbf.hip GBF_FULL_MAXE)
GBF_SLABS = os.environ.get("MMDTI_GBF_SLABS", "1") != "0"


def gbf_bias_bwd_full(g, dist, edge_type, mul, bias, means, stds, w1, b1, w2, ld, dw1, db1, dw2, db2, dmul, dbias, dmeans, dstds,
                      tile_prefix=None, row_blocks=None):
    """The whole backward of :func:`gbf_bias_fwd` in one kernel, nothing saved by the forward: all eight parameter gradients
    (fp32) are accumulated (+=) into the given buffers."""
    _chk(dist, F32, "gbf.dist"); _chk_edge(edge_type); _chk(w1, BF16, "gbf.w1"); _chk(w2, BF16, "gbf.w2")
    for t, nm in ((dw1, "dw1"), (db1, "db1"), (dw2, "dw2"), (db2, "db2"), (dmul, "dmul"), (dbias, "dbias"), (dmeans, "dmeans"), (dstds, "dstds")):
        _chk(t, F32, "gbf." + nm)
    B, N, _ = dist.shape
    Hh, Fh = w2.shape
    P = B * N * N
    # per-workgroup slabs of partial gradients, folded in a fixed order (reproducible bits; MMDTI_GBF_SLABS=0: fp32 atomics)
    ws = torch.empty(lib()._dll.mmdti_gbf_bias_bwd_full_workspace(mul.numel()), device=dist.device, dtype=torch.uint8) if GBF_SLABS else None
    t0 = kernel_timer.begin("gbf_bias_bwd")
    lib().mmdti_gbf_bias_bwd_full(_stream(), g.data_ptr(), dist.data_ptr(), edge_type.data_ptr(), edge_type.element_size(), mul.data_ptr(),
                                  bias.data_ptr(), means.data_ptr(), stds.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), B, N, ld,
                                  w1.shape[1], Fh, Hh, mul.numel(), _pair_layout_g(g, "gbf_bias_bwd_full"), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
                                  db2.data_ptr(), dmul.data_ptr(), dbias.data_ptr(), dmeans.data_ptr(), dstds.data_ptr(), _p(tile_prefix), _p(row_blocks),
                                  _p(ws), 0 if ws is None else ws.numel())
    # Work unit: MFMA flops (the kernel reads 268 B per atom pair and is nowhere near HBM): per pair the recomputed
    # pre-activation (2*F*K), dO.W2 (2*H*F), du.W1 (2*F*K) and the two weight-gradient products (2*F*K + 2*H*F)
    K = w1.shape[1]
    kept = _pair_kept if tile_prefix is not None else 1.0
    kernel_timer.end("gbf_bias_bwd", t0, kept * float(P) * (6.0 * Fh * K + 4.0 * Hh * Fh))


def gbf_bias_eligible(K, Fh, Hh, ld):
    return K == 128 and Fh == 128 and Hh == 64 and ld % 4 == 0


def pair_ld(N):
    """row stride of the internal [B,H,N,ld] pair tensors (16-byte aligned rows)."""
    return (N + 3) // 4 * 4


def pair_permute_fwd(x, B, N, H, ld):
    _chk(x, F32, "pair_permute_fwd.x")
    out = torch.empty(B, H, N, ld, device=x.device, dtype=F32)
    lib().mmdti_pair_permute_fwd(_stream(), x.data_ptr(), out.data_ptr(), B, N, H, ld)
    return out


def pair_permute_bwd(g, B, N, H, ld):
    _chk(g, F32, "pair_permute_bwd.g")
    out = torch.empty(B * N * N, H, device=g.device, dtype=BF16)
    lib().mmdti_pair_permute_bwd(_stream(), g.data_ptr(), out.data_ptr(), B, N, H, ld, int(pair_is_tiled(g)))
    return out


# --------------------------------------------------------------------------------------------- pair-bias attention
# Pair tensors (attention bias, per-layer logits S_l, their gradient G) come in two layouts:
#   standard  [B, H, N, ld]            row-major planes (ld = pair_ld(N))
#   tiled     [B, H, pair_plane(N)]    "blocked rows" (csrc/common.h): per block of 16 queries the 4-key groups follow each other, each
#             holding its vr query rows x 4 keys (vr = 16, or N - 16 qb in the last block):
#                 off(q, k) = 16 (q // 16) N4 + vr (k - k % 4) + 4 (q % 16) + k % 4,        N4 = N rounded up to 4
#             -- a 16x16 tile of a complete block is 256 contiguous elements in MFMA accumulator order (a wave's access to it is ONE
#             contiguous KiB) and nothing is stored for queries or 4-key groups past N (130 atoms: 130 x 132 slots, not 144 x 144: memory, not time -- the
#             kernels never touched most pad slots).
# The tiled form is what the hot path uses (N <= 272: the reference crops at 256 atoms, N <= 258); pair_tile / pair_untile convert at the boundary (tests, aux outputs).
def pair_is_tiled(t):
    return t.dim() == 3


def pair_plane(N):
    """elements of one (molecule, head) plane of a tiled pair tensor (a multiple of 8: 2-byte planes stay 16-byte aligned)."""
    return (N * pair_ld(N) + 7) // 8 * 8


def pair_tiles(N):
    return (N + 15) // 16


def pair_tiled_ok(N):
    return N <= 272


def pair_empty(B, H, N, device, tiled, zero=False, dtype=F32):
    shape = (B, H, pair_plane(N)) if tiled else (B, H, N, pair_ld(N))
    t = (torch.zeros if zero else torch.empty)(shape, device=device, dtype=dtype)
    if tiled and not zero and shape[2] != N * pair_ld(N):
        t[:, :, N * pair_ld(N):] = 0          # (the alignment tail of a plane -- no kernel touches it: N * N4 / 4 odd only -- holds defined bits)
    return t


# Element types.  fp32 everywhere, or -- COMPACT tiled planes, what the hot path runs -- the logits chain as fp16
# (include/mmdti_hip.h, mmdti_pair_attn_fwd: layout 3).  The gradient chain stays fp32 unless MMDTI_PAIR_G_BF16=1 (layout 7:
# measured to cost gradient fidelity).  The dtype of a tensor says which form it is.
PAIR_COMPACT = os.environ.get("MMDTI_PAIR_COMPACT", "1") != "0"
PAIR_G_BF16 = os.environ.get("MMDTI_PAIR_G_BF16", "0") == "1"


def pair_grad_dtype(s):
    """dtype of the gradient chain G that goes with a logits tensor s."""
    return BF16 if (s.dtype == F16 and PAIR_G_BF16) else F32


def _pair_layout_s(s, what):
    """layout code of a logits-like pair tensor: 0 row-major fp32, 1 tiled fp32, 3 tiled fp16."""
    tiled = pair_is_tiled(s)
    if s.dtype == F16:
        if not tiled:
            raise MMDTIError(f"{what}: fp16 pair logits exist in the tiled layout only")
        return 3
    _chk(s, F32, what)
    return int(tiled)


def _pair_layout_g(g, what):
    """layout / flag bits of a gradient pair tensor: 0 row-major fp32, 1 tiled fp32, 5 (bits 0 and 2) tiled bf16."""
    tiled = pair_is_tiled(g)
    if g.dtype == BF16:
        if not tiled:
            raise MMDTIError(f"{what}: bf16 pair gradients exist in the tiled layout only")
        return 5
    _chk(g, F32, what)
    return int(tiled)


def _tile_index(N, device, cols=None):
    """[N, cols] flat offsets inside a tiled plane of (query, key); cols = N (the real pairs) or pair_ld(N) (every slot: the pad keys too)."""
    q = torch.arange(N, device=device).view(N, 1)
    k = torch.arange(N if cols is None else cols, device=device).view(1, -1)
    vr = torch.clamp(N - q // 16 * 16, max=16)
    return (q // 16 * 16) * pair_ld(N) + vr * (k - k % 4) + (q % 16) * 4 + k % 4


def pair_slots(N, device, q_lo=0, q_hi=None, k_lo=0, k_hi=None):
    """flat offsets of every slot of a tiled plane -- pad keys N .. N4-1 included -- whose query lies in [q_lo, q_hi) and key in [k_lo, k_hi)
    (test / debugging glue: e.g. "everything behind key tile k" is k_lo = 16 k)."""
    idx = _tile_index(N, device, pair_ld(N))
    return idx[q_lo:(N if q_hi is None else q_hi), k_lo:(pair_ld(N) if k_hi is None else k_hi)].reshape(-1)


def pair_tile(x, N, pad=float("-inf")):
    """standard [B,H,N,>=N] -> tiled (layout glue for tests / API boundaries).  Pad slots: -inf for logits-like tensors
    (the invariant the attention kernels rely on), pass pad=0 for gradients."""
    B, H = x.shape[:2]
    out = torch.full((B, H, pair_plane(N)), pad, device=x.device, dtype=x.dtype)
    out[:, :, _tile_index(N, x.device).reshape(-1)] = x[..., :N, :N].reshape(B, H, N * N)
    return out


def pair_untile(t, N):
    """tiled -> standard [B,H,N,N]."""
    B, H = t.shape[:2]
    return t.reshape(B, H, -1)[:, :, _tile_index(N, t.device).reshape(-1)].view(B, H, N, N)


def pair_key_tiles_effective(kt, nt):
    """The key-tile count the ragged kernels actually cover for a molecule with ``kt`` real key tiles out of ``nt``: every count
    is supported up to 9 tiles, every 2nd up to 13, every 4th beyond (pair_attn.hip, pa_kt_effective); a count in between runs as
    the next supported one -- the extra tiles are ordinary all-padding tiles."""
    step = 1 if nt <= 9 else (2 if nt <= 13 else 4)
    k = (max(int(kt), 1) + step - 1) // step * step
    return nt if k >= nt else k


_pair_kept = 1.0       # fraction of the padded key columns the ragged kernels keep (work accounting of the timers only)


def set_pair_kept(frac):
    """MM_Model tells the kernel timers what share of the pair tiles a ragged step touches, so that bench.py's achieved
    bytes are the bytes of the tiles actually read and written (1.0 = dense)."""
    global _pair_kept
    _pair_kept = float(frac)


def pair_attn_fwd(qkv, bias_in, key_pad, B, N, H, ld, scale, drop_p=0.0, seed=0, site=0, key_tiles=None, rag_store=False, row_off=None):
    """row_off ([B+1] int32 on the device, with key_tiles): packed token rows -- qkv / o / key_pad hold molecule b's real tokens +
    at most one representative pad row at rows [row_off[b], row_off[b+1]) (packing.PackedRows); the pair planes stay positional."""
    _chk16(qkv, "pair_attn.qkv")
    layout = _pair_layout_s(bias_in, "pair_attn.bias")
    tiled = pair_is_tiled(bias_in)
    o_f16 = False
    if qkv.dtype == F16 and layout != 3:
        # (the fp16 q | k | v kernels exist for the compact tiled planes -- the hot path; the fallback layouts take the bf16 rounding of
        #  q | k | v and hand their output back in the forward operand type through two cast passes)
        qkv, o_f16 = to_bf16(qkv), True
    rows = qkv.shape[0] if row_off is not None else B * N
    if row_off is not None:
        _chk(row_off, torch.int32, "pair_attn.row_off")
        if row_off.numel() != B + 1 or key_tiles is None:
            raise MMDTIError("pair_attn_fwd: row_off must be [B+1] and come with key_tiles")
        if key_pad is not None and key_pad.numel() != rows:
            raise MMDTIError("pair_attn_fwd: with packed rows key_pad is indexed by packed row")
    elif qkv.shape[0] != B * N:
        raise MMDTIError(f"pair_attn_fwd: qkv has {qkv.shape[0]} rows, expected B*N = {B * N}")
    s_out = torch.empty_like(bias_in) if tiled else torch.empty(B, H, N, ld, device=qkv.device, dtype=F32)
    o = torch.empty(rows, H * 8, device=qkv.device, dtype=qkv.dtype)
    kp = _u8(key_pad)
    t0 = kernel_timer.begin("pair_attn_fwd")
    if key_tiles is not None:
        _chk(key_tiles, torch.int32, "pair_attn.key_tiles")
    lib().mmdti_pair_attn_fwd(_stream(), qkv.data_ptr(), bias_in.data_ptr(), s_out.data_ptr(), o.data_ptr(), _p(kp), B, N, H, ld,
                              float(scale), float(drop_p), int(seed), int(site), layout, _p(key_tiles), int(rag_store), _p(row_off),
                              int(qkv.dtype == F16))
    # per (pair, head): read the bias / previous logits, write S (4 B each; compact 2 B); per (token, head): q|k|v in (48 B), o out (16 B)
    kept, es = _pair_kept if key_tiles is not None else 1.0, float(s_out.element_size())
    kernel_timer.end("pair_attn_fwd", t0, float(H) * (B * N * N * (es * kept + es * (1.0 if rag_store else kept)) + rows * 64.0))
    if o_f16:
        o = cast_act16(cast_f32(o)) if FWD_F16 else o
    return s_out, o


def pair_attn_bwd(qkv, s, do, g, B, N, H, ld, scale, g_in_zero, drop_p=0.0, seed=0, site=0, key_tiles=None, row_off=None):
    if (row_off is None and qkv.shape[0] != B * N) or do.shape[0] != qkv.shape[0]:
        raise MMDTIError("pair_attn_bwd: qkv / do row counts do not match the layout")
    if row_off is not None:
        _chk(row_off, torch.int32, "pair_attn.row_off")
    tiled = pair_is_tiled(s)
    if qkv.dtype == F16 and not (tiled and N <= 272):
        qkv = to_bf16(qkv)         # (the per-element fallback kernel reads bf16 only)
    # (fp16 forward operands: the backward's products take the bf16 rounding of q | k | v -- converted by the kernel on the way into LDS)
    dqkv = torch.empty(qkv.shape, device=qkv.device, dtype=BF16)
    layout = _pair_layout_s(s, "pair_attn_bwd.s")
    if pair_is_tiled(g) != tiled or not (g.dtype == F32 or (g.dtype == BF16 and s.dtype == F16)):
        raise MMDTIError("pair_attn_bwd: S and G must share one pair layout (fp32 gradients, or bf16 gradients with fp16 logits)")
    if g.dtype == BF16:
        layout |= 4
    t0 = kernel_timer.begin("pair_attn_bwd")
    lib().mmdti_pair_attn_bwd(_stream(), qkv.data_ptr(), s.data_ptr(), do.data_ptr(), g.data_ptr(), dqkv.data_ptr(), B, N, H, ld,
                              float(scale), int(g_in_zero), float(drop_p), int(seed), int(site), layout, _p(key_tiles), _p(row_off), int(qkv.dtype == F16))
    # per (pair, head): read S, read + write G (4 B each, compact 2 B; the first layer reads no G); per (token, head): 7 x 16 B rows
    kept = _pair_kept if key_tiles is not None else 1.0
    kernel_timer.end("pair_attn_bwd", t0, float(H) * (B * N * N * kept * (s.element_size() + g.element_size() * (1 if g_in_zero else 2)) + qkv.shape[0] * 112.0))
    return dqkv


# --------------------------------------------------------------------------------------------- softmax over materialised scores
def softmax_fwd(s, key_add, B, heads, Lq, Lk, ld, drop_p=0.0, seed=0, site=0):
    _chk(s, F32, "softmax.s")
    p = torch.empty(B, heads, Lq, ld, device=s.device, dtype=BF16)
    pd = torch.empty_like(p) if drop_p > 0 else p
    lib().mmdti_softmax_fwd(_stream(), s.data_ptr(), _p(key_add), p.data_ptr(), pd.data_ptr(), B, heads, Lq, Lk, ld, float(drop_p),
                            int(seed), int(site))
    return p, pd


def softmax_bwd(p, dp, B, heads, Lq, Lk, ld, scale, drop_p=0.0, seed=0, site=0):
    ds = torch.empty(B, heads, Lq, ld, device=p.device, dtype=BF16)
    lib().mmdti_softmax_bwd(_stream(), p.data_ptr(), dp.data_ptr(), ds.data_ptr(), B, heads, Lq, Lk, ld, float(scale), float(drop_p),
                            int(seed), int(site))
    return ds


# --------------------------------------------------------------------------------------------- fused attention
FUSED_ATTN = True      # tests flip this to exercise the materialised-scores path that longer sequences take


def attn_eligible(Lq, Lk, hd, D):
    return FUSED_ATTN and hd in (16, 32, 64) and Lq <= 256 and Lk <= 256 and D % 8 == 0


class AttnVarlen:
    """Packed sequences for the fused attention kernels (include/mmdti_hip.h, mmdti_attn_fwd): q_off / k_off [B+1] int32 row
    offsets of the query / key side, k_cnt [B] int32 real keys per sequence (device tensors); q_rows / k_rows total rows;
    pairs: sum_b q_rows_b * k_cnt_b (work accounting)."""

    def __init__(self, q_pack, k_pack):
        self.q_off, self.k_off, self.k_cnt = q_pack.off, k_pack.off, k_pack.n_real
        self.q_rows, self.k_rows = q_pack.M, k_pack.M
        self.Lq, self.Lk = q_pack.max_rows, int(k_pack.counts_host.max())
        self.pairs = float((q_pack.rows_host * k_pack.counts_host).sum())

    def args(self):
        return (self.q_off.data_ptr(), self.k_off.data_ptr(), self.k_cnt.data_ptr(), self.q_rows)


_NO_VARLEN = (0, 0, 0, 0)


def attn_fwd(q, k, v, key_add, B, heads, Lq, Lk, scale, drop_p=0.0, seed=0, site=0, vl=None):
    """q [B*Lq, D], k/v [B*Lk, D] bf16 -> (ctx [B*Lq, D] bf16, stats [B,heads,Lq,2] fp32).  vl (AttnVarlen): packed sequences --
    q [vl.q_rows, D], k/v [vl.k_rows, D], Lq / Lk the longest sequence of each side; stats is [heads, q_rows, 2]."""
    for t_, n_ in ((q, "attn.q"), (k, "attn.k"), (v, "attn.v")):
        _chk(t_, BF16, n_, contiguous=False)
        if t_.stride(1) != 1:
            raise MMDTIError(f"{n_}: unit column stride required")
    D = q.shape[1]
    hd = D // heads
    rows_q = B * Lq if vl is None else vl.q_rows
    if q.shape[0] != rows_q or k.shape[0] != (B * Lk if vl is None else vl.k_rows) or (vl is not None and key_add is not None):
        raise MMDTIError("attn_fwd: row counts do not match the layout (packed sequences take no key_add)")
    ctx = torch.empty(rows_q, D, device=q.device, dtype=act16())         # (feeds the output projection's forward GEMM)
    stats = torch.empty((B, heads, Lq, 2) if vl is None else (heads, rows_q, 2), device=q.device, dtype=F32)
    t0 = kernel_timer.begin("attn_fwd")
    lib().mmdti_attn_fwd(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(key_add), ctx.data_ptr(), stats.data_ptr(), B, heads, Lq, Lk,
                         hd, q.stride(0), k.stride(0), D, float(scale), float(drop_p), int(seed), int(site), *(_NO_VARLEN if vl is None else vl.args()),
                         int(ctx.dtype == F16))
    kernel_timer.end("attn_fwd", t0, 4.0 * heads * hd * (B * Lq * Lk if vl is None else vl.pairs))          # flops: q.k^T and p.v
    return ctx, stats


def attn_bwd(q, k, v, key_add, dctx, stats, B, heads, Lq, Lk, scale, drop_p=0.0, seed=0, site=0, out=None, vl=None):
    """out: optional (dq, dk, dv) destination views (unit column stride; dk and dv share one row stride)."""
    _chk(dctx, BF16, "attn.dctx")
    D = q.shape[1]
    hd = D // heads
    rows_q, rows_k = (B * Lq, B * Lk) if vl is None else (vl.q_rows, vl.k_rows)
    if q.shape[0] != rows_q or k.shape[0] != rows_k or dctx.shape[0] != rows_q:
        raise MMDTIError("attn_bwd: row counts do not match the layout")
    if out is not None:
        dq, dk, dv = out
        if dk.stride(0) != dv.stride(0) or any(t.stride(1) != 1 or t.dtype != BF16 for t in out) or dq.shape[0] != rows_q or dk.shape[0] != rows_k:
            raise MMDTIError("attn_bwd: bad output views")
    else:
        dq = torch.empty(rows_q, D, device=q.device, dtype=BF16)
        dk = torch.empty(rows_k, D, device=q.device, dtype=BF16)
        dv = torch.empty(rows_k, D, device=q.device, dtype=BF16)
    drow = torch.empty((B, heads, Lq) if vl is None else (heads, rows_q), device=q.device, dtype=F32)
    t0 = kernel_timer.begin("attn_bwd")
    lib().mmdti_attn_bwd(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), _p(key_add), dctx.data_ptr(), stats.data_ptr(), drow.data_ptr(),
                         dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, heads, Lq, Lk, hd, q.stride(0), k.stride(0), dctx.stride(0), dq.stride(0), dk.stride(0),
                         float(scale), float(drop_p), int(seed), int(site), *(_NO_VARLEN if vl is None else vl.args()))
    kernel_timer.end("attn_bwd", t0, 10.0 * heads * hd * (B * Lq * Lk if vl is None else vl.pairs))         # flops of the five products an attention backward needs (scores once)
    return dq, dk, dv


# --------------------------------------------------------------------------------------------- InfoNCE pieces
def seq_mean_fwd(x, B, S, D, ld):
    out = torch.empty(B, D, device=x.device, dtype=F32)
    lib().mmdti_seq_mean_fwd(_stream(), x.data_ptr(), B, S, D, ld, out.data_ptr())
    return out


def seq_mean_bwd(dout, B, S, D, ld, aux=None, aux_mode=0):
    """dx[b*S+s] = dout[b] / S (bf16), optionally times aux (aux_mode 1) or gelu'(aux) (2) elementwise."""
    dx = torch.empty(B * S, ld, device=dout.device, dtype=BF16)
    lib().mmdti_seq_mean_bwd(_stream(), dout.data_ptr(), B, S, D, ld, dx.data_ptr(), _p(aux), 0 if aux is None else aux.stride(0), int(aux_mode))
    return dx


def seq_mean_packed_fwd(x, pack, D, ld):
    """x [pack.M, ld] bf16 -> [B, D] fp32: the unmasked mean over the S padded positions, the representative pad row weighted by
    the number of padded positions it stands for (mmdti_seq_mean_packed_fwd)."""
    _chk(x, BF16, "seq_mean_packed.x", contiguous=False)
    if x.shape[0] != pack.M:
        raise MMDTIError("seq_mean_packed_fwd: x does not have the packed row count")
    out = torch.empty(pack.B, D, device=x.device, dtype=F32)
    lib().mmdti_seq_mean_packed_fwd(_stream(), x.data_ptr(), pack.B, pack.S, D, ld, pack.off.data_ptr(), pack.n_real.data_ptr(), out.data_ptr())
    return out


def seq_mean_packed_bwd(dout, pack, D, ld, aux=None, aux_mode=0):
    dx = torch.empty(pack.M, ld, device=dout.device, dtype=BF16)
    lib().mmdti_seq_mean_packed_bwd(_stream(), dout.data_ptr(), pack.M, pack.S, D, ld, pack.off.data_ptr(), pack.n_real.data_ptr(),
                                    pack.row_seq.data_ptr(), dx.data_ptr(), _p(aux), 0 if aux is None else aux.stride(0), int(aux_mode))
    return dx


def l2norm_fwd(x):
    _chk(x, F32, "l2norm.x", contiguous=False)
    B, D = x.shape
    xh = torch.empty(B, D, device=x.device, dtype=F32)
    inv = torch.empty(B, device=x.device, dtype=F32)
    lib().mmdti_l2norm_fwd(_stream(), x.data_ptr(), B, D, x.stride(0), xh.data_ptr(), inv.data_ptr())
    return xh, inv


def l2norm_bwd(dxh, xh, inv):
    B, D = xh.shape
    dx = torch.empty(B, D, device=xh.device, dtype=F32)
    lib().mmdti_l2norm_bwd(_stream(), dxh.data_ptr(), xh.data_ptr(), inv.data_ptr(), B, D, D, dx.data_ptr())
    return dx


def infonce_dir(qh_all, kh_all, row0, Bl, temperature, loss_sum, dq_all, dk_all):
    Bg, D = qh_all.shape
    scratch = torch.empty(Bl * (Bg + 1), device=qh_all.device, dtype=F32)      # (logit gradients [Bl, Bg] + the Bl per-row loss terms)
    lib().mmdti_infonce_dir(_stream(), qh_all.data_ptr(), kh_all.data_ptr(), Bg, D, row0, Bl, float(temperature), loss_sum.data_ptr(),
                            dq_all.data_ptr(), dk_all.data_ptr(), scratch.data_ptr())


# --------------------------------------------------------------------------------------------- ConR / SupCon
def ct_loss_fwd(mode, fhat, labels_f=None, labels_i=None, pred=None, weights=None, w=0.2, t=0.07, e=0.01, coef=1.0):
    B, D = fhat.shape
    buf = torch.empty(1 + B, device=fhat.device, dtype=F32)        # loss | per-row terms (summed in row order: reproducible)
    loss = buf[:1]
    G = torch.empty(B, B, device=fhat.device, dtype=F32)
    C = 0 if labels_i is None else labels_i.shape[1]
    lib().mmdti_ct_loss_fwd(_stream(), mode, fhat.data_ptr(), B, D, _p(labels_f), _p(labels_i), C, _p(pred), _p(weights), float(w),
                            float(t), float(e), float(coef), loss.data_ptr(), G.data_ptr(), buf.data_ptr() + 4)
    return loss, G


def ct_loss_bwd(fhat, G, t=0.07):
    B, D = fhat.shape
    d = torch.empty_like(fhat)
    lib().mmdti_ct_loss_bwd(_stream(), fhat.data_ptr(), G.data_ptr(), B, D, float(t), d.data_ptr())
    return d


# --------------------------------------------------------------------------------------------- FDS
def fds_bins(labels, min_value, bin_width, bucket_start, bucket_num):
    _chk(labels, F32, "fds_bins.labels")
    n = labels.numel()
    bins = torch.empty(n, device=labels.device, dtype=torch.int32)
    flags = torch.zeros(2, device=labels.device, dtype=torch.int32)
    lib().mmdti_fds_bins(_stream(), labels.data_ptr(), n, float(min_value), float(bin_width), bucket_start, bucket_num, bins.data_ptr(),
                         flags.data_ptr())
    return bins, flags


def fds_smooth(x, bins, flags, bucket_start, bucket_num, m1, v1, m2, v2, want_scale=True):
    n, D = x.shape
    y = torch.empty_like(x)
    sc = torch.empty_like(x) if want_scale else None
    lib().mmdti_fds_smooth(_stream(), x.data_ptr(), bins.data_ptr(), flags.data_ptr(), n, D, bucket_start, bucket_num, m1.data_ptr(),
                           v1.data_ptr(), m2.data_ptr(), v2.data_ptr(), y.data_ptr(), _p(sc))
    return y, sc


def fds_update_stats(feats, bins, flags, bucket_start, bucket_num, factor, running_mean, running_var, tracked):
    n, D = feats.shape
    lib().mmdti_fds_update_stats(_stream(), feats.data_ptr(), bins.data_ptr(), flags.data_ptr(), n, D, bucket_start, bucket_num,
                                 float(factor), running_mean.data_ptr(), running_var.data_ptr(), tracked.data_ptr())


def fds_smooth_stats(stat, window):
    nb, D = stat.shape
    out = torch.empty_like(stat)
    lib().mmdti_fds_smooth_stats(_stream(), stat.data_ptr(), nb, D, window.data_ptr(), window.numel(), out.data_ptr())
    return out


# --------------------------------------------------------------------------------------------- pooling / head / losses
def masked_pool_fwd(a, t, mask_a, mask_t):
    B, Na, D = a.shape
    Nt = t.shape[1]
    out = torch.empty(B, D, device=a.device, dtype=F32)
    lib().mmdti_masked_pool_fwd(_stream(), a.data_ptr(), t.data_ptr(), mask_a.data_ptr(), mask_t.data_ptr(), B, Na, Nt, D, out.data_ptr())
    return out


def masked_pool_bwd(dp, mask_a, mask_t, Na, Nt):
    B, D = dp.shape
    da = torch.empty(B, Na, D, device=dp.device, dtype=F32)
    dt = torch.empty(B, Nt, D, device=dp.device, dtype=F32)
    lib().mmdti_masked_pool_bwd(_stream(), dp.data_ptr(), mask_a.data_ptr(), mask_t.data_ptr(), B, Na, Nt, D, da.data_ptr(), dt.data_ptr())
    return da, dt


def masked_pool_packed_fwd(a, t, pa, pt):
    """a [pa.M, D], t [pt.M, D] fp32 packed rows -> pooled [B, D]: sum of every sequence's REAL rows of both / (n_a + n_t)."""
    _chk(a, F32, "masked_pool_packed.a"); _chk(t, F32, "masked_pool_packed.t")
    if a.shape[0] != pa.M or t.shape[0] != pt.M or pa.B != pt.B:
        raise MMDTIError("masked_pool_packed_fwd: row counts do not match the packings")
    D = a.shape[1]
    out = torch.empty(pa.B, D, device=a.device, dtype=F32)
    lib().mmdti_masked_pool_packed_fwd(_stream(), a.data_ptr(), t.data_ptr(), pa.off.data_ptr(), pa.n_real.data_ptr(), pt.off.data_ptr(),
                                       pt.n_real.data_ptr(), pa.B, D, out.data_ptr())
    return out


def masked_pool_packed_bwd(dp, pa, pt):
    B, D = dp.shape
    da = torch.empty(pa.M, D, device=dp.device, dtype=F32)
    dt = torch.empty(pt.M, D, device=dp.device, dtype=F32)
    lib().mmdti_masked_pool_packed_bwd(_stream(), dp.data_ptr(), pa.off.data_ptr(), pa.n_real.data_ptr(), pa.row_seq.data_ptr(), pa.M,
                                       pt.off.data_ptr(), pt.n_real.data_ptr(), pt.row_seq.data_ptr(), pt.M, D, da.data_ptr(), dt.data_ptr())
    return da, dt


def linear_f32_fwd(x, W, b, act=ACT_NONE):
    rows, in_f = x.shape
    out_f = W.shape[0]
    y = torch.empty(rows, out_f, device=x.device, dtype=F32)
    lib().mmdti_linear_f32_fwd(_stream(), x.data_ptr(), W.data_ptr(), _p(b), rows, in_f, out_f, act, y.data_ptr())
    return y


def linear_f32_bwd(x, W, y, dy, dW, db, act=ACT_NONE, want_dx=True):
    rows, in_f = x.shape
    out_f = W.shape[0]
    dx = torch.empty_like(x) if want_dx else None
    lib().mmdti_linear_f32_bwd(_stream(), x.data_ptr(), W.data_ptr(), _p(y), dy.data_ptr(), rows, in_f, out_f, act, _p(dx), _p(dW), _p(db))
    return dx


def mse_loss(pred, target):
    loss = torch.empty(1, device=pred.device, dtype=F32)
    d = torch.empty_like(pred)
    lib().mmdti_mse_loss(_stream(), pred.data_ptr(), target.data_ptr(), pred.numel(), loss.data_ptr(), d.data_ptr())
    return loss, d


def ce_loss(logits, target):
    B, C = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=F32)
    d = torch.empty_like(logits)
    lib().mmdti_ce_loss(_stream(), logits.data_ptr(), target.data_ptr(), B, C, loss.data_ptr(), d.data_ptr())
    return loss, d


def bce_logits_loss(logits, target):
    _chk(logits, F32, "bce_logits.logits"); _chk(target, F32, "bce_logits.target")
    if target.shape != logits.shape:
        raise MMDTIError(f"bce_logits_loss: target shape {tuple(target.shape)} != logits shape {tuple(logits.shape)}")
    loss = torch.empty(1, device=logits.device, dtype=F32)
    d = torch.empty_like(logits)
    lib().mmdti_bce_logits_loss(_stream(), logits.data_ptr(), target.data_ptr(), logits.numel(), loss.data_ptr(), d.data_ptr())
    return loss, d


def sumsq(g, out, ws=None):
    """out[0] += sum of squares of g.  ws: fp32 scratch (up to 2048 values) -- per-workgroup partials folded in a fixed order: reproducible."""
    lib().mmdti_sumsq_f32(_stream(), g.data_ptr(), g.numel(), out.data_ptr(), _p(ws), 0 if ws is None else ws.numel())
    return out


def adam_step(p, g, m, v, p_bf16, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None, step_state=None, p_f16=None):
    """step_state: the device-resident schedule (step_state_advance) -- lr / step are then read on the device.  p_f16: the fp16 weight
    shadow of the fp16 forward-operand mode, refreshed in the same pass as the bf16 one."""
    lib().mmdti_adam_step(_stream(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(p_bf16), p.numel(), float(lr), float(beta1),
                          float(beta2), float(eps), float(weight_decay), int(step), _p(grad_scale), _p(step_state), _p(p_f16))


def step_state_advance(state, salt, base_lr, warmup, total, beta1=0.9, beta2=0.999):
    """state [4] fp32, salt [2] int64 (device): advance the optimizer-step counter, this step's learning rate / bias
    corrections and the dropout salt, then publish the salt to every kernel library."""
    _chk(state, F32, "step_state.state"); _chk(salt, torch.int64, "step_state.salt")
    lib().mmdti_step_state_advance(_stream(), state.data_ptr(), salt.data_ptr(), float(base_lr), int(warmup), int(total), float(beta1), float(beta2))
    lib().mmdti_seed_salt_pull(_stream(), salt.data_ptr() + 8)


_zero_salt = None


def seed_salt_reset(sync=True):
    """Back to salt 0: dropout streams are exactly what the by-value (seed, site) pairs define (the eager path's contract).
    sync=False: only enqueued (the zero word lives in a persistent device buffer) -- what FineTuner.step does when an eager step
    follows graph replays."""
    global _zero_salt
    if _zero_salt is None:
        _zero_salt = torch.zeros(1, device="cuda", dtype=torch.int64)
    lib().mmdti_seed_salt_pull(_stream(), _zero_salt.data_ptr())
    if sync:
        torch.cuda.current_stream().synchronize()


def probe_tr_read(stride):
    out = torch.empty(256, device="cuda", dtype=torch.int16)
    lib().mmdti_probe_tr_read(_stream(), int(stride), out.data_ptr())
    return out


# --------------------------------------------------------------------------------------------- live kernel timing (bench.py)
class _KernelTimer:
    """HIP-event timing of selected kernel launches on the stream they are launched on (bench.py `roofline`)."""

    def __init__(self):
        self.names = ()
        self.events = {}
        self.work = {}
        self.tags = {}

    def enable(self, names):
        self.names = tuple(names)
        self.events = {n: [] for n in self.names}
        self.work = {n: 0.0 for n in self.names}
        self.tags = {n: [] for n in self.names}

    def disable(self):
        self.names = ()

    def begin(self, name):
        if name not in self.names:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, name, e0, work=0.0, tag=None):
        """work: algorithmic flops (or bytes) of this launch, summed per name; tag: anything that identifies the launch's
        shape (by_tag() groups on it)."""
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.events[name].append((e0, e1))
        self.work[name] += work
        self.tags[name].append((tag, work))

    def by_tag(self, name):
        """{tag: {"n", "mean_ms", "total_ms", "work"}} of the launches recorded under `name`."""
        torch.cuda.synchronize()
        out = {}
        for (a, b), (tag, work) in zip(self.events.get(name, ()), self.tags.get(name, ())):
            d = out.setdefault(tag, {"n": 0, "total_ms": 0.0, "work": 0.0})
            d["n"] += 1
            d["total_ms"] += a.elapsed_time(b)
            d["work"] += work
        for d in out.values():
            d["mean_ms"] = d["total_ms"] / d["n"]
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for n, evs in self.events.items():
            if evs:
                ts = [a.elapsed_time(b) for a, b in evs]
                out[n] = {"n": len(ts), "mean_ms": sum(ts) / len(ts), "min_ms": min(ts), "total_ms": sum(ts), "work": self.work.get(n, 0.0)}
        return out


kernel_timer = _KernelTimer()
