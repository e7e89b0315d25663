"""Module-level autograd functions composed from the HIP kernels (espresso_amd/kernels.py).

Each function here is ONE autograd node that mirrors one reference sub-module, with an explicit
hand-scheduled backward (no autograd tracing inside):

  ffn_module        fairseq/modules/conformer_layer.py:134-146 + the 0.5*x + residual of
                    espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:112-114,136-139
  relpos_mhsa       LN + fairseq/modules/multihead_attention.py:544-917 (rel-pos branch) + dropout + residual
  conv_module       fairseq/modules/conformer_layer.py:79-101 + residual
  layer_norm        fairseq/modules/layer_norm.py
  linear            torch.nn.Linear
  ctc_loss          espresso/criterions/ctc_loss.py:61-94

Internal activation layout is [B*T][C] (batch-major rows, bf16); weights are fp32 master copies
with a bf16 shadow (see FlatParams in espresso_amd/optim/flat.py).  Dropout masks are never stored:
a per-call 64-bit seed is saved and the kernels re-derive the mask from (seed, element index).
"""
import collections

import ctypes
import os

import torch

from . import kernels as K

_seed_state = {"base": 0x5EED, "counter": 0}


def set_dropout_seed(seed: int):
    _seed_state["base"] = int(seed) & 0xFFFFFFFF
    _seed_state["counter"] = 0


_seed_trace = None


class trace_dropout_seeds:
    """Context manager for the parity tests: records (site, seed, p) for every dropout seed drawn inside it, in draw order
    (`.entries`).  Sites: 'subsample.out', 'ln.out', 'dropout', 'ffn.act', 'ffn.out', 'attn.probs', 'attn.out', 'conv.out'
    (per-kernel composition) and 'layer:conformer' / 'layer:transformer' / 'layer:decoder' (native layer runtime: p is the
    dict of the layer's three probabilities, per-site seeds from `ea_layer_dropout_seed`, include/espresso_amd.h).  The
    element index of each site's mask stream is the element's position in the dense activation the site writes (header,
    "Dropout sites")."""

    def __enter__(self):
        global _seed_trace
        self.prev, self.entries = _seed_trace, []
        _seed_trace = self.entries
        return self

    def __exit__(self, *a):
        global _seed_trace
        _seed_trace = self.prev


def _next_seed(site: str = "dropout", p=None) -> int:
    _seed_state["counter"] += 1
    seed = ((_seed_state["base"] << 32) | (_seed_state["counter"] & 0xFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    if _seed_trace is not None:
        _seed_trace.append([site, seed, p])
    return seed


def _layer_seed(kind: str, p_drop, p_act, p_attn) -> int:
    """EaLayerShape.seed of one native layer call: a multiple of 64, so that the site offsets (< 64) never collide."""
    seed = _next_seed() * 64 % (1 << 63)
    if _seed_trace is not None:
        _seed_trace[-1][:] = ["layer:" + kind, seed, {"p_drop": float(p_drop), "p_act": float(p_act), "p_attn": float(p_attn)}]
    return seed


def bf16_weight(p: torch.Tensor) -> torch.Tensor:
    """bf16 shadow of an fp32 parameter: the FlatParams view when attached and current, else a HIP cast."""
    sh = getattr(p, "_ea_bf16", None)
    if sh is not None:
        return sh
    if not torch.is_grad_enabled():
        # inference without the trainer's flat layout (decoders step thousands of times over frozen weights): one cast per
        # parameter version instead of one per call
        ver = (p._version, p.data_ptr())
        hit = getattr(p, "_ea_bf16_infer", None)
        if hit is not None and hit[0] == ver:
            return hit[1]
        sh = K.cast_f32_to_bf16(p.detach().contiguous())
        try:
            p._ea_bf16_infer = (ver, sh)
        except (AttributeError, RuntimeError):
            pass
        return sh
    return K.cast_f32_to_bf16(p.detach().contiguous())


def _new(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _auto_splitk(tiles, Kred):
    """Enough workgroups to fill 256 CUs a few times over when the output has few 128x128 tiles."""
    if tiles >= 384 or Kred < 1024:
        return 1
    return max(1, min((768 + tiles - 1) // tiles, Kred // 256))


def _wgrad(dy, x, M, N_out, K_in, ld_dy=None, ld_x=None, out=None, accumulate=False):
    """dW[N_out][K_in] (fp32) = dy[M][N_out]^T x[M][K_in]   (split-K over the long M reduction); accumulate: dW += ..."""
    dW = out if out is not None else torch.empty((N_out, K_in), dtype=torch.float32, device=dy.device)
    tiles = ((N_out + 127) // 128) * ((K_in + 127) // 128)
    K.gemm(dy, x, dW, N_out, K_in, M, lda=ld_dy or N_out, ldb=ld_x or K_in, ldc=K_in, a_kstrided=True, b_kstrided=True,
           splitk=_auto_splitk(tiles, M), accumulate=accumulate)
    return dW


# Small zero-initialised accumulators (bias / LayerNorm / BatchNorm parameter-gradient sums, BatchNorm statistics) are carved out
# of ONE pool that is cleared with a single fill per update step (Trainer.train_step -> begin_step) instead of one torch.zeros
# launch each (~30 per step).  A buffer handed out is only valid until the next begin_step(); everything that takes one consumes
# it within the step (autograd adds it into the flat gradient, BatchNorm finalises its statistics).  Outside a step (tests,
# inference) or when the pool is exhausted the call falls back to torch.zeros.
_zero_pool = {"buf": None, "off": 0, "live": False}
_ZERO_POOL_BYTES = 1 << 20


def begin_step(device):
    """Clear the zero pool for a new update step (one fill)."""
    zp = _zero_pool
    if zp["buf"] is None or zp["buf"].device != torch.device(device):
        zp["buf"] = torch.zeros(_ZERO_POOL_BYTES, dtype=torch.uint8, device=device)
    else:
        zp["buf"].zero_()
    zp["off"], zp["live"] = 0, True


def end_step():
    _zero_pool["live"] = False
    join_side_streams()


_side_pending = {}  # device -> optimizer-only work was put on the Python-side stream since the last join
_LINEAR_WGRAD_SIDE = os.environ.get("EA_LINEAR_WGRAD_SIDE", "1") != "0"  # (A/B switch)


def join_side_streams():
    """The compute stream waits for optimizer-only work that `_Linear.backward` put on the Python-side stream (called by
    `end_step()` before the optimizer reads the gradients; idempotent)."""
    for key in [k for k, v in _side_pending.items() if v]:
        _side_pending[key] = False
        st = _side_streams.get(key)
        if st is not None:
            dev = torch.device(key)
            torch.cuda.current_stream(dev).wait_stream(st)


def _pool_zeros(shape, dtype, device):
    zp = _zero_pool
    n = 1
    for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    if zp["live"] and zp["buf"].device == torch.device(device):
        off = (zp["off"] + 255) // 256 * 256
        if off + nbytes <= _ZERO_POOL_BYTES:
            zp["off"] = off + nbytes
            return zp["buf"][off: off + nbytes].view(dtype).view(shape)
    return torch.zeros(shape, dtype=dtype, device=device)


def _zeros_f32(n, like):
    return _pool_zeros((n,), torch.float32, like.device)


# Parameter gradients written straight into `p.grad` (the flat gradient buffer's view) by the accumulating kernels of the
# Python-composed backward passes, instead of into a pooled zero buffer that autograd then adds to `p.grad` with one ATen launch
# per parameter (22 launches at the very end of the Conformer-CTC backward, in front of the optimizer).  A parameter handled this
# way gets None from the autograd Function; autograd still runs its AccumulateGrad node — nothing to add — and with it the
# post-accumulate-grad hook the data-parallel wrapper counts bucket completion with (tests/test_host_logic.py pins that
# behaviour of torch), at the same point of the stream order as before.  Only inside an `accumulating_backward()` scope (the trainer's
# backward calls); EA_DIRECT_GRADS=0: the autograd route everywhere (A/B switch, tests).
_DIRECT_GRADS = os.environ.get("EA_DIRECT_GRADS", "1") != "0"


def set_direct_param_grads(on: bool) -> bool:
    global _DIRECT_GRADS
    old, _DIRECT_GRADS = _DIRECT_GRADS, bool(on)
    return old


_sink_scope = 0
_none_grad_hooks_ok = None  # lazily: does this torch fire post-accumulate-grad hooks for a parameter whose Function returned None?


def _none_grad_hooks_fire() -> bool:
    """Capability check behind the direct-gradient route: the data-parallel wrapper counts a bucket down from the parameter's
    post-accumulate-grad hook, which must therefore fire although the autograd Function returned None for that parameter (the
    kernel has already written `p.grad`).  torch 2.x does; the answer is measured once on a two-element CPU example instead of
    assumed, and the direct route is switched off if it ever changes (tests/test_host_logic.py pins the same behaviour)."""
    global _none_grad_hooks_ok
    if _none_grad_hooks_ok is None:
        class _Probe(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                return x * 2

            @staticmethod
            def backward(ctx, dy):
                return dy * 2, None

        fired = []
        try:
            w = torch.nn.Parameter(torch.ones(2))
            w.grad = torch.zeros(2)
            w.register_post_accumulate_grad_hook(lambda p: fired.append(1))
            with torch.enable_grad():
                _Probe.apply(torch.ones(2, requires_grad=True), w).sum().backward()
        except Exception:
            fired = []
        _none_grad_hooks_ok = bool(fired)
    return _none_grad_hooks_ok


class accumulating_backward:
    """`with accumulating_backward(): loss.backward()` — the caller declares that this backward pass ACCUMULATES into the
    parameters' `.grad` buffers (what `Tensor.backward()` does under a trainer that owns those buffers).  Only inside this scope
    do the kernels write parameter gradients behind autograd's back; anywhere else (`torch.autograd.grad`, `backward(inputs=…)`,
    a foreign training loop) every parameter gradient goes through autograd as usual."""

    def __enter__(self):
        global _sink_scope
        _none_grad_hooks_fire()  # (cached after the first call)
        _sink_scope += 1
        return self

    def __exit__(self, *exc):
        global _sink_scope
        _sink_scope -= 1
        if _sink_scope == 0:
            join_side_streams()  # (gradients that were accumulated on the Python-side stream are ordered on the compute stream again)
        return False


def _grad_sink(p, n=None):
    """The fp32 buffer a kernel may ACCUMULATE parameter `p`'s gradient into directly (flat view of p.grad), or None."""
    if not _DIRECT_GRADS or _sink_scope <= 0 or p is None or not getattr(p, "requires_grad", False):
        return None
    if not _none_grad_hooks_ok:  # (measured by accumulating_backward.__enter__, outside the autograd engine)
        return None
    if not p.is_leaf:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or not g.is_cuda or (n is not None and g.numel() != n):
        return None
    return g.view(-1)




_side_streams = {}


def new_side_stream(device):
    """A stream that does not share a HARDWARE queue with the current stream.  HIP deals a process's streams onto four hardware
    queues by load; two streams on one queue run one after the other, silently (round 5: a process group's streams pushed the
    weight-gradient stream onto the compute stream's queue, +1.5 ms per step; csrc/engine.hip `side_stream_create`).  Candidates
    come from torch's pool in turn; each is accepted when `ea_streams_share_queue` sees a tiny kernel on it finish beside a spin
    kernel on the current stream.  Blocks the host once, at set-up."""
    st = torch.cuda.Stream(device=device)
    if os.environ.get("EA_SIDE_STREAM_PROBE", "1") == "0":
        _probe_log.append({"rejected": 0, "unprobed": True})
        return st
    from . import _lib

    cur = torch.cuda.current_stream(device)
    lib = _lib.lib()
    rejected, unprobed = 0, True
    with torch.cuda.device(device):
        for _ in range(7):
            if lib.ea_streams_share_queue(ctypes.c_void_p(cur.cuda_stream), ctypes.c_void_p(st.cuda_stream)) != 1:
                unprobed = False
                break
            rejected += 1
            st = torch.cuda.Stream(device=device)
    _probe_log.append({"rejected": rejected, "unprobed": unprobed})
    return st


_probe_log = []  # one verdict per stream `new_side_stream` handed out


def side_stream_probe_report():
    """What the hardware-queue probe decided in this process: the layer runtime's side stream (csrc/engine.hip) and every stream
    `new_side_stream` created (Python-composed backward, transducer branches, the data-parallel communication stream)."""
    from . import _lib

    rej, unp = ctypes.c_int(-1), ctypes.c_int(0)
    have = _lib.lib().ea_side_stream_report(ctypes.byref(rej), ctypes.byref(unp))
    return {"layer_runtime": {"created": bool(have), "rejected": rej.value, "unprobed": bool(unp.value)},
            "python_streams": list(_probe_log)}


def _side_stream(device):
    """One auxiliary stream per device for optimizer-only work of the Python-composed backward passes."""
    key = str(device)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = new_side_stream(device)
    return st


_aux_streams = {}
_BRANCH_OVERLAP = os.environ.get("EA_BRANCH_OVERLAP", "1") != "0"


def aux_stream(device, idx):
    """Further per-device streams for whole independent branches of a model (idx 1: the transducer's predictor network, idx 2:
    the joint network's weight gradient); `_side_stream` stays reserved for work that joins inside one backward call."""
    key = (str(device), idx)
    st = _aux_streams.get(key)
    if st is None:
        st = _aux_streams[key] = new_side_stream(device)
    return st


def python_side_streams(device):
    """Every stream this module has created for `device` (the per-call side stream of the Python-composed backward passes and
    the branch streams above): gradient work may be in flight on any of them when a data-parallel bucket becomes complete."""
    key = str(torch.device(device))
    out = [st for k, st in _side_streams.items() if k == key]
    out += [st for (k, _), st in _aux_streams.items() if k == key]
    return out


def set_branch_overlap(on: bool) -> bool:
    """Run independent branches of the transducer (predictor network, joint weight gradient) on their own streams (default on;
    A/B switch and the tests' way to get the single-stream schedule).  Returns the previous setting."""
    global _BRANCH_OVERLAP
    old, _BRANCH_OVERLAP = _BRANCH_OVERLAP, bool(on)
    return old


def branch_overlap() -> bool:
    return _BRANCH_OVERLAP


# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, w16, out_f32):
        M, Kin = x.shape
        N = w.shape[0]
        # rows are padded to a multiple of 8 elements so that every later 16-byte access (dgrad /
        # wgrad of a vocabulary-sized N such as 5004) stays aligned; the result is a [M][N] view.
        ld = _pad8(N)
        buf = _new((M, ld), torch.float32 if out_f32 else torch.bfloat16, x)
        K.gemm(x, w16, buf, M, N, Kin, lda=Kin, ldb=Kin, ldc=ld, bias=b)
        ctx.save_for_backward(x, w16)
        ctx.has_bias = b is not None
        ctx.params = (w, b)
        return buf if ld == N else buf[:, :N]

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        M, Kin = x.shape
        N = w16.shape[0]
        if dy.dtype != torch.bfloat16:
            # fp32 output (vocabulary logits): one pass to bf16 at a row pitch of a multiple of 8 elements
            if dy.stride(1) != 1:
                dy = dy.contiguous()
            dy = K.cast_f32_to_bf16_rows(dy, _pad8(N)) if M <= 65535 else K.cast_f32_to_bf16(dy.contiguous())
        if dy.stride(1) != 1 or dy.stride(0) % 8 != 0:
            dy = dy.contiguous()
        ld = dy.stride(0)
        pw, pb = ctx.params
        sW = _grad_sink(pw, N * Kin)
        sb = _grad_sink(pb, N) if ctx.has_bias else None
        if sW is not None and (sb is not None or not ctx.has_bias):
            # a leaf parameter with its gradient buffer in place (fc_out, fc0): accumulate there, nothing returned to autograd.
            # Optimizer-only products: for the big projections they go to the Python-side stream (round 6: 2 x 106 us of split-K
            # GEMM per update step were on the compute stream, in front of the data gradient the backward chain waits for);
            # `end_step()` / the data-parallel wrapper join that stream before the gradients are used.
            side = _side_stream(dy.device) if (_LINEAR_WGRAD_SIDE and dy.is_cuda and 2.0 * M * N * Kin >= 4e9) else None
            if side is not None:
                cur = torch.cuda.current_stream(dy.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    _wgrad(dy, x, M, N, Kin, ld_dy=ld, out=sW.view(N, Kin), accumulate=True)
                    if ctx.has_bias:
                        K.colsum(dy, sb, M, N, ld)
                dy.record_stream(side)
                x.record_stream(side)
                _side_pending[str(dy.device)] = True
            else:
                _wgrad(dy, x, M, N, Kin, ld_dy=ld, out=sW.view(N, Kin), accumulate=True)
                if ctx.has_bias:
                    K.colsum(dy, sb, M, N, ld)
            dW, db = None, None
        else:
            dW = _wgrad(dy, x, M, N, Kin, ld_dy=ld)
            db = None
            if ctx.has_bias:
                db = K.colsum(dy, _zeros_f32(N, dy), M, N, ld)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _new((M, Kin), torch.bfloat16, x)
            K.gemm(dy, w16, dx, M, Kin, N, lda=ld, ldb=Kin, ldc=Kin, b_kstrided=True)
        return dx, dW, db, None, None


def linear(x, w, b=None, out_f32=False):
    """y = x w^T + b ; x bf16 [M][K], w fp32 [N][K] (bf16 shadow used on the MFMA)."""
    return _Linear.apply(x, w, b, bf16_weight(w), out_f32)


# ------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps, row_zero, drop_p, out_f32=False):
        seed = _next_seed("ln.out", drop_p) if drop_p > 0 else 0
        y, mean, rstd = K.layernorm_fwd(x, g, b, eps, row_zero, drop_p, seed, out_f32=out_f32)
        ctx.save_for_backward(x, g, mean, rstd, row_zero)
        ctx.drop = (drop_p, seed)
        ctx.params = (g, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd, row_zero = ctx.saved_tensors
        C = x.shape[1]
        sg, sb = _grad_sink(ctx.params[0], C), _grad_sink(ctx.params[1], C)
        direct = sg is not None and sb is not None
        dg, db = (sg, sb) if direct else (_zeros_f32(C, x), _zeros_f32(C, x))
        dx = K.layernorm_bwd(x, dy.contiguous(), g, mean, rstd, dg, db, row_zero, ctx.drop[0], ctx.drop[1])
        if direct:
            return dx, None, None, None, None, None, None
        return dx, dg, db, None, None, None, None


def layer_norm(x, g, b, eps=1e-5, row_zero=None, drop_p=0.0, out_f32=False):
    """y = zero_rows(dropout(LN(x)))  (row_zero: uint8 [M] marks padded frames).  out_f32: fp32 output and, in backward, an fp32
    incoming gradient (the reference's autocast keeps LayerNorm outputs in fp32: used where the consumer is not a GEMM)."""
    return _LayerNorm.apply(x, g, b, eps, row_zero, drop_p, out_f32)


# ------------------------------------------------------------------------------------------------
class _FFN(torch.autograd.Function):
    """y = out_scale * drop2(W2 drop1(act(W1 LN(x) + b1)) + b2) + x   (pre_ln=True)
       y = out_scale * drop2(W2 drop1(act(W1 x + b1)) + b2) + x        (ln_g None: post-LN layers, fairseq transformer_layer.py:200-226
                                                                      with normalize_before False; the caller applies the LayerNorm)"""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, w1, b1, w2, b2, w1_16, w2_16, act, p_act, p_out, out_scale, eps):
        M, C = x.shape
        Fd = w1.shape[0]
        if ln_g is not None:
            xn, mean, rstd = K.layernorm_fwd(x, ln_g, ln_b, eps)
        else:  # post-LN layer: the caller normalises the residual sum afterwards
            xn, mean, rstd = x, None, None
        s1 = _next_seed("ffn.act", p_act) if p_act > 0 else 0
        s2 = _next_seed("ffn.out", p_out) if p_out > 0 else 0
        z = _new((M, Fd), torch.bfloat16, x)
        h = _new((M, Fd), torch.bfloat16, x)
        K.gemm(xn, w1_16, z, M, Fd, C, lda=C, ldb=C, ldc=Fd, bias=b1, act=act, C2=h, ldc2=Fd, drop_p=p_act, drop_seed=s1)
        y = _new((M, C), torch.bfloat16, x)
        K.gemm(h, w2_16, y, M, C, Fd, lda=Fd, ldb=Fd, ldc=C, bias=b2, drop_p=p_out, drop_seed=s2, out_scale=out_scale,
               resid=x, ldr=C)
        ctx.save_for_backward(x, ln_g, mean, rstd, xn, z, h, w1_16, w2_16)
        ctx.cfg = (act, p_act, p_out, out_scale, s1, s2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ln_g, mean, rstd, xn, z, h, w1_16, w2_16 = ctx.saved_tensors
        act, p_act, p_out, out_scale, s1, s2 = ctx.cfg
        M, C = x.shape
        Fd = z.shape[1]
        dy = dy.contiguous()
        g2 = K.scale_dropout(dy, a=out_scale, drop_p=p_out, drop_seed=s2)
        dW2 = _wgrad(g2, h, M, C, Fd)
        db2 = K.colsum(g2, _zeros_f32(C, x), M, C, C)
        dz = _new((M, Fd), torch.bfloat16, x)
        K.gemm(g2, w2_16, dz, M, Fd, C, lda=C, ldb=Fd, ldc=Fd, b_kstrided=True, aux=z, ldaux=Fd, act=act, drop_p=p_act,
               drop_seed=s1)
        dW1 = _wgrad(dz, xn, M, Fd, C)
        db1 = K.colsum(dz, _zeros_f32(Fd, x), M, Fd, Fd)
        dxn = _new((M, C), torch.bfloat16, x)
        K.gemm(dz, w1_16, dxn, M, C, Fd, lda=Fd, ldb=C, ldc=C, b_kstrided=True)
        if ln_g is None:
            return dxn + dy, None, None, dW1, db1, dW2, db2, None, None, None, None, None, None, None
        dg, db = _zeros_f32(C, x), _zeros_f32(C, x)
        dx = K.layernorm_bwd(x, dxn, ln_g, mean, rstd, dg, db, dx_add=dy)
        return dx, dg, db, dW1, db1, dW2, db2, None, None, None, None, None, None, None


def ffn_module(x, ln_g, ln_b, w1, b1, w2, b2, act="silu", p_act=0.0, p_out=0.0, out_scale=0.5, eps=1e-5):
    return _FFN.apply(x, ln_g, ln_b, w1, b1, w2, b2, bf16_weight(w1), bf16_weight(w2), act, p_act, p_out, out_scale, eps)


# ------------------------------------------------------------------------------------------------
def _pad8(n):
    return (n + 7) // 8 * 8


class _RelPosMHSA(torch.autograd.Function):
    """y = drop(out_proj(Attn(LN(x)))) + x with Transformer-XL relative logits.

    wqkv/bqkv: fused [3C][C] / [3C] in (q, k, v) order.  Positional modes (fairseq/modules/multihead_attention.py:788-831):
      * pe bf16 [2T-1][C] + wpos: sinusoidal table slice, projected by pos_proj, queries biased by pos_bias_u / pos_bias_v;
      * pe fp32 [2T-1][C] slice of a LEARNED table (wpos None): used as is, plain scaled queries, gradient returned for the slice;
      * pe None: plain attention.
    Head dim 64 without an additive mask runs the fused flash kernels (csrc/flash_attention.hip); anything else the
    GEMM / softmax / GEMM composition."""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, wqkv, bqkv, wo, bo, u, v, wpos, wqkv16, wo16, wpos16, pe, key_len, attn_mask, B, T,
                H, p_attn, p_out, eps, pre_ln, causal=False):
        M, C = x.shape
        dh = C // H
        scaling = dh ** -0.5
        if pre_ln:
            xn, mean, rstd = K.layernorm_fwd(x, ln_g, ln_b, eps)
        else:
            xn, mean, rstd = x, None, None
        qkv = _new((M, 3 * C), torch.bfloat16, x)
        K.gemm(xn, wqkv16, qkv, M, 3 * C, C, lda=C, ldb=C, ldc=3 * C, bias=bqkv)
        relpos = pe is not None
        learned = relpos and wpos is None
        if learned:
            qu, _ = K.relpos_q_prep(qkv, 3 * C, None, None, M, C, scaling, want_qv=False)
            qv = qu
        else:
            qu, qv = K.relpos_q_prep(qkv, 3 * C, u if relpos else None, v if relpos else None, M, C, scaling, want_qv=relpos)
        Z = H * B
        Sp = _pad8(T)
        R = 2 * T - 1
        Rp = _pad8(R)
        pp = None
        if relpos:
            if learned:
                pp = K.cast_f32_to_bf16(pe.detach().contiguous())
            else:
                pp = _new((R, C), torch.bfloat16, x)
                K.gemm(pe, wpos16, pp, R, C, C, lda=C, ldb=C, ldc=C)
        sa = _next_seed("attn.probs", p_attn) if p_attn > 0 else 0
        fused = attn_mask is None and K.flash_attention_supported(dh, T, T, relpos)
        P = Pd = lse = bits = None
        if fused:
            o, lse, bits = K.flash_attention_fwd(qu, qv if relpos else None, qkv[:, C:], qkv[:, 2 * C:], pp, key_len, H, B, T, T, C,
                                                 3 * C, C, causal=causal, drop_p=p_attn, drop_seed=sa, want_bits=True)
        else:
            ac = _new((Z * T, Sp), torch.float32, x)
            K.gemm(qu, qkv, ac, T, T, dh, lda=C, ldb=3 * C, ldc=Sp, batch=Z, zdiv=B, sA=(dh, T * C), sB=(dh, T * 3 * C),
                   b_off=C, sC=(B * T * Sp, T * Sp))
            bd = None
            if relpos:
                bd = _new((Z * T, Rp), torch.float32, x)
                K.gemm(qv, pp, bd, T, R, dh, lda=C, ldb=C, ldc=Rp, batch=Z, zdiv=B, sA=(dh, T * C), sB=(dh, 0),
                       sC=(B * T * Rp, T * Rp))
            P, Pd = K.relpos_softmax_fwd(ac, bd, key_len, attn_mask, H, B, T, T, Sp, Rp, Sp, causal, p_attn, sa)
            del ac, bd
            o = _new((M, C), torch.bfloat16, x)
            K.gemm(Pd, qkv, o, T, dh, T, lda=Sp, ldb=3 * C, ldc=C, b_kstrided=True, batch=Z, zdiv=B, sA=(B * T * Sp, T * Sp),
                   sB=(dh, T * 3 * C), b_off=2 * C, sC=(dh, T * C))
        so = _next_seed("attn.out", p_out) if p_out > 0 else 0
        y = _new((M, C), torch.bfloat16, x)
        K.gemm(o, wo16, y, M, C, C, lda=C, ldb=C, ldc=C, bias=bo, drop_p=p_out, drop_seed=so, resid=x, ldr=C)
        ctx.save_for_backward(x, ln_g, mean, rstd, xn, qkv, qu, qv, pp, P, Pd, o, wqkv16, wo16, wpos16, pe, lse, key_len, bits)
        ctx.cfg = (B, T, H, p_attn, p_out, sa, so, pre_ln, relpos, learned, fused, causal)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, ln_g, mean, rstd, xn, qkv, qu, qv, pp, P, Pd, o, wqkv16, wo16, wpos16, pe, lse, key_len, bits) = ctx.saved_tensors
        B, T, H, p_attn, p_out, sa, so, pre_ln, relpos, learned, fused, causal = ctx.cfg
        M, C = x.shape
        dh = C // H
        scaling = dh ** -0.5
        Z = H * B
        Sp, R = _pad8(T), 2 * T - 1
        Rp = _pad8(R)
        dy = dy.contiguous()
        g = K.scale_dropout(dy, a=1.0, drop_p=p_out, drop_seed=so) if p_out > 0 else dy
        dWo = _wgrad(g, o, M, C, C)
        dbo = K.colsum(g, _zeros_f32(C, x), M, C, C)
        do = _new((M, C), torch.bfloat16, x)
        K.gemm(g, wo16, do, M, C, C, lda=C, ldb=C, ldc=C, b_kstrided=True)
        dqkv = _new((M, 3 * C), torch.bfloat16, x)
        t2 = dBD = None
        if fused:
            t1, t2, dBD = K.flash_attention_bwd(qu, qv if relpos else None, qkv[:, C:], qkv[:, 2 * C:], pp, key_len, o, do, lse,
                                                dqkv[:, C:], dqkv[:, 2 * C:], H, B, T, T, C, 3 * C, 3 * C, ldpp=C, causal=causal,
                                                scaling=scaling, drop_p=p_attn, drop_seed=sa, keep_bits=bits,
                                                dq=dqkv if relpos else None, lddq=3 * C)
        else:
            # dPd[z][i][j] = sum_d do[(b,i),h,d] v[(b,j),h,d]
            dPd = _new((Z * T, Sp), torch.float32, x)
            K.gemm(do, qkv, dPd, T, T, dh, lda=C, ldb=3 * C, ldc=Sp, batch=Z, zdiv=B, sA=(dh, T * C), sB=(dh, T * 3 * C),
                   b_off=2 * C, sC=(B * T * Sp, T * Sp))
            # dV[(b,j),h,d] = sum_i Pd[z][i][j] do[(b,i),h,d]
            K.gemm(Pd, do, dqkv, T, dh, T, lda=Sp, ldb=C, ldc=3 * C, a_kstrided=True, b_kstrided=True, batch=Z, zdiv=B,
                   sA=(B * T * Sp, T * Sp), sB=(dh, T * C), sC=(dh, T * 3 * C), c_off=2 * C)
            dAC, dBD = K.relpos_softmax_bwd(P, dPd, H, B, T, T, Sp, Sp, Rp, want_bd=relpos, drop_p=p_attn, drop_seed=sa)
            del dPd
            # dK[(b,j),h,d] = sum_i dAC[z][i][j] qu[(b,i),h,d]
            K.gemm(dAC, qu, dqkv, T, dh, T, lda=Sp, ldb=C, ldc=3 * C, a_kstrided=True, b_kstrided=True, batch=Z, zdiv=B,
                   sA=(B * T * Sp, T * Sp), sB=(dh, T * C), sC=(dh, T * 3 * C), c_off=C)
            # dq (through qu): s * sum_j dAC[z][i][j] k[(b,j),h,d]
            t1 = _new((M, C), torch.bfloat16, x)
            K.gemm(dAC, qkv, t1, T, dh, T, lda=Sp, ldb=3 * C, ldc=C, b_kstrided=True, batch=Z, zdiv=B, sA=(B * T * Sp, T * Sp),
                   sB=(dh, T * 3 * C), b_off=C, sC=(dh, T * C), alpha=scaling)
            if relpos:
                t2 = _new((M, C), torch.bfloat16, x)
                K.gemm(dBD, pp, t2, T, dh, R, lda=Rp, ldb=C, ldc=C, b_kstrided=True, batch=Z, zdiv=B, sA=(B * T * Rp, T * Rp),
                       sB=(dh, 0), sC=(dh, T * C), alpha=scaling)
        du = dv = dWpos = dpe = None
        if relpos:
            # dpp[r][h*dh+d] = sum_{b,i} dBD[h][(b,i)][r] qv[(b,i),h,d]
            dpp32 = _new((R, C), torch.float32, x)
            K.gemm(dBD, qv, dpp32, R, dh, B * T, lda=Rp, ldb=C, ldc=C, a_kstrided=True, b_kstrided=True, batch=H, zdiv=1,
                   sA=(B * T * Rp, 0), sB=(dh, 0), sC=(dh, 0), splitk=_auto_splitk(((R + 127) // 128) * H, B * T))
            if learned:
                dpe = dpp32
            else:
                dWpos = _wgrad(K.cast_f32_to_bf16(dpp32), pe, R, C, C)
                du = K.colsum(t1, _zeros_f32(C, x), M, C, C)
                dv = K.colsum(t2, _zeros_f32(C, x), M, C, C)
            if not fused:  # (the fused backward wrote dq = t1 + t2 into the q third itself)
                K.add2_strided(t1, C, t2, C, dqkv, 3 * C, M, C)
        else:
            K.add2_strided(t1, C, torch.zeros_like(t1), C, dqkv, 3 * C, M, C)
        dWqkv = _wgrad(dqkv, xn, M, 3 * C, C)
        dbqkv = K.colsum(dqkv, _zeros_f32(3 * C, x), M, 3 * C, 3 * C)
        dxn = _new((M, C), torch.bfloat16, x)
        K.gemm(dqkv, wqkv16, dxn, M, C, 3 * C, lda=3 * C, ldb=C, ldc=C, b_kstrided=True)
        if pre_ln:
            dg, db = _zeros_f32(C, x), _zeros_f32(C, x)
            dx = K.layernorm_bwd(x, dxn, ln_g, mean, rstd, dg, db, dx_add=dy)
        else:
            dg = db = None
            dx = K.scale_dropout(dxn, a=1.0, y=dy, b=1.0)
        return (dx, dg, db, dWqkv, dbqkv, dWo, dbo, du, dv, dWpos, None, None, None, dpe) + (None,) * 10


def relpos_mhsa(x, ln_g, ln_b, wqkv, bqkv, wo, bo, u, v, wpos, pe, key_len, attn_mask, B, T, H, p_attn=0.0, p_out=0.0,
                eps=1e-5, pre_ln=True, wqkv16=None, causal=False):
    return _RelPosMHSA.apply(
        x, ln_g, ln_b, wqkv, bqkv, wo, bo, u, v, wpos, wqkv16 if wqkv16 is not None else bf16_weight(wqkv),
        bf16_weight(wo), bf16_weight(wpos) if wpos is not None else None, pe, key_len, attn_mask, B, T, H, p_attn, p_out,
        eps, pre_ln, causal)


# ------------------------------------------------------------------------------------------------
class _ConvModule(torch.autograd.Function):
    """y = drop(PW2(SiLU(BN(DWConv(GLU(PW1(LN(x))))))) ) + x"""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, wpw1, wdw, bn_g, bn_b, wpw2, wpw1_16, wpw2_16, running_mean, running_var, B, T, KW,
                p_out, eps, bn_eps, bn_momentum, training):
        M, C = x.shape
        xn, mean, rstd = K.layernorm_fwd(x, ln_g, ln_b, eps)
        Y = _new((M, 2 * C), torch.bfloat16, x)
        K.gemm(xn, wpw1_16, Y, M, 2 * C, C, lda=C, ldb=C, ldc=2 * C)
        wdw2 = wdw.detach().reshape(C, KW).contiguous()
        stats = torch.zeros(2 * C, dtype=torch.float64, device=x.device) if training else None
        U, Zt = K.glu_dwconv_fwd(Y, wdw2, B, T, C, KW, stats)
        if training:
            mr = K.bn_finalize(stats, C, M, bn_eps, bn_momentum, running_mean, running_var)
        else:
            mr = K.bn_from_running(running_mean, running_var, bn_eps)
        Hh = K.bn_act_fwd(Zt, mr, bn_g, bn_b, "silu")
        so = _next_seed("conv.out", p_out) if p_out > 0 else 0
        y = _new((M, C), torch.bfloat16, x)
        K.gemm(Hh, wpw2_16, y, M, C, C, lda=C, ldb=C, ldc=C, drop_p=p_out, drop_seed=so, resid=x, ldr=C)
        ctx.save_for_backward(x, ln_g, mean, rstd, xn, Y, U, Zt, mr, Hh, bn_g, bn_b, wdw2, wpw1_16, wpw2_16)
        ctx.cfg = (B, T, KW, p_out, so, training)
        ctx.wdw_shape = wdw.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, ln_g, mean, rstd, xn, Y, U, Zt, mr, Hh, bn_g, bn_b, wdw2, wpw1_16, wpw2_16) = ctx.saved_tensors
        B, T, KW, p_out, so, training = ctx.cfg
        M, C = x.shape
        dy = dy.contiguous()
        g = K.scale_dropout(dy, a=1.0, drop_p=p_out, drop_seed=so) if p_out > 0 else dy
        dWpw2 = _wgrad(g, Hh, M, C, C)
        dH = _new((M, C), torch.bfloat16, x)
        K.gemm(g, wpw2_16, dH, M, C, C, lda=C, ldb=C, ldc=C, b_kstrided=True)
        dbn_g, dbn_b = _zeros_f32(C, x), _zeros_f32(C, x)
        dZ = K.bn_act_bwd(Zt, dH, mr, bn_g, bn_b, dbn_g, dbn_b, "silu", training)
        dwdw = _zeros_f32(C * KW, x)
        dY = K.glu_dwconv_bwd(dZ, Y, U, wdw2, dwdw, B, T, C, KW)
        dWpw1 = _wgrad(dY, xn, M, 2 * C, C)
        dxn = _new((M, C), torch.bfloat16, x)
        K.gemm(dY, wpw1_16, dxn, M, C, 2 * C, lda=2 * C, ldb=C, ldc=C, b_kstrided=True)
        dg, db = _zeros_f32(C, x), _zeros_f32(C, x)
        dx = K.layernorm_bwd(x, dxn, ln_g, mean, rstd, dg, db, dx_add=dy)
        return (dx, dg, db, dWpw1.view(2 * C, C, 1), dwdw.view(ctx.wdw_shape), dbn_g, dbn_b, dWpw2.view(C, C, 1)) + (None,) * 12


def conv_module(x, ln_g, ln_b, wpw1, wdw, bn_g, bn_b, wpw2, running_mean, running_var, B, T, p_out=0.0, eps=1e-5,
                bn_eps=1e-5, bn_momentum=0.1, training=True):
    C = x.shape[1]
    KW = wdw.shape[-1]
    return _ConvModule.apply(x, ln_g, ln_b, wpw1, wdw, bn_g, bn_b, wpw2, bf16_weight(wpw1).view(2 * C, C),
                             bf16_weight(wpw2).view(C, C), running_mean, running_var, B, T, KW, p_out, eps, bn_eps,
                             bn_momentum, training)


# ------------------------------------------------------------------------------------------------
class _CTCLoss(torch.autograd.Function):
    """Per-utterance -log p(target | logits).  logits: bf16 or fp32 [B*T][V] (batch-major rows; may be
    a row-padded view).  Returns (nll [B] fp32, lprobs fp32 [B*T][V]); d/dlogits comes from the
    alpha/beta lattice kept in the workspace."""

    @staticmethod
    def forward(ctx, logits, targets, in_len, tgt_len, B, T, blank, zero_infinity):
        M, V = logits.shape
        assert M == B * T
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        Lmax = targets.shape[1]
        lprobs = K.log_softmax(logits, M, V, logits.stride(0))
        nll, ws = K.ctc_loss_fwd(lprobs, targets, in_len, tgt_len, B, T, V, Lmax, blank)
        ctx.save_for_backward(lprobs, ws, nll, targets, in_len, tgt_len)
        ctx.cfg = (B, T, V, Lmax, blank, zero_infinity, logits.dtype)
        ctx.mark_non_differentiable(lprobs)
        return nll, lprobs

    @staticmethod
    def backward(ctx, dnll, _dlp):
        lprobs, ws, nll, targets, in_len, tgt_len = ctx.saved_tensors
        B, T, V, Lmax, blank, zero_infinity, dt = ctx.cfg
        # The criterion reduces nll with a plain sum, so dnll is one scalar broadcast over B; it is
        # applied on the device from dnll[0] (no host sync).
        scale_dev = dnll.float().contiguous()
        bf = dt == torch.bfloat16
        ld = _pad8(V) if bf else V
        dl = K.ctc_loss_grad(lprobs, ws, nll, targets, in_len, tgt_len, B, T, V, Lmax, blank, ld_out=ld, grad_bf16=bf,
                             grad_scale=1.0, grad_scale_dev=scale_dev, zero_infinity=zero_infinity)
        return (dl if ld == V else dl[:, :V]), None, None, None, None, None, None, None


def ctc_loss(logits, targets, in_len, tgt_len, B, T, blank=0, zero_infinity=True):
    """(nll [B], lprobs [B*T][V]).  NOTE: backward assumes a uniform upstream weight over utterances
    (reduction='sum' as in espresso/criterions/ctc_loss.py:85-94)."""
    return _CTCLoss.apply(logits, targets, in_len, tgt_len, B, T, blank, zero_infinity)


class _LabelSmoothedCE(torch.autograd.Function):
    """(sum loss, sum nll) over non-pad rows — espresso/criterions/label_smoothed_cross_entropy_v2.py:94-119."""

    @staticmethod
    def forward(ctx, logits, target, pad_idx, eps, smoothing="uniform", prior=None, tgt_len=0):
        M, V = logits.shape
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        out, _ = K.label_smoothed_ce(logits, logits.stride(0), target, M, V, pad_idx, eps, want_grad=False, smoothing=smoothing,
                                     prior=prior, tgt_len=tgt_len)
        ctx.save_for_backward(logits, target, prior)
        ctx.cfg = (pad_idx, eps, smoothing, tgt_len)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, dloss, _dnll):
        logits, target, prior = ctx.saved_tensors
        pad_idx, eps, smoothing, tgt_len = ctx.cfg
        M, V = logits.shape
        bf = logits.dtype == torch.bfloat16
        # grad scale read on the host would sync; the loss scale is 1 in fp32/bf16 training, so apply
        # it lazily with the streaming kernel only when a non-unit scale tensor is passed.
        _, dl = K.label_smoothed_ce(logits, logits.stride(0), target, M, V, pad_idx, eps, want_grad=True, grad_bf16=bf,
                                    grad_ld=_pad8(V) if bf else V, smoothing=smoothing, prior=prior, tgt_len=tgt_len)
        g = dl if dl.shape[1] == V else dl[:, :V]
        return g * dloss.to(g.dtype), None, None, None, None, None, None


def label_smoothed_ce(logits, target, pad_idx, eps, smoothing="uniform", prior=None, tgt_len=0):
    """smoothing "uniform" | "unigram" (prior fp32 [V]) | "temporal" (rows are b*tgt_len + u)."""
    return _LabelSmoothedCE.apply(logits, target, pad_idx, eps, smoothing, prior, tgt_len)


_CONV_IGEMM = True  # A/B switch: implicit-GEMM 3x3 convolutions (False: the round-1 im2col + GEMM + col2im lowering)


_CONV1_FUSED_BWD = True


def set_conv1_fused_backward(on: bool) -> bool:
    """First sub-sampler layer: fused BatchNorm-backward + weight-gradient kernel (default) vs separate kernels.  Returns the old setting."""
    global _CONV1_FUSED_BWD
    old, _CONV1_FUSED_BWD = _CONV1_FUSED_BWD, bool(on)
    return old


def set_conv_implicit_gemm(on: bool):
    global _CONV_IGEMM
    _CONV_IGEMM = bool(on)


def _conv_igemm_ok(cin, cout, sy, sx):
    """A layer takes the implicit-GEMM kernels only if ALL THREE of them accept it (forward: Cin % 64, Cout in {64, 128}; data
    gradient: Cin in {64, 128}, Cout % 64, strides <= 2; weight gradient: both % 64): forward and backward must agree on the
    lowering, and anything else (the reference accepts arbitrary channel lists) takes im2col + GEMM + col2im."""
    return _CONV_IGEMM and cin in (64, 128) and cout in (64, 128) and sy in (1, 2) and sx in (1, 2)


# ------------------------------------------------------------------------------------------------
class _ConvSubsample(torch.autograd.Function):
    """ConvBNReLU stack of espresso/modules/speech_convolutions.py:78-102 in channels-last form.

    X fp32 [B][T][F] (in_channels == 1) -> bf16 [B*T'][F'*C_last] with feature index f*C_last + c, padded
    frames zeroed and (training) input dropout of fc0 applied
    (espresso/models/transformer/speech_transformer_encoder.py:341-342).
    params: flat list [w_1, b_1, g_1, beta_1, ..., w_L, b_L, g_L, beta_L]; bufs: [rm_1, rv_1, ...]."""

    @staticmethod
    def forward(ctx, X, row_zero, strides, bufs, p_drop, training, bn_eps, bn_momentum, *params):
        B, T, F = X.shape
        L = len(params) // 4
        saved = []
        cfgs = []
        A = None
        Tc, Fc, Cc = T, F, 1
        for i in range(L):
            w, b, g, beta = params[4 * i: 4 * i + 4]
            rm, rv = bufs[2 * i], bufs[2 * i + 1]
            sy, sx = strides[i]
            Co = w.shape[0]
            To, Fo = (Tc - 1) // sy + 1, (Fc - 1) // sx + 1
            stats = _pool_zeros((2 * Co,), torch.float64, X.device) if training else None
            if i == 0:
                assert w.shape[1] == 1 and tuple(w.shape[2:]) == (3, 3)
                Zi = K.conv1_fwd(X, w.detach().reshape(Co, 9).contiguous(), b, B, Tc, Fc, Co, sy, sx, stats)
                col, w16 = None, None
            elif _conv_igemm_ok(Cc, Co, sy, sx):
                # implicit GEMM (csrc/conv_igemm.hip): the tap tiles are gathered straight from the channels-last activation,
                # BatchNorm sums come out of the epilogue; what backward needs is the layer's INPUT, not a 9x larger im2col matrix
                w16 = K.cast_f32_to_bf16(w.detach().permute(0, 2, 3, 1).reshape(Co, 9 * Cc).contiguous())
                Zi = K.conv3x3_fwd(A, w16, b, B, Tc, Fc, Cc, Co, sy, sx, stats=stats)
                col = A
            else:
                col = K.im2col3x3(A, B, Tc, Fc, Cc, sy, sx)
                w16 = K.cast_f32_to_bf16(w.detach().permute(0, 2, 3, 1).reshape(Co, 9 * Cc).contiguous())
                Zi = _new((B * To * Fo, Co), torch.bfloat16, X)
                K.gemm(col, w16, Zi, B * To * Fo, Co, 9 * Cc, lda=9 * Cc, ldb=9 * Cc, ldc=Co, bias=b)
                if training:
                    K.colstats(Zi, stats)
            n = B * To * Fo
            mr = K.bn_finalize(stats, Co, n, bn_eps, bn_momentum, rm, rv) if training else K.bn_from_running(rm, rv, bn_eps)
            A = K.bn_act_fwd(Zi, mr, g, beta, "relu")
            saved += [Zi, mr, col, w16, g, beta]
            cfgs.append((Tc, Fc, Cc, To, Fo, Co, sy, sx))
            Tc, Fc, Cc = To, Fo, Co
        out = A.view(B * Tc, Fc * Cc)
        seed = _next_seed("subsample.out", p_drop) if (training and p_drop > 0) else 0
        if seed:
            out = K.scale_dropout(out, a=1.0, drop_p=p_drop, drop_seed=seed)
        if row_zero is not None:
            K.zero_rows(out, row_zero)
        ctx.save_for_backward(X, row_zero, *[t for t in saved if t is not None])
        ctx.layout = [[t is not None for t in saved[6 * i: 6 * i + 6]] for i in range(L)]
        ctx.cfg = (B, cfgs, p_drop, seed, training, [tuple(p.shape) for p in params[0::4]])
        ctx.igemm = [i > 0 and _conv_igemm_ok(cfgs[i][2], cfgs[i][5], cfgs[i][6], cfgs[i][7]) for i in range(L)]
        ctx.weights = [params[4 * i] for i in range(L)]
        ctx.params = list(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        X, row_zero, *flat = ctx.saved_tensors
        B, cfgs, p_drop, seed, training, wshapes = ctx.cfg
        L = len(cfgs)
        it = iter(flat)
        per = []
        for i in range(L):
            per.append([next(it) if has else None for has in ctx.layout[i]])
        dout = dout.contiguous()
        if seed:
            dout = K.scale_dropout(dout, a=1.0, drop_p=p_drop, drop_seed=seed)
        elif row_zero is not None:
            dout = dout.clone()
        if row_zero is not None:
            K.zero_rows(dout, row_zero)
        grads = [None] * (4 * L)
        Tc, Fc, Cc, To, Fo, Co, sy, sx = cfgs[-1]
        dA = dout.view(B * To * Fo, Co)
        # weight / bias gradients only feed the optimizer: they run on a side stream next to the data-gradient chain
        # (GEMM + col2im + BatchNorm backward of the next layer down).  Outputs are allocated on the main stream, every
        # tensor a side kernel reads is kept alive until the join below.
        cur = torch.cuda.current_stream(X.device)
        side = _side_stream(X.device)
        keep = []
        for i in range(L - 1, -1, -1):
            Zi, mr, col, w16, g, beta = per[i]
            Tc, Fc, Cc, To, Fo, Co, sy, sx = cfgs[i]
            pw, pb, pg, pbe = ctx.params[4 * i: 4 * i + 4]
            # running statistics (training=False): the conv bias gradient below is derived from THIS call's dbeta, so dbeta must
            # not be the live p.grad view (it may already hold earlier micro-batches: update_freq > 1 / no_sync accumulation)
            sg, sbe = (_grad_sink(pg, Co), _grad_sink(pbe, Co)) if training else (None, None)
            if sg is not None and sbe is not None:  # BatchNorm parameter gradients accumulate straight into p.grad
                dg, dbeta = sg, sbe
            else:
                sg = sbe = None
                dg, dbeta = _zeros_f32(Co, X), _zeros_f32(Co, X)
            if i == 0 and Co % 64 == 0 and _CONV1_FUSED_BWD:
                # first layer: BatchNorm backward + conv1 weight gradient in one pass (csrc/convmodule.hip conv1_bn_bwd_wgrad_kernel):
                # no dZ tensor, and the weight gradient is no longer the last, un-overlapped kernel of the backward pass
                sW, sb = _grad_sink(pw, Co * 9), _grad_sink(pb, Co)
                if sW is not None and sb is not None:
                    db, dW = sb, sW
                else:
                    sW = None
                    db, dW = _zeros_f32(Co, X), _zeros_f32(Co * 9, X)
                K.conv1_bn_bwd(X, Zi, dA, mr, g, beta, dg, dbeta, dW, db, B, Tc, Fc, Co, sy, sx, "relu", training)
                if sW is None:
                    grads[0], grads[1] = dW.view(wshapes[0]), db
                if sg is None:
                    grads[2], grads[3] = dg, dbeta
                continue
            dZ = K.bn_act_bwd(Zi, dA, mr, g, beta, dg, dbeta, "relu", training)
            keep.append(dZ)
            n = B * To * Fo
            db = _zeros_f32(Co, X)
            if i == 0:
                dW = _zeros_f32(Co * 9, X)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    K.conv1_wgrad(X, dZ, dW, db, B, Tc, Fc, Co, sy, sx)
                grads[0] = dW.view(wshapes[0])
            elif ctx.igemm[i]:
                # `col` is the layer's input activation here.  Weight gradient (side stream): transposing-LDS-read kernel over
                # the input and dZ; data gradient: gather GEMM per parity class of the input position.  The bias gradient of a
                # convolution feeding BatchNorm comes from BatchNorm's own sums: sum_p dZ = gamma * rstd * (dbeta - dbeta) = 0 with
                # batch statistics (the batch mean removes the bias), gamma * rstd * dbeta with running statistics.
                sW = _grad_sink(pw, Co * 9 * Cc)
                side.wait_stream(cur)
                if sW is not None:  # the slab reduce scatters into the parameter's [Co][Cc][3][3] gradient itself
                    with torch.cuda.stream(side):
                        K.conv3x3_wgrad(col, dZ, sW, B, Tc, Fc, Cc, Co, sy, sx, param_layout=True)
                else:
                    dWp = _zeros_f32(Co * 9 * Cc, X).view(Co, 9 * Cc)
                    with torch.cuda.stream(side):
                        K.conv3x3_wgrad(col, dZ, dWp, B, Tc, Fc, Cc, Co, sy, sx)
                    grads[4 * i] = dWp.view(Co, 3, 3, Cc).permute(0, 3, 1, 2)
                if not training:
                    db = g.detach() * mr[1] * dbeta
                elif _grad_sink(pb, Co) is not None:
                    db = None  # exactly zero with batch statistics: nothing to add to p.grad
                wd16 = K.cast_f32_to_bf16(ctx.weights[i].detach().permute(1, 2, 3, 0).reshape(Cc, 9 * Co).contiguous())
                dA = K.conv3x3_dgrad(dZ, wd16, B, Tc, Fc, Cc, Co, sy, sx)
            else:
                dWp = torch.empty((Co, 9 * Cc), dtype=torch.float32, device=X.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    _wgrad(dZ, col, n, Co, 9 * Cc, out=dWp)
                    K.colsum(dZ, db, n, Co, Co)
                grads[4 * i] = dWp.view(Co, 3, 3, Cc).permute(0, 3, 1, 2)
                dcol = _new((n, 9 * Cc), torch.bfloat16, X)
                K.gemm(dZ, w16, dcol, n, 9 * Cc, Co, lda=Co, ldb=9 * Cc, ldc=9 * Cc, b_kstrided=True)
                dA = K.col2im3x3(dcol, B, Tc, Fc, Cc, sy, sx)
            grads[4 * i + 1] = db
            if sg is None:
                grads[4 * i + 2] = dg
                grads[4 * i + 3] = dbeta
        cur.wait_stream(side)
        del keep
        return (None,) * 8 + tuple(grads)


def conv_subsample(X, row_zero, strides, params, bufs, p_drop=0.0, training=True, bn_eps=1e-5, bn_momentum=0.1):
    return _ConvSubsample.apply(X, row_zero, tuple(strides), list(bufs), p_drop, training, bn_eps, bn_momentum, *params)


# ------------------------------------------------------------------------------------------------
# Native layer runtime (csrc/engine.hip): the whole Conformer layer forward / backward as ONE C-ABI call.
_grad_ready_callback = None


def set_grad_ready_callback(fn):
    """fn(list_of_parameters) is invoked when a native backward has finished accumulating the gradients
    of those parameters (the data-parallel wrapper uses it to launch bucket all-reduces)."""
    global _grad_ready_callback
    _grad_ready_callback = fn


def _ptr(t):
    return None if t is None else t.data_ptr()


class _LayerBinding:
    """ctypes view of one ConformerWithRelativePositionalEmbeddingEncoderLayer's parameters / gradients.
    Keeps the tensors it points to alive; rebuilt when the underlying storage changes."""

    def __init__(self, m, kind="conformer"):
        import ctypes

        from ._lib import EaConformerLayer

        self.keep = []
        self.kind = kind
        L = EaConformerLayer()

        def w16(p, shape=None):
            t = bf16_weight(p)
            t = t.reshape(shape) if shape is not None else t
            self.keep.append(t)
            return t.data_ptr()

        def grad(p):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            return p.grad.data_ptr()

        def ffn(dst, gdst, f):
            dst.ln_g, dst.ln_b = _ptr(f.layer_norm.weight), _ptr(f.layer_norm.bias)
            dst.w1, dst.b1, dst.w2, dst.b2 = w16(f.w_1.weight), _ptr(f.w_1.bias), w16(f.w_2.weight), _ptr(f.w_2.bias)
            gdst.ln_g, gdst.ln_b = grad(f.layer_norm.weight), grad(f.layer_norm.bias)
            gdst.w1, gdst.b1, gdst.w2, gdst.b2 = grad(f.w_1.weight), grad(f.w_1.bias), grad(f.w_2.weight), grad(f.w_2.bias)

        if kind == "conformer":
            ffn(L.ffn1, L.grads.ffn1, m.ffn1)
            ffn(L.ffn2, L.grads.ffn2, m.ffn2)
            # k-contiguous copies of eight weights for the backward data-gradient GEMMs (refreshed by every training forward)
            Cm, Fm = m.embed_dim, m.ffn1.w_1.weight.shape[0]
            self.wt = torch.empty(4 * Cm * Fm + 7 * Cm * Cm, dtype=torch.bfloat16, device=m.ffn1.w_1.weight.device)
        else:  # Transformer layer: (final_layer_norm, fc1, fc2) is the FFN block
            f1, gf = L.ffn1, L.grads.ffn1
            f1.ln_g, f1.ln_b = _ptr(m.final_layer_norm.weight), _ptr(m.final_layer_norm.bias)
            f1.w1, f1.b1, f1.w2, f1.b2 = w16(m.fc1.weight), _ptr(m.fc1.bias), w16(m.fc2.weight), _ptr(m.fc2.bias)
            gf.ln_g, gf.ln_b = grad(m.final_layer_norm.weight), grad(m.final_layer_norm.bias)
            gf.w1, gf.b1, gf.w2, gf.b2 = grad(m.fc1.weight), grad(m.fc1.bias), grad(m.fc2.weight), grad(m.fc2.bias)
            Cm, Fm = m.embed_dim, m.fc1.weight.shape[0]
            self.wt = torch.empty(2 * Cm * Fm + 4 * Cm * Cm, dtype=torch.bfloat16, device=m.fc1.weight.device)
        L.wt = self.wt.data_ptr()
        a = m.self_attn
        qw, kw, vw = a.q_proj.weight, a.k_proj.weight, a.v_proj.weight
        qb, kb, vb = a.q_proj.bias, a.k_proj.bias, a.v_proj.bias
        for p in (qw, kw, vw, qb, kb, vb):
            grad(p)
        fused16 = getattr(qw, "_ea_fused_qkv", None)
        n = qw.numel()
        grads_contig = (qw.grad is not None and kw.grad is not None and vw.grad is not None
                        and kw.grad.data_ptr() == qw.grad.data_ptr() + 4 * n and vw.grad.data_ptr() == kw.grad.data_ptr() + 4 * n)
        if fused16 is not None and grads_contig:
            wqkv16, self.split_w = fused16, None
            gw = qw.grad.data_ptr()
        else:  # unfused storage: work on packed copies, scatter the gradient back after backward
            wqkv16 = torch.cat([bf16_weight(qw), bf16_weight(kw), bf16_weight(vw)], 0)
            gbuf = torch.zeros(3 * qw.shape[0], qw.shape[1], dtype=torch.float32, device=qw.device)
            self.split_w = (gbuf, (qw, kw, vw))
            gw = gbuf.data_ptr()
        self.keep.append(wqkv16)
        nb = qb.numel()
        bias_contig = (kb.data_ptr() == qb.data_ptr() + 4 * nb and vb.data_ptr() == kb.data_ptr() + 4 * nb
                       and qb.grad is not None and kb.grad is not None and vb.grad is not None
                       and kb.grad.data_ptr() == qb.grad.data_ptr() + 4 * nb and vb.grad.data_ptr() == kb.grad.data_ptr() + 4 * nb)
        if bias_contig:
            self.split_b = None
            bq_ptr, gb_ptr = qb.data_ptr(), qb.grad.data_ptr()
        else:
            bq = torch.cat([qb.detach(), kb.detach(), vb.detach()], 0)
            gb = torch.zeros_like(bq)
            self.split_b = (gb, (qb, kb, vb))
            self.keep += [bq, gb]
            bq_ptr, gb_ptr = bq.data_ptr(), gb.data_ptr()
        self.cacheable = self.split_w is None and self.split_b is None
        self.key = (qw.data_ptr(), qw.grad.data_ptr() if qw.grad is not None else 0)
        A, GA = L.attn, L.grads.attn
        A.ln_g, A.ln_b = _ptr(m.self_attn_layer_norm.weight), _ptr(m.self_attn_layer_norm.bias)
        A.wqkv, A.bqkv, A.wo, A.bo = wqkv16.data_ptr(), bq_ptr, w16(a.out_proj.weight), _ptr(a.out_proj.bias)
        GA.ln_g, GA.ln_b = grad(m.self_attn_layer_norm.weight), grad(m.self_attn_layer_norm.bias)
        GA.wqkv, GA.bqkv, GA.wo, GA.bo = gw, gb_ptr, grad(a.out_proj.weight), grad(a.out_proj.bias)
        if a.pos_proj is not None:  # sinusoidal table projected per layer; a learned table needs none of these
            A.pos_u, A.pos_v, A.wpos = _ptr(a.pos_bias_u), _ptr(a.pos_bias_v), w16(a.pos_proj.weight)
            GA.pos_u, GA.pos_v, GA.wpos = grad(a.pos_bias_u), grad(a.pos_bias_v), grad(a.pos_proj.weight)
        for p in (qw, kw, vw, qb, kb, vb):
            grad(p)
        self.saved_buf, self.saved_busy = None, False
        self.L = L
        if kind != "conformer":
            # parameters whose gradients the runtime writes itself (a learned positional table gets its gradient through
            # autograd instead and must not be reported as ready here)
            mods = [a.q_proj, a.k_proj, a.v_proj, a.out_proj, m.self_attn_layer_norm, m.fc1, m.fc2, m.final_layer_norm]
            self.params = [p for mod in mods for p in mod.parameters()]
            if a.pos_proj is not None:
                self.params += [a.pos_bias_u, a.pos_bias_v, a.pos_proj.weight]
            return
        cm = m.conv_module
        Cc, GC = L.conv, L.grads.conv
        Cdim = m.embed_dim
        Cc.ln_g, Cc.ln_b = _ptr(cm.layer_norm.weight), _ptr(cm.layer_norm.bias)
        Cc.pw1 = w16(cm.pointwise_conv1.weight, (2 * Cdim, Cdim))
        Cc.dw = _ptr(cm.depthwise_conv.weight)
        Cc.bn_g, Cc.bn_b = _ptr(cm.batch_norm.weight), _ptr(cm.batch_norm.bias)
        Cc.bn_rm, Cc.bn_rv = _ptr(cm.batch_norm.running_mean), _ptr(cm.batch_norm.running_var)
        Cc.pw2 = w16(cm.pointwise_conv2.weight, (Cdim, Cdim))
        GC.ln_g, GC.ln_b = grad(cm.layer_norm.weight), grad(cm.layer_norm.bias)
        GC.pw1, GC.dw = grad(cm.pointwise_conv1.weight), grad(cm.depthwise_conv.weight)
        GC.bn_g, GC.bn_b, GC.pw2 = grad(cm.batch_norm.weight), grad(cm.batch_norm.bias), grad(cm.pointwise_conv2.weight)
        L.final_ln_g, L.final_ln_b = _ptr(m.final_layer_norm.weight), _ptr(m.final_layer_norm.bias)
        L.grads.final_ln_g, L.grads.final_ln_b = grad(m.final_layer_norm.weight), grad(m.final_layer_norm.bias)
        self.L = L
        self.params = [p for p in m.parameters()]
        self.saved_buf, self.saved_busy = None, False

    def finish_backward(self):
        """Scatter packed q/k/v gradients back to their parameters and signal readiness."""
        if self.split_w is not None:
            gbuf, ps = self.split_w
            n = ps[0].shape[0]
            for i, p in enumerate(ps):
                p.grad += gbuf[i * n:(i + 1) * n]
        if self.split_b is not None:
            gb, ps = self.split_b
            n = ps[0].shape[0]
            for i, p in enumerate(ps):
                p.grad += gb[i * n:(i + 1) * n]
        if _grad_ready_callback is not None:
            _grad_ready_callback(self.params)


def _conformer_binding(module):
    bind = getattr(module, "_ea_binding", None)
    qw = module.self_attn.q_proj.weight
    if bind is None or bind.key != (qw.data_ptr(), qw.grad.data_ptr() if qw.grad is not None else 0):
        bind = _LayerBinding(module)
        module._ea_binding = bind if bind.cacheable else None
    return bind


_WT_PREFETCH = os.environ.get("EA_WT_PREFETCH", "1") != "0"  # (A/B switch)


def refresh_layer_transposes(layers, B, T):
    """Before a TRAINING forward pass over native Conformer layers: the transposed (k-contiguous) copies of every layer's weights —
    read only by the backward pass — are refreshed on the Python-side stream, under the sub-sampler's / first layers' forward
    kernels, instead of by each layer's forward call on the compute stream (12 launches of 12 us per update step at the recipe
    size).  Returns the event the compute stream must wait for before the backward pass (None: nothing was done).  The layers'
    next forward call is told through `_ea_wt_fresh` (consumed there; a binding rebuilt in between invalidates it)."""
    import ctypes

    from . import _lib
    from ._lib import EaLayerShape

    layers = list(layers)
    if not _WT_PREFETCH or not layers:
        return None
    dev = layers[0].ffn1.w_1.weight.device
    if dev.type != "cuda":
        return None
    binds = [_conformer_binding(m) for m in layers]
    if not all(b.cacheable for b in binds):
        return None
    cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
    side.wait_stream(cur)  # the bf16 weight shadows were written by the optimizer step on the compute stream
    lib = _lib.lib()
    for m, bind in zip(layers, binds):
        sh = EaLayerShape()
        sh.B, sh.T, sh.C, sh.H = B, T, m.embed_dim, m.num_heads
        sh.F = m.ffn1.w_1.weight.shape[0]
        sh.KW = m.conv_module.depthwise_conv.weight.shape[-1]
        sh.training = 1
        _lib.check(lib.ea_conformer_layer_refresh_wt(ctypes.byref(bind.L), ctypes.byref(sh), ctypes.c_void_p(side.cuda_stream)),
                   "ea_conformer_layer_refresh_wt")
        m.__dict__["_ea_wt_fresh"] = bind
    return side.record_event()


_LAYER_CHAIN = os.environ.get("EA_LAYER_CHAIN", "1") != "0"  # (A/B switch)


def set_layer_chain(on: bool):
    """A/B switch: chained Conformer layer calls (one kernel for layer k's final LayerNorm + layer k+1's first, both passes)."""
    global _LAYER_CHAIN
    _LAYER_CHAIN = bool(on)


_chain_pre = {}


def _chain_pre_buffer(like):
    """The buffer a chained backward call fills for the PREVIOUS layer's ffn2 block (EaLayerChain.prev_pre).  That layer's deferred
    side work (its weight-gradient launch reads the buffer) is only joined at the end of the backward call after its own, so the
    buffer must outlive two more calls: three persistent buffers per shape, used in rotation — not a pooled temporary, which the
    allocator would hand out again while the side stream is still reading it."""
    key = (str(like.device), like.numel(), like.dtype)
    ring = _chain_pre.get(key)
    if ring is None:
        ring = _chain_pre[key] = [[torch.empty_like(like) for _ in range(3)], 0]
    ring[1] = (ring[1] + 1) % 3
    return ring[0][ring[1]]


def _chain_key(sh):
    return (sh.B, sh.T, sh.C, sh.H, sh.F, sh.KW, sh.training, sh.has_attn_mask, sh.p_drop, sh.p_act, sh.p_attn)


class _ConformerLayerNative(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, key_len, attn_mask, pe, B, T, p_drop, p_act, p_attn, training, next_module=None):
        import ctypes

        from . import _lib
        from ._lib import EaLayerChain, EaLayerShape

        bind = _conformer_binding(module)
        sh = EaLayerShape()
        sh.B, sh.T, sh.C, sh.H = B, T, module.embed_dim, module.num_heads
        sh.F = module.ffn1.w_1.weight.shape[0]
        sh.KW = module.conv_module.depthwise_conv.weight.shape[-1]
        sh.training = int(training)
        # (the encoder refreshed this layer's transposed weight copies on a side stream before its layer loop: refresh_layer_transposes)
        sh.wt_fresh = int(bool(training) and module.__dict__.pop("_ea_wt_fresh", None) is bind)
        sh.p_drop, sh.p_act, sh.p_attn = p_drop, p_act, p_attn
        sh.seed = _layer_seed("conformer", p_drop, p_act, p_attn)
        sh.has_attn_mask = int(attn_mask is not None)
        nb_saved, nb_scratch = ctypes.c_long(0), ctypes.c_long(0)
        lib = _lib.lib()
        _lib.check(lib.ea_conformer_layer_workspace(ctypes.byref(sh), ctypes.byref(nb_saved), ctypes.byref(nb_scratch)), "workspace")
        # saved activations live in a grow-only arena owned by the layer binding: no allocator traffic in steady state
        # (a fresh private buffer only when the previous forward of this layer is still waiting for its backward)
        needs_bwd = ctx.needs_input_grad[0]
        if bind.saved_busy or not bind.cacheable:
            saved = torch.empty(nb_saved.value, dtype=torch.uint8, device=x.device)
            ctx.owns_arena = False
        else:
            if bind.saved_buf is None or bind.saved_buf.numel() < nb_saved.value or bind.saved_buf.device != x.device:
                bind.saved_buf = torch.empty(int(nb_saved.value * 1.1) + 4096, dtype=torch.uint8, device=x.device)
            saved = bind.saved_buf
            bind.saved_busy = bool(needs_bwd)
            ctx.owns_arena = bool(needs_bwd)
        scratch = _scratch_buffer(nb_scratch.value, x.device)
        _scratch_forward_touch(str(x.device))  # the forward overwrites what a backward pass left in the arena
        y = torch.empty_like(x)
        stream = K._stream()
        # Chained calls (EaLayerChain): the previous layer's call left this layer's first LayerNorm in the binding's arena — valid only
        # if this call's input IS that call's output (the token keeps the tensor alive, so the address identifies it) and the arena is
        # the one it wrote to; and this call does the same for `next_module` (the encoder passes it when the layer's output has no
        # other consumer).
        ch = EaLayerChain()
        token = module.__dict__.pop("_ea_chain_in", None)
        ctx.chain_prev = None
        if (token is not None and token[0] is bind and saved is bind.saved_buf and token[1].data_ptr() == x.data_ptr()
                and token[2] == _chain_key(sh)):
            ch.ln1_done = 1
            if needs_bwd:
                ctx.chain_prev = token[3]  # the previous layer's autograd context (None: it records no backward)
        if next_module is not None and _LAYER_CHAIN and sh.C <= 512 and next_module is not module:
            nbind = _conformer_binding(next_module)
            if nbind.cacheable and not nbind.saved_busy:
                if nbind.saved_buf is None or nbind.saved_buf.numel() < nb_saved.value or nbind.saved_buf.device != x.device:
                    nbind.saved_buf = torch.empty(int(nb_saved.value * 1.1) + 4096, dtype=torch.uint8, device=x.device)
                ch.next = ctypes.addressof(nbind.L)
                ch.next_saved, ch.next_saved_bytes = nbind.saved_buf.data_ptr(), nbind.saved_buf.numel()
                next_module.__dict__["_ea_chain_in"] = (nbind, y, _chain_key(sh), ctx if needs_bwd else None)
        _lib.check(lib.ea_conformer_layer_fwd_chained(ctypes.byref(bind.L), ctypes.byref(sh), ctypes.byref(ch), _ptr(x), _ptr(y), _ptr(key_len),
                                                      _ptr(attn_mask), _ptr(pe), _ptr(saved), saved.numel(), _ptr(scratch), scratch.numel(),
                                                      stream), "ea_conformer_layer_fwd")
        ctx.save_for_backward(x, saved, pe, key_len)
        ctx.bind, ctx.sh, ctx.nb_scratch = bind, sh, nb_scratch.value
        ctx.chain_saved, ctx.chain_in = saved, None
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes

        from . import _lib

        x, saved, pe, key_len = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        scratch = _scratch_buffer(ctx.nb_scratch, x.device)
        sh = ctx.sh
        # everything the arena layout depends on (shape, which optional buffers exist, runtime switches via the byte count)
        tag = (scratch.data_ptr(), ctx.nb_scratch, sh.B, sh.T, sh.C, sh.H, sh.F, sh.KW, sh.training, sh.has_attn_mask,
               sh.p_drop > 0, sh.p_act > 0, sh.p_attn > 0)
        dev = str(x.device)
        half = _native_bwd_begin(sh, dev, tag, deferrable=ctx.owns_arena)
        stream = K._stream()
        ch = _lib.EaLayerChain()
        if ctx.chain_in is not None:  # the next layer's chained call already ran this layer's final-LayerNorm backward
            ch.final_ln_done, ch.pre_in = 1, ctx.chain_in.data_ptr()
        prev = ctx.chain_prev
        if prev is not None and _LAYER_CHAIN and _chain_key(prev.sh) == _chain_key(sh):
            # ... and this call ends with the previous layer's: `dx` is then the gradient w.r.t. ITS pre-final-norm activations
            pre = _chain_pre_buffer(x)
            ch.prev, ch.prev_saved, ch.prev_saved_bytes = ctypes.addressof(prev.bind.L), prev.chain_saved.data_ptr(), prev.chain_saved.numel()
            ch.prev_seed, ch.prev_pre = prev.sh.seed, pre.data_ptr()
            prev.chain_in = pre
        ctx.chain_prev = None
        _lib.check(_lib.lib().ea_conformer_layer_bwd_chained(ctypes.byref(ctx.bind.L), ctypes.byref(ctx.sh), ctypes.byref(ch), _ptr(x), _ptr(dy),
                                                             _ptr(dx), _ptr(key_len), _ptr(pe), _ptr(saved), saved.numel(), _ptr(scratch),
                                                             scratch.numel(), stream), "ea_conformer_layer_bwd")
        ctx.chain_in = None
        if ctx.owns_arena:
            ctx.bind.saved_busy = False
        _native_bwd_end(ctx.bind, dev, half, stream)
        return (dx,) + (None,) * 11


_LAYER_STACK = os.environ.get("EA_LAYER_STACK", "1") != "0"  # (A/B switch)


def set_layer_stack(on: bool):
    """A/B switch: a run of native Conformer layers as ONE C call per direction (ea_conformer_stack_fwd / _bwd)."""
    global _LAYER_STACK
    _LAYER_STACK = bool(on)


def conformer_stack_supported(modules, x) -> bool:
    """Can this run of native Conformer layers go through the stack call?  Needs the bindings' own arenas (cached bindings, none
    busy with an earlier forward that still awaits its backward), one shared positional table, and no per-layer gradient-ready
    callback (the overlapped data-parallel wrapper launches a bucket's all-reduce as soon as ITS layers are done: that needs the
    Python loop; on one rank nothing listens)."""
    if not _LAYER_STACK or len(modules) < 2 or not x.is_cuda or _grad_ready_callback is not None:
        return False
    if len({id(m) for m in modules}) != len(modules):
        return False
    pe0 = modules[0].positional_embedding[0]
    for m in modules:
        if m.positional_embedding[0] is not pe0 or m.embed_dim != modules[0].embed_dim or m.cfg is not modules[0].cfg:
            return False
        bind = _conformer_binding(m)
        if not bind.cacheable or bind.saved_busy:
            return False
    return True


class _ConformerStackNative(torch.autograd.Function):
    """n native Conformer layers, one C call per direction (include/espresso_amd.h EaStackLayer).  Same launches, dropout seeds and
    bookkeeping as n `_ConformerLayerNative` nodes in a row."""

    @staticmethod
    def forward(ctx, x, modules, key_len, attn_mask, pe, B, T, p_drop, p_act, p_attn, training):
        import ctypes

        from . import _lib
        from ._lib import EaLayerShape, EaStackLayer

        n = len(modules)
        lib = _lib.lib()
        needs_bwd = ctx.needs_input_grad[0]
        arr = (EaStackLayer * n)()
        binds = [_conformer_binding(m) for m in modules]
        m0 = modules[0]
        xs = torch.empty((n - 1,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        nb_saved, nb_scratch = ctypes.c_long(0), ctypes.c_long(0)
        for k, (m, bind) in enumerate(zip(modules, binds)):
            sh = arr[k].shape
            sh.B, sh.T, sh.C, sh.H = B, T, m.embed_dim, m.num_heads
            sh.F = m.ffn1.w_1.weight.shape[0]
            sh.KW = m.conv_module.depthwise_conv.weight.shape[-1]
            sh.training = int(training)
            sh.wt_fresh = int(bool(training) and m.__dict__.pop("_ea_wt_fresh", None) is bind)
            sh.p_drop, sh.p_act, sh.p_attn = p_drop, p_act, p_attn
            sh.seed = _layer_seed("conformer", p_drop, p_act, p_attn)
            sh.has_attn_mask = int(attn_mask is not None)
            if k == 0:
                _lib.check(lib.ea_conformer_layer_workspace(ctypes.byref(sh), ctypes.byref(nb_saved), ctypes.byref(nb_scratch)), "workspace")
            if bind.saved_buf is None or bind.saved_buf.numel() < nb_saved.value or bind.saved_buf.device != x.device:
                bind.saved_buf = torch.empty(int(nb_saved.value * 1.1) + 4096, dtype=torch.uint8, device=x.device)
            bind.saved_busy = bool(needs_bwd)
            m.__dict__.pop("_ea_chain_in", None)
            arr[k].layer = ctypes.addressof(bind.L)
            arr[k].saved, arr[k].saved_bytes = bind.saved_buf.data_ptr(), bind.saved_buf.numel()
            arr[k].x_in = x.data_ptr() if k == 0 else xs[k - 1].data_ptr()
        scratch = _scratch_buffer(nb_scratch.value, x.device)
        _scratch_forward_touch(str(x.device))
        y = torch.empty_like(x)
        stream = K._stream()
        chain = int(_LAYER_CHAIN and m0.embed_dim <= 512)
        _lib.check(lib.ea_conformer_stack_fwd(arr, n, _ptr(y), _ptr(key_len), _ptr(attn_mask), _ptr(pe), _ptr(scratch), scratch.numel(), chain,
                                              stream), "ea_conformer_stack_fwd")
        ctx.save_for_backward(x, xs, pe, key_len)
        ctx.arr, ctx.binds, ctx.n, ctx.nb_scratch, ctx.chain = arr, binds, n, nb_scratch.value, chain
        ctx.keep = [b.saved_buf for b in binds]
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes

        from . import _lib

        x, xs, pe, key_len = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        n, arr = ctx.n, ctx.arr
        scratch = _scratch_buffer(ctx.nb_scratch, x.device)
        dev = str(x.device)
        halves = [None] * n
        for k in range(n - 1, -1, -1):  # the bookkeeping of n consecutive layer calls: alternating halves, clean-scratch flags
            sh = arr[k].shape
            tag = (scratch.data_ptr(), ctx.nb_scratch, sh.B, sh.T, sh.C, sh.H, sh.F, sh.KW, sh.training, sh.has_attn_mask,
                   sh.p_drop > 0, sh.p_act > 0, sh.p_attn > 0)
            halves[k] = _native_bwd_begin(sh, dev, tag, deferrable=True)
        dbuf = torch.empty((2,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        dptr = (ctypes.c_void_p * 2)(dbuf[0].data_ptr(), dbuf[1].data_ptr())
        pre = [_chain_pre_buffer(x) for _ in range(3)] if ctx.chain else None
        pptr = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in pre]) if pre else None
        stream = K._stream()
        _lib.check(_lib.lib().ea_conformer_stack_bwd(arr, n, _ptr(dy), _ptr(dx), dptr, pptr, _ptr(key_len), _ptr(pe), _ptr(scratch),
                                                     scratch.numel(), ctx.chain, stream), "ea_conformer_stack_bwd")
        for k in range(n - 1, -1, -1):
            ctx.binds[k].saved_busy = False
            _native_bwd_end(ctx.binds[k], dev, halves[k], stream)
        return (dx,) + (None,) * 10


def conformer_stack_native(x, modules, key_len, attn_mask, pe, B, T, p_drop, p_act, p_attn, training):
    """The layer loop over `modules` (native Conformer layers, conformer_stack_supported) as one autograd node."""
    return _ConformerStackNative.apply(x, list(modules), key_len, attn_mask, pe, B, T, p_drop, p_act, p_attn, training)


_scratch = {}
_scratch_tag = {}

# ---- deferred side work of the native layer backward (csrc/engine.hip "Deferred") ------------------------------------------
# A backward call in deferred mode returns with its optimizer-only work (all weight / bias / norm-parameter gradients of the
# layer, one grouped GEMM launch) still queued on the side stream; the NEXT backward call joins it.  So the binding whose
# gradients became complete — and may be reported to the data-parallel wrapper — is the one of the PREVIOUS call, and the last
# layer of a backward pass is finished by a callback the autograd engine runs at the end of that pass.
_pending = {}        # device -> (binding awaiting the join, raw stream handle)
_bwd_half = {}       # device -> scratch half the next deferred call uses
_defer_enabled = True


def _scratch_forward_touch(dev):
    _scratch_tag[dev] = None
    _scratch_tag.pop((dev, 0), None)
    _scratch_tag.pop((dev, 1), None)


def set_backward_deferred(on: bool):
    """A/B switch (also tells the library): False = every layer backward forks and joins its side work inside the call."""
    global _defer_enabled
    from . import _lib

    _defer_enabled = bool(on)
    _lib.lib().ea_set_backward_deferred(int(on))


def _native_bwd_begin(sh, dev, tag, deferrable):
    """Fills sh.defer / sh.scratch_clean for the backward call that follows; returns the scratch half or None (immediate mode).
    Deferred mode needs the saved-activation arena to outlive the call (the binding's grow-only arena, not a private buffer).
    (A learned positional table's gradient `dpe`, consumed by autograd right after the call, is computed on the main stream.)"""
    deferred = _defer_enabled and deferrable
    if not deferred:
        sh.defer = 0
        sh.scratch_clean = int(_scratch_tag.get(dev) == tag)  # consecutive layers of one backward pass share the layout
        _scratch_tag[dev] = tag
        _scratch_tag.pop((dev, 0), None)
        _scratch_tag.pop((dev, 1), None)
        return None
    half = _bwd_half.get(dev, 0)
    _bwd_half[dev] = half ^ 1
    sh.defer = 1 + half
    sh.scratch_clean = int(_scratch_tag.get((dev, half)) == tag)
    _scratch_tag[(dev, half)] = tag
    _scratch_tag[dev] = None
    return half


def _flush_native_backward():
    """End of a backward pass (autograd engine callback), or explicit: join the last layer's side work and report it."""
    from . import _lib

    for dev in list(_pending):
        bind, stream = _pending.pop(dev)
        _lib.check(_lib.lib().ea_backward_flush(stream), "ea_backward_flush")
        bind.finish_backward()


def _native_bwd_end(bind, dev, half, stream):
    prev = _pending.pop(dev, None)
    if half is None:  # immediate mode: the library joined everything before running this call
        if prev is not None:
            prev[0].finish_backward()
        bind.finish_backward()
        return
    if prev is not None:
        prev[0].finish_backward()  # this call ended with the main stream joining the previous call's side work
    _pending[dev] = (bind, stream)
    # runs when the engine has finished this backward pass (queued by every deferred call: a pass that died half-way must not
    # leave the next one without its flush; all but the first invocation find nothing pending)
    torch.autograd.Variable._execution_engine.queue_callback(_flush_native_backward)


def _scratch_buffer(nbytes, device):
    """One reusable scratch arena per device (kernels of consecutive layers run in stream order)."""
    key = str(device)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


class _TransformerLayerNative(torch.autograd.Function):
    """Whole pre-LN Transformer encoder layer (rel-pos MHA block + FFN block) per C-ABI call: ea_transformer_layer_fwd / _bwd.
    `pe`: bf16 constant sinusoidal slice, or the fp32 autograd slice of a learned table (its gradient is returned)."""

    @staticmethod
    def forward(ctx, x, pe, module, key_len, attn_mask, B, T, p_drop, p_act, p_attn, training, act):
        import ctypes

        from . import _lib
        from ._lib import EaLayerShape

        bind = getattr(module, "_ea_binding", None)
        qw = module.self_attn.q_proj.weight
        if bind is None or bind.key != (qw.data_ptr(), qw.grad.data_ptr() if qw.grad is not None else 0):
            bind = _LayerBinding(module, kind="transformer")
            module._ea_binding = bind if bind.cacheable else None
        learned = pe.dtype == torch.float32
        pe16 = K.cast_f32_to_bf16(pe.detach().contiguous()) if learned else pe
        sh = EaLayerShape()
        sh.B, sh.T, sh.C, sh.H = B, T, module.embed_dim, module.num_heads
        sh.F, sh.KW = module.fc1.weight.shape[0], 0
        sh.training = int(training)
        sh.p_drop, sh.p_act, sh.p_attn = p_drop, p_act, p_attn
        sh.seed = _layer_seed("transformer", p_drop, p_act, p_attn)
        sh.has_attn_mask = int(attn_mask is not None)
        sh.pos_mode = 1 if learned else 0
        sh.act = K._ACT[act] if isinstance(act, str) else int(act)
        nb_saved, nb_scratch = ctypes.c_long(0), ctypes.c_long(0)
        lib = _lib.lib()
        _lib.check(lib.ea_transformer_layer_workspace(ctypes.byref(sh), ctypes.byref(nb_saved), ctypes.byref(nb_scratch)), "workspace")
        needs_bwd = ctx.needs_input_grad[0]
        if bind.saved_busy or not bind.cacheable:
            saved = torch.empty(nb_saved.value, dtype=torch.uint8, device=x.device)
            ctx.owns_arena = False
        else:
            if bind.saved_buf is None or bind.saved_buf.numel() < nb_saved.value or bind.saved_buf.device != x.device:
                bind.saved_buf = torch.empty(int(nb_saved.value * 1.1) + 4096, dtype=torch.uint8, device=x.device)
            saved = bind.saved_buf
            bind.saved_busy = bool(needs_bwd)
            ctx.owns_arena = bool(needs_bwd)
        scratch = _scratch_buffer(nb_scratch.value, x.device)
        _scratch_forward_touch(str(x.device))
        y = torch.empty_like(x)
        stream = K._stream()
        _lib.check(lib.ea_transformer_layer_fwd(ctypes.byref(bind.L), ctypes.byref(sh), _ptr(x), _ptr(y), _ptr(key_len), _ptr(attn_mask),
                                                _ptr(pe16), _ptr(saved), saved.numel(), _ptr(scratch), scratch.numel(), stream),
                   "ea_transformer_layer_fwd")
        ctx.save_for_backward(x, saved, pe16, key_len)
        ctx.bind, ctx.sh, ctx.nb_scratch, ctx.learned = bind, sh, nb_scratch.value, learned
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes

        from . import _lib

        x, saved, pe16, key_len = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dpe = torch.empty(pe16.shape, dtype=torch.float32, device=x.device) if ctx.learned else None
        scratch = _scratch_buffer(ctx.nb_scratch, x.device)
        sh = ctx.sh
        tag = ("transformer", scratch.data_ptr(), ctx.nb_scratch, sh.B, sh.T, sh.C, sh.H, sh.F, sh.training, sh.has_attn_mask,
               sh.pos_mode, sh.p_drop > 0, sh.p_act > 0, sh.p_attn > 0)
        dev = str(x.device)
        half = _native_bwd_begin(sh, dev, tag, deferrable=ctx.owns_arena)
        stream = K._stream()
        _lib.check(_lib.lib().ea_transformer_layer_bwd(ctypes.byref(ctx.bind.L), ctypes.byref(sh), _ptr(x), _ptr(dy), _ptr(dx), _ptr(key_len),
                                                       _ptr(pe16), _ptr(dpe), _ptr(saved), saved.numel(), _ptr(scratch), scratch.numel(),
                                                       stream), "ea_transformer_layer_bwd")
        if ctx.owns_arena:
            ctx.bind.saved_busy = False
        _native_bwd_end(ctx.bind, dev, half, stream)
        return (dx, dpe) + (None,) * 10


def transformer_layer_native(x, pe, module, key_len, attn_mask, B, T, p_drop, p_act, p_attn, training, act):
    return _TransformerLayerNative.apply(x, pe, module, key_len, attn_mask, B, T, p_drop, p_act, p_attn, training, act)


def _adjacent(ts):
    """True when the tensors lie back to back in memory (same dtype), in the given order."""
    for a, b in zip(ts[:-1], ts[1:]):
        if a is None or b is None or b.data_ptr() != a.data_ptr() + a.numel() * a.element_size():
            return False
    return True


class _DecoderLayerBinding:
    """ctypes view of a TransformerDecoderLayer for ea_decoder_layer_fwd/bwd.  Needs the FlatParams layout (q, k, v weights,
    biases, their bf16 shadows and gradients adjacent): `ok` is False otherwise and the module keeps the per-kernel path."""

    def __init__(self, m):
        from ._lib import EaDecoderLayer

        self.ok = False
        self.keep = []
        L = EaDecoderLayer()
        sa, ca = m.self_attn, m.encoder_attn
        groups = {"self": [sa.q_proj, sa.k_proj, sa.v_proj], "cross": [ca.k_proj, ca.v_proj]}
        every = [p for mod in (sa.q_proj, sa.k_proj, sa.v_proj, sa.out_proj, ca.q_proj, ca.k_proj, ca.v_proj, ca.out_proj, m.fc1, m.fc2,
                               m.self_attn_layer_norm, m.encoder_attn_layer_norm, m.final_layer_norm) for p in mod.parameters()]
        if any(p.grad is None or getattr(p, "_ea_bf16", None) is None for p in every if p.dim() == 2) or any(p.grad is None for p in every):
            return
        for g in groups.values():
            ws, bs = [x.weight for x in g], [x.bias for x in g]
            if not (_adjacent([w._ea_bf16 for w in ws]) and _adjacent([w.grad for w in ws]) and _adjacent(bs)
                    and _adjacent([b.grad for b in bs])):
                return
        w16 = lambda p: p._ea_bf16.data_ptr()  # noqa: E731
        gp = lambda p: p.grad.data_ptr()  # noqa: E731
        S, GS = L.self_attn, L.g_self
        S.ln_g, S.ln_b = _ptr(m.self_attn_layer_norm.weight), _ptr(m.self_attn_layer_norm.bias)
        S.wqkv, S.bqkv, S.wo, S.bo = w16(sa.q_proj.weight), _ptr(sa.q_proj.bias), w16(sa.out_proj.weight), _ptr(sa.out_proj.bias)
        GS.ln_g, GS.ln_b = gp(m.self_attn_layer_norm.weight), gp(m.self_attn_layer_norm.bias)
        GS.wqkv, GS.bqkv, GS.wo, GS.bo = gp(sa.q_proj.weight), gp(sa.q_proj.bias), gp(sa.out_proj.weight), gp(sa.out_proj.bias)
        X, GX = L.cross, L.g_cross
        X.ln_g, X.ln_b = _ptr(m.encoder_attn_layer_norm.weight), _ptr(m.encoder_attn_layer_norm.bias)
        X.wq, X.bq, X.wkv, X.bkv = w16(ca.q_proj.weight), _ptr(ca.q_proj.bias), w16(ca.k_proj.weight), _ptr(ca.k_proj.bias)
        X.wo, X.bo = w16(ca.out_proj.weight), _ptr(ca.out_proj.bias)
        GX.ln_g, GX.ln_b = gp(m.encoder_attn_layer_norm.weight), gp(m.encoder_attn_layer_norm.bias)
        GX.wq, GX.bq, GX.wkv, GX.bkv = gp(ca.q_proj.weight), gp(ca.q_proj.bias), gp(ca.k_proj.weight), gp(ca.k_proj.bias)
        GX.wo, GX.bo = gp(ca.out_proj.weight), gp(ca.out_proj.bias)
        F_, GF = L.ffn, L.g_ffn
        F_.ln_g, F_.ln_b = _ptr(m.final_layer_norm.weight), _ptr(m.final_layer_norm.bias)
        F_.w1, F_.b1, F_.w2, F_.b2 = w16(m.fc1.weight), _ptr(m.fc1.bias), w16(m.fc2.weight), _ptr(m.fc2.bias)
        GF.ln_g, GF.ln_b = gp(m.final_layer_norm.weight), gp(m.final_layer_norm.bias)
        GF.w1, GF.b1, GF.w2, GF.b2 = gp(m.fc1.weight), gp(m.fc1.bias), gp(m.fc2.weight), gp(m.fc2.bias)
        C, Fd = m.embed_dim, m.fc1.weight.shape[0]
        self.wt = torch.empty(2 * C * Fd + 8 * C * C, dtype=torch.bfloat16, device=m.fc1.weight.device)
        L.wt = self.wt.data_ptr()
        self.L, self.params = L, every
        self.key = (sa.q_proj.weight.data_ptr(), sa.q_proj.weight.grad.data_ptr())
        self.saved_buf, self.saved_busy = None, False
        self.ok = True

    def finish_backward(self):
        if _grad_ready_callback is not None:
            _grad_ready_callback(self.params)


def decoder_layer_binding(module):
    """Cached binding of a decoder layer, or None when the parameters are not in the flat layout (then: per-kernel path)."""
    bind = getattr(module, "_ea_binding", None)
    qw = module.self_attn.q_proj.weight
    if bind is None or bind.key != (qw.data_ptr(), qw.grad.data_ptr() if qw.grad is not None else 0):
        bind = _DecoderLayerBinding(module)
        if not bind.ok:
            return None
        module._ea_binding = bind
    return bind


class _DecoderLayerNative(torch.autograd.Function):
    """Whole pre-LN Transformer decoder layer (causal self-attention, encoder-decoder attention, FFN) per C-ABI call."""

    @staticmethod
    def forward(ctx, x, enc, bind, module, enc_len, B, U, S, p_drop, p_act, p_attn, training, act):
        import ctypes

        from . import _lib
        from ._lib import EaLayerShape

        sh = EaLayerShape()
        sh.B, sh.T, sh.S, sh.C, sh.H = B, U, S, module.embed_dim, module.num_heads
        sh.F, sh.KW = module.fc1.weight.shape[0], 0
        sh.training = int(training)
        sh.p_drop, sh.p_act, sh.p_attn = p_drop, p_act, p_attn
        sh.seed = _layer_seed("decoder", p_drop, p_act, p_attn)
        sh.act = K._ACT[act] if isinstance(act, str) else int(act)
        nb_saved, nb_scratch = ctypes.c_long(0), ctypes.c_long(0)
        lib = _lib.lib()
        _lib.check(lib.ea_decoder_layer_workspace(ctypes.byref(sh), ctypes.byref(nb_saved), ctypes.byref(nb_scratch)), "workspace")
        needs_bwd = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if bind.saved_busy:
            saved = torch.empty(nb_saved.value, dtype=torch.uint8, device=x.device)
            ctx.owns_arena = False
        else:
            if bind.saved_buf is None or bind.saved_buf.numel() < nb_saved.value or bind.saved_buf.device != x.device:
                bind.saved_buf = torch.empty(int(nb_saved.value * 1.1) + 4096, dtype=torch.uint8, device=x.device)
            saved = bind.saved_buf
            bind.saved_busy = bool(needs_bwd)
            ctx.owns_arena = bool(needs_bwd)
        scratch = _scratch_buffer(nb_scratch.value, x.device)
        _scratch_forward_touch(str(x.device))
        y = torch.empty_like(x)
        stream = K._stream()
        _lib.check(lib.ea_decoder_layer_fwd(ctypes.byref(bind.L), ctypes.byref(sh), _ptr(x), _ptr(enc), _ptr(y), _ptr(enc_len), _ptr(saved),
                                            saved.numel(), _ptr(scratch), scratch.numel(), stream), "ea_decoder_layer_fwd")
        ctx.save_for_backward(x, enc, saved, enc_len)
        ctx.bind, ctx.sh, ctx.nb_scratch = bind, sh, nb_scratch.value
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes

        from . import _lib

        x, enc, saved, enc_len = ctx.saved_tensors
        dy = dy.contiguous()
        dx, denc = torch.empty_like(x), torch.empty_like(enc)
        scratch = _scratch_buffer(ctx.nb_scratch, x.device)
        stream = K._stream()
        _lib.check(_lib.lib().ea_decoder_layer_bwd(ctypes.byref(ctx.bind.L), ctypes.byref(ctx.sh), _ptr(x), _ptr(enc), _ptr(dy), _ptr(dx),
                                                   _ptr(denc), _ptr(enc_len), _ptr(saved), saved.numel(), _ptr(scratch), scratch.numel(),
                                                   stream), "ea_decoder_layer_bwd")
        _scratch_forward_touch(str(x.device))
        if ctx.owns_arena:
            ctx.bind.saved_busy = False
        ctx.bind.finish_backward()
        return (dx, denc) + (None,) * 11


def decoder_layer_native(x, enc, bind, module, enc_len, B, U, S, p_drop, p_act, p_attn, training, act):
    return _DecoderLayerNative.apply(x, enc, bind, module, enc_len, B, U, S, p_drop, p_act, p_attn, training, act)


def conformer_layer_native(x, module, key_len, attn_mask, pe, B, T, p_drop, p_act, p_attn, training, next_module=None):
    """`next_module`: the native Conformer layer that consumes this call's output AND IS ITS ONLY CONSUMER (a layer stack without
    collected hidden states): the two calls are chained (EaLayerChain in espresso_amd.h)."""
    return _ConformerLayerNative.apply(x, module, key_len, attn_mask, pe, B, T, p_drop, p_act, p_attn, training, next_module)


# ------------------------------------------------------------------------------------------------
class _CrossMHA(torch.autograd.Function):
    """Encoder-decoder attention of fairseq/modules/transformer_layer.py:470-497 with its pre-LayerNorm, dropout and
    residual: y = drop(out_proj(Attn(q = LN(x), k = v = enc))) + x.   x: [B*U][C], enc: [B*S][C] (batch-major rows).
    wkv/bkv: fused [2C][C] / [2C] in (k, v) order."""

    @staticmethod
    def forward(ctx, x, enc, ln_g, ln_b, wq, bq, wkv, bkv, wo, bo, wq16, wkv16, wo16, key_len, B, U, S, H, p_attn, p_out, eps):
        M, C = x.shape
        Ms = enc.shape[0]
        dh = C // H
        scaling = dh ** -0.5
        xn, mean, rstd = K.layernorm_fwd(x, ln_g, ln_b, eps)
        q = _new((M, C), torch.bfloat16, x)
        K.gemm(xn, wq16, q, M, C, C, lda=C, ldb=C, ldc=C, bias=bq)
        qs, _ = K.relpos_q_prep(q, C, None, None, M, C, scaling, want_qv=False)
        kv = _new((Ms, 2 * C), torch.bfloat16, x)
        K.gemm(enc, wkv16, kv, Ms, 2 * C, C, lda=C, ldb=C, ldc=2 * C, bias=bkv)
        Z = H * B
        Sp = _pad8(S)
        ac = _new((Z * U, Sp), torch.float32, x)
        K.gemm(qs, kv, ac, U, S, dh, lda=C, ldb=2 * C, ldc=Sp, batch=Z, zdiv=B, sA=(dh, U * C), sB=(dh, S * 2 * C),
               sC=(B * U * Sp, U * Sp))
        sa = _next_seed("attn.probs", p_attn) if p_attn > 0 else 0
        P, Pd = K.relpos_softmax_fwd(ac, None, key_len, None, H, B, U, S, Sp, 0, Sp, False, p_attn, sa)
        del ac
        o = _new((M, C), torch.bfloat16, x)
        K.gemm(Pd, kv, o, U, dh, S, lda=Sp, ldb=2 * C, ldc=C, b_kstrided=True, batch=Z, zdiv=B, sA=(B * U * Sp, U * Sp),
               sB=(dh, S * 2 * C), b_off=C, sC=(dh, U * C))
        so = _next_seed("attn.out", p_out) if p_out > 0 else 0
        y = _new((M, C), torch.bfloat16, x)
        K.gemm(o, wo16, y, M, C, C, lda=C, ldb=C, ldc=C, bias=bo, drop_p=p_out, drop_seed=so, resid=x, ldr=C)
        ctx.save_for_backward(x, enc, ln_g, mean, rstd, xn, qs, kv, P, Pd, o, wq16, wkv16, wo16)
        ctx.cfg = (B, U, S, H, p_attn, p_out, sa, so)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, enc, ln_g, mean, rstd, xn, qs, kv, P, Pd, o, wq16, wkv16, wo16 = ctx.saved_tensors
        B, U, S, H, p_attn, p_out, sa, so = ctx.cfg
        M, C = x.shape
        Ms = enc.shape[0]
        dh = C // H
        scaling = dh ** -0.5
        Z = H * B
        Sp = _pad8(S)
        dy = dy.contiguous()
        g = K.scale_dropout(dy, a=1.0, drop_p=p_out, drop_seed=so) if p_out > 0 else dy
        dWo = _wgrad(g, o, M, C, C)
        dbo = K.colsum(g, _zeros_f32(C, x), M, C, C)
        do = _new((M, C), torch.bfloat16, x)
        K.gemm(g, wo16, do, M, C, C, lda=C, ldb=C, ldc=C, b_kstrided=True)
        dPd = _new((Z * U, Sp), torch.float32, x)
        K.gemm(do, kv, dPd, U, S, dh, lda=C, ldb=2 * C, ldc=Sp, batch=Z, zdiv=B, sA=(dh, U * C), sB=(dh, S * 2 * C),
               b_off=C, sC=(B * U * Sp, U * Sp))
        dkv = _new((Ms, 2 * C), torch.bfloat16, x)
        # dV[(b,j),h,d] = sum_i Pd[z][i][j] do[(b,i),h,d]
        K.gemm(Pd, do, dkv, S, dh, U, lda=Sp, ldb=C, ldc=2 * C, a_kstrided=True, b_kstrided=True, batch=Z, zdiv=B,
               sA=(B * U * Sp, U * Sp), sB=(dh, U * C), sC=(dh, S * 2 * C), c_off=C)
        dAC, _ = K.relpos_softmax_bwd(P, dPd, H, B, U, S, Sp, Sp, 0, want_bd=False, drop_p=p_attn, drop_seed=sa)
        del dPd
        # dK[(b,j),h,d] = sum_i dAC[z][i][j] qs[(b,i),h,d]
        K.gemm(dAC, qs, dkv, S, dh, U, lda=Sp, ldb=C, ldc=2 * C, a_kstrided=True, b_kstrided=True, batch=Z, zdiv=B,
               sA=(B * U * Sp, U * Sp), sB=(dh, U * C), sC=(dh, S * 2 * C))
        # dq = s * sum_j dAC[z][i][j] k[(b,j),h,d]
        dq = _new((M, C), torch.bfloat16, x)
        K.gemm(dAC, kv, dq, U, dh, S, lda=Sp, ldb=2 * C, ldc=C, b_kstrided=True, batch=Z, zdiv=B, sA=(B * U * Sp, U * Sp),
               sB=(dh, S * 2 * C), sC=(dh, U * C), alpha=scaling)
        dWq = _wgrad(dq, xn, M, C, C)
        dbq = K.colsum(dq, _zeros_f32(C, x), M, C, C)
        dWkv = _wgrad(dkv, enc, Ms, 2 * C, C)
        dbkv = K.colsum(dkv, _zeros_f32(2 * C, x), Ms, 2 * C, 2 * C)
        denc = _new((Ms, C), torch.bfloat16, x)
        K.gemm(dkv, wkv16, denc, Ms, C, 2 * C, lda=2 * C, ldb=C, ldc=C, b_kstrided=True)
        dxn = _new((M, C), torch.bfloat16, x)
        K.gemm(dq, wq16, dxn, M, C, C, lda=C, ldb=C, ldc=C, b_kstrided=True)
        dg, db = _zeros_f32(C, x), _zeros_f32(C, x)
        dx = K.layernorm_bwd(x, dxn, ln_g, mean, rstd, dg, db, dx_add=dy)
        return (dx, denc, dg, db, dWq, dbq, dWkv, dbkv, dWo, dbo) + (None,) * 11


def cross_mha(x, enc, ln_g, ln_b, wq, bq, wkv, bkv, wo, bo, key_len, B, U, S, H, p_attn=0.0, p_out=0.0, eps=1e-5):
    return _CrossMHA.apply(x, enc, ln_g, ln_b, wq, bq, wkv, bkv, wo, bo, bf16_weight(wq), bf16_weight(wkv), bf16_weight(wo),
                           key_len, B, U, S, H, p_attn, p_out, eps)


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, tokens, positions, pos_table, scale, pad_idx):
        out = K.embedding_fwd(tokens, positions, W, pos_table, scale)
        ctx.save_for_backward(tokens)
        ctx.cfg = (W.shape, scale, pad_idx)
        return out

    @staticmethod
    def backward(ctx, dy):
        (tokens,) = ctx.saved_tensors
        shape, scale, pad_idx = ctx.cfg
        dW = torch.zeros(shape, dtype=torch.float32, device=dy.device)
        K.embedding_bwd(tokens, dy.contiguous(), dW, scale, pad_idx)
        return dW, None, None, None, None, None


def embedding(W, tokens, positions=None, pos_table=None, scale=1.0, pad_idx=-1):
    """bf16 [M][C] = scale * W[tokens] + pos_table[positions] ; tokens/positions int32 [M]."""
    return _Embedding.apply(W, tokens, positions, pos_table, scale, pad_idx)


class _AddPositions(torch.autograd.Function):
    """bf16 [M][C] = scale * x + table[positions] — the absolute positional embedding of the legacy encoder presets
    (espresso/models/transformer/speech_transformer_encoder.py:345-347; positions from fairseq utils.make_positions: 1..len on
    frames, padding_idx on padded frames).  A one-off element-wise pass outside the layer stack: plain tensor ops."""

    @staticmethod
    def forward(ctx, x, table, positions, scale):
        ctx.save_for_backward(positions)
        ctx.scale, ctx.tshape, ctx.tgrad = scale, table.shape, table.requires_grad
        return (x.float() * scale + table.float().index_select(0, positions)).to(torch.bfloat16)

    @staticmethod
    def backward(ctx, dy):
        (positions,) = ctx.saved_tensors
        dt = None
        if ctx.tgrad:
            dt = torch.zeros(ctx.tshape, dtype=torch.float32, device=dy.device).index_add_(0, positions, dy.float())
        return (dy if ctx.scale == 1.0 else (dy.float() * ctx.scale).to(dy.dtype)), dt, None, None


def add_positions(x, table, positions, scale=1.0):
    return _AddPositions.apply(x, table, positions, scale)


class _ZeroRows(torch.autograd.Function):
    """x with the rows marked in row_zero (uint8 [M]) set to 0 (padded frames, speech_transformer_encoder.py:353-355)."""

    @staticmethod
    def forward(ctx, x, row_zero):
        ctx.save_for_backward(row_zero)
        y = x.contiguous().clone()
        K.zero_rows(y, row_zero)
        return y

    @staticmethod
    def backward(ctx, dy):
        (row_zero,) = ctx.saved_tensors
        g = dy.contiguous().clone()
        K.zero_rows(g, row_zero)
        return g, None


def zero_rows(x, row_zero):
    return _ZeroRows.apply(x, row_zero)


def dropout(x, p):
    """FairseqDropout on a bf16 activation (mask re-derived from the seed in backward)."""
    return _Dropout.apply(x, p)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        seed = _next_seed("dropout", p)
        ctx.cfg = (p, seed)
        return K.scale_dropout(x.contiguous(), a=1.0, drop_p=p, drop_seed=seed)

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.cfg
        return K.scale_dropout(dy.contiguous(), a=1.0, drop_p=p, drop_seed=seed), None


class _RNNTLoss(torch.autograd.Function):
    """Per-utterance transducer negative log-likelihood (torchaudio.functional.rnnt_loss semantics, reduction 'none')."""

    @staticmethod
    def forward(ctx, logits, targets, logit_lengths, target_lengths, blank):
        if logits.dtype != torch.bfloat16:
            logits = logits.float()
        st = logits.stride()
        if not (st[3] == 1 and st[2] >= logits.shape[3] and st[1] == logits.shape[2] * st[2] and st[0] == logits.shape[1] * st[1]):
            logits = logits.contiguous()  # (a [..., :V] view of rows with a padded pitch is used as it is: no copy of B*T*U*V)
        loss, ws = K.rnnt_loss_fwd(logits, targets, logit_lengths, target_lengths, blank)
        ctx.save_for_backward(logits, targets, logit_lengths, target_lengths, loss, ws)
        ctx.blank = blank
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, targets, logit_lengths, target_lengths, loss, ws = ctx.saved_tensors
        # the criterion sums the per-utterance losses: dloss is a broadcast scalar, applied on the device
        g = K.rnnt_loss_grad(logits, targets, logit_lengths, target_lengths, loss, ws, ctx.blank,
                             grad_scale_dev=dloss.float().contiguous(), grad_bf16=logits.dtype == torch.bfloat16)
        return g, None, None, None, None


def rnnt_loss(logits, targets, logit_lengths, target_lengths, blank=0):
    return _RNNTLoss.apply(logits, targets, logit_lengths, target_lengths, blank)


# ------------------------------------------------------------------------------------------------ LSTM
_LSTM_PERSISTENT = True
_lstm_counters = collections.deque(maxlen=64)  # barrier scratch (most recent launches) of the persistent launches ([1] != 0: a wait timed out); checked by lstm_barrier_timeouts()


def set_lstm_persistent(on: bool) -> bool:
    """Persistent whole-sequence LSTM kernels (default) vs one recurrent GEMM + cell kernel per step.  Returns the old setting."""
    global _LSTM_PERSISTENT
    old, _LSTM_PERSISTENT = _LSTM_PERSISTENT, bool(on)
    return old


def lstm_barrier_timeouts() -> int:
    """Number of persistent LSTM launches since the last call whose grid barrier gave up (synchronises; diagnostics / tests)."""
    n = sum(int(c[1].item() != 0) for c in _lstm_counters)
    _lstm_counters.clear()
    return n


class _LSTMLayer(torch.autograd.Function):
    """One LSTM layer over a whole (teacher-forced) sequence — the time loop of
    espresso/models/speech_lstm.py:846-893 for one `LSTMCell` (fairseq/models/lstm.py:LSTMCell = torch.nn.LSTMCell), and one
    direction of the packed `nn.LSTM` of the BiLSTM encoder (:470-520).

    x: bf16 [U*B][I] time-major rows (t*B + b).  Returns hs bf16 [U*B][H] (and the final (h, c) fp32).
    reverse: walk t = U-1 .. 0.  frozen: uint8 [U][B], 1 where t >= length[b] (packed-sequence semantics: the state of such a
    row does not advance, its output is 0 and it receives no gradient; requires zero initial state).
    The input projection of ALL steps is one GEMM; each step then costs one small recurrent GEMM (fp32 output with the
    input projection as fp32 residual) + one element-wise cell kernel.  Backward walks the steps in the opposite order with
    one recurrent dgrad GEMM per step and computes every weight gradient with a single GEMM over all steps."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih16, w_hh16, h0, c0, B, U, reverse=False, frozen=None):
        H = w_hh.shape[1]
        I = w_ih.shape[1]
        dev = x.device
        x = x.contiguous()
        bias = (b_ih + b_hh).detach().float().contiguous() if b_ih is not None else None
        gx = torch.empty(U * B, 4 * H, dtype=torch.float32, device=dev)
        K.gemm(x, w_ih16, gx, U * B, 4 * H, I, lda=I, ldb=I, ldc=4 * H, bias=bias)
        hs = torch.empty(U * B, H, dtype=torch.bfloat16, device=dev)
        cs = torch.empty(U, B, H, dtype=torch.float32, device=dev)
        act = torch.empty(U, B, 4 * H, dtype=torch.float32, device=dev)
        G = torch.empty(B, 4 * H, dtype=torch.float32, device=dev)
        h_last = torch.empty(B, H, dtype=torch.float32, device=dev)
        h0_16 = K.cast_f32_to_bf16(h0.float().contiguous()) if h0 is not None else None
        c0 = c0.float().contiguous() if c0 is not None else None
        assert frozen is None or (h0 is None and c0 is None), "packed sequences start from the zero state"
        order = range(U - 1, -1, -1) if reverse else range(U)
        prev = None
        persistent = _LSTM_PERSISTENT and K.lstm_seq_supported(B, H)
        if persistent:  # the whole time loop in one launch (csrc/lstm_seq.hip)
            counter = torch.empty(2, dtype=torch.int32, device=dev)
            K.lstm_seq_fwd(gx, w_hh16, h0_16, c0, frozen, hs, cs, act, h_last, counter, B, U, H, reverse=reverse,
                           frozen_out_zero=frozen is not None)
            prev = 0 if reverse else U - 1
            order = ()
            _lstm_counters.append(counter)
        for n, t in enumerate(order):
            if prev is None and h0_16 is None:
                Gt = gx[t * B:(t + 1) * B]
            else:
                hp = h0_16 if prev is None else hs[prev * B:(prev + 1) * B]
                K.gemm(hp, w_hh16, G, B, 4 * H, H, lda=H, ldb=H, ldc=4 * H, resid=gx, ldr=4 * H, r_off=t * B * 4 * H)
                Gt = G
            K.lstm_cell_fwd(Gt, c0 if prev is None else cs[prev], cs[t], h_last if n == U - 1 else None, hs[t * B:(t + 1) * B], H,
                            act[t], B, H, keep_row=frozen[t] if frozen is not None else None, frozen_out_zero=frozen is not None)
            prev = t
        ctx.save_for_backward(x, hs, cs, act, w_ih16, w_hh16, h0_16, c0, frozen)
        ctx.dims = (B, U, H, I, b_ih is not None, h0 is not None, reverse)
        ctx.persistent = persistent
        return hs, h_last, cs[prev]

    @staticmethod
    def backward(ctx, dhs, dh_last, dc_last):
        x, hs, cs, act, w_ih16, w_hh16, h0_16, c0, frozen = ctx.saved_tensors
        B, U, H, I, has_bias, has_h0, reverse = ctx.dims
        dev = x.device
        dhs = dhs.contiguous() if dhs is not None else None
        dG = torch.empty(U * B, 4 * H, dtype=torch.bfloat16, device=dev)
        dh_rec = dh_last.float().contiguous().clone() if dh_last is not None else None
        dc = dc_last.float().contiguous().clone() if dc_last is not None else None
        order = list(range(U - 1, -1, -1) if reverse else range(U))  # forward processing order
        sk_rec = max(1, min(16, (4 * H) // 512)) if B <= 64 else 1
        steps = range(U - 1, -1, -1)
        if ctx.persistent:
            counter = torch.empty(2, dtype=torch.int32, device=dev)
            dh0 = torch.empty(B, H, dtype=torch.float32, device=dev) if has_h0 else None
            dc0 = torch.empty(B, H, dtype=torch.float32, device=dev) if c0 is not None else None
            K.lstm_seq_bwd(dhs, dh_rec, dc, act, cs, c0, frozen, w_hh16.t().contiguous(), dG, dh0, dc0, counter, B, U, H,
                           reverse=reverse)
            dh_rec, dc = dh0, dc0
            steps = ()
            _lstm_counters.append(counter)
        for n in steps:
            t = order[n]
            tp = order[n - 1] if n > 0 else None  # the step whose state fed this one
            dc_new = torch.empty(B, H, dtype=torch.float32, device=dev)
            K.lstm_cell_bwd(dhs[t * B:(t + 1) * B] if dhs is not None else None, H, dh_rec, dc, act[t],
                            c0 if tp is None else cs[tp], cs[t], dG[t * B:(t + 1) * B], 4 * H, dc_new, B, H,
                            frozen=frozen[t] if frozen is not None else None)
            dc = dc_new
            if tp is not None or has_h0:
                # gradient of the recurrent input h_{prev} = dG_t W_hh
                # (a handful of rows, 4H-long reduction, H/128 output tiles: split the reduction so that more than H/128
                # workgroups share the 4H x H weight read — 57 -> ~15 us per step at H = 1024)
                dh_rec = torch.empty(B, H, dtype=torch.float32, device=dev)
                K.gemm(dG, w_hh16, dh_rec, B, H, 4 * H, lda=4 * H, ldb=H, ldc=H, b_kstrided=True, a_off=t * B * 4 * H,
                       splitk=sk_rec)
        dx = torch.empty(U * B, I, dtype=torch.bfloat16, device=dev)
        K.gemm(dG, w_ih16, dx, U * B, I, 4 * H, lda=4 * H, ldb=I, ldc=I, b_kstrided=True)
        dw_ih = _wgrad(dG, x, U * B, 4 * H, I)
        # dW_hh = sum_t dG_t^T h_{prev(t)}: in forward order steps 1..U-1 pair with hs of steps 0..U-2 (reverse: dG rows of
        # steps 0..U-2 pair with hs rows 1..U-1); the first processed step pairs with h0 (zero when absent)
        dw_hh = torch.zeros(4 * H, H, dtype=torch.float32, device=dev)
        if U > 1:
            if reverse:
                K.gemm(dG, hs, dw_hh, 4 * H, H, (U - 1) * B, lda=4 * H, ldb=H, ldc=H, a_kstrided=True, b_kstrided=True, b_off=B * H)
            else:
                K.gemm(dG, hs, dw_hh, 4 * H, H, (U - 1) * B, lda=4 * H, ldb=H, ldc=H, a_kstrided=True, b_kstrided=True, a_off=B * 4 * H)
        if has_h0:
            K.gemm(dG, h0_16, dw_hh, 4 * H, H, B, lda=4 * H, ldb=H, ldc=H, a_kstrided=True, b_kstrided=True, accumulate=True,
                   a_off=order[0] * B * 4 * H)
        db = K.colsum(dG, torch.zeros(4 * H, dtype=torch.float32, device=dev), U * B, 4 * H, 4 * H) if has_bias else None
        return (dx, dw_ih, dw_hh, db, db.clone() if db is not None else None, None, None,
                dh_rec if has_h0 else None, dc if c0 is not None else None, None, None, None, None)


def lstm_layer(x, cell, B, U, h0=None, c0=None):
    """x bf16 [U*B][I] time-major; cell: LSTMCellParams.  Returns (hs bf16 [U*B][H], h_last fp32, c_last fp32)."""
    return _LSTMLayer.apply(x, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, bf16_weight(cell.weight_ih),
                            bf16_weight(cell.weight_hh), h0, c0, B, U)

def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, B, U, reverse=False, frozen=None):
    """One direction of a packed nn.LSTM layer (weights named weight_ih_l0[_reverse] ...).  Returns hs bf16 [U*B][H]."""
    return _LSTMLayer.apply(x, w_ih, w_hh, b_ih, b_hh, bf16_weight(w_ih), bf16_weight(w_hh), None, None, B, U, reverse, frozen)[0]


class _GradSink(torch.autograd.Function):
    """Identity on its tensor arguments whose backward returns gradients that later (per-step) backward passes have
    ACCUMULATED IN PLACE into `acc[i]` (fp32 buffers created on first use).  It sits upstream of every per-step node of a
    recurrent decoder, so autograd runs it after all of them: per-step weight / key / value gradients then cost one
    accumulate-GEMM or one in-place kernel each instead of a fresh full-size tensor and an add per step."""

    @staticmethod
    def forward(ctx, holder, *tensors):
        ctx.holder = holder
        holder.shapes = [(t.shape, t.device) for t in tensors]
        holder.acc = [None] * len(tensors)
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        out = []
        for a, g, (shape, dev) in zip(ctx.holder.acc, grads, ctx.holder.shapes):
            if a is None:
                out.append(g)
            else:
                out.append(a.view(shape) if g is None else a.view(shape) + g.to(a.dtype))
        return (None,) + tuple(out)


class GradSink:
    def __init__(self, *tensors):
        self.acc, self.shapes = None, None
        self.views = _GradSink.apply(self, *tensors)

    def buf(self, i):
        if self.acc[i] is None:
            shape, dev = self.shapes[i]
            self.acc[i] = torch.zeros(shape, dtype=torch.float32, device=dev)
        return self.acc[i]


class _LSTMCellStep(torch.autograd.Function):
    """One LSTMCell step with autograd (attention decoders, where every step depends on the previous context).
    x16 bf16 [B][I], h_prev16 bf16 [B][H], c_prev fp32 [B][H] -> (h16 bf16, c fp32).  Weight / bias gradients are accumulated
    into the GradSink `sink` (views order: w_ih, w_hh, b_ih, b_hh)."""

    @staticmethod
    def forward(ctx, x16, h_prev16, c_prev, w_ih_v, w_hh_v, b_ih_v, b_hh_v, w_ih16, w_hh16, bias_sum, sink):
        B, I = x16.shape
        H = w_hh16.shape[1]
        dev = x16.device
        x16, h_prev16 = x16.contiguous(), h_prev16.contiguous()
        G = torch.empty(B, 4 * H, dtype=torch.float32, device=dev)
        K.gemm(x16, w_ih16, G, B, 4 * H, I, lda=I, ldb=I, ldc=4 * H, bias=bias_sum)
        K.gemm(h_prev16, w_hh16, G, B, 4 * H, H, lda=H, ldb=H, ldc=4 * H, resid=G, ldr=4 * H)
        h16 = torch.empty(B, H, dtype=torch.bfloat16, device=dev)
        c = torch.empty(B, H, dtype=torch.float32, device=dev)
        act = torch.empty(B, 4 * H, dtype=torch.float32, device=dev)
        c_prev = c_prev.contiguous()
        K.lstm_cell_fwd(G, c_prev, c, None, h16, H, act, B, H)
        ctx.save_for_backward(x16, h_prev16, c_prev, c, act, w_ih16, w_hh16)
        ctx.sink = sink
        return h16, c

    @staticmethod
    def backward(ctx, dh, dc):
        x16, h_prev16, c_prev, c, act, w_ih16, w_hh16 = ctx.saved_tensors
        B, I = x16.shape
        H = w_hh16.shape[1]
        dev = x16.device
        dG = torch.empty(B, 4 * H, dtype=torch.bfloat16, device=dev)
        dc_prev = torch.empty(B, H, dtype=torch.float32, device=dev)
        K.lstm_cell_bwd(dh.contiguous() if dh is not None else None, H, None, dc.contiguous() if dc is not None else None, act, c_prev, c,
                        dG, 4 * H, dc_prev, B, H)
        dx = torch.empty(B, I, dtype=torch.bfloat16, device=dev)
        K.gemm(dG, w_ih16, dx, B, I, 4 * H, lda=4 * H, ldb=I, ldc=I, b_kstrided=True)
        dhp = torch.empty(B, H, dtype=torch.bfloat16, device=dev)
        K.gemm(dG, w_hh16, dhp, B, H, 4 * H, lda=4 * H, ldb=H, ldc=H, b_kstrided=True)
        sk = ctx.sink
        K.gemm(dG, x16, sk.buf(0), 4 * H, I, B, lda=4 * H, ldb=I, ldc=I, a_kstrided=True, b_kstrided=True, accumulate=True)
        K.gemm(dG, h_prev16, sk.buf(1), 4 * H, H, B, lda=4 * H, ldb=H, ldc=H, a_kstrided=True, b_kstrided=True, accumulate=True)
        K.colsum(dG, sk.buf(2), B, 4 * H, 4 * H)
        sk.acc[3] = sk.acc[2]  # bias_hh receives the same gradient as bias_ih
        return dx, dhp, dc_prev, None, None, None, None, None, None, None, None


def lstm_cell_ag(x16, h_prev16, c_prev, cell, sink, bias_sum):
    """Autograd LSTMCell step; `sink` = GradSink(cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh) shared by all steps."""
    v = sink.views
    return _LSTMCellStep.apply(x16, h_prev16, c_prev, v[0], v[1], v[2], v[3], bf16_weight(cell.weight_ih), bf16_weight(cell.weight_hh),
                               bias_sum, sink)


class _BahdanauStep(torch.autograd.Function):
    """Additive attention of one decoder step (espresso/modules/speech_attention.py:38-87).  qp bf16 [B][A] (query_proj of
    the layer-0 hidden), key bf16 [T*B][A] / value bf16 [T*B][Cv] time-major views coming out of a GradSink (gradients are
    accumulated there, buffers 0 / 1), nv fp32 [A] = g v/||v||, bias fp32 [A].  Returns (context bf16 [B][Cv], p fp32 [T][B])."""

    @staticmethod
    def forward(ctx, qp, key_v, value_v, nv, bias, lens, sink, T, B):
        qp = qp.contiguous()
        nv32 = nv.detach().float().contiguous()
        b32 = bias.detach().float().contiguous() if bias is not None else None
        p, c = K.bahdanau_fwd(qp, key_v, value_v, nv32, b32, lens, T, B)
        ctx.save_for_backward(qp, key_v, value_v, nv32, b32, lens, p)
        ctx.sink, ctx.dims = sink, (T, B)
        ctx.mark_non_differentiable(p)
        return c, p

    @staticmethod
    def backward(ctx, dctx, _dp):
        qp, key_v, value_v, nv32, b32, lens, p = ctx.saved_tensors
        T, B = ctx.dims
        sk = ctx.sink
        A = qp.shape[1]
        dnv = torch.zeros(A, dtype=torch.float32, device=qp.device)
        dbias = torch.zeros(A, dtype=torch.float32, device=qp.device) if b32 is not None else None
        dqp = K.bahdanau_bwd(dctx.contiguous(), qp, key_v, value_v, nv32, b32, lens, p, sk.buf(0), sk.buf(1), dnv, dbias, T, B)
        return dqp, None, None, dnv, dbias, None, None, None, None


def bahdanau_step(qp, sink, nv, bias, lens, T, B):
    """sink = GradSink(key [T*B][A] bf16, value [T*B][Cv] bf16).  Returns (context bf16 [B][Cv], attention fp32 [T][B])."""
    return _BahdanauStep.apply(qp, sink.views[0], sink.views[1], nv, bias, lens, sink, T, B)



def lstm_cell_step(x16, cell, h_prev16, h_prev32, c_prev, keep_row=None):
    """One inference step (no autograd): x16 bf16 [N][I]; state fp32 [N][H] (+ bf16 copy of h).  Returns (h16, h32, c)."""
    N, I = x16.shape
    H = cell.weight_hh.shape[1]
    dev = x16.device
    bias = getattr(cell, "_ea_bias_sum", None)
    if bias is None or bias.device != dev:
        bias = (cell.bias_ih + cell.bias_hh).detach().float().contiguous()
        cell._ea_bias_sum = bias
    G = torch.empty(N, 4 * H, dtype=torch.float32, device=dev)
    K.gemm(x16.contiguous(), bf16_weight(cell.weight_ih), G, N, 4 * H, I, lda=I, ldb=I, ldc=4 * H, bias=bias)
    if h_prev16 is not None:
        K.gemm(h_prev16, bf16_weight(cell.weight_hh), G, N, 4 * H, H, lda=H, ldb=H, ldc=4 * H, resid=G, ldr=4 * H)
    h16 = torch.empty(N, H, dtype=torch.bfloat16, device=dev)
    h32 = torch.empty(N, H, dtype=torch.float32, device=dev)
    c = torch.empty(N, H, dtype=torch.float32, device=dev)
    K.lstm_cell_fwd(G, c_prev, c, h32, h16, H, None, N, H, keep_row=keep_row, h_prev_f32=h_prev32)
    return h16, h32, c


# ------------------------------------------------------------------------------------------------ transducer joint
_JOINT_LOGITS_F32 = os.environ.get("EA_JOINT_LOGITS_F32", "0") == "1"


class _TransducerJoint(torch.autograd.Function):
    """logits[b,t,u,:] = fc_out(relu(E[b,t] + D[b,u]))  — espresso/models/transformer/speech_transformer_transducer_base.py:276-299
    after the two LayerNorm'd projections.  E fp32 [B*T][J], D fp32 [B*U1][J] (LayerNorm outputs: fp32 under the reference's
    autocast, so the sum, the ReLU and its derivative mask are evaluated in fp32 — with bf16 E / D the mask flipped within an ulp
    of the kink and the gradients were 9 - 13 % noisy, rounds 3 - 5), w fp32 [V][J] (the effective, weight-normed matrix);
    returns bf16 [B][T][U1][V] (what the reference's fc_out yields under bf16 autocast).  dE / dD leave as fp32 sums."""

    @staticmethod
    def forward(ctx, E, D, w, b, w16, B, T, U1, late=None):
        V, J = w.shape
        ctx.late = late
        Z = K.joint_add_relu(E.contiguous(), D.contiguous(), B, T, U1)
        n = B * T * U1
        # rows padded to a multiple of 64 columns (5004 -> 5056): 16-byte aligned rows for the epilogue stores and the loss kernels,
        # and a reduction length the direct-to-LDS GEMM takes when the logits' gradient is the A operand of the data gradient
        Vp = (V + 63) // 64 * 64
        # EA_JOINT_LOGITS_F32=1 (diagnostic, round 6): fp32 lattice logits — isolates how much of the gradient noise between two
        # bf16 realisations is the bf16 rounding of the logits themselves (the reference's fc_out rounds there too)
        buf = torch.empty(n, Vp, dtype=torch.float32 if _JOINT_LOGITS_F32 else torch.bfloat16, device=E.device)
        K.gemm(Z, w16, buf, n, V, J, lda=J, ldb=J, ldc=Vp, bias=b)
        ctx.save_for_backward(Z, w16)
        ctx.dims = (B, T, U1, V, J, b is not None, Vp)
        ctx.ed_f32 = E.dtype == torch.float32
        return buf.view(B, T, U1, Vp)[..., :V]

    @staticmethod
    def backward(ctx, dlogits):
        Z, w16 = ctx.saved_tensors
        B, T, U1, V, J, has_bias, Vp = ctx.dims
        n = B * T * U1
        st = dlogits.stride()
        if dlogits.dtype == torch.bfloat16 and st[3] == 1 and st[2] == Vp and st[1] == U1 * Vp and st[0] == T * U1 * Vp:
            dl = dlogits.as_strided((n, Vp), (Vp, 1))  # the loss kernel's padded gradient (pad columns are zeros)
        else:
            dl = torch.zeros(n, Vp, dtype=torch.bfloat16, device=dlogits.device)
            dl[:, :V] = dlogits.reshape(n, V)
        dE, dD, dw, db = _joint_backward_from_dl(dl, Z, w16, ctx.dims, ctx.late, ctx.ed_f32)
        return dE, dD, dw, db, None, None, None, None, None


def _joint_backward_from_dl(dl, Z, w16, dims, late, ed_f32):
    """Everything behind the gradient of the lattice logits `dl` (bf16 [n][Vp], pad columns zero): dZ through the output layer and
    the ReLU, its two reductions dE / dD, and the output layer's weight / bias gradient (on a side stream when `late`)."""
    B, T, U1, V, J, has_bias, Vp = dims
    n = B * T * U1
    # data gradient: dZ = dl W through the k-contiguous form (the weight's transposed, zero-padded bf16 copy [J][Vp]):
    # both operands k-contiguous and K = Vp a multiple of 64 -> direct-to-LDS ring kernel
    wt = torch.zeros(J, Vp, dtype=torch.bfloat16, device=dl.device)
    wt[:, :V] = w16.t()
    dZ = torch.empty(n, J, dtype=torch.bfloat16, device=dl.device)
    # relu'(pre) == (Z > 0): the post-activation tensor doubles as the derivative mask
    K.gemm(dl, wt, dZ, n, J, Vp, lda=Vp, ldb=Vp, ldc=J, aux=Z, ldaux=J, act="relu")
    dE, dD = K.joint_reduce(dZ, B, T, U1, out_f32=ed_f32)
    # dW [V][J] = dl^T Z over all B*T*U1 lattice nodes: split-K GEMM on the aligned (padded-pitch) gradient.  (The grouped
    # weight-gradient kernel was tried here: 632 tiles each walking 45 000 rows of a 10 KB-pitch operand ran at 115 TFLOP/s,
    # slower than the split-K launch.)
    if late is not None and has_bias:
        # optimizer-only product, 2 ms at the recipe's batch: launched on its own stream behind the data gradient (two
        # device-filling GEMMs side by side only slow each other down), handed to autograd by the `_JointWeightLate` node,
        # which the engine reaches after the encoder's and the predictor's backward passes
        cur, side = torch.cuda.current_stream(dl.device), aux_stream(dl.device, 2)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            dw, db = _joint_wgrad(dl, Z, n, V, J, Vp)
            late.event = side.record_event()
        dw.record_stream(cur)
        db.record_stream(cur)
        late.dw, late.db, late.keep = dw, db, (dl, Z)  # operands stay allocated until the join
        dw = db = None
    elif has_bias:
        dw, db = _joint_wgrad(dl, Z, n, V, J, Vp)
    else:
        dw, db = _wgrad(dl, Z, n, V, J, ld_dy=Vp), None
    return dE, dD, dw, db


class LazyJointLogits:
    """What the transducer model hands the `transducer_loss` criterion instead of the (B, T', U+1, V) logits when asked to
    (`model(..., lazy_joint=True)`): the joint network's two fp32 branches and its output layer.  `joint_rnnt_loss` consumes it
    without ever writing the logits (csrc/joint_rnnt.hip); `materialize()` is the ordinary tensor for any other consumer."""

    def __init__(self, E, D, w, b, B, T, U1, late=None):
        self.E, self.D, self.w, self.b, self.B, self.T, self.U1, self.late = E, D, w, b, B, T, U1, late

    @property
    def shape(self):
        return (self.B, self.T, self.U1, self.w.shape[0])

    def materialize(self):
        return transducer_joint(self.E, self.D, self.w, self.b, self.B, self.T, self.U1, late=self.late)


_JOINT_FUSED = os.environ.get("EA_JOINT_FUSED", "1") != "0"  # (A/B switch: 0 = always materialise the logits)


def set_joint_fused(on: bool) -> bool:
    """A/B switch of `joint_rnnt_loss` (tests, tools): False = materialise the logits and run the unfused loss kernels."""
    global _JOINT_FUSED
    old, _JOINT_FUSED = _JOINT_FUSED, bool(on)
    return old


class _JointRNNTLoss(torch.autograd.Function):
    """loss[b] = RNN-T negative log-likelihood of logits[b,t,u,:] = fc_out(relu(E[b,t] + D[b,u])) with the logits never written:
    espresso/models/transformer/speech_transformer_transducer_base.py:276-299 + espresso/criterions/transducer_loss.py:130-140.
    Forward: Z = relu(E + D) (bf16, saved), the vocabulary projection on the 8-wave GEMM with a log-sum-exp epilogue, the
    alpha / beta sweep.  Backward: the projection again with the loss-gradient epilogue -> dl bf16, then the joint's backward."""

    @staticmethod
    def forward(ctx, E, D, w, b, w16, targets, logit_lengths, target_lengths, B, T, U1, blank, late):
        V, J = w.shape
        Z = K.joint_add_relu(E.contiguous(), D.contiguous(), B, T, U1)
        loss, ws = K.joint_rnnt_loss_fwd(Z, w16, b, targets, logit_lengths, target_lengths, B, T, U1, blank)
        ctx.save_for_backward(Z, w16, targets, logit_lengths, target_lengths, loss, ws)
        ctx.bias = b
        ctx.dims = (B, T, U1, V, J, b is not None, (V + 63) // 64 * 64)
        ctx.blank, ctx.late, ctx.ed_f32 = blank, late, E.dtype == torch.float32
        return loss

    @staticmethod
    def backward(ctx, dloss):
        Z, w16, targets, logit_lengths, target_lengths, loss, ws = ctx.saved_tensors
        B, T, U1, V, J, has_bias, Vp = ctx.dims
        # the criterion sums the per-utterance losses: dloss is a broadcast scalar, applied on the device
        dl = K.joint_rnnt_loss_grad(Z, w16, ctx.bias, targets, logit_lengths, target_lengths, loss, ws, B, T, U1, ctx.blank, Vp,
                                    grad_scale_dev=dloss.float().contiguous())
        dE, dD, dw, db = _joint_backward_from_dl(dl, Z, w16, ctx.dims, ctx.late, ctx.ed_f32)
        return (dE, dD, dw, db) + (None,) * 9


def joint_rnnt_loss(lazy, targets, logit_lengths, target_lengths, blank=0):
    """Per-utterance RNN-T loss of a `LazyJointLogits`: fused (no logits in HBM) for the shapes the kernels take, else
    materialise + `rnnt_loss`."""
    w16 = K.cast_f32_to_bf16(lazy.w.detach().contiguous())
    J = lazy.w.shape[1]
    ok = (_JOINT_FUSED and lazy.E.is_cuda and J % 64 == 0 and lazy.E.dtype == torch.float32 and lazy.D.dtype == torch.float32
          and lazy.U1 <= 512 and lazy.B * lazy.T * lazy.U1 * J < 2 ** 31)
    if not ok:
        return rnnt_loss(lazy.materialize(), targets, logit_lengths, target_lengths, blank)
    return _JointRNNTLoss.apply(lazy.E, lazy.D, lazy.w, lazy.b, w16, targets, logit_lengths, target_lengths, lazy.B, lazy.T, lazy.U1,
                                blank, lazy.late)


def _joint_wgrad(dl, Z, n, V, J, Vp):
    """dW [V][J] = dl^T Z and db = column sums of dl over all n lattice nodes.  The reduction is cut into row slabs that go
    through ONE grouped weight-gradient launch (direct-to-LDS kernel with transposing reads; the bias sums ride along as an
    extra MFMA against ones), each slab into its own fp32 output, summed afterwards: no split-K workspace pass, no separate
    1.5 GB column-sum pass."""
    # as many slabs as put one 256 x 256 tile of the 8-wave kernel (csrc/wgrad_w8.hip) on every CU: 40 tiles x 6 slabs for V = 5004,
    # J = 512 (with the 64 x 128 tiles of the 4-wave kernel the same slabs are ~1 900 short tiles: it takes either)
    tiles = ((V + 255) // 256) * ((J + 255) // 256)
    slabs = max(1, min(8, 256 // tiles, n // 4096))
    if slabs == 1 or (Vp * 2) % 16 or (J * 2) % 16:
        return _wgrad(dl, Z, n, V, J, ld_dy=Vp), K.colsum(dl, torch.zeros(V, dtype=torch.float32, device=dl.device), n, V, Vp)
    dws = torch.zeros(slabs, V, J, dtype=torch.float32, device=dl.device)
    dbs = torch.zeros(slabs, V, dtype=torch.float32, device=dl.device)
    per = (n + slabs - 1) // slabs
    probs = []
    for i in range(slabs):
        r0, r1 = i * per, min(n, (i + 1) * per)
        probs.append((dl[r0:r1], Z[r0:r1], dws[i], dbs[i], r1 - r0, V, J, Vp, J, J))
    K.wgrad_group(probs)
    return dws.sum(0), dbs.sum(0)


class _LateGrad:
    """What `_TransducerJoint.backward` leaves for `_JointWeightLate.backward`: the gradients, the event that marks them
    complete, and the operands the side stream is still reading."""
    __slots__ = ("dw", "db", "event", "keep")

    def __init__(self):
        self.dw = self.db = self.event = self.keep = None


class _JointWeightLate(torch.autograd.Function):
    """Identity on the joint network's output weight and bias, created BEFORE the encoder and the predictor run.  The autograd
    engine orders ready nodes by creation time, latest first, so this node's backward runs after the encoder's and the
    predictor's; the weight gradient that `_TransducerJoint.backward` launched on a side stream is joined only here.
    (If the engine ever reaches this node earlier the result is the same, the overlap is just shorter.)"""

    @staticmethod
    def forward(ctx, w, b, holder):
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        return w.view_as(w), b.view_as(b)

    @staticmethod
    def backward(ctx, gw, gb):
        h = ctx.holder
        if h.event is None:
            return gw, gb, None
        dev = h.dw.device
        torch.cuda.current_stream(dev).wait_event(h.event)
        dw, db = h.dw, h.db
        h.dw = h.db = h.event = h.keep = None
        if gw is not None:
            dw = dw + gw
        if gb is not None:
            db = db + gb
        return dw, db, None


def joint_weight_late(w, b):
    """-> (w', b', holder) for `transducer_joint(..., late=holder)`; call it before the branches that feed the joint network"""
    holder = _LateGrad()
    w2, b2 = _JointWeightLate.apply(w, b, holder)
    return w2, b2, holder


def transducer_joint(E, D, w, b, B, T, U1, late=None):
    return _TransducerJoint.apply(E, D, w, b, K.cast_f32_to_bf16(w.detach().contiguous()), B, T, U1, late)
