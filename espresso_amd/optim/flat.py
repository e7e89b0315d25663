"""Flat parameter storage sized for HBM3E: ONE fp32 master buffer, ONE fp32 gradient buffer and ONE
bf16 shadow buffer for the whole model (≈ 1.3 GB with Adam moments for the 80 M-parameter Conformer).

This is the MI355X-native replacement of the per-parameter storage the reference optimises and of
the flat copy-in/copy-out buffer of fairseq/distributed/legacy_distributed_data_parallel.py:41-50,
82-120: parameters *live* in the flat buffer (every nn.Parameter is a view), so the optimizer is one
streaming kernel, the gradient norm is one reduction, and data-parallel buckets are plain slices
that RCCL reduces in place."""
from typing import Dict, List

import torch

from .. import kernels as K

_ALIGN = 64  # elements; keeps every parameter 256-byte (fp32) / 128-byte (bf16) aligned


class FlatParams:
    def __init__(self, module: torch.nn.Module, device=None):
        from ..modules.conformer_layer import MultiheadAttentionParams

        params: List[torch.nn.Parameter] = []
        seen = set()
        # group q,k,v projection weights contiguously so the fused QKV GEMM reads one [3C][C] operand
        fused = []
        for m in module.modules():
            if isinstance(m, MultiheadAttentionParams) and m.q_proj.weight.shape == m.k_proj.weight.shape == m.v_proj.weight.shape:
                grp = [m.q_proj.weight, m.k_proj.weight, m.v_proj.weight]
                if all(id(p) not in seen for p in grp) and grp[0].numel() % _ALIGN == 0:
                    fused.append(grp)
                    for p in grp:
                        seen.add(id(p))
                        params.append(p)
                bgrp = [m.q_proj.bias, m.k_proj.bias, m.v_proj.bias]
                if all(b is not None and id(b) not in seen for b in bgrp) and bgrp[0].numel() % _ALIGN == 0:
                    for p in bgrp:  # adjacent, no padding: the runtime reads them as one [3C] vector
                        seen.add(id(p))
                        params.append(p)
        for p in module.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        self.params = params
        device = device or params[0].device
        self.offsets: Dict[int, int] = {}
        off = 0
        for p in params:
            self.offsets[id(p)] = off
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.p32 = torch.zeros(off, dtype=torch.float32, device=device)
        self.g32 = torch.zeros(off, dtype=torch.float32, device=device)
        self.p16 = torch.zeros(off, dtype=torch.bfloat16, device=device)
        with torch.no_grad():
            for p in params:
                o, n = self.offsets[id(p)], p.numel()
                self.p32[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.p32[o:o + n].view(p.shape)
                p.grad = self.g32[o:o + n].view(p.shape)
                p._ea_bf16 = self.p16[o:o + n].view(p.shape)
        for grp in fused:
            o = self.offsets[id(grp[0])]
            n = grp[0].numel()
            grp[0]._ea_fused_qkv = self.p16[o:o + 3 * n].view(3 * grp[0].shape[0], grp[0].shape[1])
        self.sync_bf16()

    def sync_bf16(self):
        """Refresh the bf16 shadow from the fp32 master (after init / checkpoint load).  The shadow only
        feeds the HIP GEMMs; on a CPU buffer (host-logic tests of bucketing / hooks under gloo) there is
        nothing to feed and the shadow is left untouched."""
        if self.p32.is_cuda:
            K.cast_f32_to_bf16(self.p32, self.p16)

    def rebind_grads(self):
        """Point every .grad back at the flat gradient buffer (after anything set them to None)."""
        for p in self.params:
            o, n = self.offsets[id(p)], p.numel()
            p.grad = self.g32[o:o + n].view(p.shape)

    def zero_grad(self):
        self.g32.zero_()
        self.rebind_grads()

    def slices_in_backward_order(self, bucket_elems: int, last_elems: int = 0):
        """[(start, end)] element ranges covering the buffer, last parameters first, ~bucket_elems each.  `last_elems` > 0: the
        FINAL range (the first parameters of the model, whose gradients the backward pass produces last, so that nothing is left
        to overlap its all-reduce with) holds at most that many elements."""
        out = []
        end = self.numel
        tail = min(max(0, int(last_elems)), self.numel) if last_elems and self.numel > bucket_elems else 0
        while end > tail:
            start = max(tail, end - bucket_elems)
            out.append((start, end))
            end = start
        if tail:
            out.append((0, tail))
        return out
