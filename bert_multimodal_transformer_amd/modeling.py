"""MAG -- drop-in for /root/reference/modeling.py:6-51, running on the HIP library.

    MAG(hidden_size, beta_shift, dropout_prob).forward(text_embedding, visual, acoustic) -> Tensor

Same constructor/forward signature, parameter names (W_hv, W_ha, W_v, W_a, LayerNorm) and init as the reference
(nn.Linear / nn.LayerNorm defaults); visual_dim / acoustic_dim are keyword arguments instead of module globals.
Layout-agnostic like the reference (last-dim ops only): [B,L,*] for BERT, [L,B,*] for XLNet.
"""
import torch
import torch.nn as nn

from . import _lib
from .global_configs import ACOUSTIC_DIM, VISUAL_DIM

_STEP = [0]


class _MagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text, visual, acoustic, W_hv, b_hv, W_ha, b_ha, W_v, b_v, W_a, b_a, ln_w, ln_b, beta_shift, p, training,
                compute_dtype):
        if not text.is_cuda:
            raise _lib.MagbertError("MAG runs on the HIP path only: inputs must be on a ROCm device (no CPU fallback)")
        L = _lib.lib()
        H = text.shape[-1]
        V, A = visual.shape[-1], acoustic.shape[-1]
        T = text.numel() // H
        dt = _lib.DT_BF16 if compute_dtype == torch.bfloat16 else _lib.DT_F32
        tx = text.detach().to(compute_dtype).contiguous().view(T, H)
        vis = visual.detach().float().contiguous().view(T, V)
        aco = acoustic.detach().float().contiguous().view(T, A)
        params = [t.detach().float().contiguous() for t in (W_hv, b_hv, W_ha, b_ha, W_v, b_v, W_a, b_a, ln_w, ln_b)]
        ws = torch.empty(L.mb_mag_workspace_bytes(dt, T, H, V, A), dtype=torch.uint8, device=text.device)
        out = torch.empty(T, H, dtype=compute_dtype, device=text.device)
        _STEP[0] += 1
        key = _lib.make_dropkey(torch.initial_seed(), _STEP[0], 1, p) if (training and p > 0) else _lib.no_drop()
        st = torch.cuda.current_stream(text.device).cuda_stream
        _lib.check(L.mb_mag_forward(dt, _lib.ptr(tx), _lib.ptr(vis), _lib.ptr(aco), *[_lib.ptr(t) for t in params],
                                    float(beta_shift), key, _lib.ptr(out), _lib.ptr(ws), T, H, V, A, st))
        ctx.save_for_backward(tx, ws, *params)
        ctx.meta = (dt, T, H, V, A, float(beta_shift), key, text.shape, visual.shape, acoustic.shape, text.dtype)
        return out.view(text.shape).to(text.dtype)

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        tx, ws, W_hv, b_hv, W_ha, b_ha, W_v, b_v, W_a, b_a, ln_w, ln_b = ctx.saved_tensors
        dt, T, H, V, A, beta_shift, key, tshape, vshape, ashape, in_dtype = ctx.meta
        dev = tx.device
        do = dout.detach().to(tx.dtype).contiguous().view(T, H)
        d_text = torch.empty(T, H, dtype=tx.dtype, device=dev)
        d_vis = torch.empty(T, V, dtype=torch.float32, device=dev)
        d_aco = torch.empty(T, A, dtype=torch.float32, device=dev)
        g = [torch.zeros_like(t) for t in (W_hv, b_hv, W_ha, b_ha, W_v, b_v, W_a, b_a, ln_w, ln_b)]
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.mb_mag_backward(dt, _lib.ptr(do), _lib.ptr(tx), _lib.ptr(W_hv), _lib.ptr(b_hv), _lib.ptr(W_ha),
                                     _lib.ptr(b_ha), _lib.ptr(W_v), _lib.ptr(b_v), _lib.ptr(W_a), _lib.ptr(b_a),
                                     _lib.ptr(ln_w), beta_shift, key, _lib.ptr(ws), _lib.ptr(d_text), _lib.ptr(d_vis),
                                     _lib.ptr(d_aco), *[_lib.ptr(t) for t in g], T, H, V, A, st))
        return (d_text.view(tshape).to(in_dtype), d_vis.view(vshape), d_aco.view(ashape), *g, None, None, None, None)


class MAG(nn.Module):
    def __init__(self, hidden_size, beta_shift, dropout_prob, visual_dim=VISUAL_DIM, acoustic_dim=ACOUSTIC_DIM,
                 compute_dtype=torch.float32):
        super(MAG, self).__init__()
        if hidden_size % 256 or not 256 <= hidden_size <= 1024:
            raise NotImplementedError("the HIP MAG row kernels take hidden_size = 256, 512, 768 (TEXT_DIM, global_configs.py:11) or 1024")
        self.W_hv = nn.Linear(visual_dim + hidden_size, hidden_size)       # modeling.py:15
        self.W_ha = nn.Linear(acoustic_dim + hidden_size, hidden_size)     # modeling.py:16
        self.W_v = nn.Linear(visual_dim, hidden_size)                      # modeling.py:18
        self.W_a = nn.Linear(acoustic_dim, hidden_size)                    # modeling.py:19
        self.beta_shift = beta_shift
        self.LayerNorm = nn.LayerNorm(hidden_size)                         # modeling.py:22
        self.dropout = nn.Dropout(dropout_prob)                            # modeling.py:23 (p is read from here)
        self.compute_dtype = compute_dtype

    def forward(self, text_embedding, visual, acoustic):
        return _MagFn.apply(text_embedding, visual, acoustic, self.W_hv.weight, self.W_hv.bias, self.W_ha.weight,
                            self.W_ha.bias, self.W_v.weight, self.W_v.bias, self.W_a.weight, self.W_a.bias,
                            self.LayerNorm.weight, self.LayerNorm.bias, self.beta_shift, self.dropout.p, self.training,
                            self.compute_dtype)
