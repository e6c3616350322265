"""Operands in the PRE-SPLIT layout the ablation builds (-DDC_WG_ABL=16, -DDC_PP_ABL=32) read: every float4 (4 consecutive channels of a
pixel) holds [hi01 | hi23 | lo01 | lo23], hi = fp16(v), lo = fp16(v - hi) -- exactly what the producers' two v_fma_mix per element would
have made of the fp32 values v (scale 1), so the MFMA operands, and the result, are those of the shipped kernel on v."""
import torch


def presplit_pack(v):
    """v: float32 (..., C) with C % 4 == 0 -> float32 tensor of the same shape holding the fp16 pairs."""
    h = v.half()
    l = (v - h.float()).half()
    hq = h.view(torch.int16).to(torch.int32).bitwise_and(0xffff).reshape(*v.shape[:-1], v.shape[-1] // 4, 4)
    lq = l.view(torch.int16).to(torch.int32).bitwise_and(0xffff).reshape(*v.shape[:-1], v.shape[-1] // 4, 4)
    w = torch.stack([hq[..., 0] | (hq[..., 1] << 16), hq[..., 2] | (hq[..., 3] << 16),
                     lq[..., 0] | (lq[..., 1] << 16), lq[..., 2] | (lq[..., 3] << 16)], dim=-1)
    return w.reshape(v.shape).contiguous().view(torch.float32)
