"""CPU emulation of the MX-FP8 operand format of BASELINE config 5 (test infrastructure, like the rest of ``oracle/``).

Not reference math - the reference (fp32 PyTorch) has no low-precision path.  This restates the PUBLISHED format the
HIP quantiser implements (OCP Microscaling Formats v1.0, MXFP8 with e4m3 elements, section 6.3 conversion): blocks of 32
consecutive elements of a row share the scale 2^(floor(log2(max|x|)) - 8), stored as an E8M0 byte (exponent + 127);
elements are x / scale, clamped to the e4m3 range +-448 and rounded to nearest even.  Bit patterns are compared
one-to-one with ``avf_quant_mx8`` (tests/test_gpu_mx8.py); parity unpinned by the reference (nothing to pin against).
"""
import torch


def mx8_quant(x: torch.Tensor):
    """x [rows, cols] (cols % 32 == 0) -> (e4m3 bit patterns uint8 [rows, cols], E8M0 bytes uint8 [rows, cols/32])."""
    x = x.detach().float().cpu().contiguous()
    rows, cols = x.shape
    xb = x.view(rows, cols // 32, 32)
    amax = xb.abs().amax(dim=-1)
    eb = (amax.view(torch.int32) >> 23) & 255  # biased exponent of the block maximum
    sb = torch.clamp(eb - 8, min=0)
    inv = ((254 - sb) << 23).to(torch.int32).view(torch.float32)  # 2^(127 - sb)
    t = (xb * inv.unsqueeze(-1)).clamp(-448.0, 448.0)
    q = t.to(torch.float8_e4m3fn).view(torch.uint8).view(rows, cols)
    return q, sb.to(torch.uint8)


def mx8_dequant(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """Inverse image: fp32 [rows, cols] (exact: every MX-FP8 value is an fp32 value)."""
    q = q.detach().cpu()
    s = s.detach().cpu()
    rows, cols = q.shape
    v = q.view(torch.float8_e4m3fn).float().view(rows, cols // 32, 32)
    scale = torch.ldexp(torch.ones(()), s.to(torch.int32) - 127)
    return (v * scale.unsqueeze(-1)).view(rows, cols)
