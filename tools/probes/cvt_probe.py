"""What does v_cvt_scalef32_pk_fp8_f32 do with its scale, and does v_fma_mix_f32 give x - fp16(x)?"""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libcvtprobe.so"))
dev = torch.device("cuda:0")

def dec_e4m3(b):
    b = int(b); s = -1.0 if b & 0x80 else 1.0; e = (b >> 3) & 15; m = b & 7
    if e == 15 and m == 7: return float("nan")
    return s * (m / 8 * 2.0 ** -6 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7))

x = torch.tensor([1.0, 2.0, 0.75, -3.0, 2.0 ** -11, 3 * 2.0 ** -12, 100.0, 448.0], device=dev)
n = x.numel()
vp = ctypes.c_void_p
for scale in (1.0, 2.0, 0.5, 2.0 ** -11, 3.0 * 2.0 ** -11):
    qs = torch.zeros(n // 2, dtype=torch.int32, device=dev); qp = torch.zeros_like(qs)
    r = torch.zeros(n, device=dev)
    lib.run_cvt.argtypes = [vp, ctypes.c_int, ctypes.c_float, vp, vp, vp, ctypes.c_int]
    assert lib.run_cvt(vp(x.data_ptr()), n, scale, vp(qs.data_ptr()), vp(qp.data_ptr()), vp(r.data_ptr()), 0) == 0
    dq = [dec_e4m3(v & 255) for v in qs.cpu().tolist()] , [dec_e4m3((v >> 8) & 255) for v in qs.cpu().tolist()]
    scaled = [v for pair in zip(*dq) for v in pair]
    print("scale %-12g -> %s" % (scale, scaled))
print("plain            ->", [v for p in zip([dec_e4m3(v & 255) for v in qp.cpu().tolist()], [dec_e4m3((v >> 8) & 255) for v in qp.cpu().tolist()]) for v in p])
y = torch.randn(4096, device=dev) * 3
qs = torch.zeros(2048, dtype=torch.int32, device=dev); qp = torch.zeros_like(qs); r = torch.zeros(4096, device=dev)
lib.run_cvt(vp(y.data_ptr()), 4096, 1.0, vp(qs.data_ptr()), vp(qp.data_ptr()), vp(r.data_ptr()), 0)
ref = y - y.half().float()
print("fma_mix residual exact:", bool((r == ref).all()), float((r - ref).abs().max()))

# overflow behaviour with and without MODE.FP16_OVFL
big = torch.tensor([500.0, -1000.0, 70000.0, -1e6, 448.0, 460.0, 1e30, 3.0], device=dev)
for ovfl in (0, 1):
    qs = torch.zeros(4, dtype=torch.int32, device=dev); qp = torch.zeros_like(qs); r = torch.zeros(8, device=dev)
    lib.run_cvt(vp(big.data_ptr()), 8, 0.5, vp(qs.data_ptr()), vp(qp.data_ptr()), vp(r.data_ptr()), ovfl)
    dec = lambda t: [v for p in zip([dec_e4m3(v & 255) for v in t.cpu().tolist()], [dec_e4m3((v >> 8) & 255) for v in t.cpu().tolist()]) for v in p]
    print("FP16_OVFL=%d plain fp8 %s\n            scaled(0.5) %s\n            %s %s" % (ovfl, dec(qp), dec(qs), "fp16" if ovfl else "resid", r.cpu().tolist()))
