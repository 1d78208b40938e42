#!/usr/bin/env python3
"""numpy model of the split-precision linear: fp16 main product + block-scaled (MX) correction products in
fp8 e4m3 / fp6 e2m3 / bf6 e3m2 / fp4 e2m1, with fixed or per-block power-of-two scales.  Development tool:
prices a correction format before a kernel is written for it (config-2 statistics: x ~ N(0,1), xavier W)."""
import sys
import numpy as np


def quant(v, ebits, mbits, emax_val, has_sub=True):
    """RNE to a tiny float format: `mbits` explicit mantissa bits, min normal exponent emin, max value emax_val."""
    bias = (1 << (ebits - 1)) - 1
    emin = 1 - bias
    a = np.abs(v)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.maximum(e, emin)
    step = np.exp2(e - mbits)
    q = np.round(a / step) * step      # numpy rounds half to even
    q = np.minimum(q, emax_val)
    return np.sign(v) * q


FORMATS = {
    "fp8": lambda v: quant(v, 4, 3, 448.0),
    "fp6": lambda v: quant(v, 2, 3, 7.5),
    "bf6": lambda v: quant(v, 3, 2, 28.0),
    "fp4": lambda v: quant(v, 2, 1, 6.0),
}
FMAX = {"fp8": 448.0, "fp6": 7.5, "bf6": 28.0, "fp4": 6.0}


def block_scale(v, fmt, axis_blocks):
    """power-of-two scale per 32-block along the K axis so that the block maximum lands in (fmax/2, fmax]"""
    m = np.abs(v).max(axis=axis_blocks, keepdims=True)
    m = np.maximum(m, 1e-30)
    return np.exp2(np.ceil(np.log2(m / FMAX[fmt])))


def run(M=1024, K=768, F=768, seed=0, xscale=1.0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((M, K)) * xscale).astype(np.float32)
    lim = np.sqrt(6.0 / (K + F))
    w = rng.uniform(-lim, lim, (K, F)).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    xh = x.astype(np.float16).astype(np.float64)
    xl = x.astype(np.float64) - xh
    wh = w.astype(np.float16).astype(np.float64)
    wl = w.astype(np.float64) - wh
    main = xh @ wh
    print("x scale %g: |y| rms %.3f;  fp16 product alone: max %.2e rms %.2e" % (
        xscale, ref.std(), np.abs(main - ref).max(), (main - ref).std()))
    xb = lambda a: a.reshape(M, K // 32, 32)
    wb = lambda a: a.reshape(K // 32, 32, F)

    def corr(fmt, a_mode, w_fmt=None):
        w_fmt = w_fmt or fmt
        q, qw = FORMATS[fmt], FORMATS[w_fmt]
        # W side: true per-(32 k, column) scales
        s_wh = block_scale(wb(wh), w_fmt, 1)
        s_wl = block_scale(wb(wl), w_fmt, 1)
        wh_q = (qw(wb(wh) / s_wh) * s_wh).reshape(K, F)
        wl_q = (qw(wb(wl) / s_wl) * s_wl).reshape(K, F)
        if a_mode == "fixed":      # xl stored * 2^11, xh as it is (the f16mx8 kernel)
            xl_q = q(xl * 2048.0) / 2048.0
            xh_q = q(x.astype(np.float64))
        elif a_mode == "block":    # per-(row, 32 k) scale from the block maximum of |x|, xl scale = that * 2^-11
            s_x = block_scale(xb(x.astype(np.float64)), fmt, 2)
            xh_q = (q(xb(x.astype(np.float64)) / s_x) * s_x).reshape(M, K)
            s_l = s_x / 2048.0
            xl_q = (q(xb(xl) / s_l) * s_l).reshape(M, K)
        elif a_mode == "block2":   # separate maxima for xl and xh
            s_x = block_scale(xb(x.astype(np.float64)), fmt, 2)
            xh_q = (q(xb(x.astype(np.float64)) / s_x) * s_x).reshape(M, K)
            s_l = block_scale(xb(xl), fmt, 2)
            xl_q = (q(xb(xl) / s_l) * s_l).reshape(M, K)
        elif a_mode in ("exp2_11", "exp2_12", "exp1_11"):   # what the kernel can afford: scale from the EXPONENT of the block
            # maximum (2^(E-2): the maximum lands in [4, 8) and saturates above 7.5; or 2^(E-1): [2, 4)), xl scale = that * 2^-11 / 2^-12
            m = np.abs(xb(x.astype(np.float64))).max(axis=2, keepdims=True)
            E = np.floor(np.log2(np.maximum(m, 1e-30)))
            s_x = np.exp2(E - (2 if a_mode.startswith("exp2") else 1))
            xh_q = (q(xb(x.astype(np.float64)) / s_x) * s_x).reshape(M, K)
            s_l = s_x / (2048.0 if a_mode.endswith("11") else 4096.0)
            xl_q = (q(xb(xl) / s_l) * s_l).reshape(M, K)
        y = main + xl_q @ wh_q + xh_q @ wl_q
        e = y - ref
        return np.abs(e).max(), e.std()

    for fmt, a_mode, w_fmt in (("fp8", "fixed", None), ("fp8", "block", None), ("fp6", "fixed", None), ("fp6", "block", None),
                               ("fp6", "block2", None), ("fp6", "exp2_11", None), ("fp6", "exp2_12", None), ("fp6", "exp1_11", None), ("bf6", "fixed", None), ("bf6", "block", None), ("fp4", "block", None)):
        mx, rms = corr(fmt, a_mode, w_fmt)
        print("  correction %-4s  A scales %-6s  linear error max %.2e  rms %.2e" % (fmt, a_mode, mx, rms))


if __name__ == "__main__":
    for xs in (1.0, 0.1, 4.0):
        run(xscale=xs)
