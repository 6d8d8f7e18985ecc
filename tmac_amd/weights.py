"""Offline weight transform: quantised uint weights -> the reference's bit-interleaved tile layout.

Same signature and result as ``python/t_mac/weights.py:preprocess_weights`` of the reference (lines
5-88) — the blob format GGUF files converted for T-MAC already contain and that
``tmac_hip_register_weights`` accepts — but derived from the closed-form byte addresses of
SURVEY.md Appendix A.3 instead of a chain of reshapes/transposes:

    M-space row     r   = (o // 8) * 8 * bits + p * 8 + o % 8          (o: weight row, p: bit-plane)
    nibble(r, t)        = sum_ig  bit_p(w[o, 4t + ig]) << ig
    byte(r, t)          = tile*(bm/2 * K/4) + ((t // kfactor) * (bm/32) + rr // 32) * kfactor*16
                          + (t % kfactor) * 16 + rr % 16,     tile = r // bm, rr = r % bm
    nibble position     = low for rr % 32 < 16, high otherwise
"""
from typing import Optional, Tuple

import numpy as np


def preprocess_weights(w: np.ndarray, scales: np.ndarray, zeros: Optional[np.ndarray] = None, bits: int = 4,
                       g: int = 4, bm: int = 512, kfactor: int = 16, simd_n_in: int = 16,
                       simd_n_out: int = 8) -> Tuple[np.ndarray, np.ndarray]:
    """See module docstring.  Returns (A uint8 [M/bm][K/g][bm/2], scales [M/bm][K/gs][bm/bits*(2|1)])."""
    if w.dtype != np.uint8:
        raise TypeError("w must be uint8 in [0, 2**bits)")
    if g != 4 or simd_n_in != 16 or simd_n_out != 8:
        raise NotImplementedError("only g=4, simd_n_in=16, simd_n_out=8 (all shipped T-MAC configurations)")
    Mw, K = w.shape
    M = Mw * bits
    if M % bm or bm % 32 or bm % bits or (bm // bits) % 8 or K % 4 or (K // 4) % kfactor:
        raise ValueError(f"shape (Mw={Mw}, K={K}, bits={bits}) is not tileable by bm={bm}, kfactor={kfactor}")
    T = K // 4
    # nibbles[o, p, t]
    planes = ((w[:, None, :] >> np.arange(bits, dtype=np.uint8)[None, :, None]) & 1).astype(np.uint8)
    nib = (planes.reshape(Mw, bits, T, 4) << np.arange(4, dtype=np.uint8)).sum(-1).astype(np.uint8)
    o = np.arange(Mw)[:, None]
    p = np.arange(bits)[None, :]
    r = (o // 8) * 8 * bits + p * 8 + o % 8                      # [Mw, bits] M-space row
    by_row = np.empty((M, T), np.uint8)
    by_row[r.reshape(-1)] = nib.reshape(M, T)
    tile, rr = np.divmod(np.arange(M), bm)
    t = np.arange(T)
    byte = (tile[:, None] * (bm // 2 * T) + ((t[None, :] // kfactor) * (bm // 32) + rr[:, None] // 32) * kfactor * 16
            + (t[None, :] % kfactor) * 16 + rr[:, None] % 16)
    shift = (4 * ((rr % 32) // 16)).astype(np.uint8)[:, None]
    A = np.zeros(M * T // 2, np.uint8)
    np.bitwise_or.at(A, byte.reshape(-1), (by_row << shift).reshape(-1))
    A = A.reshape(M // bm, T, bm // 2)

    if scales.size >= Mw:
        SG = scales.shape[1]
        rpt = bm // bits

        def tile_view(x):
            return x.reshape(M // bm, rpt // 8, 8, SG).transpose(0, 3, 1, 2)   # [tile][sg][m/8][8]
        s_t = tile_view(np.asarray(scales))
        if zeros is not None:
            s_t = np.stack([s_t, tile_view(np.asarray(zeros))], axis=-2)       # [tile][sg][m/8][2][8]
        scales_out = np.ascontiguousarray(s_t).reshape(M // bm, SG, -1)
    else:
        scales_out = np.concatenate([scales, zeros]) if zeros is not None else scales
    return A, scales_out
