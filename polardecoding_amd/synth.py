"""Synthetic BPSK-AWGN frames on the GPU (torch), shaped like the reference's transmit chain.
Used by bench.py and the full-size GPU tests; the bit-exact sequential chain lives in host/polar_sim.c."""
import torch


def polar_transform_(x):
    """x: [B, N] uint8 on device; in-place x = u * F^{(x)n}, natural order (SCL_1024.c:242-250)."""
    B, n_ = x.shape
    s = 1
    while s < n_:
        v = x.view(B, n_ // (2 * s), 2, s)
        v[:, :, 0, :] ^= v[:, :, 1, :]
        s *= 2
    return x


def make_batch(B, N, K, crc_taps, snr_db, info_order, device, gen, dtype=torch.float64):
    """Synthetic frames of the reference's transmit chain shape: random payload -> CRC multiply by g(D)
    (CASCL_1024_L8.c:245-266) -> u[I[i]] = w[i] -> polar encode -> BPSK + AWGN -> LLR = 2y/s/s."""
    R = max(crc_taps) if crc_taps else 0
    v = torch.randint(0, 2, (B, K), device=device, dtype=torch.uint8, generator=gen)
    w = torch.zeros((B, K + R), device=device, dtype=torch.uint8)
    for t in (crc_taps if crc_taps else (0,)):
        w[:, t:t + K] ^= v
    u = torch.zeros((B, N), device=device, dtype=torch.uint8)
    u[:, info_order] = w
    x = polar_transform_(u.clone())
    sigma = 10.0 ** (-snr_db / 20.0)  # R = 1/2 (CASCL_1024_L8.c:237)
    noise = torch.randn((B, N), device=device, dtype=torch.float64, generator=gen)
    y = (1.0 - 2.0 * x.to(torch.float64)) + sigma * noise
    llr = (2.0 * y / sigma / sigma).to(dtype).contiguous()
    # pack u into words for the device error counter
    ub = u.view(B, N // 32, 32).to(torch.int64)
    weights = (1 << torch.arange(32, device=device, dtype=torch.int64))
    words = (ub * weights).sum(-1)
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()
    return llr, words
