"""CRC generator-matrix file format of the reference (CRC_6.dat).

CRC_6.dat is a 64x6 0/1 matrix: row i holds the coefficients of D^(r+i) mod g(D), column j the coefficient
of D^j -- the systematic-CRC parity generator for (K = 64, r = 6), the N = 128 analogue of the ``Gc[K][r]``
literal in CASCL_1024_sys.c:48-561.  On disk it is UTF-16LE with a BOM, CRLF line ends, single-space
separated, no trailing newline.  No program of the reference opens it; this module reads, writes and
regenerates it so that a systematic-CRC front end can be fed from the same file.
"""
import numpy as np


def systematic_parity_matrix(K, taps):
    """[K][r] uint8: row i = D^(r+i) mod g(D), column j = coefficient of D^j.  taps: exponents of g incl. 0 and r."""
    r = max(taps)
    glow = 0
    for t in taps:
        if t < r:
            glow |= 1 << t
    rem = glow  # D^r mod g
    out = np.zeros((K, r), dtype=np.uint8)
    for i in range(K):
        for j in range(r):
            out[i, j] = (rem >> j) & 1
        rem <<= 1
        if rem >> r & 1:
            rem = (rem ^ (1 << r)) ^ glow
    return out


def dumps(m):
    lines = [" ".join(str(int(v)) for v in row) for row in np.asarray(m)]
    return b"\xff\xfe" + "\r\n".join(lines).encode("utf-16-le")


def loads(data):
    text = data.decode("utf-16")  # honours the BOM
    rows = [[int(x) for x in line.split()] for line in text.replace("\r", "").split("\n") if line.strip()]
    return np.asarray(rows, dtype=np.uint8)


def load(path):
    with open(path, "rb") as f:
        return loads(f.read())


def save(path, m):
    with open(path, "wb") as f:
        f.write(dumps(m))


def systematic_crc(v, parity):
    """CRC bits of payload v (length K) for a parity matrix [K][r]: c = v * parity over GF(2)."""
    return (np.asarray(v, dtype=np.uint8) @ parity.astype(np.uint8)) & 1
