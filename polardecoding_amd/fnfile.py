"""The generator-matrix input of the reference's simulators ("Fn (stdin)").

Every simulator reads an N x N matrix of whitespace-separated decimal 0/1 from stdin before it starts
(SCL_1024.c:207-217: `scanf("%d", &temp)` N*N times, row-major; any other value prints "Illegal input!" and is stored
anyway) and encodes with its rows, x = u Fn (SCL_1024.c:242-250).  The file is not shipped with the reference; the matrix is
the n-fold Kronecker power of F = [1 0; 1 1], Fn[i][j] = ((i & j) == j) -- with it the programs reproduce their published
logs.  This module writes that file, reads one back the way the reference does, and says whether a given matrix is the
Kronecker power: the encoders and decoders here (and the factor graph of the reference's own decoders, connectBCB
SCL_1024.c:377-401) are built for exactly that matrix, so polar_sim --fn refuses any other."""
import numpy as np


def kronecker(N):
    """Fn[i][j] = ((i & j) == j), uint8 [N][N]."""
    i = np.arange(N)
    return ((i[:, None] & i[None, :]) == i[None, :]).astype(np.uint8)


def write_fn(path, N):
    """N lines of N single-space separated 0/1 (what `./SCL_1024 < Fn_1024.txt` expects)."""
    m = kronecker(N)
    with open(path, "w") as f:
        for row in m:
            f.write(" ".join("1" if v else "0" for v in row) + "\n")


def read_fn(path, N=None):
    """Read like the reference: whitespace-separated decimal integers, row-major.  Returns (matrix int64 [N][N], illegal)
    where illegal = number of entries that are neither 0 nor 1 (each would print "Illegal input!", SCL_1024.c:213-214).
    N defaults to the square root of the number of entries."""
    with open(path) as f:
        vals = np.array(f.read().split(), dtype=np.int64)
    if N is None:
        N = int(round(vals.size ** 0.5))
    if vals.size < N * N:
        raise ValueError(f"{path}: {vals.size} entries, need {N * N}")
    m = vals[:N * N].reshape(N, N)
    return m, int(((m != 0) & (m != 1)).sum())


def is_kronecker(m):
    m = np.asarray(m)
    return m.ndim == 2 and m.shape[0] == m.shape[1] and np.array_equal(m, kronecker(m.shape[0]))
