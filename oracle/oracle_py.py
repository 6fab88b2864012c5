"""ctypes bindings for the CPU oracle (oracle/libpolar_oracle.so) and, when built, the real
reference decoders (oracle/_ref/lib*.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by polardecoding_amd/."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

CRC6_TAPS = [0, 5, 6]  # g(D) = D^6 + D^5 + 1 (CASCL_128.c:3, :212-214)
CRC24C_TAPS = [0, 1, 2, 4, 8, 12, 13, 15, 17, 20, 21, 23, 24]  # CASCL_1024_L8.c:2-4, :253-265

ALGO = {"SC": 0, "BP": 1, "SCL": 2, "CASCL": 3}


def load_q(nmax=1024):
    path = os.path.join(REPO, "polardecoding_amd", "data", "q5g_nmax1024.txt")
    vals = []
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                continue
            vals += [int(x) for x in line.split()]
    assert len(vals) == 1024
    return vals


def q_for(N):
    return [x for x in load_q() if x < N]


class _Code(C.Structure):
    _fields_ = [("N", C.c_int), ("n", C.c_int), ("K", C.c_int), ("r", C.c_int), ("A", C.c_int),
                ("ntaps", C.c_int), ("taps", C.c_int * 32),
                ("info_order", C.POINTER(C.c_int)), ("frozen", C.POINTER(C.c_ubyte)),
                ("systematic", C.c_int)]


class _Sim(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("ranv", C.c_uint64), ("rani", C.c_int),
                ("pn", C.c_int * 63), ("m", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libpolar_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libpolar_oracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.po_code_create.restype = C.POINTER(_Code)
        L.po_code_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        L.po_code_destroy.argtypes = [C.POINTER(_Code)]
        L.po_code_set_systematic.argtypes = [C.POINTER(_Code), C.c_int]
        dp = C.POINTER(C.c_double)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        L.po_sc_decode_f64.argtypes = [C.POINTER(_Code), dp, ip]
        L.po_bp_decode_f64.argtypes = [C.POINTER(_Code), dp, C.c_int, ip]
        L.po_scl_decode_f64.argtypes = [C.POINTER(_Code), dp, C.c_int, C.c_int, ip, dp, ip]
        L.po_sc_decode_f32.argtypes = [C.POINTER(_Code), fp, ip]
        L.po_bp_decode_f32.argtypes = [C.POINTER(_Code), fp, C.c_int, ip]
        L.po_scl_decode_f32.argtypes = [C.POINTER(_Code), fp, C.c_int, C.c_int, ip, fp, ip]
        L.po_decode_batch_f64.argtypes = [C.POINTER(_Code), C.c_int, C.c_int, C.c_int, dp, C.c_size_t,
                                          C.POINTER(C.c_uint32)]
        L.po_sim_init.argtypes = [C.POINTER(_Sim), C.c_uint64]
        L.po_sim_frame.argtypes = [C.POINTER(_Sim), C.POINTER(_Code), C.c_double, ip, dp]
        L.po_sigma_from_db.restype = C.c_double
        L.po_sigma_from_db.argtypes = [C.c_double]
        L.po_llr_from_y.argtypes = [dp, C.c_double, dp, C.c_int]
        L.po_run_sweep.argtypes = [C.POINTER(_Code), C.c_int, C.c_int, C.c_int, C.c_uint64, dp, C.c_int,
                                   C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.po_count_bit_errors.argtypes = [C.POINTER(_Code), ip, ip]
        L.po_lit_create.restype = C.c_void_p
        L.po_lit_create.argtypes = [C.POINTER(_Code), C.c_int]
        L.po_lit_destroy.argtypes = [C.c_void_p]
        L.po_lit_reset.argtypes = [C.c_void_p]
        L.po_lit_poison.argtypes = [C.c_void_p, C.c_uint64]
        L.po_lit_decode.argtypes = [C.c_void_p, dp, C.c_int, ip, dp]
        L.po_lit_diag.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.c_int]
        L.po_lit_path_metrics.argtypes = [C.c_void_p, dp]
        _lib = L
    return _lib


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Code:
    """N, K, CRC taps -> frozen set in 5G reliability order (CASCL_1024_L8.c:209-217)."""

    def __init__(self, N, K, crc_taps=None, Q=None, systematic=False):
        self.N, self.K = N, K
        self.systematic = bool(systematic)
        self.taps = list(crc_taps) if crc_taps else []
        self.r = max(self.taps) if self.taps else 0
        q = np.asarray(Q if Q is not None else q_for(N), dtype=np.int32)
        if q.size != N:
            raise ValueError(f"reliability order has {q.size} entries, need N = {N} (the 5G table stops at 1024: pass Q)")
        t = np.asarray(self.taps if self.taps else [0], dtype=np.int32)
        self._h = lib().po_code_create(N, K, self.r, _ip(t), len(self.taps), _ip(q))
        if not self._h:
            raise ValueError("bad code parameters")
        if systematic:
            lib().po_code_set_systematic(self._h, 1)
        self.A = K + self.r
        self.n = int(np.log2(N))
        self.info_order = np.array([self._h.contents.info_order[i] for i in range(self.A)], dtype=np.int32)
        self.frozen = np.ones(N, dtype=np.uint8)
        self.frozen[self.info_order] = 0

    def __del__(self):
        try:
            lib().po_code_destroy(self._h)
        except Exception:
            pass


def sigma_from_db(db):
    return lib().po_sigma_from_db(db)


def llr_from_y(y, sigma):
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(y)
    lib().po_llr_from_y(_dp(y), sigma, _dp(out), y.size)
    return out


def decode(code, llr, algo, L=8, bp_iters=100, dtype="f64", stats=None):
    """Decode one frame or a [B][N] batch.  Returns (u_hat [B][N] int32, pm [B], ties [B]).
    stats (optional int32 [B][2], list decoders): per frame, leaves where the 32-bit ranking keys are not enough,
    leaves with three or more equal candidates (po_scl_last_stats)."""
    L_ = lib()
    f32 = dtype == "f32"
    llr = np.ascontiguousarray(llr, dtype=np.float32 if f32 else np.float64)
    single = llr.ndim == 1
    llr2 = llr.reshape(-1, code.N)
    B = llr2.shape[0]
    uh = np.zeros((B, code.N), dtype=np.int32)
    pm = np.zeros(B, dtype=np.float32 if f32 else np.float64)
    ties = np.zeros(B, dtype=np.int32)
    rp = C.POINTER(C.c_float if f32 else C.c_double)
    for b in range(B):
        lp = llr2[b].ctypes.data_as(rp)
        up = uh[b].ctypes.data_as(C.POINTER(C.c_int))
        if algo == "SC":
            rc = (L_.po_sc_decode_f32 if f32 else L_.po_sc_decode_f64)(code._h, lp, up)
        elif algo == "BP":
            rc = (L_.po_bp_decode_f32 if f32 else L_.po_bp_decode_f64)(code._h, lp, bp_iters, up)
        else:
            fn = L_.po_scl_decode_f32 if f32 else L_.po_scl_decode_f64
            rc = fn(code._h, lp, L, 1 if algo == "CASCL" else 0, up,
                    pm[b:].ctypes.data_as(rp), ties[b:].ctypes.data_as(C.POINTER(C.c_int)))
            if stats is not None:
                L_.po_scl_last_stats(stats.reshape(-1, 2)[b].ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError(f"oracle decode failed rc={rc}")
    if single:
        return uh[0], pm[0], ties[0]
    return uh, pm, ties


def math(op, a, b, dtype=np.float64):
    """CHK / T / PHI of the oracle element-wise: op 0 CHK(a, b), 1 T(|a|), 2 PHI(a, u = (b != 0))."""
    a = np.ascontiguousarray(a, dtype=dtype).ravel()
    b = np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=dtype), a.shape), dtype=dtype).ravel()
    out = np.empty_like(a)
    fn = lib().po_math_f32 if dtype == np.float32 else lib().po_math_f64
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    fn(int(op), a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
    return out


class Literal:
    """The reference's list decoder WITH its persistent node records (oracle/polar_oracle_literal.c): what
    SCLdecode / CASCL do frame after frame, including after a median tie.  decode() -> (u_hat, pm, diag[3]) with
    diag = counts of "Oops!", "Wrong propagation order!", "Error!" for that frame."""

    def __init__(self, code, L=8, crc=False):
        self.code, self.L, self.crc = code, L, bool(crc)
        self._h = lib().po_lit_create(code._h, L)
        if not self._h:
            raise ValueError("bad list size")

    def __del__(self):
        try:
            lib().po_lit_destroy(self._h)
        except Exception:
            pass

    def reset(self):
        lib().po_lit_reset(self._h)

    def poison(self, seed):
        """arbitrary leftovers in every record a frame does not initialise (stands for an unknown history)"""
        lib().po_lit_poison(self._h, seed)

    def decode(self, llr):
        llr = np.ascontiguousarray(llr, dtype=np.float64)
        uh = np.zeros(self.code.N, dtype=np.int32)
        pm = C.c_double(0)
        rc = lib().po_lit_decode(self._h, _dp(llr), 1 if self.crc else 0, _ip(uh), C.byref(pm))
        d = (C.c_long * 3)()
        lib().po_lit_diag(self._h, d, 1)
        if rc not in (0, -3, -5):
            raise RuntimeError(f"po_lit_decode rc={rc}")
        self.last_rc = rc   # -5: the reference would never return from this frame (Partition, SCL_1024.c:518-544)
        return uh, pm.value, list(d)

    def path_metrics(self):
        out = np.zeros(self.L)
        lib().po_lit_path_metrics(self._h, _dp(out))
        return out


class Sim:
    """The reference's transmit chain and RNG (SCL_1024.c:184-197, :238-261, :295-326)."""

    def __init__(self, seed):
        self._s = _Sim()
        lib().po_sim_init(C.byref(self._s), seed)

    def frame(self, code, sigma):
        u = np.zeros(code.N, dtype=np.int32)
        y = np.zeros(code.N, dtype=np.float64)
        lib().po_sim_frame(C.byref(self._s), code._h, sigma, _ip(u), _dp(y))
        return u, y

    def frames(self, code, sigma, count):
        us = np.zeros((count, code.N), dtype=np.int32)
        ys = np.zeros((count, code.N), dtype=np.float64)
        for i in range(count):
            lib().po_sim_frame(C.byref(self._s), code._h, sigma, _ip(us[i]), _dp(ys[i]))
        return us, ys

    @property
    def state(self):
        return (int(self._s.ranv), int(self._s.rani), int(self._s.m))


def bp_readout(code, llr, u, iters, checkpoints):
    """BPr_128.c: BP with per-stage read-outs.  llr [B][N], u [B][N] -> (u_hat [B][N], E [len(cp)][n+1] summed over B)."""
    L_ = lib()
    L_.po_bpr_decode_f64.argtypes = [C.POINTER(_Code), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_int)]
    llr = np.ascontiguousarray(llr, dtype=np.float64).reshape(-1, code.N)
    u = np.ascontiguousarray(u, dtype=np.int32).reshape(-1, code.N)
    cp = np.asarray(checkpoints, dtype=np.int32)
    E = np.zeros((len(cp), code.n + 1), dtype=np.int64)
    uh = np.zeros_like(u)
    for b in range(llr.shape[0]):
        rc = L_.po_bpr_decode_f64(code._h, _dp(llr[b]), iters, _ip(cp), len(cp), _ip(u[b]),
                                  E.ctypes.data_as(C.POINTER(C.c_long)), _ip(uh[b]))
        if rc != 0:
            raise RuntimeError(f"po_bpr_decode rc={rc}")
    return uh, E


def run_sweep(code, algo, snr_db, ble, seed, L=8, bp_iters=100, min_run=0, want_blocks=False):
    """min_run: the rule `errBlock < BLE || run < min_run` of the published L = 32 logs."""
    snr = np.asarray(snr_db, dtype=np.float64)
    run = (C.c_long * len(snr))()
    eb = (C.c_long * len(snr))()
    blk = (C.c_long * len(snr))()
    fn = lib().po_run_sweep_min
    fn.argtypes = [C.POINTER(_Code), C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_double), C.c_int, C.c_int,
                   C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    rc = fn(code._h, ALGO[algo], L, bp_iters, seed, _dp(snr), len(snr), ble, min_run, run, eb, blk)
    if rc != 0:
        raise RuntimeError(f"po_run_sweep rc={rc}")
    if want_blocks:
        return list(run), list(eb), list(blk)
    return list(run), list(eb)


def count_bit_errors(code, u, u_hat):
    u = np.ascontiguousarray(u, dtype=np.int32)
    u_hat = np.ascontiguousarray(u_hat, dtype=np.int32)
    return lib().po_count_bit_errors(code._h, _ip(u), _ip(u_hat))


# ---- the real reference, when oracle/_ref has been built (oracle/Makefile `ref`) -------------

REF_PROGRAMS = {
    # name: (N, K, crc taps, algo, L)
    "SC_128": (128, 64, None, "SC", 1),
    "SC_1024": (1024, 512, None, "SC", 1),
    "BP_128": (128, 64, None, "BP", 1),
    "BP_1024": (1024, 512, None, "BP", 1),
    # BP_1024.c with its `#define iterMax 100` reading 50 (BASELINE.json config 2; oracle/Makefile target ref50)
    "BP_1024_it50": (1024, 512, None, "BP", 1),
    "SCL_128": (128, 64, None, "SCL", 8),
    "SCL_1024": (1024, 512, None, "SCL", 8),
    "CASCL_128": (128, 64, CRC6_TAPS, "CASCL", 8),
    "CASCL_1024_L8": (1024, 512, CRC24C_TAPS, "CASCL", 8),
    # systematic CRC, decoder on the bit-reversed graph (same decisions as the natural-order decoder)
    "CASCL_1024_sys": (1024, 512, CRC24C_TAPS, "CASCL", 8),
}
SYSTEMATIC_PROGRAMS = ("CASCL_1024_sys",)
BP_ITERS = {"BP_128": 100, "BP_1024": 100, "BP_1024_it50": 50}   # iterMax, BP_1024.c:16


def ref_available(name):
    return os.path.exists(os.path.join(HERE, "_ref", f"lib{name}.so"))


class Ref:
    """One compiled reference program (its decode function, unmodified)."""

    def __init__(self, name):
        path = os.path.join(HERE, "_ref", f"lib{name}.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle ref` where /root/reference exists")
        self.name = name
        self.l = C.CDLL(path)
        self.l.ref_decode.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int)]
        self.l.ref_last_pm.restype = C.c_double
        self.l.ref_time_decode.restype = C.c_double
        self.l.ref_time_decode.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_long, C.POINTER(C.c_int)]
        self.l.ref_init()
        self.N = self.l.ref_block_length()
        self.L = self.l.ref_list_size()
        self.l.ref_diag.argtypes = [C.POINTER(C.c_long), C.c_int]

    def diag(self, reset=True):
        """counts of "Oops!" (SCL_1024.c:622), "Wrong propagation order!" (:418), "Error!" (:651) printed since the last reset"""
        d = (C.c_long * 3)()
        self.l.ref_diag(d, 1 if reset else 0)
        return list(d)

    def reset_state(self):
        """list decoders only: node records back to their calloc() state (what the decoder does after a median
        tie depends on what earlier frames left in them)"""
        self.l.ref_reset_state()

    def path_metrics(self):
        out = np.zeros(self.L)
        self.l.ref_path_metrics(_dp(out))
        return out

    def decode(self, y, sigma):
        y = np.ascontiguousarray(y, dtype=np.float64)
        uh = np.zeros(self.N, dtype=np.int32)
        self.l.ref_decode(_dp(y), sigma, _ip(uh))
        return uh, self.l.ref_last_pm()

    # --- BPr_128.c only (oracle/Makefile kind 4): BPr(y, u_hat, u) and its read-out table E[7][n+1] ---
    def readout_iters(self):
        cp = (C.c_int * 6)()
        iters = self.l.ref_readout_iters(cp)
        return iters, list(cp)

    def decode_u(self, y, sigma, u):
        y = np.ascontiguousarray(y, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.int32)
        uh = np.zeros(self.N, dtype=np.int32)
        self.l.ref_decode_u.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.l.ref_decode_u(_dp(y), sigma, _ip(u), _ip(uh))
        return uh

    def readout(self, reset=True):
        n = int(np.log2(self.N))
        e = np.zeros((7, n + 1), dtype=np.int32)
        self.l.ref_readout(_ip(e), 1 if reset else 0)
        return e[:6]

    def time_decode(self, ys, sigma):
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        uh = np.zeros(self.N, dtype=np.int32)
        return self.l.ref_time_decode(_dp(ys), sigma, ys.shape[0], _ip(uh))
