"""polardecoding_amd -- MI355X-native polar decoders (SC / BP / SCL / CRC-aided SCL).

The product is the C-ABI shared library ``polardecoding_amd/lib/libpolar_hip.so`` (include/polar_hip.h):
hand-written HIP kernels for gfx950.  This package is the thin host-side mirror of the reference's
per-frame decode functions (``SCdecode`` / ``BP`` / ``SCLdecode`` / ``CASCL``), plus torch plumbing for
device buffers and multi-GPU sharding.  There is no CPU fallback: everything raises if the HIP library
is missing.
"""
from .api import (ALGO_BP, ALGO_CASCL, ALGO_SC, ALGO_SCL, CRC6_TAPS, CRC24C_TAPS, F32, F64, FLAG_CRC_PASS,
                  FLAG_RERANK, FLAG_TIE, BP, CASCL, Decoder, Group, PolarError, SCdecode, SCLdecode, decode, lib_path, load_crc_matrix, load_library,
                  q_sequence, save_crc_matrix)

__all__ = ["Decoder", "Group", "SCdecode", "BP", "SCLdecode", "CASCL", "decode", "PolarError", "load_library", "lib_path",
           "q_sequence", "load_crc_matrix", "save_crc_matrix", "ALGO_SC", "ALGO_BP", "ALGO_SCL", "ALGO_CASCL", "F64", "F32", "CRC6_TAPS", "CRC24C_TAPS",
           "FLAG_TIE", "FLAG_CRC_PASS", "FLAG_RERANK"]
