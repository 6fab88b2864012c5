// polar_params.h -- kernel argument blocks shared by the kernel translation units and the host layer.
#pragma once
#include <stdint.h>

namespace polar {

struct SclParams {
    const void *in;            // [B][N] double or float: LLRs, or y when sigma > 0
    double sigma;              // > 0: input is y, llr = 2*y/sigma/sigma
    uint32_t *out_bits;        // [B][N/32]
    double *pm;                // [B] or null
    uint32_t *flags;           // [B] or null
    const uint32_t *frozen;    // [N/32] bit j = leaf j frozen
    const uint32_t *crc_tab;   // [N] D^{pos(j)} mod g for unfrozen leaf j (0 for frozen); null = no CRC
    int N, n;
    int B;
    int sc_mode;               // 1: plain SC decisions (SCdecode), L must be 1
    void *scratch;             // k_scl_fast, N = 1024: per-wave global scratch (FastCfg::scratch_elems each)
    unsigned long long *dbg;   // diagnostic builds only (-DPOLAR_STAMPS): per-section cycle sums
};

struct BpParams {
    const void *in;          // [B][N] double or float (LLR, or y when sigma > 0)
    double sigma;
    uint32_t *out_bits;      // [B][N/32]
    const uint32_t *frozen;  // [N/32]
    int N, n, B, iters;
};

// BP with per-stage read-outs (reference: BPr, BPr_128.c:373-575), see bp_kernel.h
struct BpReadoutParams {
    const void *in;           // [B][N] LLR, or y when sigma > 0
    double sigma;
    uint32_t *out_bits;       // [B][N/32] final decisions (may be null)
    const uint32_t *frozen;   // [N/32]
    const uint32_t *info;     // [N/32] 1 = information position
    const uint32_t *u_bits;   // [B][N/32] sent bits
    unsigned long long *E;    // [ncp][n+1], accumulated over the frames of the launch
    int cp[8];                // iteration counts (1-based), ascending
    int ncp;
    int N, n, B, iters;
};

}  // namespace polar
