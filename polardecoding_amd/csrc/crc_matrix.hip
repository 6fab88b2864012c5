// crc_matrix.hip -- the reference's CRC generator-matrix file format (host code only, no kernel).
//
// /root/reference/CRC_6.dat: a K x r 0/1 matrix, K = 64 rows of r = 6 entries; row i holds the coefficients of
// D^(r+i) mod g(D), column j the coefficient of D^j -- the redundant part of the systematic CRC generator matrix.
// The literal `const int Gc[K][r] = { {1, 1, 1, 0, ...}, ... }` of CASCL_1024_sys.c:48-561 is the same object for
// K = 512, r = 24 (used at :776-789: w[j] += Gc[i][j] for every payload bit v[i] = 1).  On disk CRC_6.dat is UTF-16LE
// with a byte-order mark, CRLF line ends, single-space separated, no newline after the last row.  The loader also
// takes the same matrix as UTF-8 / ASCII text and with the punctuation of the C literal (braces, commas, semicolon),
// so the body of `Gc` pasted into a file loads as well.  One text line = one row.
//
// A matrix is accepted only if it IS such a generator: g(D) = D^r + (row 0) must have a D^0 term and every row i must
// equal D^(r+i) mod g.  The caller gets g's exponents (polar_cfg.crc_taps) and the rows as bit words.
#include "../../include/polar_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// file bytes -> 8-bit text (every character the format uses is ASCII; anything else becomes '?', which is rejected)
bool decode_text(const std::vector<unsigned char> &raw, std::string &out)
{
    size_t n = raw.size();
    if (n >= 2 && raw[0] == 0xFF && raw[1] == 0xFE) {          // UTF-16LE with BOM (the reference's file)
        if ((n - 2) % 2) return false;
        for (size_t i = 2; i + 1 < n; i += 2) out.push_back(raw[i + 1] ? '?' : (char)raw[i]);
        return true;
    }
    if (n >= 2 && raw[0] == 0xFE && raw[1] == 0xFF) {          // UTF-16BE with BOM
        if ((n - 2) % 2) return false;
        for (size_t i = 2; i + 1 < n; i += 2) out.push_back(raw[i] ? '?' : (char)raw[i + 1]);
        return true;
    }
    size_t i = (n >= 3 && raw[0] == 0xEF && raw[1] == 0xBB && raw[2] == 0xBF) ? 3 : 0;   // UTF-8 BOM
    for (; i < n; ++i) out.push_back(raw[i] < 0x80 ? (char)raw[i] : '?');
    return true;
}

}  // namespace

extern "C" {

int polar_crc_matrix_load(const char *path, polar_crc_matrix *out)
{
    if (!path || !out) return POLAR_EINVAL;
    std::memset(out, 0, sizeof *out);
    FILE *f = std::fopen(path, "rb");
    if (!f) return POLAR_EINVAL;
    std::vector<unsigned char> raw;
    unsigned char buf[4096];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) {
        raw.insert(raw.end(), buf, buf + got);
        if (raw.size() > (1u << 24)) break;   // a 4096 x 32 matrix in UTF-16 is < 1 MB
    }
    std::fclose(f);
    if (raw.size() > (1u << 24)) return POLAR_EINVAL;
    std::string text;
    if (!decode_text(raw, text)) return POLAR_EINVAL;

    std::vector<uint32_t> rows;
    int r = -1;
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        uint32_t word = 0;
        int cnt = 0;
        for (size_t i = pos; i < eol; ++i) {
            const char ch = text[i];
            if (ch == '0' || ch == '1') {
                // an entry is a single digit: "10" or "01" is not a 0/1 entry
                if (i + 1 < eol && text[i + 1] >= '0' && text[i + 1] <= '9') return POLAR_EINVAL;
                if (cnt >= 32) return POLAR_EINVAL;
                if (ch == '1') word |= 1u << cnt;
                ++cnt;
            } else if (ch == ' ' || ch == '\t' || ch == '\r' || ch == ',' || ch == '{' || ch == '}' || ch == ';') {
                continue;
            } else {
                return POLAR_EINVAL;   // any other character (a 2, a letter, a non-ASCII code unit)
            }
        }
        if (cnt > 0) {
            if (r < 0) r = cnt;
            if (cnt != r) return POLAR_EINVAL;           // ragged row
            if (rows.size() >= 4096) return POLAR_EINVAL;
            rows.push_back(word);
        }
        pos = eol + 1;
    }
    if (rows.empty() || r < 1) return POLAR_EINVAL;

    // g(D) = D^r + row 0; every row must be D^(r+i) mod g
    const uint64_t glow = rows[0];
    if (!(glow & 1u)) return POLAR_EINVAL;               // no D^0 term: not a CRC generator polynomial
    const uint64_t top = 1ull << r;
    uint64_t rem = glow;
    for (size_t i = 0; i < rows.size(); ++i) {
        if (rows[i] != (uint32_t)rem) return POLAR_EINVAL;
        rem <<= 1;
        if (rem & top) rem = (rem ^ top) ^ glow;
    }
    out->K = (int)rows.size();
    out->r = r;
    out->n_taps = 0;
    for (int j = 0; j < r; ++j)
        if ((glow >> j) & 1u) out->taps[out->n_taps++] = j;
    out->taps[out->n_taps++] = r;
    out->rows = (uint32_t *)std::malloc(rows.size() * sizeof(uint32_t));
    if (!out->rows) {
        std::memset(out, 0, sizeof *out);
        return POLAR_ENOMEM;
    }
    std::memcpy(out->rows, rows.data(), rows.size() * sizeof(uint32_t));
    return POLAR_OK;
}

void polar_crc_matrix_free(polar_crc_matrix *m)
{
    if (!m) return;
    std::free(m->rows);
    std::memset(m, 0, sizeof *m);
}

// Writes the K x r matrix of g(D) in exactly the bytes of the reference's file: UTF-16LE, BOM, "0 1 ... 1" rows,
// CRLF between rows, nothing after the last one (tests/golden/CRC_6.dat is reproduced byte for byte).
int polar_crc_matrix_save(const char *path, int K, const int *taps, int n_taps)
{
    if (!path || !taps || n_taps < 2 || K < 1 || K > 4096) return POLAR_EINVAL;
    int r = 0;
    uint64_t glow = 0;
    bool has0 = false;
    for (int i = 0; i < n_taps; ++i) {
        if (taps[i] < 0 || taps[i] > 32) return POLAR_EINVAL;
        if (taps[i] > r) r = taps[i];
        has0 |= taps[i] == 0;
    }
    if (r < 1 || !has0) return POLAR_EINVAL;
    for (int i = 0; i < n_taps; ++i)
        if (taps[i] < r) glow |= 1ull << taps[i];
    std::vector<unsigned char> o = {0xFF, 0xFE};
    auto put = [&](char c) { o.push_back((unsigned char)c); o.push_back(0); };
    const uint64_t top = 1ull << r;
    uint64_t rem = glow;
    for (int i = 0; i < K; ++i) {
        if (i) { put('\r'); put('\n'); }
        for (int j = 0; j < r; ++j) {
            if (j) put(' ');
            put(((rem >> j) & 1u) ? '1' : '0');
        }
        rem <<= 1;
        if (rem & top) rem = (rem ^ top) ^ glow;
    }
    FILE *f = std::fopen(path, "wb");
    if (!f) return POLAR_EINVAL;
    const bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
    return (std::fclose(f) == 0 && ok) ? POLAR_OK : POLAR_EINVAL;
}

// polar_create with the CRC taken from a generator-matrix file: cfg->crc_taps / n_taps / crc_r are ignored and
// replaced by g(D) of the file; cfg->K must not exceed the file's row count (rows are D^(r+i) mod g for i < K, so a
// shorter payload uses a prefix of the matrix).  cfg->crc_systematic keeps its meaning: 1 = the encoder the matrix
// belongs to (CASCL_1024_sys.c:776-789), 0 = the same g(D) in the multiply-by-g encoder (CASCL_1024_L8.c:251-266).
int polar_create_crc_file(const polar_cfg *cfg, const char *path, polar_ctx **out)
{
    if (!cfg || !out) return POLAR_EINVAL;
    *out = nullptr;
    if (cfg->algo != POLAR_ALGO_CASCL) return POLAR_EINVAL;
    polar_crc_matrix m;
    int rc = polar_crc_matrix_load(path, &m);
    if (rc) return rc;
    if (cfg->K > m.K) {
        polar_crc_matrix_free(&m);
        return POLAR_EINVAL;
    }
    polar_cfg c = *cfg;
    c.crc_r = m.r;
    c.crc_taps = m.taps;
    c.n_taps = m.n_taps;
    rc = polar_create(&c, out);   // copies the taps
    polar_crc_matrix_free(&m);
    return rc;
}

}  // extern "C"
