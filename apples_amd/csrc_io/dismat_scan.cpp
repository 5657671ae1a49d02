// Distance-table text scanner behind include/apples_io.h (reader semantics of run_apples.py:43-54).
#include <cstdlib>
#include <cstring>

#include "apples_io.h"

namespace {

inline bool is_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r') || (c >= 0x1c && c <= 0x1f); }  // Python's \s on ASCII

// next line [b, e) of the image, universal newlines (the reference opens the file in text mode)
bool next_line(const uint8_t *&p, const uint8_t *end, const uint8_t *&b, const uint8_t *&e) {
    if (p >= end) return false;
    b = p;
    const uint8_t *q = p;
    while (q < end && *q != '\n' && *q != '\r') ++q;
    e = q;
    p = q == end ? end : ((*q == '\r' && q + 1 < end && q[1] == '\n') ? q + 2 : q + 1);
    return true;
}

// a spelling on which strtod and Python's float() agree: [+-] digits [. digits] [e [+-] digits]
bool plain_decimal(const uint8_t *b, const uint8_t *e) {
    const uint8_t *p = b;
    if (p < e && (*p == '+' || *p == '-')) ++p;
    int digits = 0;
    while (p < e && *p >= '0' && *p <= '9') { ++p; ++digits; }
    if (p < e && *p == '.') {
        ++p;
        while (p < e && *p >= '0' && *p <= '9') { ++p; ++digits; }
    }
    if (!digits) return false;
    if (p < e && (*p == 'e' || *p == 'E')) {
        ++p;
        if (p < e && (*p == '+' || *p == '-')) ++p;
        int ed = 0;
        while (p < e && *p >= '0' && *p <= '9') { ++p; ++ed; }
        if (!ed) return false;
    }
    return p == e;
}

// Clinger's fast path: a decimal with at most 15 significant digits and a power of ten within 10^22 is
// one exact integer times or divided by one exact power of ten: a single correctly rounded operation, the
// value strtod and Python's float() return.  Anything else goes to strtod.
const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                           1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
bool fast_decimal(const uint8_t *b, const uint8_t *e, double *out) {
    const uint8_t *p = b;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; ++p; }
    uint64_t w = 0;
    int nd = 0, exp10 = 0;
    bool lead = true;
    while (p < e && *p >= '0' && *p <= '9') {
        if (!(lead && *p == '0')) { lead = false; if (++nd > 15) return false; w = w * 10 + (*p - '0'); }
        ++p;
    }
    if (p < e && *p == '.') {
        ++p;
        while (p < e && *p >= '0' && *p <= '9') {
            if (!(lead && *p == '0')) { lead = false; if (++nd > 15) return false; }
            w = w * 10 + (*p - '0');
            --exp10;
            ++p;
        }
    }
    if (p < e && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < e && (*p == '+' || *p == '-')) { eneg = *p == '-'; ++p; }
        int x = 0;
        while (p < e && *p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); if (x > 400) return false; ++p; }
        exp10 += eneg ? -x : x;
    }
    if (p != e || exp10 < -22 || exp10 > 22) return false;
    double v = (double)w;
    v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];
    *out = neg ? -v : v;
    return true;
}

}  // namespace

extern "C" int apples_dismat_scan(const uint8_t *data, int64_t n_bytes, double *out, int64_t n_tags_cap, int64_t n_rows_cap,
                                  int64_t *n_tags, int64_t *n_rows, int64_t *tag_off, int32_t *tag_len, int64_t *name_off,
                                  int32_t *name_len) {
    const uint8_t *p = data, *end = data + n_bytes, *b, *e;
    *n_tags = 0;
    *n_rows = 0;
    if (!next_line(p, end, b, e)) return 0;
    // header: rstrip, split on white space, drop the first field (run_apples.py:50)
    while (e > b && is_space(e[-1])) --e;
    {
        const uint8_t *q = b;
        int64_t field = 0;
        bool first = true;
        while (true) {
            const uint8_t *fb = q;
            while (q < e && !is_space(*q)) ++q;
            // re.split gives an empty first field when the line starts with white space; it is dropped like any first field
            if (!first) {
                if (out) {
                    if (field >= n_tags_cap) return 2;
                    tag_off[field] = fb - data;
                    tag_len[field] = (int32_t)(q - fb);
                }
                ++field;
            }
            first = false;
            if (q >= e) break;
            while (q < e && is_space(*q)) ++q;
            if (q >= e) { break; }
        }
        *n_tags = field;
    }
    const int64_t nt = *n_tags;
    int64_t row = 0;
    int rc = 0;
    char buf[64];
    while (next_line(p, end, b, e)) {
        while (b < e && is_space(*b)) ++b;  // line.strip()
        while (e > b && is_space(e[-1])) --e;
        const uint8_t *q = b;
        while (q < e && !is_space(*q)) ++q;
        if (out) {
            if (row >= n_rows_cap) return 2;
            name_off[row] = b - data;
            name_len[row] = (int32_t)(q - b);
            double *dst = out + row * nt;
            for (int64_t k = 0; k < nt; ++k) dst[k] = -1.0;  // a tag without a value: absent from the reference's dict
            int64_t k = 0;
            while (q < e && k < nt) {
                while (q < e && is_space(*q)) ++q;
                if (q >= e) break;
                const uint8_t *vb = q;
                while (q < e && !is_space(*q)) ++q;
                const size_t len = (size_t)(q - vb);
                if (len >= sizeof buf || !plain_decimal(vb, q)) { rc = 1; ++k; continue; }  // the caller's own reader decides
                if (fast_decimal(vb, q, &dst[k])) { ++k; continue; }
                memcpy(buf, vb, len);
                buf[len] = 0;
                dst[k++] = strtod(buf, nullptr);
            }
        }
        ++row;
    }
    *n_rows = row;
    return rc;
}
