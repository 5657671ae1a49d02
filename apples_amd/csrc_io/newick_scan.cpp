// Newick tokenizer behind include/apples_io.h: the token loop of apples_amd/tree.py:parse_newick
// (itself the reader contract of SURVEY.md Appendix B for apples/prepareTree.py:24-36) over the
// file image, leaving per-node arrays in creation (pre-) order.  Anything it does not take on the
// fast path -- malformed input, a branch length that is not a plain decimal number -- is reported
// back so that the caller's record-by-record parser decides (and words the error).
#include <cstdlib>
#include <cstring>

#include "apples_io.h"

namespace {

// what Python's str.strip() removes, as far as ASCII goes (the caller keeps non-ASCII text away)
inline bool is_space(uint8_t c) { return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f); }

inline bool is_delim(uint8_t c) {
    return c == '(' || c == ')' || c == ',' || c == ':' || c == ';' || c == '[' || c == ']' || c == '\'';
}

// [+-]digits[.digits][(e|E)[+-]digits] or [+-].digits[...]: the spellings on which strtod and
// Python's float() are the same correctly rounded function
bool plain_decimal(const uint8_t *b, const uint8_t *e) {
    const uint8_t *p = b;
    if (p < e && (*p == '+' || *p == '-')) ++p;
    int digits = 0;
    while (p < e && *p >= '0' && *p <= '9') { ++p; ++digits; }
    if (p < e && *p == '.') {
        ++p;
        while (p < e && *p >= '0' && *p <= '9') { ++p; ++digits; }
    }
    if (digits == 0) return false;
    if (p < e && (*p == 'e' || *p == 'E')) {
        ++p;
        if (p < e && (*p == '+' || *p == '-')) ++p;
        int ed = 0;
        while (p < e && *p >= '0' && *p <= '9') { ++p; ++ed; }
        if (ed == 0) return false;
    }
    return p == e;
}

}  // namespace

extern "C" int apples_newick_scan(const uint8_t *text, int64_t n_bytes, int64_t cap, int32_t *parent, int32_t *depth,
                                  int32_t *size, int64_t *label_off, int32_t *label_len, uint8_t *label_quoted,
                                  double *length, uint8_t *length_state, int64_t *length_off, int32_t *length_len,
                                  int64_t *n_nodes) {
    if (cap < 1) return 2;
    int64_t n = 1;
    auto init = [&](int64_t k, int32_t par, int32_t d) {
        parent[k] = par; depth[k] = d; size[k] = 1;
        label_off[k] = 0; label_len[k] = -1; label_quoted[k] = 0;
        length[k] = 0.0; length_state[k] = 0; length_off[k] = 0; length_len[k] = 0;
    };
    init(0, -1, 0);
    int64_t cur = 0;
    int32_t d = 0;
    bool expect_len = false, done = false;
    const uint8_t *p = text, *end = text + n_bytes;
    while (p < end && !done) {
        const uint8_t c = *p;
        // ---- one token [tb, te) of the reader's grammar; characters that start no token are skipped
        const uint8_t *tb = p, *te = p + 1;
        enum { PUNCT, QUOTED, COMMENT, OTHER } kind = PUNCT;
        if (c == '\'') {
            const uint8_t *q = p + 1;
            bool closed = false;
            while (q < end) {
                if (*q == '\'') {
                    if (q + 1 < end && q[1] == '\'') { q += 2; continue; }
                    closed = true;
                    break;
                }
                ++q;
            }
            if (!closed) return 1;  // how much a regular expression would still match here is the caller's parser's business
            kind = QUOTED; te = q + 1;
        } else if (c == '[') {
            const uint8_t *q = p + 1;
            while (q < end && *q != ']') ++q;
            if (q == end) { ++p; continue; }
            kind = COMMENT; te = q + 1;
        } else if (c == ']') {
            ++p; continue;
        } else if (!is_delim(c)) {
            const uint8_t *q = p + 1;
            while (q < end && !is_delim(*q)) ++q;
            kind = OTHER; te = q;
        }
        p = te;
        if (expect_len) {
            if (kind != OTHER) return 1;  // missing branch length (or a quoted one): the caller's parser words the error
            if (length_state[cur] == 2) return 1;  // a second length over one float() has not seen yet: it might have been an error
            const uint8_t *b = tb, *e = te;
            while (b < e && is_space(*b)) ++b;
            while (e > b && is_space(e[-1])) --e;
            if (plain_decimal(b, e) && e - b < 64) {
                char buf[64];
                std::memcpy(buf, b, (size_t)(e - b));
                buf[e - b] = 0;
                length[cur] = std::strtod(buf, nullptr);
                length_state[cur] = 1;
            } else {  // nan, inf, 1_000, hex, garbage: float() decides
                length_state[cur] = 2;
                length_off[cur] = b - text;
                length_len[cur] = (int32_t)(e - b);
            }
            expect_len = false;
            continue;
        }
        if (kind == COMMENT) continue;
        if (kind == QUOTED) {
            label_off[cur] = (tb + 1) - text; label_len[cur] = (int32_t)(te - tb - 2); label_quoted[cur] = 1;
            continue;
        }
        if (kind == OTHER) {
            const uint8_t *b = tb, *e = te;
            while (b < e && is_space(*b)) ++b;
            while (e > b && is_space(e[-1])) --e;
            if (e > b) { label_off[cur] = b - text; label_len[cur] = (int32_t)(e - b); label_quoted[cur] = 0; }
            continue;
        }
        switch (c) {
            case '(':
                if (n >= cap) return 2;
                ++d;
                init(n, (int32_t)cur, d);
                cur = n++;
                break;
            case ',': {
                const int32_t par = parent[cur];
                if (par < 0) return 1;
                if (n >= cap) return 2;
                init(n, par, d);
                cur = n++;
                break;
            }
            case ')':
                cur = parent[cur];
                if (cur < 0) return 1;
                --d;
                size[cur] = (int32_t)(n - cur);
                break;
            case ':': expect_len = true; break;
            case ';': done = true; break;
        }
    }
    if (cur != 0) return 1;
    *n_nodes = n;
    return 0;
}
