// jplace placement rows as text, behind include/apples_io.h: the bytes json.dumps(..., sort_keys=True, indent=4) gives for
// {"n": [name], "p": [[edge, likelihood, 1, distal, pendant]]} (run_apples.py:106-118), numbers spelled as Python spells them
// (float.__repr__: shortest digits that round-trip, exponent form below 1e-4 and from 1e16; ints without a point).
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "apples_io.h"

namespace {

// repr(float) into p; returns the end
char *py_float(char *p, double x) {
    if (std::isnan(x)) { memcpy(p, "NaN", 3); return p + 3; }                 // json's allow_nan spellings
    if (std::isinf(x)) {
        if (x < 0) *p++ = '-';
        memcpy(p, "Infinity", 8);
        return p + 8;
    }
    char b[40];
    const auto r = std::to_chars(b, b + sizeof b, x, std::chars_format::scientific);  // [-]d[.ddd]e[+-]XX, shortest round trip
    const char *s = b, *e = r.ptr;
    if (*s == '-') { *p++ = '-'; ++s; }
    char dig[24];
    int nd = 0;
    const char *q = s;
    for (; q < e && *q != 'e'; ++q)
        if (*q != '.') dig[nd++] = *q;
    int ex = 0;
    {
        const char *t = q + 1;
        const bool neg = *t == '-';
        if (*t == '+' || *t == '-') ++t;
        for (; t < e; ++t) ex = ex * 10 + (*t - '0');
        if (neg) ex = -ex;
    }
    const int decpt = ex + 1;
    if (decpt > -4 && decpt <= 16) {  // fixed notation
        if (decpt <= 0) {
            *p++ = '0'; *p++ = '.';
            for (int i = 0; i < -decpt; ++i) *p++ = '0';
            memcpy(p, dig, nd); p += nd;
        } else if (decpt >= nd) {
            memcpy(p, dig, nd); p += nd;
            for (int i = nd; i < decpt; ++i) *p++ = '0';
            *p++ = '.'; *p++ = '0';
        } else {
            memcpy(p, dig, decpt); p += decpt;
            *p++ = '.';
            memcpy(p, dig + decpt, nd - decpt); p += nd - decpt;
        }
        return p;
    }
    *p++ = dig[0];
    if (nd > 1) { *p++ = '.'; memcpy(p, dig + 1, nd - 1); p += nd - 1; }
    *p++ = 'e';
    int x10 = decpt - 1;
    *p++ = x10 < 0 ? '-' : '+';
    if (x10 < 0) x10 = -x10;
    char eb[8];
    int ne = 0;
    do { eb[ne++] = (char)('0' + x10 % 10); x10 /= 10; } while (x10);
    if (ne < 2) eb[ne++] = '0';
    while (ne) *p++ = eb[--ne];
    return p;
}

char *put(char *p, const char *s) {
    const size_t n = strlen(s);
    memcpy(p, s, n);
    return p + n;
}

}  // namespace

extern "C" int64_t apples_format_double(double x, char *out) { return py_float(out, x) - out; }

// kind[i]: 0 = [edge, error, 1, distal, pendant] with floats; 1 = [edge, 0, 1, 0, 0] (exact hit / cannot be placed); 2 = floats with
// the pendant printed as the int 0 (the clamped branch of util.py:32-50).  Rows with keep[i] == 0 are skipped.  `first` != 0:
// the first row written opens the "placements" list.  Names are copied as they are between quotes: the caller passes only
// names that json.dumps would not escape.  Returns the bytes written, or -1 when `cap` could not hold them.
extern "C" int64_t apples_jplace_rows(const uint8_t *names, const int64_t *name_off, const int32_t *name_len, int64_t n,
                                      const int32_t *edge, const double *err, const double *distal, const double *pendant,
                                      const uint8_t *kind, const uint8_t *keep, int first, char *out, int64_t cap) {
    char *p = out;
    const char *ind5 = "                    ";  // 20 spaces
    for (int64_t i = 0; i < n; ++i) {
        if (!keep[i]) continue;
        if (cap - (p - out) < 512 + name_len[i]) return -1;
        p = put(p, first ? "    \"placements\": [" : ",");
        first = 0;
        p = put(p, "\n        {\n            \"n\": [\n                \"");
        memcpy(p, names + name_off[i], (size_t)name_len[i]);
        p += name_len[i];
        p = put(p, "\"\n            ],\n            \"p\": [\n                [\n");
        p = put(p, ind5);
        p = std::to_chars(p, p + 16, edge[i]).ptr;
        p = put(p, ",\n"); p = put(p, ind5);
        if (kind[i] == 1) *p++ = '0'; else p = py_float(p, err[i]);
        p = put(p, ",\n"); p = put(p, ind5);
        *p++ = '1';
        p = put(p, ",\n"); p = put(p, ind5);
        if (kind[i] == 1) *p++ = '0'; else p = py_float(p, distal[i]);
        p = put(p, ",\n"); p = put(p, ind5);
        if (kind[i] != 0) *p++ = '0'; else p = py_float(p, pendant[i]);
        p = put(p, "\n                ]\n            ]\n        }");
    }
    return p - out;
}

// The jplace tree string (apples/jutil.py:22-96): Newick with "{edge_index}" after every node but the root, a branch length
// printed as str(int(x)) when integral and as str(float(x)) otherwise (:80-87).  Nodes are numbered in left-to-right post-order
// (= edge_index); children in file order as CSR; labels as byte ranges of one blob (label_len < 0: none), copied as they are.
// Returns the bytes written (without the closing ';'), -1 if cap is too small, -2 on a length this routine leaves to Python
// (integral beyond 2^63).
extern "C" int64_t apples_extended_newick(int32_t n_nodes, const int32_t *child_off, const int32_t *child_idx, int32_t root,
                                          const double *edge_len, const uint8_t *has_len, const uint8_t *labels,
                                          const int64_t *label_off, const int32_t *label_len, char *out, int64_t cap) {
    char *p = out, *end = out + cap;
    // iterative depth-first walk: (node, next child position)
    int32_t *stack_node = new int32_t[(size_t)n_nodes + 1];
    int32_t *stack_pos = new int32_t[(size_t)n_nodes + 1];
    int sp = 0;
    int64_t rc = 0;
    stack_node[0] = root; stack_pos[0] = child_off[root];
    if (child_off[root + 1] > child_off[root]) { if (p < end) *p++ = '('; }
    while (sp >= 0) {
        const int32_t v = stack_node[sp];
        if (stack_pos[sp] < child_off[v + 1]) {
            const int32_t c = child_idx[stack_pos[sp]];
            if (stack_pos[sp] > child_off[v]) { if (end - p < 2) { rc = -1; break; } *p++ = ','; }
            ++stack_pos[sp];
            ++sp;
            stack_node[sp] = c; stack_pos[sp] = child_off[c];
            if (child_off[c + 1] > child_off[c]) { if (end - p < 2) { rc = -1; break; } *p++ = '('; }
            continue;
        }
        // all children written: close, label, then (unless the root) the length and the edge index
        const int32_t ll = label_len[v];
        if (end - p < 96 + (ll > 0 ? ll : 0)) { rc = -1; break; }
        if (child_off[v + 1] > child_off[v]) *p++ = ')';
        if (ll > 0) { memcpy(p, labels + label_off[v], (size_t)ll); p += ll; }
        if (v != root) {
            if (has_len[v]) {
                *p++ = ':';
                const double x = edge_len[v];
                if (std::isnan(x)) p = put(p, "nan");
                else if (std::isinf(x)) p = put(p, x < 0 ? "-inf" : "inf");
                else if (x == std::floor(x)) {
                    if (std::fabs(x) >= 9.2e18) { rc = -2; break; }
                    p = std::to_chars(p, p + 24, (long long)x).ptr;
                } else p = py_float(p, x);
            }
            *p++ = '{';
            p = std::to_chars(p, p + 12, v).ptr;
            *p++ = '}';
        }
        --sp;
    }
    delete[] stack_node;
    delete[] stack_pos;
    return rc < 0 ? rc : p - out;
}
