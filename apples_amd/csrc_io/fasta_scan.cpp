// FASTA/FASTQ scanner behind include/apples_io.h (reader semantics of apples/fasta2dic.py:4-39).
#include <cstring>

#include "apples_io.h"

namespace {

// Line cursor over the file image: universal newlines, and the reference's `l[:-1]`, which drops
// the newline -- or, on an unterminated last line, the last character.
struct Lines {
    const uint8_t *p, *end;
    // next line as [b, e); false at end of data
    bool next(const uint8_t *&b, const uint8_t *&e) {
        if (p >= end) return false;
        b = p;
        const uint8_t *q = p;
        while (q < end && *q != '\n' && *q != '\r') ++q;
        if (q == end) {  // no newline: the line loses its last character
            e = q > b ? q - 1 : q;
            p = end;
        } else {
            e = q;
            p = (*q == '\r' && q + 1 < end && q[1] == '\n') ? q + 2 : q + 1;
        }
        return true;
    }
};

}  // namespace

extern "C" int apples_fasta_scan(const uint8_t *data, int64_t n_bytes, const uint8_t *translate, uint8_t *rows,
                                 int64_t n_rows, int64_t *n_records, int64_t *length, int64_t *name_off,
                                 int32_t *name_len, int64_t *bad_record, int64_t *bad_length) {
    Lines ln{data, data + n_bytes};
    const uint8_t *b = nullptr, *e = nullptr;
    const uint8_t *hdr_b = nullptr, *hdr_e = nullptr;  // pending header line ("last" in readfq)
    bool have_hdr = false;
    int64_t rec = 0;
    const int64_t L = *length;
    int rc = 0;
    while (true) {
        if (!have_hdr) {
            while (ln.next(b, e)) {
                if (e > b && (*b == '>' || *b == '@')) { hdr_b = b; hdr_e = e; have_hdr = true; break; }
            }
        }
        if (!have_hdr) break;
        // name = header[1:] up to the first space
        const uint8_t *nb = hdr_b + 1, *ne = nb;
        while (ne < hdr_e && *ne != ' ') ++ne;
        if (rows) {
            if (rec >= n_rows) { rc = 2; break; }
            name_off[rec] = nb - data;
            name_len[rec] = (int32_t)(ne - nb);
        }
        uint8_t *row = rows ? rows + rec * L : nullptr;
        int64_t len = 0;
        have_hdr = false;
        bool plus = false;
        while (ln.next(b, e)) {
            if (e > b && (*b == '@' || *b == '+' || *b == '>')) {
                hdr_b = b; hdr_e = e; have_hdr = true;
                plus = (*b == '+');
                break;
            }
            const int64_t k = e - b;
            if (row && len + k <= L)
                for (int64_t i = 0; i < k; ++i) row[len + i] = translate[b[i]];
            len += k;
        }
        if (!rows && rec == 0) *length = len;
        if (rows && len != L && rc == 0) { rc = 1; *bad_record = rec; *bad_length = len; }
        ++rec;
        if (!have_hdr) break;
        if (plus) {  // FASTQ: skip the quality block (at least `len` characters of lines)
            have_hdr = false;
            int64_t got = 0;
            bool complete = false;
            while (ln.next(b, e)) {
                got += e - b;
                if (got >= len) { complete = true; break; }
            }
            if (!complete) break;
        }
    }
    *n_records = rec;
    return rc;
}
