// FASTA/FASTQ scanner behind include/apples_io.h (reader semantics of apples/fasta2dic.py:4-39).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

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

// ---------------------------------------------------------------------------------------------------------------------
// The same for the common shape -- plain FASTA, "\n" line ends -- in one threaded pass over the file image (at
// 200 000 x 1 000 the line-by-line scanner above, run twice for the count and the fill, took 0.7 s; the whole GPU pass over the
// file's 100 000 queries takes 0.05 s).  Phase 1, by byte ranges: the offsets of the record headers = '>' at a line start
// (memchr for the line ends); anything the reader above treats specially -- a '\r', a line starting with '@' or '+' -- makes
// the caller take the general scanner.  Phase 2, by record ranges: name ranges, and the sequence lines copied through the
// translation table straight into their rows (memchr per line, 16 bytes of table look-ups per unrolled round).
// Returns 0; 3 = not the plain shape (nothing written that the caller may use); 1 / 2 as apples_fasta_scan.  *length is taken
// from the first record.  rows == NULL: records and length only.
extern "C" int apples_fasta_scan_mt(const uint8_t *data, int64_t n_bytes, const uint8_t *translate, uint8_t *rows, int64_t n_rows,
                                    int64_t *n_records, int64_t *length, int64_t *name_off, int32_t *name_len,
                                    int64_t *bad_record, int64_t *bad_length, int32_t n_threads) {
    if (n_bytes <= 0) { *n_records = 0; return 0; }
    int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    T = std::max(1, std::min(T, 64));
    if (n_bytes < (1 << 20)) T = 1;
    // phase 1: header offsets per byte range (a line start = offset 0 or the byte after a '\n')
    std::vector<std::vector<int64_t>> starts(T);
    std::atomic<int> odd{0};
    auto index = [&](int t) {
        const int64_t lo = n_bytes * t / T, hi = n_bytes * (t + 1) / T;
        std::vector<int64_t> &out = starts[t];
        if (memchr(data + lo, '\r', (size_t)(hi - lo))) { odd = 1; return; }
        int64_t p = lo;
        if (p > 0) {  // first line start inside the range
            const uint8_t *q = (const uint8_t *)memchr(data + p - 1, '\n', (size_t)(hi - p + 1));
            if (!q) return;
            p = (q - data) + 1;
        }
        while (p < hi) {
            const uint8_t c = data[p];
            if (c == '>') out.push_back(p);
            else if (c == '@' || c == '+') { odd = 1; return; }
            const uint8_t *q = (const uint8_t *)memchr(data + p, '\n', (size_t)(n_bytes - p));
            if (!q) break;
            p = (q - data) + 1;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(index, t);
        index(0);
        for (auto &x : th) x.join();
    }
    if (odd) return 3;
    // (a lone '>' as the file's unterminated last line ends a sequence but opens no record: the reader's `l[:-1]` leaves nothing of it)
    if (data[n_bytes - 1] == '>' && (n_bytes == 1 || data[n_bytes - 2] == '\n')) return 3;
    std::vector<int64_t> rec;
    for (auto &v : starts) rec.insert(rec.end(), v.begin(), v.end());
    const int64_t n = (int64_t)rec.size();
    *n_records = n;
    if (n == 0) return 0;
    rec.push_back(n_bytes);
    // sequence length of one record (lines lose their last byte: the '\n', or the last character of an unterminated last line)
    auto walk = [&](int64_t r, uint8_t *row, int64_t L, int64_t *noff, int32_t *nlen) -> int64_t {
        const uint8_t *p = data + rec[r], *end = data + rec[r + 1];
        const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
        const uint8_t *he = q ? q : (end > p ? end - 1 : end);  // header line without its last byte
        if (noff) {
            const uint8_t *nb = p + 1, *ne = nb;
            while (ne < he && *ne != ' ') ++ne;
            if (nb > he) nb = ne = he;
            *noff = nb - data;
            *nlen = (int32_t)(ne - nb);
        }
        int64_t len = 0;
        p = q ? q + 1 : end;
        while (p < end) {
            q = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
            const uint8_t *le = q ? q : end - 1;
            const int64_t k = le - p;
            if (row && len + k <= L) {
                uint8_t *d = row + len;
                int64_t i = 0;
                for (; i + 16 <= k; i += 16)
                    for (int j = 0; j < 16; ++j) d[i + j] = translate[p[i + j]];
                for (; i < k; ++i) d[i] = translate[p[i]];
            }
            len += k;
            p = q ? q + 1 : end;
        }
        return len;
    };
    const int64_t L = walk(0, nullptr, 0, nullptr, nullptr);
    *length = L;
    if (!rows) return 0;
    if (n > n_rows) return 2;
    std::atomic<int64_t> first_bad{n};
    std::vector<int64_t> bad_len(T, 0), bad_at(T, n);
    auto fill = [&](int t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t len = walk(r, rows + r * L, L, name_off + r, name_len + r);
            if (len != L && bad_at[t] == n) { bad_at[t] = r; bad_len[t] = len; }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(fill, t);
        fill(0);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; ++t)
        if (bad_at[t] < n) { *bad_record = bad_at[t]; *bad_length = bad_len[t]; return 1; }
    return 0;
}
