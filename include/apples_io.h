/* Host-side input scanning for the APPLES hot path's callers: plain C ABI, no
 * device code.  Built as apples_amd/libapples_io.so by `python -m apples_amd.build`.
 *
 * apples_fasta_scan replaces the per-record Python of the reference's reader
 * (apples/fasta2dic.py:4-39 readfq, :42-72 fasta2dic) on the way to the dense N x L byte matrix the
 * distance kernels take: one pass over the file image, sequences written straight into their rows
 * through the 256-entry byte translation the caller derived from the alphabet flags
 * (fasta2dic.py:56-67).  Reader semantics kept: a record starts at a line whose first byte is '>'
 * or '@'; its name is the header up to the first space; sequence lines run until a line starting
 * with '@', '+' or '>'; after '+' a FASTQ quality block of at least the sequence's length is
 * skipped; every line loses its last character (so an unterminated final line loses a base);
 * "\r\n" and lone "\r" end a line (the reference opens the file in text mode).
 */
#ifndef APPLES_IO_H
#define APPLES_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Pass 1 (rows == NULL): counts records, reports the first record's sequence length in *length.
 * Pass 2 (rows != NULL, n_rows x length bytes): fills rows and name_off/name_len (byte range of
 * every name inside `data`).  Returns 0, or 1 if a record's length differs from *length
 * (*bad_record = its index, *bad_length = its length), or 2 if n_rows is too small. */
int apples_fasta_scan(const uint8_t *data, int64_t n_bytes, const uint8_t *translate /*[256]*/,
                      uint8_t *rows, int64_t n_rows, int64_t *n_records, int64_t *length,
                      int64_t *name_off, int32_t *name_len, int64_t *bad_record, int64_t *bad_length);

#ifdef __cplusplus
}
#endif
#endif
