/*
 * raft_host.h -- C ABI of the host-side text layer around the engine (libraft_host.so):
 * the FASTA/FASTQ and PAF readers that feed include/raft_hip.h, and the writers of the
 * reference's four output files.  Used by the `raft` CLI (raft_amd/host/raft_main.cpp) and,
 * without a GPU, by the CPU tests that check tokenisation and byte-exact formatting.
 *
 * Reference behaviour reproduced (own code; kseq.h is third-party and is not copied):
 *   loadFASTA          chop.hpp:88-131   name = first whitespace-delimited token, multi-line
 *                                        sequences joined, FASTQ qualities dropped, gz via zlib;
 *                                        first read decides real/simulated mode (regex :101)
 *   paf_read/paf_parse paf.hpp:50-100    split on TAB only, >= 10 fields else the line is skipped,
 *                                        columns 3,4,8,9 through strtol -> uint32 -> int
 *   create_pileup      chop.hpp:157-163  name -> read id (FASTA order)
 *   writers            repeat.hpp:105-108,180-203 ; chop.hpp:250-322
 * Inputs on which the reference is undefined are rejected: duplicate FASTA names, PAF names that
 * are not in the FASTA ("Inputs on which the reference is undefined").
 */
#ifndef RAFT_HOST_H
#define RAFT_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RAFT_HOST_OK = 0,
    RAFT_HOST_ERR_OPEN = 1,       /* file missing / unreadable */
    RAFT_HOST_ERR_DUP_NAME = 2,   /* two reads share a name (ids would be ambiguous, chop.hpp:108) */
    RAFT_HOST_ERR_UNKNOWN_NAME = 3,/* PAF names a read that is not in the reads file (chop.hpp:162-165 OOB) */
    RAFT_HOST_ERR_IO = 4,         /* write failed */
    RAFT_HOST_ERR_ARG = 5,
    RAFT_HOST_ERR_COORD = 6,      /* raft_host_pack_windows: a negative coordinate (the engine's RAFT_HIP_ERR_COORD) */
    RAFT_HOST_ERR_RANGE = 7       /* raft_host_pack_windows: a window index beyond 16 bits -- stay with the coordinate columns */
};

typedef struct raft_host_reads raft_host_reads;
typedef struct raft_host_paf raft_host_paf;

/* Worker threads used by the loaders and writers below (the reference is single-threaded; output bytes do not
 * depend on the count).  Default: RAFT_HOST_THREADS, else the hardware threads, capped at 128.  n = 0 restores the
 * default, n = 1 runs everything on the calling thread. */
int            raft_host_set_threads(int n);
int            raft_host_get_threads(void);

/* reads */
int            raft_host_reads_load(const char *path, raft_host_reads **out);
void           raft_host_reads_free(raft_host_reads *r);
int32_t        raft_host_reads_count(const raft_host_reads *r);
const int32_t *raft_host_reads_lengths(const raft_host_reads *r);          /* [count] */
const char    *raft_host_reads_name(const raft_host_reads *r, int32_t i);
const char    *raft_host_reads_bases(const raft_host_reads *r, int32_t i); /* not NUL-terminated; lengths[i] bytes */
int            raft_host_reads_real(const raft_host_reads *r);             /* algoParams::real_reads */

/* overlaps: name -> id resolution against `reads`; err_name (may be NULL) receives the offending name */
int            raft_host_paf_load(const char *path, const raft_host_reads *reads, raft_host_paf **out,
                                  char *err_name, int err_name_cap);
/* The same in two steps: the file's bytes (read by all workers, or inflated when it is .gz, paf.hpp:29) need nothing of
 * the reads and can be fetched on a thread of the caller's while raft_host_reads_load is still running; the parse
 * (paf.hpp:50-87, chop.hpp:155-165) tokenises the text in place and consumes it.  load == read + parse. */
typedef struct raft_host_text raft_host_text;
int            raft_host_text_read(const char *path, raft_host_text **out);
void           raft_host_text_free(raft_host_text *t);
int            raft_host_paf_parse(raft_host_text *text, const raft_host_reads *reads, raft_host_paf **out,
                                   char *err_name, int err_name_cap);
void           raft_host_paf_free(raft_host_paf *p);
int64_t        raft_host_paf_count(const raft_host_paf *p);                /* accepted records */
const int32_t *raft_host_paf_column(const raft_host_paf *p, int k);        /* k: 0 qid 1 qs 2 qe 3 tid 4 ts 5 te */

/* algoParams::symmetric_overlaps as create_pileup leaves it (chop.hpp:171-184): 1 when some accepted record after the
 * first mirrors the first.  Found while the lines are tokenised; passed to the engine as symmetric_mode, which then
 * neither runs its own detection nor needs the three target columns. */
int            raft_host_paf_symmetric(const raft_host_paf *p);

/* The grouped form of a tokenised query column, for raft_hip_run_*_grouped (include/raft_hip.h): hifiasm writes its PAF
 * grouped by query (reference README.md:36-38; create_pileup meets the records in that order, chop.hpp:147-169), so the
 * column is a handful of runs sorted by read id.  When it is at most max_runs such runs and every id lies in
 * [0, n_reads), *n_runs receives their number and rec_offset[k * (n_reads + 1) + r] the index of the first record of
 * read r in run k (entry n_reads closes the run); otherwise *n_runs = 0 and the caller stays with the per-record ids.
 * rec_offset holds max_runs * (n_reads + 1) entries (page-locked when it is to be uploaded at the link's rate). */
int raft_host_group_offsets(int32_t n_reads, int64_t n_rec, const int32_t *qid, int32_t max_runs, int32_t *n_runs,
                            int64_t *rec_offset);

/* ... and in the four-bit step encoding (include/raft_hip.h, cov_width = RAFT_HIP_COV_DELTA4): nib[] holds a step + 8 per
 * window (low nibble = even window) or 0 for a window listed with its value, anchor[k] = cov[1024 k - 1].  exc_index must
 * be ascending (as the engine hands it out). */
int raft_host_unpack_coverage_d4(int64_t n_bins, const uint8_t *nib, const int32_t *anchor, int64_t n_exc, const int64_t *exc_index,
                                 const int32_t *exc_value, int32_t *cov);
int raft_host_write_coverage_d4(const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset, const uint8_t *nib,
                                const int32_t *anchor, int64_t n_exc, const int64_t *exc_index, const int32_t *exc_value);

/* Window records for raft_hip_run_*_windows (include/raft_hip.h): win[i] = first | last1 << 16 with first = qs / reso and
 * last1 = (qe - 1) / reso + 1 -- the windows profileCoverage adds the interval to (repeat.hpp:69-72) -- or 0 for an
 * interval without windows (qe == 0, or last1 <= first).  The integer divisions of the pileup, done where the
 * coordinates are tokenised; 4 bytes per record instead of 8.
 *   RAFT_HOST_ERR_COORD  a coordinate is negative (*bad_index = the first such record): what the engine reports as
 *                        RAFT_HIP_ERR_COORD with that index;
 *   RAFT_HOST_ERR_RANGE  some interval ends beyond window 65,535 (*bad_index = the first such record): not expressible --
 *                        the caller keeps the coordinate columns (raft_hip_run_*_grouped);
 * win[] is undefined after either.  reso must be in [1, 32767]. */
int raft_host_pack_windows(int64_t n_rec, const int32_t *qs, const int32_t *qe, int32_t reso, uint32_t *win, int64_t *bad_index);

/* Coverage in the engine's transfer encoding (raft_hip_fetch_packed: one byte per window, 255 = look up the ascending
 * exception list): back to int32, and straight to coverage.txt (repeat.hpp:105-108) without the int32 detour. */
int raft_host_unpack_coverage(int64_t n_bins, const uint8_t *cov8, int64_t n_exc, const int64_t *exc_index,
                              const int32_t *exc_value, int32_t *cov);
int raft_host_write_coverage_packed(const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset,
                                    const uint8_t *cov8, int64_t n_exc, const int64_t *exc_index, const int32_t *exc_value);
/* ... and for either width of the encoding (1: uint8 codes, escape 255; 2: uint16 codes, escape 65535) */
int raft_host_unpack_coverage_w(int32_t width, int64_t n_bins, const void *cov_packed, int64_t n_exc, const int64_t *exc_index,
                                const int32_t *exc_value, int32_t *cov);
int raft_host_write_coverage_packed_w(int32_t width, const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset,
                                      const void *cov_packed, int64_t n_exc, const int64_t *exc_index, const int32_t *exc_value);

/* writers (CSR arrays as returned by raft_hip_fetch) */
int raft_host_write_coverage(const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset, const int32_t *cov);
int raft_host_write_repeats(const char *txt_path, const char *bed_path, const raft_host_reads *reads,
                            const int64_t *rep_offset, const int32_t *rep_s, const int32_t *rep_e);
int raft_host_write_fasta(const char *path, const raft_host_reads *reads, const int64_t *frag_offset,
                          const int32_t *frag_begin, const int32_t *frag_end);

/* The reference's stand-alone comparator tool (split_naive.cpp:10-44): every read of in_path cut into consecutive
 * pieces of split_len bases, written to out_path as ">name_k" records, k from 1.  n_reads (may be NULL) receives the
 * number of input records. */
int raft_host_split_naive(const char *in_path, const char *out_path, int32_t split_len, int32_t *n_reads);

#ifdef __cplusplus
}
#endif
#endif
