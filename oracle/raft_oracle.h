/*
 * raft_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Single-threaded plain-C restatement of the RAFT hot path
 *   PAF overlap records -> per-read buckets -> binned coverage -> high-coverage
 *   repeat runs -> cut points -> fragment table
 * as the reference computes it (at-cg/RAFT @ 2024_10_08):
 *   chop.hpp:133-191  create_pileup      (bucketing + symmetric-PAF detection)
 *   repeat.hpp:28-79  profileCoverage    (event select, sort by start, sweep)
 *   repeat.hpp:89-171 repeat_annotate    (run scan, flank/clamp, totals)
 *   chop.hpp:209-321  break_reads        (markers, repeat mask, fragment bounds)
 *
 * It is the parity checker for the HIP path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it; the product (libraft_hip.so, the
 * raft CLI, raft_amd/) never links, loads or falls back to anything in oracle/.
 *
 * Parity pin: validated in the build container against the compiled reference
 * (oracle/_ref/raft, oracle/_ref/libraft_ref.so; recipe: oracle/Makefile) on the
 * survey's micro-vectors G1-G3 and on seeded synthetic sets; expected outputs of
 * those runs are committed under tests/golden/ (see tests/golden/make_golden.py).
 */
#ifndef RAFT_ORACLE_H
#define RAFT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* param.hpp:4-31 (the scalars the path reads) */
typedef struct {
    int32_t reso;            /* -r, default 50    */
    int32_t est_cov;         /* -e, mandatory > 0 */
    double  cov_mul;         /* -m, default 1.5   */
    int32_t repeat_length;   /* -p, default 10000 */
    int32_t interval_length; /* -p, default 10000 */
    int32_t read_length;     /* -l, default 20000 */
    int32_t overlap_length;  /* -v, default 500   */
    int32_t flanking_length; /* -f, default 1000  */
} raft_oracle_params;

/* Same numeric values as include/raft_hip.h's error codes. */
enum {
    RAFT_ORACLE_OK = 0,
    RAFT_ORACLE_ERR_PARAM = 1,    /* reso/interval/repeat length <= 0, read_length/interval_length == 0 (SIGFPE in the reference, chop.hpp:270) */
    RAFT_ORACLE_ERR_READ_ID = 2,  /* record names a read id outside [0, n_reads) (OOB bucket in the reference, chop.hpp:165) */
    RAFT_ORACLE_ERR_COORD = 3,    /* negative coordinate, or interval reaches a bin >= ceil(len/reso) (heap overflow in the reference, repeat.hpp:69-72) */
    RAFT_ORACLE_ERR_FRAGMENT = 4, /* fragment begin F[pos]-overlap_length < 0 (std::out_of_range in the reference, chop.hpp:318) */
    RAFT_ORACLE_ERR_NOMEM = 5
};

typedef struct {
    int32_t  n_reads;
    int32_t  symmetric;          /* final value of algoParams::symmetric_overlaps (chop.hpp:182) */
    int32_t  high_cov;           /* (int)(est_cov * cov_mul), repeat.hpp:89-90 */
    int64_t  n_intervals;        /* events that took part in the pileup */
    int64_t  total_coverage;     /* repeat.hpp:93,116 */
    int64_t  total_windows;      /* repeat.hpp:95,117 (the reference keeps this in an int) */
    int64_t  total_repeat_length;/* repeat.hpp:96,127,152 (unclamped end-start) */
    int64_t  total_read_length;  /* repeat.hpp:97,101 */
    int64_t *cov_offset;         /* [n_reads+1] prefix of ceil(len/reso) */
    int32_t *cov;                /* [cov_offset[n_reads]] counts; bin j of read i starts at j*reso */
    int64_t *rep_offset;         /* [n_reads+1] */
    int32_t *rep_s, *rep_e;      /* flanked+clamped repeats = Read::long_repeats */
    int64_t *cut_offset;         /* [n_reads+1] */
    int32_t *cuts;               /* final_stars per read (chop.hpp:212,225-246) */
    int64_t *frag_offset;        /* [n_reads+1]; read_num of a read's first fragment = frag_offset[i]+1 */
    int32_t *frag_read;          /* fragment table rows */
    int32_t *frag_begin, *frag_end;
} raft_oracle_result;

int  raft_oracle_run(const raft_oracle_params *p, int32_t n_reads, const int32_t *read_len,
                     int64_t n_rec, const int32_t *qid, const int32_t *qs, const int32_t *qe,
                     const int32_t *tid, const int32_t *ts, const int32_t *te,
                     raft_oracle_result *out);
void raft_oracle_free(raft_oracle_result *out);

#ifdef __cplusplus
}
#endif
#endif
