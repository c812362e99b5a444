/*
 * raft_hip.h -- C ABI of the MI355X (gfx950) engine for RAFT's hot path:
 *
 *   PAF overlap records -> per-read interval buckets -> binned coverage pileup
 *   -> high-coverage repeat runs -> cut points -> fragment table
 *
 * The reference (at-cg/RAFT @ 2024_10_08) has no FFI; the in-process seam this
 * library replaces is the three calls in break_long_reads():
 *
 *   create_pileup(paf, reads, idx_pileup, umap, param)   chop.hpp:366  (-> :133-191)
 *   repeat_annotate(reads, idx_pileup, param)            chop.hpp:370  (-> repeat.hpp:81-204,
 *                                                                         profileCoverage :28-79)
 *   break_reads(param, n_read, reads, reads_final)       chop.hpp:372  (-> :193-324, integer half)
 *
 * i.e. everything between "a PAF record has been tokenised into 2 resolved read
 * ids + 4 coordinates" (chop.hpp:157-163) and "a read's final_stars / fragment
 * bounds exist" (chop.hpp:242-321).  Text parsing and text/FASTA emission stay on
 * the host (raft_amd/host/, the `raft` CLI).
 *
 * Conventions: plain pointers and sizes only; every entry point returns an
 * RAFT_HIP_* code and never throws or exits; one context per host thread, one
 * HIP stream per context.  The caller owns all inputs; the context owns all
 * outputs until the next run or raft_hip_destroy().  Inputs on which the
 * reference has undefined behaviour are rejected with a defined error instead
 * (SURVEY.md §5.3): unknown read id, interval reaching past the last bin,
 * read_length < interval_length, fragment start before 0.
 */
#ifndef RAFT_HIP_H
#define RAFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: what this header declares is everything it exports. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define RAFT_HIP_ABI_VERSION 11

/* error codes (0..5 are shared with oracle/raft_oracle.h) */
enum {
    RAFT_HIP_OK = 0,
    RAFT_HIP_ERR_PARAM = 1,    /* reso/interval_length/repeat_length/est_cov <= 0, read_length/interval_length == 0, negative read length */
    RAFT_HIP_ERR_READ_ID = 2,  /* a record names a read id outside [0, n_reads)            (chop.hpp:165 OOB)       */
    RAFT_HIP_ERR_COORD = 3,    /* negative coordinate or interval reaching a bin >= ceil(len/reso) (repeat.hpp:69-72 OOB) */
    RAFT_HIP_ERR_FRAGMENT = 4, /* a fragment would start before base 0 (F[pos] - overlap_length < 0, chop.hpp:318 throws) */
    RAFT_HIP_ERR_NOMEM = 5,    /* host or device allocation failed */
    RAFT_HIP_ERR_DEVICE = 6,   /* no usable gfx950 device / HIP runtime error (see raft_hip_last_error) */
    RAFT_HIP_ERR_STATE = 7,    /* call order violated (e.g. fetch before run) */
    RAFT_HIP_ERR_TOO_LARGE = 8 /* more than 2^31-1 reads, or a per-read quantity overflowing int32 */
};

/* algoParams (param.hpp:4-31): the scalars the path reads.  symmetric_mode:
 * -1 = detect as the reference does (first later record that mirrors record 0,
 * chop.hpp:175-184); 0 / 1 = caller asserts the final value of
 * algoParams::symmetric_overlaps and the detection pass is skipped. */
typedef struct raft_hip_params {
    int32_t reso;            /* -r  (param.hpp:20, default 50)    */
    int32_t est_cov;         /* -e  (mandatory > 0, main.cpp:65)  */
    double  cov_mul;         /* -m  (default 1.5); high_cov = (int)(est_cov*cov_mul), repeat.hpp:89-90 */
    int32_t repeat_length;   /* -p  (default 10000)               */
    int32_t interval_length; /* -p  (default 10000)               */
    int32_t read_length;     /* -l  (default 20000)               */
    int32_t overlap_length;  /* -v  (default 500)                 */
    int32_t flanking_length; /* -f  (default 1000)                */
    int32_t symmetric_mode;  /* -1 auto | 0 | 1                   */
} raft_hip_params;

/* Scalars of one finished run (host values). */
typedef struct raft_hip_summary {
    int32_t n_reads;
    int32_t symmetric;          /* resolved algoParams::symmetric_overlaps (chop.hpp:189) */
    int32_t high_cov;           /* repeat.hpp:90-91 */
    int32_t interval_path;      /* 0 = sorted-segment fast path (records already grouped by ascending query id in
                                   <= 4 runs, query side only); 1 = counting-sort bucketing path */
    int32_t n_segments;         /* sorted runs found in the record stream (fast path) */
    int64_t n_records;          /* PAF records consumed ("length of alignments", chop.hpp:190) */
    int64_t n_intervals;        /* intervals piled up (query sides + target sides when not symmetric) */
    int64_t n_bins;             /* sum of ceil(len/reso) */
    int64_t n_repeats, n_cuts, n_fragments;
    int64_t total_coverage;     /* repeat.hpp:93,116 */
    int64_t total_windows;      /* repeat.hpp:95,117 (int in the reference; int64 here) */
    int64_t total_repeat_length;/* repeat.hpp:96,127,152 */
    int64_t total_read_length;  /* repeat.hpp:97,101 */
    int64_t error_index;        /* what raised the returned error, -1 if none: the read index for a negative length or
                                   RAFT_HIP_ERR_FRAGMENT; the PAF record index for RAFT_HIP_ERR_READ_ID and, on the
                                   sorted-segment path, RAFT_HIP_ERR_COORD; on the counting-sort path (interval_path
                                   == 1) RAFT_HIP_ERR_COORD reports the index into the BUCKETED interval array */
    int32_t n_devices_used;     /* host-to-host entry points: contexts that took part in the job (1 for a one-piece pass) */
    int32_t flags;              /* RAFT_HIP_SUM_*: bit 0 -- the general bucketing (interval_path == 1) sorted its sides as window-record items
                                   and the pileup kernel read those (4 bytes per interval) instead of coordinate columns;
                                   bit 1 -- the pass was built, without its host wait, on what the context's previous pass over a stream
                                   of the same shape had found (sizes, sorted runs), and the device confirmed it (raft_hip_run_device);
                                   bits 2, 3 -- see RAFT_HIP_SUM_DEEP_TILES, RAFT_HIP_SUM_RERUN */
} raft_hip_summary;
#define RAFT_HIP_SUM_BUCKET_WINDOWS 1
#define RAFT_HIP_SUM_SPECULATED 2
#define RAFT_HIP_SUM_DEEP_TILES 4   /* tiles of 2^15 intervals or more took the 32-bit side kernel (pileup_deep.hpp) */
#define RAFT_HIP_SUM_RERUN 8        /* raft_hip_finish ran the pass more than once (a refuted guess or assumption, a list that had to grow) */

/* Device-resident outputs of the last run (valid until the next run/destroy).
 * Layout is CSR per read, FASTA-index order (= reference output order):
 *   cov[cov_offset[i] + j]  = coverage of window j of read i   (repeat.hpp:105-108: "pos,cov" with pos = j*reso)
 *   rep_s/rep_e[rep_offset[i] ..]  = Read::long_repeats of read i (repeat.hpp:142,167)
 *   cuts[cut_offset[i] ..]         = final_stars of read i        (chop.hpp:225-246)
 *   frag_*[frag_offset[i] ..]      = fragments of read i; read_num = row index + 1 (chop.hpp:195,266,319);
 *                                    sequence = bases[frag_begin, frag_end)          (chop.hpp:265,318) */
typedef struct raft_hip_outputs {
    const int64_t *cov_offset;  /* [n_reads+1] */
    const int32_t *cov;         /* [n_bins]    */
    const int64_t *rep_offset;  /* [n_reads+1] */
    const int32_t *rep_s, *rep_e;
    const int64_t *cut_offset;  /* [n_reads+1] */
    const int32_t *cuts;
    const int64_t *frag_offset; /* [n_reads+1] */
    const int32_t *frag_read, *frag_begin, *frag_end;
} raft_hip_outputs;

typedef struct raft_hip_ctx raft_hip_ctx;

int         raft_hip_abi_version(void);
const char *raft_hip_strerror(int code);
/* Text of the last HIP runtime failure seen by this context ("" if none). */
const char *raft_hip_last_error(const raft_hip_ctx *ctx);

/* Binds a context to HIP device `device_id` (must be gfx950) and validates params. */
int  raft_hip_create(int device_id, const raft_hip_params *params, raft_hip_ctx **out);
void raft_hip_destroy(raft_hip_ctx *ctx);
int  raft_hip_set_params(raft_hip_ctx *ctx, const raft_hip_params *params);

/* Use `stream` (a hipStream_t; NULL = the device's default stream, which is what
 * torch.cuda.current_stream() is unless the caller changed it) for all work of this
 * context; raft_hip_use_own_stream() switches back to the context's private stream. */
int  raft_hip_set_stream(raft_hip_ctx *ctx, void *stream);
int  raft_hip_use_own_stream(raft_hip_ctx *ctx);
void *raft_hip_get_stream(raft_hip_ctx *ctx);

/* One pass of the hot path over inputs that already live in device memory
 * (int32 SoA columns, 4-byte aligned; read ids are FASTA indices).  Enqueues
 * every kernel; returns after the last launch, not after completion.  It waits
 * for the device once on the way (sizes of the coverage array and choice of
 * interval path come back together), which is part of the cost of a pass.
 * Every input column must stay valid and unchanged until raft_hip_finish() has
 * returned: a pass that a kernel refutes (a stream that is not what its samples
 * suggested, an exception list that overflowed) is run again from the same
 * pointers inside raft_hip_finish().  d_read_len must stay valid until the outputs
 * have been fetched: the cut points (chop.hpp's final_stars) are materialised by
 * the first raft_hip_fetch() / raft_hip_outputs_device() that asks for them, not
 * by the pass. */
int  raft_hip_run_device(raft_hip_ctx *ctx, int32_t n_reads, const int32_t *d_read_len,
                         int64_t n_rec, const int32_t *d_qid, const int32_t *d_qs, const int32_t *d_qe,
                         const int32_t *d_tid, const int32_t *d_ts, const int32_t *d_te);

/* Same pass from host memory: stages the columns to the device first.  With symmetric_mode = 1 only read_len and the
 * three query columns are uploaded; tid/ts/te are not read and may be NULL (also in raft_hip_run_device). */
int  raft_hip_run_host(raft_hip_ctx *ctx, int32_t n_reads, const int32_t *read_len,
                       int64_t n_rec, const int32_t *qid, const int32_t *qs, const int32_t *qe,
                       const int32_t *tid, const int32_t *ts, const int32_t *te);

/* The same pass over GROUPED input.  hifiasm writes its PAF grouped by query (reference README.md:36-38; the records
 * reach create_pileup in that order, chop.hpp:147-169) and a tokeniser that resolves every name knows where each read's
 * records begin: the record stream is n_runs (1..RAFT_HIP_MAX_RUNS = 16) runs, each sorted by query id (more than four are merged into
 * one on the device first, 24 bytes of traffic per record), and
 *     rec_offset[k * (n_reads + 1) + r]  =  index of the first record of read r in run k,
 * entry n_reads of a run closing it: rec_offset[0] = 0, run k + 1 begins where run k ends, the last run ends at n_rec,
 * offsets never step back (raft_host_paf_grouped() builds this from the tokenised columns).  What create_pileup's
 * bucketing has to find out is then handed over -- query sides only, so the context must assert symmetric_mode = 1:
 *   * no look at the record stream before the pass, no searches for the tile cuts (look-ups in rec_offset);
 *   * d_qid may be NULL: the ids ARE the offsets and are rebuilt on the device (4 of the 12 bytes per record need not
 *     cross PCIe -- the host forms below never upload them).  When it is given, every record is checked against the
 *     reads of the tile that processes it, and a record that does not sit where the offsets say sends the pass to the
 *     plain form above (results as from raft_hip_run_device, whatever the offsets were);
 *   * n_bins >= 0: the caller's sum of ceil(len / reso) over the reads (the loader of the reads has the lengths).  The
 *     host then sizes every buffer without waiting for the device -- the pass is one uninterrupted sequence of launches;
 *     a count that is not what the lengths give is noticed on the device and costs a second pass.  n_bins = -1: unknown,
 *     the pass waits for the device once, as raft_hip_run_device does.
 * Errors: offsets that step back or do not chain from 0 to n_rec -> RAFT_HIP_ERR_PARAM (error_index = the read). */
int  raft_hip_run_device_grouped(raft_hip_ctx *ctx, int32_t n_reads, const int32_t *d_read_len, int64_t n_rec, int32_t n_runs,
                                 const int64_t *d_rec_offset, const int32_t *d_qid, const int32_t *d_qs, const int32_t *d_qe,
                                 int64_t n_bins);
/* ... from host memory: uploads read_len, rec_offset, qs, qe (8 bytes per record instead of 12); n_bins = -1: counted here. */
int  raft_hip_run_host_grouped(raft_hip_ctx *ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec, int32_t n_runs,
                               const int64_t *rec_offset, const int32_t *qs, const int32_t *qe, int64_t n_bins);

/* Grouped input as WINDOW RECORDS: one 32-bit word per record instead of (qs, qe) --
 *     win[i] = first | last1 << 16,   first = qs / reso,   last1 = qe > 0 ? (qe - 1) / reso + 1 : 0   (repeat.hpp:69-72: the
 *     windows first .. last1 - 1 of the query get +1; an interval without windows is stored as 0)
 * -- cut from the coordinates where they are tokenised (raft_host_pack_windows; possible while every window index fits 16
 * bits, i.e. reads below 65,535 * reso bases, 3.2 Mbp at the default).  The pileup reads nothing else of a record: its read
 * is where rec_offset says.  4 bytes per record cross PCIe and are read by the pileup kernel instead of 8 + the 4 of the
 * rebuilt ids.  Same outputs; a record reaching past the last window of its read is RAFT_HIP_ERR_COORD with its index as
 * before (negative coordinates cannot be expressed: raft_host_pack_windows reports them).  Requires symmetric_mode = 1 and
 * reso <= 32767.  n_runs > 2, non-default tuning variants and the fallbacks of the pass are served by unpacking the
 * records into coordinate columns on the device first. */
int  raft_hip_run_device_windows(raft_hip_ctx *ctx, int32_t n_reads, const int32_t *d_read_len, int64_t n_rec, int32_t n_runs,
                                 const int64_t *d_rec_offset, const uint32_t *d_win, int64_t n_bins);
int  raft_hip_run_host_windows(raft_hip_ctx *ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec, int32_t n_runs,
                               const int64_t *rec_offset, const uint32_t *win, int64_t n_bins);

/* Waits for the pass, reads back its scalars and reports data errors found on
 * the device (RAFT_HIP_ERR_READ_ID / _COORD / _FRAGMENT). */
int  raft_hip_finish(raft_hip_ctx *ctx, raft_hip_summary *summary);

/* Device pointers of the finished pass. */
int  raft_hip_outputs_device(raft_hip_ctx *ctx, raft_hip_outputs *out);

/* Copies outputs of the finished pass to caller-provided host arrays sized from
 * the summary (any pointer may be NULL to skip that array). */
int  raft_hip_fetch(raft_hip_ctx *ctx, int64_t *cov_offset, int32_t *cov,
                    int64_t *rep_offset, int32_t *rep_s, int32_t *rep_e,
                    int64_t *cut_offset, int32_t *cuts,
                    int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end);

/* The same outputs with the coverage array in its transfer encoding: one byte per window, cov8[i] = min(cov[i], 255),
 * plus the windows with cov >= 255 as (exc_index, exc_value) pairs, ascending by window index (cov[] is two thirds of
 * the bytes that cross PCIe; the reference's consumer is the text formatter of repeat.hpp:105-108, which
 * raft_host_write_coverage_packed() serves from this form; raft_host_unpack_coverage() restores the int32 array).
 * *n_exc receives the number of exceptions; when it exceeds exc_cap and any of cov8 / exc_index / exc_value is asked
 * for, nothing is copied and RAFT_HIP_ERR_TOO_LARGE is returned -- call again with larger arrays (the size query, all
 * three NULL, always succeeds).  Any pointer except n_exc may be NULL to skip that array. */
int  raft_hip_fetch_packed(raft_hip_ctx *ctx, int64_t *cov_offset, uint8_t *cov8, int64_t exc_cap, int64_t *exc_index,
                           int32_t *exc_value, int64_t *n_exc, int64_t *rep_offset, int32_t *rep_s, int32_t *rep_e,
                           int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end);
/* The same with the width of the encoding chosen by the caller: width = 1 as above (limit 255), width = 2 -- uint16 per
 * window, limit 65535 -- for deep sets whose repeats pile up beyond a byte (at 60x with six-copy tandem arrays half of
 * all windows are at or above 255 and a byte per window has to list them one by one).  cov_packed holds
 * n_bins * width bytes. */
int  raft_hip_fetch_packed_w(raft_hip_ctx *ctx, int32_t width, int64_t *cov_offset, void *cov_packed, int64_t exc_cap,
                             int64_t *exc_index, int32_t *exc_value, int64_t *n_exc, int64_t *rep_offset, int32_t *rep_s,
                             int32_t *rep_e, int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end);

/* "delta4": four bits per window.  Once the records cross PCIe as one word each (raft_hip_run_*_windows), the coverage
 * array is two thirds of a job's bytes, and it barely moves from one window to the next -- the step cov[w] - cov[w-1] is
 * the pileup's own difference array, within +-7 for 99.8 % of the windows of a 32x set.  Over the concatenated array
 * cov[0 .. n_bins) (cov[-1] = 0):
 *     cov_nib[w >> 1], low nibble = even w:  step + 8 for a step in [-7, 7], 0 = escape
 *     (exc_index, exc_value), ascending:      every escaped window with its ABSOLUTE value (large steps -- mostly where one
 *                                             read ends and the next begins -- and each pileup tile's first window)
 *     cov_anchor[k] = cov[1024 k - 1]:        a decoder starts at any multiple of 1024 windows (cov_anchor[0] = 0; where
 *                                             window 1024 k itself is escaped the entry is not needed and may be 0)
 * cov_nib holds (n_bins + 1) / 2 bytes, cov_anchor (n_bins + 1023) / 1024 entries.  raft_host_unpack_coverage_d4 /
 * raft_host_write_coverage_d4 decode it; exceptions and the size query behave as in raft_hip_fetch_packed. */
#define RAFT_HIP_COV_DELTA4 8
int  raft_hip_fetch_delta4(raft_hip_ctx *ctx, int64_t *cov_offset, uint8_t *cov_nib, int32_t *cov_anchor, int64_t exc_cap,
                           int64_t *exc_index, int32_t *exc_value, int64_t *n_exc, int64_t *rep_offset, int32_t *rep_s,
                           int32_t *rep_e, int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end);

/* Output width of the context's later passes: 4 (the default) -- cov[] is written as int32; 1 or 2 -- the pileup kernel
 * writes the transfer encoding above directly (four fifths of a pass's HBM traffic is this array, and its consumer,
 * repeat.hpp:105-108, is a text formatter) and the int32 array exists only if somebody asks for it: raft_hip_fetch() and
 * raft_hip_outputs_device() decode it on the device at their first call, raft_hip_fetch_packed_w() of the same width is a
 * plain copy, raft_hip_packed_device() hands out the device arrays.  Results are the same in every width; a width whose
 * limit most windows reach (width 1 on a 60x set) costs a second pass, because the list of exceptions is sized for the
 * usual case first.  The host pipelines below set the width their caller's buffers ask for by themselves.
 * width = RAFT_HIP_COV_DELTA4: the pileup kernel writes the four-bit step encoding above (raft_hip_fetch_delta4 is then a
 * plain copy). */
int  raft_hip_set_output_width(raft_hip_ctx *ctx, int32_t width);

/* Cut points (chop.hpp:225-246 final_stars; `cuts` / `cut_offset` of raft_hip_outputs and raft_hip_fetch): on = 1 (the
 * default) -- every later pass of the context writes them itself, in the kernel that also writes the fragment table;
 * on = 0 -- they are left out of the pass and written by the first raft_hip_fetch / raft_hip_outputs_device that asks for
 * them (the fragment bounds are derived without them; the host pipelines, whose outputs hold no cut points, run this way). */
int  raft_hip_set_emit_cuts(raft_hip_ctx *ctx, int32_t on);

/* Device memory for the caller's input columns (ABI 8), placed the way the engine places its own large arrays: a virtual
 * range backed by 32 MiB physical chunks in a shuffled order.  What a stream of loads and stores gets from this part
 * depends on where its buffers lie (tools/membench: the int32 pass's mix of traffic takes 2.0 ms on such chunks, 2.2-2.3 ms
 * on one physically contiguous block, and anything in between on hipMalloc memory, depending on what earlier processes left
 * behind); columns that live in memory from here make a pass's time reproducible.  Plain hipMalloc memory stays valid
 * input everywhere.  The buffer belongs to the context's device and lives until raft_hip_device_free or
 * raft_hip_destroy; falls back to hipMalloc where the mapping calls are unavailable.  (Replaces nothing in the
 * reference: chop.hpp:155-169 fills host vectors.) */
int  raft_hip_device_alloc(raft_hip_ctx *ctx, int64_t bytes, void **dptr);
int  raft_hip_device_free(raft_hip_ctx *ctx, void *dptr);

/* The placement's memory (ABI 10).  Physical chunks no buffer maps at the moment wait in a per-device pool for the next
 * buffer; the pool holds RAFT_VMM_POOL_GB GiB at most (default 64), spare chunks are only made while an eighth of the device's
 * memory (8 GiB at least) stays free behind them, an allocation that fails hands the pool back and is made once more, and the
 * last context of a device to be destroyed hands all of it back.  raft_hip_trim does that on demand: every pooled chunk of
 * the device beyond keep_bytes goes back to the driver (returns the bytes released, or a negative RAFT_HIP_ERR_*);
 * raft_hip_pool_bytes reports what the pool holds.  (Replaces nothing in the reference.) */
int64_t raft_hip_trim(int device_id, int64_t keep_bytes);
int64_t raft_hip_pool_bytes(int device_id);
/* Where buffers made from now on lie (process-wide; returns the setting before): 0 -- plain hipMalloc; k >= 1 -- shuffled 32 MiB
 * chunks, and for a buffer of 1 GiB or more k times as many made as used (every k-th taken: the default is 8, RAFT_VMM_SPREAD /
 * RAFT_NO_VMM set the start value).  Buffers that exist keep their memory.  bench.py's `placement_ab` times the pileup kernel
 * under 8, 1 and 0 in one process. */
int32_t raft_hip_set_placement(int32_t spread);
/* The placement trial of the context's coverage array (ABI 10; OPT-IN since ABI 11).  What the pileup kernel gets from this part
 * follows the array it stores into -- by the draw, not by the kind of memory: two hipMalloc blocks of one process gave 2.24 and
 * 2.63 ms.  A context that asked for it (raft_hip_set_placement_trial(ctx, k), k = 2..8 candidates; or
 * RAFT_PLACEMENT_TRIALS=<k> for every context of the process; 0 / 1 = off, the default) draws k - 1 more arrays at the first pass
 * that makes an int32 coverage array of 1 GiB or more -- plain blocks and chunk mappings in turn, each only while an eighth of
 * the device's memory (8 GiB at least) stays free behind it --, runs the kernel into each of them warm in that pass (2 k - 1
 * more launches, one host wait) and keeps the fastest; not after raft_hip_set_placement / RAFT_NO_VMM / RAFT_VMM_SPREAD chose
 * by hand.  Off by default because a one-shot caller has nothing to amortise it over and the bench's own A/B of the policies
 * shows differences of 0.2 % on most leases.  raft_hip_placement_trial reports the kernel's ms into the first placement and
 * into the best other candidate, and what was kept (0 the first placement, 1 a plain block, 2 another chunk mapping);
 * RAFT_HIP_ERR_STATE when no trial has run. */
int  raft_hip_set_placement_trial(raft_hip_ctx *ctx, int32_t candidates);
int  raft_hip_placement_trial(raft_hip_ctx *ctx, double *first_ms, double *best_other_ms, int32_t *kept);

/* Device arrays of the encoding the finished pass holds (width 0: none -- the pass wrote int32; call
 * raft_hip_fetch_packed_w once to have it encoded).  The exceptions are in no particular order. */
int  raft_hip_packed_device(raft_hip_ctx *ctx, int32_t *width, const void **cov_packed, const int64_t **exc_index,
                            const int32_t **exc_value, int64_t *n_exc);
/* ... and the block anchors when that encoding is delta4 (width RAFT_HIP_COV_DELTA4; NULL / 0 otherwise). */
int  raft_hip_packed_anchor_device(raft_hip_ctx *ctx, const int32_t **cov_anchor, int64_t *n_anchor);

/* Caller-owned host arrays (page-locked for full PCIe rate) that receive the outputs of raft_hip_run_pipelined, with
 * their capacities in elements.  Upper bounds the caller can compute from read_len alone, with W = sum ceil(len/reso)
 * and N = n_reads:
 *   cov8_cap >= W;   frag_cap >= (sum len) / interval_length + 2 N;   rep_cap >= (W + N) / (ceil(repeat_length/reso) + 1).
 * cov_offset / rep_offset / frag_offset hold n_reads + 1 entries.  frag_read is not returned: fragment f of read i is
 * every f in [frag_offset[i], frag_offset[i+1]).  n_exc is written by the call; exc_cap is one limit for the whole job
 * however many devices share it, and when it is too small the call returns RAFT_HIP_ERR_TOO_LARGE with the number of
 * exceptions the job has in n_exc (one retry with that much room suffices). */
typedef struct raft_hip_host_outputs {
    int64_t *cov_offset;  uint8_t *cov8;      int64_t cov8_cap;
    int64_t *exc_index;   int32_t *exc_value; int64_t exc_cap;   int64_t n_exc;
    int64_t *rep_offset;  int32_t *rep_s, *rep_e;               int64_t rep_cap;
    int64_t *frag_offset; int32_t *frag_begin, *frag_end;       int64_t frag_cap;
    int32_t cov_width;    /* bytes per window of the coverage encoding: 0 or 1 = one (cov8 as declared), 2 = cov8 points
                             at cov8_cap uint16 codes (limit 65535; see raft_hip_fetch_packed_w);
                             RAFT_HIP_COV_DELTA4 = four-bit steps (see raft_hip_fetch_delta4): cov8 holds (cov8_cap + 1) / 2
                             bytes, cov_anchor anchor_cap >= (W + 1023) / 1024 entries; exc_value are absolute values, and
                             exc_cap should allow for W / 128 of them (0.2-0.3 % of the windows of a 32x set).  Chunks are cut
                             at reads that begin on a multiple of four windows; a job that would take the host-routed path
                             (a stream that is not a handful of sorted runs) is served in one piece in this encoding */
    int32_t reserved;
    int32_t *cov_anchor;  int64_t anchor_cap;
} raft_hip_host_outputs;

/* One end-to-end pass, host memory to host memory: raft_hip_run_host + raft_hip_finish + raft_hip_fetch_packed in one
 * call, with the three stages overlapped.  With symmetric_mode = 1 and a record stream of at most four runs sorted by
 * query id (hifiasm's shape), the reads are cut into n_chunks ranges (0 = chosen from the record count) whose upload,
 * pass and download run concurrently on separate streams; results are identical to the one-piece pass, to which the
 * call falls back for any other input (and for any chunk that reports a data error, so that errors are reported
 * exactly as by raft_hip_run_host).  tid/ts/te may be NULL when symmetric_mode = 1.  RAFT_HIP_ERR_TOO_LARGE: a
 * capacity in `out` was too small (size them by the bounds above).  Device-resident outputs of the context are NOT valid
 * after this call (raft_hip_fetch / raft_hip_outputs_device return RAFT_HIP_ERR_STATE).
 * Derived input (ABI 7).  With symmetric_mode = 1, a stream of one or two sorted runs and reads below 65,535 windows, the
 * engine's host side derives -- chunk by chunk, on threads of its own, beside the uploads of the chunks before -- what
 * raft_hip_run_multi_windows would have to be handed: where every read's records begin (from qid) and the records as
 * window records (from qs / qe; repeat.hpp:69-72 uses nothing else of an interval).  4 bytes per record cross the link
 * instead of 12, the pass needs no look at the stream, and the caller prepares nothing: this is the call SURVEY.md §8(d)'s
 * clock brackets ("int32 SoA in pinned host memory ... to outputs on the host").  The ids are checked on the way; anything
 * the derivation cannot take (ids out of place, a negative coordinate, a window beyond 16 bits) sends the job to the
 * one-piece pass over the columns, which reports or handles it as before.  RAFT_NO_DERIVE=1 uploads the columns as they are. */
int  raft_hip_run_pipelined(raft_hip_ctx *ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                            const int32_t *qid, const int32_t *qs, const int32_t *qe,
                            const int32_t *tid, const int32_t *ts, const int32_t *te,
                            int32_t n_chunks, raft_hip_host_outputs *out, raft_hip_summary *summary);

/* The same job spread over several contexts -- one per GPU of the node (north_star: "reads and their overlaps shard
 * embarrassingly across the 8 GPUs of one node"; SURVEY.md §8(e) host-routed mode, no collective): the host cuts the
 * reads into consecutive ranges of equal record counts, context d gets the d-th group of ranges and runs the chunked
 * pipeline on it with its own upload / download streams, all devices at once; outputs land in the caller's arrays in read
 * order exactly as from one device.  ctxs[0]'s parameters and tuning apply to all.  Contexts are made with
 * raft_hip_create(device_ids[d], ...) -- one call per device instead of SURVEY §8(b)'s create(device_ids[], n_dev): a
 * context IS one device + its streams and buffers, and two contexts may share a device (how the single-GPU tests drive
 * this path).  rep_cap / frag_cap must reach the bounds stated above (each device writes at its bound and the gaps are
 * closed afterwards).  Falls back to ctxs[0] alone, one piece, exactly where raft_hip_run_pipelined does. */
int  raft_hip_run_multi(raft_hip_ctx *const *ctxs, int32_t n_ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                        const int32_t *qid, const int32_t *qs, const int32_t *qe,
                        const int32_t *tid, const int32_t *ts, const int32_t *te,
                        int32_t n_chunks, raft_hip_host_outputs *out, raft_hip_summary *summary);

/* raft_hip_run_multi (n_ctx = 1: raft_hip_run_pipelined) over grouped input (see raft_hip_run_device_grouped): the
 * caller's offsets replace the query column -- a third of the upload that bounds the host-to-host rate -- and the host's
 * plan needs no search in the record stream.  Contexts must assert symmetric_mode = 1.  Same outputs, same errors. */
int  raft_hip_run_multi_grouped(raft_hip_ctx *const *ctxs, int32_t n_ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                                int32_t n_runs, const int64_t *rec_offset, const int32_t *qs, const int32_t *qe,
                                int32_t n_chunks, raft_hip_host_outputs *out, raft_hip_summary *summary);

/* ... over window records (see raft_hip_run_device_windows): half the upload again. */
int  raft_hip_run_multi_windows(raft_hip_ctx *const *ctxs, int32_t n_ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                                int32_t n_runs, const int64_t *rec_offset, const uint32_t *win,
                                int32_t n_chunks, raft_hip_host_outputs *out, raft_hip_summary *summary);

/* ---- pre-split PAF: the exchange step (BASELINE.json configs[3]; SURVEY.md §8e "pre-split" mode) ----------------------
 * Every rank holds a contiguous slice of the record stream, in its grouped form (see raft_hip_run_device_grouped): per
 * sorted run of the slice -- a slice of a hifiasm PAF has at most two -- where every read of the WHOLE set begins.  Reads
 * are owned in contiguous ranges: rank g owns [bounds[g], bounds[g+1]), bounds[0] = 0, bounds[world] = n_reads_total.
 * A run sorted by read id is sorted by owner too: what a rank has for another is one contiguous piece per run, sent from
 * where it lies together with the matching slice of the run's offsets; the query ids never travel.  What arrives is
 * grouped input again -- one run per (peer, run) with records for this rank, at most RAFT_HIP_MAX_RUNS -- held in
 * buffers of the context until its next exchange:
 *     raft_hip_run_device_grouped(ctx, got.n_reads, d_read_len_of_my_reads, got.n_rec, got.n_runs, got.d_rec_offset,
 *                                 NULL, got.d_qs, got.d_qe, n_bins_of_my_reads);
 * (more than four runs are merged on the device by that pass).  Cross-read state afterwards: the fragment counter
 * (chop.hpp:195) and the stdout sums (repeat.hpp:93-97) -- one all-gather of five integers per rank, the caller's.
 *   raft_hip_exchange        one process per GPU: RCCL over xGMI -- piece sizes by ncclAllGather, payload by grouped
 *                            ncclSend / ncclRecv on the context's stream; returns in stream order (no host wait at the end).
 *                            Errors are COLLECTIVE: whatever one rank finds wrong on its side -- bounds or offsets that do not fit its
 *                            slice, a device allocation or staging copy that failed -- travels in its row of the gather, and every rank
 *                            returns the same code (RAFT_HIP_ERR_PARAM / _NOMEM / _TOO_LARGE) before any send or receive is posted; no
 *                            rank is left waiting in a collective for a peer that has returned.
 *                            `comm`: an ncclComm_t of the caller's, or one made by raft_hip_comm_create from an id that rank 0
 *                            obtained with raft_hip_comm_unique_id (128 bytes) and handed to the others by its own means.
 *                            librccl.so.1 is loaded when first used (a single-GPU run never maps it).
 *   raft_hip_exchange_local  one process, one context per rank (raft_hip_create on each device): peer copies.
 * Window records travel the same way: a slice with d_qe = NULL carries them in d_qs (one 32-bit word per record,
 * raft_host_pack_windows -- every rank of an exchange in the same form); what arrives then has d_qe = NULL and goes to
 * raft_hip_run_device_windows(ctx, got.n_reads, ..., got.n_runs, got.d_rec_offset, (const uint32_t *)got.d_qs, n_bins): half the
 * bytes over xGMI.
 * The slices' device columns must be complete when the call is made (raft_hip_exchange: on the context's stream, or
 * their producers synchronised; raft_hip_exchange_local: synchronised), and stay untouched until the exchange is done. */
#define RAFT_HIP_MAX_RUNS 16
typedef struct raft_hip_slice {
    int64_t n_rec;              /* records of this rank's slice */
    int32_t n_runs;             /* sorted runs of the slice, 1..4 */
    const int64_t *rec_offset;  /* HOST: [n_runs * (n_reads_total + 1)], first record (index into the slice) of every read in every run */
    const int32_t *d_qs, *d_qe; /* DEVICE: the slice's query coordinates */
    const int64_t *d_rec_offset;/* DEVICE, optional (ABI 11): the same offsets on the rank's device.  raft_hip_exchange sends the slices of
                                   the offsets from there and otherwise uploads rec_offset at every call (8 bytes per read and run: 53 MB
                                   for 3.3 M reads in two runs); a caller that exchanges the same slice again keeps a copy.  NULL: uploaded */
} raft_hip_slice;
typedef struct raft_hip_received {
    int32_t n_reads;            /* reads this rank owns */
    int32_t n_runs;             /* 1..RAFT_HIP_MAX_RUNS */
    int64_t n_rec;
    const int64_t *d_rec_offset;/* DEVICE: [n_runs * (n_reads + 1)] */
    const int32_t *d_qs, *d_qe; /* DEVICE */
} raft_hip_received;
int  raft_hip_comm_unique_id(void *id128);
int  raft_hip_comm_create(int device_id, const void *id128, int32_t rank, int32_t world, void **comm);
void raft_hip_comm_destroy(void *comm);
int  raft_hip_exchange(raft_hip_ctx *ctx, void *comm, int32_t rank, int32_t world, int32_t n_reads_total, const int64_t *bounds,
                       const raft_hip_slice *mine, raft_hip_received *out);
int  raft_hip_exchange_local(raft_hip_ctx *const *ctxs, int32_t world, int32_t n_reads_total, const int64_t *bounds,
                             const raft_hip_slice *slices, raft_hip_received *outs);

/* A slice of a NON-symmetric (or not id-sorted) PAF, made ready for the exchange above on the device (ABI 9; replaces
 * raft_amd/dist.py's torch sort + all-to-all for target sides).  The reference piles up, for every record, its query side on
 * the query read and -- unless the PAF is symmetric -- its target side on the target read when the two reads differ
 * (chop.hpp:165-169, repeat.hpp:48-58).  raft_hip_group_sides writes exactly those (read, start, end) intervals of the
 * slice's records, sorted by read id (the device radix sort the engine's general bucketing uses), and hands them back as
 * a slice in grouped form with ONE sorted run: `out->rec_offset` (host, n_reads_total + 1 entries: where every read's
 * intervals begin), `out->d_qs / d_qe` (device), `out->n_rec` = the number of intervals.  A run sorted by read id is
 * sorted by owner, so raft_hip_exchange / raft_hip_exchange_local route it as they route a symmetric slice -- one
 * contiguous piece per destination -- and the receiver runs raft_hip_run_device_grouped on what arrives (the sides are
 * already expanded: a grouped pass piles up what it is given).  symmetric != 0: query sides only.
 * The arrays belong to the context and live until its next raft_hip_group_sides or pass of the general bucketing path;
 * the record columns are device arrays, complete when the call is made.  Errors: RAFT_HIP_ERR_READ_ID (an id outside
 * [0, n_reads_total): error_index via raft_hip_last_error text), RAFT_HIP_ERR_TOO_LARGE (2^31 intervals or more). */
int  raft_hip_group_sides(raft_hip_ctx *ctx, int32_t n_reads_total, int64_t n_rec, const int32_t *d_qid, const int32_t *d_qs,
                          const int32_t *d_qe, const int32_t *d_tid, const int32_t *d_ts, const int32_t *d_te, int32_t symmetric,
                          raft_hip_slice *out);

/* ... and the flag such a job needs first: is the pre-split PAF symmetric (chop.hpp:171-184: some record i > 0 is record 0
 * with query and target swapped)?  Record 0 is rank 0's first record; every rank looks for its mirror in its own slice (one
 * kernel over the six columns) and the verdict is the OR over the ranks -- two all-gathers of a few words (RCCL), or host
 * copies between the contexts of one process.  Every rank returns the same *symmetric.  A stream without records is not
 * symmetric.  (Replaces raft_amd/dist.py global_symmetric_flag: torch broadcast + all-reduce.) */
typedef struct raft_hip_records {
    int64_t n_rec;                                       /* records of this rank's slice */
    const int32_t *d_qid, *d_qs, *d_qe, *d_tid, *d_ts, *d_te;   /* DEVICE */
} raft_hip_records;
int  raft_hip_presplit_symmetric(raft_hip_ctx *ctx, void *comm, int32_t rank, int32_t world, const raft_hip_records *mine,
                                 int32_t *symmetric);
int  raft_hip_presplit_symmetric_local(raft_hip_ctx *const *ctxs, int32_t world, const raft_hip_records *slices, int32_t *symmetric);

/* The whole pre-split job of ONE process (ABI 10; the C++ caller BASELINE configs[3] / SURVEY.md §8e "pre-split" asks for: main.cpp:21-87
 * + chop.hpp:331-373 with the record stream cut into `world` contiguous slices).  ctxs[r] is rank r -- contexts of this process on
 * whatever devices the caller made them (one GPU per rank on a node; several ranks on one GPU is how the one-GPU tests run it).
 * Rank r holds records [n_rec r / world, n_rec (r + 1) / world) of the six HOST columns as they come (any order, symmetric or
 * not); the call uploads the slices, finds the symmetric flag (raft_hip_presplit_symmetric_local), expands and groups every
 * slice's sides (raft_hip_group_sides), routes them to the owners of their reads (raft_hip_exchange_local: contiguous read
 * ranges of equal window counts), runs every rank's grouped pass and writes the ranks' outputs into `out` in read order -- the
 * same arrays, bounds and errors as raft_hip_run_multi; out->cov_width 1 or 2.  ctxs[0]'s parameters apply to all.
 * One process per GPU: the same steps with raft_hip_presplit_symmetric / raft_hip_exchange over RCCL. */
int  raft_hip_run_presplit_local(raft_hip_ctx *const *ctxs, int32_t world, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                                 const int32_t *qid, const int32_t *qs, const int32_t *qe,
                                 const int32_t *tid, const int32_t *ts, const int32_t *te,
                                 raft_hip_host_outputs *out, raft_hip_summary *summary);

/* Page-locks / releases a range of the caller's host memory (hipHostRegister, every device of the node).  Arrays handed to
 * the host-to-host entry points move at the link's rate (53 GB/s each way on MI355X) only from page-locked memory; pages
 * that have been written before are pinned at ~120 GB/s, untouched ones at the cost of their first touch.  A caller that
 * cannot register (not its memory, already registered) may ignore the error: the copies then take the pageable path. */
int  raft_hip_host_register(void *ptr, uint64_t bytes);
int  raft_hip_host_unregister(void *ptr);

/* Optional: pays now what the context's first host-to-host job would pay inside its own clock -- the engine's code going
 * to the device at its first launch, the pipeline's lanes (sub-contexts, streams, page-locked blocks): 70-80 ms on the
 * MI355X box.  The CLI calls it on a helper thread while it tokenises its inputs. */
int  raft_hip_warm_up(raft_hip_ctx *ctx);
/* Optional: allocates the device buffers of a coming host-to-host job from what its caller knows early -- the reads' lengths
 * and an estimate of the record count (the CLI: from the size of the overlaps file) -- for a job shared by n_ctx contexts
 * in coverage width cov_width, and the page-locked staging a job over plain columns derives its chunks into (three chunks' worth
 * of window records).  Buffers only grow: a short estimate costs what no estimate would have cost. */
int  raft_hip_reserve(raft_hip_ctx *ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec_estimate, int32_t n_ctx,
                      int32_t cov_width);

/* Device seconds spent in the dominant kernel (coverage pileup + run scan) and
 * in all kernels of the last finished pass, from HIP events recorded on the
 * context's stream around them. */
int  raft_hip_last_timing(raft_hip_ctx *ctx, double *pileup_seconds, double *pass_seconds);

/* Tuning knobs (optional): the quantum -- windows per range of reads a worker of the pileup kernel draws (0 = chosen from the
 * set's size) --, force the general bucketing path, and `variant`: rounds 1-5 kept several pileup kernels and chose among them
 * here; since ABI 11 there is one (raft_amd/csrc/pileup_wave.hpp), named by -1 or 5, and any other value is RAFT_HIP_ERR_PARAM. */
int  raft_hip_set_tuning(raft_hip_ctx *ctx, int32_t tile_bins, int32_t force_bucket_path, int32_t variant);

/* On-device self test of the wavefront primitives (scan, ballots); 0 = pass. */
int  raft_hip_selftest(int device_id);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif
