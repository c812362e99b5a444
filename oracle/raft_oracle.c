/*
 * raft_oracle.c -- TEST INFRASTRUCTURE ONLY (see raft_oracle.h).
 *
 * Plain-C, single-thread restatement of the reference's hot path.  Each stage
 * keeps the reference's algorithm (bucket push order, event sort + sweep,
 * sequential run scan, two-pointer marker mask) rather than a closed form, so
 * that it can be timed as a "reference-shaped" CPU baseline and so that the
 * closed forms used by the HIP kernels are checked against an independent
 * statement of the same computation.
 */
#include "raft_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct { int32_t start, end; } event_t;

static int event_by_start(const void *a, const void *b)
{
    /* repeat.hpp:14-17,60: ordered by start only; ties are free */
    int32_t x = ((const event_t *)a)->start, y = ((const event_t *)b)->start;
    return (x > y) - (x < y);
}

/*
 * repeat.hpp:170 -- std::sort(long_repeats, compare_start_repeat): ordered by the (flanked, clamped) start only.
 * Runs are emitted left to right and the flank/clamp is monotone, so the list arrives non-decreasing; but several
 * repeats can clamp to start 0 (flanking_length larger than their distance from the read's begin), and std::sort is
 * not stable: with more than 16 elements libstdc++'s introsort permutes such ties (with <= 16 its insertion sort
 * leaves them alone).  The reference binary is built with GCC's libstdc++ (Makefile:2-6), whose algorithm has been
 * the same since GCC 4: introsort loop (median of first+1 / middle / last-1 moved to first, unguarded partition,
 * recursion on the right part, depth limit 2*floor(log2 n), heap sort fallback) followed by the final insertion sort
 * with threshold 16.  This is a restatement of that published algorithm (bits/stl_algo.h, bits/stl_heap.h), pinned
 * against the real std::sort through oracle/_ref/libraft_ref.so in tests/test_oracle_golden.py.
 */
typedef struct { int32_t s, e; } rep_t;
#define REP_LESS(a, b) ((a).s < (b).s)

static void rep_swap(rep_t *a, rep_t *b) { rep_t t = *a; *a = *b; *b = t; }

static void rep_push_heap(rep_t *first, int64_t hole, int64_t top, rep_t value)
{
    int64_t parent = (hole - 1) / 2;
    while (hole > top && REP_LESS(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void rep_adjust_heap(rep_t *first, int64_t hole, int64_t len, rep_t value)
{
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (REP_LESS(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    rep_push_heap(first, hole, top, value);
}

static void rep_heap_sort(rep_t *first, rep_t *last)       /* __partial_sort(first, last, last) */
{
    const int64_t len = last - first;
    if (len >= 2)
        for (int64_t parent = (len - 2) / 2;; parent--) {
            rep_adjust_heap(first, parent, len, first[parent]);
            if (parent == 0) break;
        }
    while (last - first > 1) {
        --last;
        rep_t value = *last;
        *last = *first;
        rep_adjust_heap(first, 0, last - first, value);
    }
}

static void rep_introsort_loop(rep_t *first, rep_t *last, int depth_limit)
{
    while (last - first > 16) {
        if (depth_limit == 0) { rep_heap_sort(first, last); return; }
        --depth_limit;
        /* __unguarded_partition_pivot */
        rep_t *mid = first + (last - first) / 2;
        rep_t *a = first + 1, *b = mid, *c = last - 1;
        if (REP_LESS(*a, *b)) {
            if (REP_LESS(*b, *c)) rep_swap(first, b);
            else if (REP_LESS(*a, *c)) rep_swap(first, c);
            else rep_swap(first, a);
        } else if (REP_LESS(*a, *c)) rep_swap(first, a);
        else if (REP_LESS(*b, *c)) rep_swap(first, c);
        else rep_swap(first, b);
        rep_t *lo = first + 1, *hi = last;
        for (;;) {
            while (REP_LESS(*lo, *first)) ++lo;
            --hi;
            while (REP_LESS(*first, *hi)) --hi;
            if (!(lo < hi)) break;
            rep_swap(lo, hi);
            ++lo;
        }
        rep_introsort_loop(lo, last, depth_limit);
        last = lo;
    }
}

static void rep_unguarded_linear_insert(rep_t *last)
{
    rep_t val = *last;
    rep_t *next = last - 1;
    while (REP_LESS(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}

static void rep_insertion_sort(rep_t *first, rep_t *last)
{
    if (first == last) return;
    for (rep_t *i = first + 1; i != last; ++i) {
        if (REP_LESS(*i, *first)) {
            rep_t val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(rep_t));
            *first = val;
        } else rep_unguarded_linear_insert(i);
    }
}

static void rep_std_sort(rep_t *first, rep_t *last)
{
    if (first == last) return;
    int lg = 0;
    for (int64_t n = last - first; n > 1; n >>= 1) lg++;
    rep_introsort_loop(first, last, 2 * lg);
    if (last - first > 16) {
        rep_insertion_sort(first, first + 16);
        for (rep_t *i = first + 16; i != last; ++i) rep_unguarded_linear_insert(i);
    } else rep_insertion_sort(first, last);
}

/* growable int32 column ------------------------------------------------------ */
typedef struct { int32_t *v; int64_t n, cap; } ivec;

static int ivec_push(ivec *a, int32_t x)
{
    if (a->n == a->cap) {
        int64_t nc = a->cap ? a->cap * 2 : 1024;
        int32_t *nv = (int32_t *)realloc(a->v, (size_t)nc * sizeof(int32_t));
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = x;
    return 0;
}

void raft_oracle_free(raft_oracle_result *o)
{
    if (!o) return;
    free(o->cov_offset); free(o->cov);
    free(o->rep_offset); free(o->rep_s); free(o->rep_e);
    free(o->cut_offset); free(o->cuts);
    free(o->frag_offset); free(o->frag_read); free(o->frag_begin); free(o->frag_end);
    memset(o, 0, sizeof(*o));
}

/*
 * Stage 1 -- chop.hpp:133-191 create_pileup.
 * Every record is appended to its query's bucket; it is also appended to its
 * target's bucket when query != target and the symmetric flag is still 0 at the
 * moment of the push (chop.hpp:165-169).  The flag flips for good at the first
 * record i >= 1 that mirrors record 0 (chop.hpp:175-184); the push of record i
 * itself happens before that test.  Buckets are returned as CSR (start[], items[])
 * holding record indices in push order.
 */
static int build_buckets(int32_t n_reads, int64_t n_rec,
                         const int32_t *qid, const int32_t *qs, const int32_t *qe,
                         const int32_t *tid, const int32_t *ts, const int32_t *te,
                         int64_t **start_out, int64_t **items_out, int32_t *sym_out)
{
    int64_t flip = -1; /* index of the record at which the flag flipped */
    for (int64_t i = 0; i < n_rec; i++) {
        if (qid[i] < 0 || qid[i] >= n_reads || tid[i] < 0 || tid[i] >= n_reads)
            return RAFT_ORACLE_ERR_READ_ID;
        if (flip < 0 && i > 0 &&
            qid[0] == tid[i] && tid[0] == qid[i] &&
            qs[0] == ts[i] && qe[0] == te[i] &&
            ts[0] == qs[i] && te[0] == qe[i])
            flip = i;
    }
    int64_t *start = (int64_t *)calloc((size_t)n_reads + 1, sizeof(int64_t));
    if (!start) return RAFT_ORACLE_ERR_NOMEM;
    int64_t last_b = (flip < 0) ? n_rec - 1 : flip; /* last record whose target side is pushed */
    for (int64_t i = 0; i < n_rec; i++) {
        start[qid[i] + 1]++;
        if (i <= last_b && qid[i] != tid[i]) start[tid[i] + 1]++;
    }
    for (int32_t r = 0; r < n_reads; r++) start[r + 1] += start[r];
    int64_t total = start[n_reads];
    int64_t *items = (int64_t *)malloc((size_t)(total ? total : 1) * sizeof(int64_t));
    int64_t *cur = (int64_t *)malloc((size_t)(n_reads ? n_reads : 1) * sizeof(int64_t));
    if (!items || !cur) { free(start); free(items); free(cur); return RAFT_ORACLE_ERR_NOMEM; }
    memcpy(cur, start, (size_t)n_reads * sizeof(int64_t));
    for (int64_t i = 0; i < n_rec; i++) {
        items[cur[qid[i]]++] = i;
        if (i <= last_b && qid[i] != tid[i]) items[cur[tid[i]]++] = i;
    }
    free(cur);
    *start_out = start; *items_out = items; *sym_out = (flip >= 0);
    return RAFT_ORACLE_OK;
}

/*
 * Stage 2 -- repeat.hpp:28-79 profileCoverage for one read.
 * Select the read's events (query side; target side only when the final flag is
 * 0, repeat.hpp:50-57), sort by start, then sweep: the event is entered at bin
 * i = first window with start < (i+1)*reso and increments bins k = i, i+1, ...
 * while end >= k*reso (repeat.hpp:62-77).  The reference has no bound on k; a
 * write past the last bin is reported as RAFT_ORACLE_ERR_COORD instead.
 */
static int pile_one_read(int32_t r, int32_t nbins, int32_t reso, int32_t sym,
                         const int64_t *items, int64_t n_items,
                         const int32_t *qid, const int32_t *qs, const int32_t *qe,
                         const int32_t *tid, const int32_t *ts, const int32_t *te,
                         event_t *ev, int32_t *cov, int64_t *n_events)
{
    int64_t n = 0;
    for (int64_t a = 0; a < n_items; a++) {
        int64_t i = items[a];
        if (qid[i] == r)              { ev[n].start = qs[i]; ev[n].end = qe[i] - 1; n++; }
        else if (!sym && tid[i] == r) { ev[n].start = ts[i]; ev[n].end = te[i] - 1; n++; }
    }
    *n_events = n;
    for (int64_t a = 0; a < n; a++)
        if (ev[a].start < 0 || ev[a].end < -1) return RAFT_ORACLE_ERR_COORD;
    qsort(ev, (size_t)n, sizeof(event_t), event_by_start);
    int64_t pos = 0, i = 0;
    while (pos < n) {
        while (pos < n && (int64_t)ev[pos].start < (i + 1) * (int64_t)reso) {
            int64_t k = i;
            while ((int64_t)ev[pos].end >= k * (int64_t)reso) {
                if (k >= nbins) return RAFT_ORACLE_ERR_COORD;
                cov[k]++;
                k++;
            }
            pos++;
        }
        i++;
    }
    return RAFT_ORACLE_OK;
}

int raft_oracle_run(const raft_oracle_params *p, int32_t n_reads, const int32_t *read_len,
                    int64_t n_rec, const int32_t *qid, const int32_t *qs, const int32_t *qe,
                    const int32_t *tid, const int32_t *ts, const int32_t *te,
                    raft_oracle_result *o)
{
    memset(o, 0, sizeof(*o));
    if (p->reso <= 0 || p->interval_length <= 0 || p->repeat_length <= 0 || p->est_cov <= 0 ||
        p->read_length / p->interval_length <= 0 || n_reads < 0 || n_rec < 0)
        return RAFT_ORACLE_ERR_PARAM;
    for (int32_t r = 0; r < n_reads; r++)
        if (read_len[r] < 0) return RAFT_ORACLE_ERR_PARAM;

    const int32_t reso = p->reso, L = p->interval_length;
    const int32_t high_cov = (int32_t)((int32_t)p->est_cov * p->cov_mul); /* repeat.hpp:89-90 */
    int rc;

    int64_t *bstart = NULL, *bitems = NULL;
    int32_t sym = 0;
    rc = build_buckets(n_reads, n_rec, qid, qs, qe, tid, ts, te, &bstart, &bitems, &sym);
    if (rc) return rc;

    o->n_reads = n_reads; o->symmetric = sym; o->high_cov = high_cov;
    o->cov_offset  = (int64_t *)calloc((size_t)n_reads + 1, sizeof(int64_t));
    o->rep_offset  = (int64_t *)calloc((size_t)n_reads + 1, sizeof(int64_t));
    o->cut_offset  = (int64_t *)calloc((size_t)n_reads + 1, sizeof(int64_t));
    o->frag_offset = (int64_t *)calloc((size_t)n_reads + 1, sizeof(int64_t));
    if (!o->cov_offset || !o->rep_offset || !o->cut_offset || !o->frag_offset) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail; }

    int64_t max_bucket = 0;
    for (int32_t r = 0; r < n_reads; r++) {
        int32_t nb = read_len[r] / reso + (read_len[r] % reso ? 1 : 0); /* repeat.hpp:32-37 */
        o->cov_offset[r + 1] = o->cov_offset[r] + nb;
        if (bstart[r + 1] - bstart[r] > max_bucket) max_bucket = bstart[r + 1] - bstart[r];
    }
    {
        int64_t B = o->cov_offset[n_reads];
        o->cov = (int32_t *)calloc((size_t)(B ? B : 1), sizeof(int32_t));
        if (!o->cov) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail; }
    }
    event_t *ev = (event_t *)malloc((size_t)(max_bucket ? max_bucket : 1) * sizeof(event_t));
    if (!ev) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail; }

    ivec reps = {0}, repe = {0}, cuts = {0}, fr = {0}, fb = {0}, fe = {0};
    ivec init = {0};
    const int32_t div = p->read_length / L; /* chop.hpp:248 */

    for (int32_t r = 0; r < n_reads; r++) {
        const int32_t len = read_len[r];
        const int32_t nb = (int32_t)(o->cov_offset[r + 1] - o->cov_offset[r]);
        int32_t *cov = o->cov + o->cov_offset[r];
        int64_t nev = 0;
        rc = pile_one_read(r, nb, reso, sym, bitems + bstart[r], bstart[r + 1] - bstart[r],
                           qid, qs, qe, tid, ts, te, ev, cov, &nev);
        if (rc) goto fail_ev;
        o->n_intervals += nev;
        o->total_read_length += len;

        /* Stage 3 -- repeat.hpp:111-168 run scan over the read's windows */
        int64_t rep_first = reps.n;
        int32_t start = 0, end = 0;
        for (int32_t j = 0; j <= nb; j++) {
            int is_high = 0;
            if (j < nb) {
                o->total_coverage += cov[j];
                o->total_windows++;
                is_high = cov[j] >= high_cov;
            }
            if (is_high) { end = j * reso + reso; continue; }
            /* a window below threshold (or the end of the read, repeat.hpp:150) closes the run */
            if (end - start >= p->repeat_length) {
                o->total_repeat_length += end - start;
                int32_t s = start - p->flanking_length, e = end + p->flanking_length;
                if (s <= 0) s = 0;
                if (e >= len) e = len;
                if (ivec_push(&reps, s) || ivec_push(&repe, e)) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
            }
            if (j < nb) { start = j * reso + reso; end = start; }
        }
        o->rep_offset[r + 1] = reps.n;
        /* repeat.hpp:170: std::sort by the clamped start.  The list arrives non-decreasing; ties (several repeats
         * clamped to 0) are permuted by libstdc++ once the read has more than 16 repeats: rep_std_sort above. */
        if (reps.n - rep_first > 16) {
            const int64_t nrep = reps.n - rep_first;
            rep_t *tmp = (rep_t *)malloc((size_t)nrep * sizeof(rep_t));
            if (!tmp) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
            for (int64_t k = 0; k < nrep; k++) { tmp[k].s = reps.v[rep_first + k]; tmp[k].e = repe.v[rep_first + k]; }
            rep_std_sort(tmp, tmp + nrep);
            for (int64_t k = 0; k < nrep; k++) { reps.v[rep_first + k] = tmp[k].s; repe.v[rep_first + k] = tmp[k].e; }
            free(tmp);
        }

        /* Stage 4a -- chop.hpp:209-223 candidate markers */
        init.n = 0;
        if (ivec_push(&init, 0)) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
        for (int32_t j = 1; j <= len / L; j++)
            if (ivec_push(&init, j * L)) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
        if (len % L)
            if (ivec_push(&init, len)) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }

        /* Stage 4b -- chop.hpp:225-246 two-pointer mask against the repeats */
        int64_t cut_first = cuts.n;
        if (ivec_push(&cuts, init.v[0])) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
        int64_t pos = 1;
        for (int64_t k = rep_first; k < reps.n; k++) {
            while (reps.v[k] > init.v[pos] && pos < init.n - 1) {
                if (ivec_push(&cuts, init.v[pos])) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
                pos++;
            }
            while (repe.v[k] >= init.v[pos] && pos < init.n - 1) pos++;
        }
        while (pos < init.n) {
            if (ivec_push(&cuts, init.v[pos])) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
            pos++;
        }
        o->cut_offset[r + 1] = cuts.n;

        /* Stage 4c -- chop.hpp:248-321 fragment bounds */
        const int32_t *F = cuts.v + cut_first;
        const int64_t nF = cuts.n - cut_first;
        if (nF <= (int64_t)div + 1) {
            if (ivec_push(&fr, r) || ivec_push(&fb, 0) || ivec_push(&fe, len)) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
        } else {
            int64_t nf = 1 + (nF - (div + 1)) / div;
            if ((nF - (div + 1)) % div) nf++;
            int64_t q = 0;
            for (int64_t j = 1; j <= nf; j++) {
                int32_t ovl = (j == 1) ? 0 : p->overlap_length;
                int32_t last = (j == nf) ? F[nF - 1] : F[q + div];
                int32_t begin = F[q] - ovl;
                if (begin < 0 || begin > len) { rc = RAFT_ORACLE_ERR_FRAGMENT; goto fail_ev; }
                if (ivec_push(&fr, r) || ivec_push(&fb, begin) || ivec_push(&fe, last)) { rc = RAFT_ORACLE_ERR_NOMEM; goto fail_ev; }
                q += div;
            }
        }
        o->frag_offset[r + 1] = fr.n;
    }

    o->rep_s = reps.v; o->rep_e = repe.v; o->cuts = cuts.v;
    o->frag_read = fr.v; o->frag_begin = fb.v; o->frag_end = fe.v;
    free(init.v); free(ev); free(bstart); free(bitems);
    return RAFT_ORACLE_OK;

fail_ev:
    free(ev); free(init.v);
    free(reps.v); free(repe.v); free(cuts.v); free(fr.v); free(fb.v); free(fe.v);
fail:
    free(bstart); free(bitems);
    raft_oracle_free(o);
    return rc;
}
