// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver around the UNMODIFIED reference sources, compiled from
// where they lie (-I$(REF), i.e. /root/reference) into oracle/_ref/libraft_ref.so
// by oracle/Makefile.  Nothing of the reference is copied into this repository:
// this file only builds the reference's own in-memory inputs (Read / Overlap
// objects, the idx_pileup buckets) from int32 SoA columns and then calls
//   profileCoverage()   repeat.hpp:28-79
//   repeat_annotate()   repeat.hpp:81-204
// themselves.  It exists so that (a) the C restatement in raft_oracle.c can be
// checked against the reference's real code on arbitrary in-memory inputs, and
// (b) bench.py can time the reference's integer path ("cpu_baseline.kind":
// "reference") on the GPU box, where /root/reference is absent but this
// prebuilt .so travels.
//
// The bucket fill below restates chop.hpp:155-184 (one heap Overlap per record,
// pointer pushed to the query's bucket and, while the symmetric flag is 0 and
// query != target, to the target's bucket; flag flips at the first mirror of
// record 0).  PAF text parsing and name hashing (chop.hpp:147,162-163) are not
// part of the integer path and are not exercised here; the stock binary
// oracle/_ref/raft covers them for the golden fixtures.
#include "chop.hpp"

#include <chrono>
#include <cstdint>

extern "C" {

struct raft_ref_params {
    int32_t reso, est_cov;
    double cov_mul;
    int32_t repeat_length, interval_length, read_length, overlap_length, flanking_length;
};

// Returns 0 on success.  cov (may be NULL) must hold sum(ceil(len/reso)) ints;
// rep_count (may be NULL) n_reads ints; rep_s/rep_e (may be NULL) rep_cap ints.
// seconds[0] = bucket build, seconds[1] = repeat_annotate (profileCoverage + run
// scan; its three text streams are pointed at an unopenable path so that every
// operator<< returns at the sentry without formatting).
int raft_ref_run(const raft_ref_params *p, int32_t n_reads, const int32_t *read_len,
                 int64_t n_rec, const int32_t *qid, const int32_t *qs, const int32_t *qe,
                 const int32_t *tid, const int32_t *ts, const int32_t *te,
                 int32_t *symmetric_out, int32_t *cov, int32_t *rep_count,
                 int32_t *rep_s, int32_t *rep_e, int64_t rep_cap, int64_t *n_rep_out,
                 double *seconds)
{
    algoParams param;
    param.initParams();
    param.reso = p->reso; param.est_cov = p->est_cov; param.cov_mul = p->cov_mul;
    param.repeat_length = p->repeat_length; param.interval_length = p->interval_length;
    param.read_length = p->read_length; param.overlap_length = p->overlap_length;
    param.flanking_length = p->flanking_length;
    param.outputfilename = "/nonexistent-dir-for-raft-ref/x";

    std::vector<Read *> reads;
    reads.reserve(n_reads);
    for (int32_t i = 0; i < n_reads; i++)
        reads.push_back(new Read(i, read_len[i], std::string(), std::string()));

    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::vector<Overlap *>> idx_pileup;
    for (int32_t i = 0; i < n_reads; i++) idx_pileup.push_back(std::vector<Overlap *>());
    Overlap *first = nullptr;
    int check = 1;
    std::vector<Overlap *> all;
    all.reserve((size_t)n_rec);
    for (int64_t i = 0; i < n_rec; i++) {
        Overlap *o = new Overlap();
        all.push_back(o);
        o->read_A_match_start_ = qs[i]; o->read_B_match_start_ = ts[i];
        o->read_A_match_end_ = qe[i];   o->read_B_match_end_ = te[i];
        o->read_A_id_ = qid[i];         o->read_B_id_ = tid[i];
        idx_pileup[o->read_A_id_].push_back(o);
        if (o->read_A_id_ != o->read_B_id_ && !param.symmetric_overlaps)
            idx_pileup[o->read_B_id_].push_back(o);
        if (i == 0) first = o;
        else if (check && first->read_A_id_ == o->read_B_id_ && first->read_B_id_ == o->read_A_id_ &&
                 first->read_A_match_start_ == o->read_B_match_start_ &&
                 first->read_A_match_end_ == o->read_B_match_end_ &&
                 first->read_B_match_start_ == o->read_A_match_start_ &&
                 first->read_B_match_end_ == o->read_A_match_end_) {
            param.symmetric_overlaps = 1;
            check = 0;
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (symmetric_out) *symmetric_out = param.symmetric_overlaps;

    if (cov) {
        int64_t base = 0;
        for (int32_t i = 0; i < n_reads; i++) {
            std::vector<std::tuple<int, int>> c;
            profileCoverage(idx_pileup[i], c, reads[i], param);
            for (size_t j = 0; j < c.size(); j++) cov[base + (int64_t)j] = std::get<1>(c[j]);
            base += (int64_t)c.size();
        }
    }

    auto t2 = std::chrono::steady_clock::now();
    {
        // silence the reference's stdout statistics for the duration of the call
        fflush(stdout);
        FILE *keep = stdout;
        FILE *nul = fopen("/dev/null", "w");
        if (nul) stdout = nul;
        repeat_annotate(reads, idx_pileup, param);
        if (nul) { fflush(nul); stdout = keep; fclose(nul); }
    }
    auto t3 = std::chrono::steady_clock::now();

    int64_t n_rep = 0;
    int rc = 0;
    for (int32_t i = 0; i < n_reads; i++) {
        if (rep_count) rep_count[i] = (int32_t)reads[i]->long_repeats.size();
        for (auto &pr : reads[i]->long_repeats) {
            if (rep_s && rep_e) {
                if (n_rep >= rep_cap) { rc = 1; break; }
                rep_s[n_rep] = pr.first; rep_e[n_rep] = pr.second;
            }
            n_rep++;
        }
    }
    if (n_rep_out) *n_rep_out = n_rep;
    if (seconds) {
        seconds[0] = std::chrono::duration<double>(t1 - t0).count();
        seconds[1] = std::chrono::duration<double>(t3 - t2).count();
    }
    // the reference never frees its Overlap objects (chop.hpp:155); the harness does
    for (Overlap *o : all) delete o;
    for (Read *r : reads) delete r;
    return rc;
}

} // extern "C"
