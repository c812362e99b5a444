// gen_set.cpp -- text inputs of the `raft` CLI at scale: a FASTA of random bases and a hifiasm-shaped PAF (cis file then
// trans file, each grouped by ascending query, symmetric) for a synthetic read set -- the model of raft_amd/synth.py
// (log-normal lengths, uniform placement on a genome of sum(len)/coverage, every pair sharing >= 500 bp overlaps, two-copy
// repeat families whose reads overlap across copies), written by a native tool because at 500 k reads the files are 10 GB
// + 3 GB.  Used by tools/cli_big.py; not part of the product.
//   usage: gen_set <n_reads> <mean_len> <coverage> <seed> <reads.fa> <overlaps.paf>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545F4914F6CDD1Dull; }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double normal() { const double u = std::max(uni(), 1e-300), v = uni(); return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v); }
};

struct Rec { int32_t tid, qs, qe, ts, te; uint8_t trans; };

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: gen_set <n_reads> <mean_len> <coverage> <seed> <reads.fa> <overlaps.paf>\n"); return 2; }
    const int64_t N = atoll(argv[1]);
    const double mean_len = atof(argv[2]), coverage = atof(argv[3]);
    Rng rng((uint64_t)atoll(argv[4]));
    const int64_t min_len = 2000, max_len = 200000, min_ovl = 500;
    const double sigma = 0.5, mu = std::log(mean_len) - 0.5 * sigma * sigma;
    std::vector<int64_t> len((size_t)N), start((size_t)N);
    std::vector<uint8_t> hap((size_t)N), strand((size_t)N);
    int64_t total = 0, longest = 0;
    for (int64_t i = 0; i < N; ++i) {
        len[(size_t)i] = std::min(max_len, std::max(min_len, (int64_t)std::exp(mu + sigma * rng.normal())));
        total += len[(size_t)i]; longest = std::max(longest, len[(size_t)i]);
    }
    const int64_t G = std::max((int64_t)((double)total / coverage), longest + 1);
    for (int64_t i = 0; i < N; ++i) {
        start[(size_t)i] = (int64_t)(rng.uni() * (double)(G - len[(size_t)i]));
        hap[(size_t)i] = rng.uni() < 0.5; strand[(size_t)i] = rng.uni() < 0.5;
    }
    std::vector<int32_t> order((size_t)N);
    for (int64_t i = 0; i < N; ++i) order[(size_t)i] = (int32_t)i;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return start[(size_t)a] < start[(size_t)b] || (start[(size_t)a] == start[(size_t)b] && a < b); });
    std::vector<int64_t> S((size_t)N), E((size_t)N);
    for (int64_t k = 0; k < N; ++k) { S[(size_t)k] = start[(size_t)order[(size_t)k]]; E[(size_t)k] = S[(size_t)k] + len[(size_t)order[(size_t)k]]; }
    std::vector<std::vector<Rec>> recs((size_t)N);
    // read-local coordinates of genome-relative offsets [a, b) (reverse reads flip)
    auto put = [&](int32_t q, int64_t qa, int64_t qb, int32_t t, int64_t ta, int64_t tb) {
        auto loc = [&](int32_t r, int64_t a, int64_t b, int32_t &oa, int32_t &ob) {
            if (strand[(size_t)r]) { oa = (int32_t)(len[(size_t)r] - b); ob = (int32_t)(len[(size_t)r] - a); } else { oa = (int32_t)a; ob = (int32_t)b; }
        };
        Rec r{};
        r.tid = t; r.trans = hap[(size_t)q] != hap[(size_t)t];
        loc(q, qa, qb, r.qs, r.qe); loc(t, ta, tb, r.ts, r.te);
        recs[(size_t)q].push_back(r);
    };
    // positional overlaps: sorted ranks i < j with S[j] < E[i] - min_ovl share [S[j], min(E[i], E[j])); both directions
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = i + 1; j < N && S[(size_t)j] < E[(size_t)i] - min_ovl; ++j) {
            const int64_t a = S[(size_t)j], b = std::min(E[(size_t)i], E[(size_t)j]);
            const int32_t ri = order[(size_t)i], rj = order[(size_t)j];
            put(ri, a - S[(size_t)i], b - S[(size_t)i], rj, a - S[(size_t)j], b - S[(size_t)j]);
            put(rj, a - S[(size_t)j], b - S[(size_t)j], ri, a - S[(size_t)i], b - S[(size_t)i]);
        }
    // repeat families: two copies of 15-50 kb; reads over different copies overlap inside the repeat
    const int64_t n_fam = std::max<int64_t>(N / 250, 1);
    struct Hit { int64_t rank, u, v; };
    for (int64_t f = 0; f < n_fam; ++f) {
        const int64_t L = std::min<int64_t>(15000 + (int64_t)(rng.uni() * 35001.0), std::max(G / 4, min_ovl + 1));
        std::vector<Hit> hits[2];
        int64_t pos[2];
        for (int c = 0; c < 2; ++c) {
            const int64_t p = (int64_t)(rng.uni() * (double)(G - L));
            pos[c] = p;
            int64_t k = std::lower_bound(S.begin(), S.end(), p - max_len) - S.begin();
            for (; k < N && S[(size_t)k] < p + L - min_ovl; ++k) {
                const int64_t u = std::max(S[(size_t)k], p) - p, v = std::min(E[(size_t)k], p + L) - p;
                if (v - u >= min_ovl) hits[c].push_back(Hit{k, u, v});
            }
        }
        for (const Hit &x : hits[0])
            for (const Hit &y : hits[1]) {
                if (x.rank == y.rank) continue;
                const int64_t u = std::max(x.u, y.u), v = std::min(x.v, y.v);
                if (v - u < min_ovl) continue;
                const int32_t ra = order[(size_t)x.rank], rb = order[(size_t)y.rank];
                const int64_t a0 = pos[0] + u - S[(size_t)x.rank], a1 = pos[0] + v - S[(size_t)x.rank];
                const int64_t b0 = pos[1] + u - S[(size_t)y.rank], b1 = pos[1] + v - S[(size_t)y.rank];
                put(ra, a0, a1, rb, b0, b1);
                put(rb, b0, b1, ra, a0, a1);
            }
    }
    // ---- FASTA: one record per read, bases on one line
    {
        FILE *fa = fopen(argv[5], "wb");
        if (!fa) { perror(argv[5]); return 1; }
        static char iobuf[1 << 22];
        setvbuf(fa, iobuf, _IOFBF, sizeof iobuf);
        std::string line;
        for (int64_t i = 0; i < N; ++i) {
            fprintf(fa, ">r%lld\n", (long long)i);
            line.resize((size_t)len[(size_t)i] + 1);
            uint64_t bits = 0;
            int have = 0;
            for (int64_t k = 0; k < len[(size_t)i]; ++k) {
                if (!have) { bits = rng.next(); have = 32; }
                line[(size_t)k] = "ACGT"[bits & 3]; bits >>= 2; --have;
            }
            line[(size_t)len[(size_t)i]] = '\n';
            fwrite(line.data(), 1, line.size(), fa);
        }
        if (fclose(fa) != 0) { perror("fclose"); return 1; }
    }
    // ---- PAF: the cis file, then the trans file, each grouped by ascending query (and ascending target inside a query)
    {
        FILE *pf = fopen(argv[6], "wb");
        if (!pf) { perror(argv[6]); return 1; }
        static char iobuf2[1 << 22];
        setvbuf(pf, iobuf2, _IOFBF, sizeof iobuf2);
        int64_t n_rec = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int64_t q = 0; q < N; ++q) {
                std::vector<Rec> &v = recs[(size_t)q];
                if (pass == 0) std::stable_sort(v.begin(), v.end(), [](const Rec &a, const Rec &b) { return a.tid < b.tid; });
                for (const Rec &r : v) {
                    if ((int)r.trans != pass) continue;
                    fprintf(pf, "r%lld\t%lld\t%d\t%d\t+\tr%d\t%lld\t%d\t%d\t%d\t%d\t255\n", (long long)q, (long long)len[(size_t)q], r.qs, r.qe, r.tid,
                            (long long)len[(size_t)r.tid], r.ts, r.te, r.qe - r.qs, r.qe - r.qs);
                    ++n_rec;
                }
            }
        if (fclose(pf) != 0) { perror("fclose"); return 1; }
        fprintf(stderr, "gen_set: %lld reads, %lld bases, %lld records\n", (long long)N, (long long)total, (long long)n_rec);
    }
    return 0;
}
