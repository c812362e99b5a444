// ref_binding.cpp -- TEST INFRASTRUCTURE ONLY: the reference-side binding of INTEGRATION.md §B, compiled.
//
// What a maintainer of at-cg/RAFT would add to use the MI355X engine from the reference's own program: the reference's
// loaders, types and name table stay (loadFASTA chop.hpp:88-131, paf_open/paf_read paf.hpp:24-100, addStringToMap
// chop.hpp:73-85, Read, algoParams -- all used here from the UNMODIFIED headers under /root/reference, -I$(REF));
// the three hot calls of break_long_reads() (chop.hpp:366,370,372: create_pileup, repeat_annotate, break_reads) become
// one call into the C ABI of include/raft_hip.h, and the four files are written from what it returns.
// oracle/Makefile builds this into oracle/_ref/raft_bound (build container only; the binary travels to the GPU box like
// the other _ref artefacts and never enters history).  tests/test_gpu_cli.py runs it beside raft_amd/bin/raft: same
// inputs, the reference's expected files, byte for byte.  Nothing in the product links or includes this file.
#include "chop.hpp"                 // the reference (brings repeat.hpp, paf.hpp, read.hpp, param.hpp, kseq.h)
#include "../include/raft_hip.h"

#include <chrono>
#include <cstdint>

static void break_long_reads_hip(const char *readfilename, const char *paffilename, algoParams &param)
{
    std::ofstream reads_final(param.outputfilename + ".reads.fasta");
    for (const char *fn : {readfilename, paffilename}) {            // chop.hpp:336-349
        std::ifstream file(fn);
        if (!file || file_is_empty(file)) {
            std::cout << "ERROR, break_long_reads(), " << fn << " input file either does not exist or is empty\n";
            exit(1);
        }
    }
    std::vector<Read *> reads;
    std::unordered_map<std::string, int> umap;
    const int n_read = loadFASTA(readfilename, reads, umap, param);   // chop.hpp:357, unchanged

    // ---- chop.hpp:366 create_pileup: the reference's PAF reader and name table, records kept as int32 columns
    std::vector<int32_t> len(n_read), qid, qs, qe, tid, ts, te;
    for (int i = 0; i < n_read; i++) len[i] = reads[i]->len;
    paf_file_t *fp = paf_open(paffilename);
    paf_rec_t r;
    while (paf_read(fp, &r) >= 0) {
        qid.push_back(addStringToMap(std::string(r.qn), umap));
        tid.push_back(addStringToMap(std::string(r.tn), umap));
        qs.push_back(r.qs); qe.push_back(r.qe); ts.push_back(r.ts); te.push_back(r.te);
    }
    umap.clear();

    // ---- chop.hpp:366-372 on the MI355X
    raft_hip_params hp = {param.reso, param.est_cov, param.cov_mul, param.repeat_length, param.interval_length,
                          param.read_length, param.overlap_length, param.flanking_length, /*symmetric_mode=*/-1};
    raft_hip_ctx *ctx = nullptr;
    int rc = raft_hip_create(0, &hp, &ctx);
    raft_hip_summary s{};
    if (rc == RAFT_HIP_OK)
        rc = raft_hip_run_host(ctx, n_read, len.data(), (int64_t)qid.size(), qid.data(), qs.data(), qe.data(), tid.data(),
                               ts.data(), te.data());
    if (rc == RAFT_HIP_OK) rc = raft_hip_finish(ctx, &s);
    if (rc != RAFT_HIP_OK) { std::cout << "ERROR, raft_hip, " << raft_hip_strerror(rc) << "\n"; exit(1); }
    param.symmetric_overlaps = s.symmetric;
    fprintf(stdout, "INFO, Symmetric overlaps %d \n", param.symmetric_overlaps);          // chop.hpp:189-190
    fprintf(stdout, "INFO, length of alignments  %d()\n", (int)s.n_records);
    fprintf(stdout, "high_cov %d\n", s.high_cov);                                          // repeat.hpp:91

    std::vector<int64_t> cov_off(n_read + 1), rep_off(n_read + 1), frag_off(n_read + 1);
    std::vector<int32_t> cov(s.n_bins), rep_s(s.n_repeats), rep_e(s.n_repeats), fb(s.n_fragments), fe(s.n_fragments);
    rc = raft_hip_fetch(ctx, cov_off.data(), cov.data(), rep_off.data(), rep_s.data(), rep_e.data(), nullptr, nullptr,
                        frag_off.data(), nullptr, fb.data(), fe.data());
    if (rc != RAFT_HIP_OK) { std::cout << "ERROR, raft_hip_fetch, " << raft_hip_strerror(rc) << "\n"; exit(1); }
    raft_hip_destroy(ctx);

    // ---- the reference's output formats from the returned arrays
    std::ofstream cov_txt(param.outputfilename + ".coverage.txt");                         // repeat.hpp:105-108
    std::ofstream long_repeats(param.outputfilename + ".long_repeats.txt");
    std::ofstream long_repeats_bed(param.outputfilename + ".long_repeats.bed");
    for (int i = 0; i < n_read; i++) {
        cov_txt << "read " << i << " ";
        for (int64_t j = cov_off[i]; j < cov_off[i + 1]; j++) cov_txt << (j - cov_off[i]) * param.reso << "," << cov[j] << " ";
        cov_txt << std::endl;
        for (int64_t k = rep_off[i]; k < rep_off[i + 1]; k++)                              // Read::long_repeats
            reads[i]->long_repeats.push_back(std::pair<int, int>(rep_s[k], rep_e[k]));
    }
    const double coverage_per_window = (double)s.total_coverage / (int)s.total_windows;    // repeat.hpp:173-178
    fprintf(stdout, "coverage per window is %f \n", coverage_per_window);
    fprintf(stdout, "coverage per window/average coverage is %f \n", coverage_per_window / param.est_cov);
    fprintf(stdout, "fraction_of_repeat_length %f \n", (double)s.total_repeat_length / s.total_read_length);
    for (int i = 0; i < n_read; i++) {                                                     // repeat.hpp:180-203
        long_repeats << "read " << i << ", ";
        for (auto &lr : reads[i]->long_repeats) {
            long_repeats << lr.first << "," << lr.second << "    ";
            if (!param.real_reads) {
                if (reads[i]->align.compare("forward") == 0)
                    long_repeats_bed << reads[i]->chr << "\t" << reads[i]->start_pos + lr.first << "\t" << reads[i]->start_pos + lr.second << std::endl;
                else if (reads[i]->align.compare("reverse") == 0)
                    long_repeats_bed << reads[i]->chr << "\t" << reads[i]->end_pos - lr.second << "\t" << reads[i]->end_pos - lr.first << std::endl;
            }
        }
        long_repeats << std::endl;
    }
    for (int i = 0; i < n_read; i++) {                                                     // chop.hpp:250-322
        const Read *rd = reads[i];
        const bool whole = frag_off[i + 1] - frag_off[i] == 1;
        for (int64_t f = frag_off[i]; f < frag_off[i + 1]; f++) {
            const int read_num = (int)f + 1, b = fb[f], e = fe[f];
            if (!param.real_reads) {
                const std::string tail = rd->name.substr(rd->name.find_last_of(','));
                if (whole)
                    reads_final << ">read=" << read_num << "," << rd->align << ",position=" << rd->start_pos << "-" << rd->end_pos
                                << ",length=" << rd->len << tail << "\n";
                else if (rd->align.compare("forward") == 0)
                    reads_final << ">read=" << read_num << "," << rd->align << ",position=" << rd->start_pos + b << "-"
                                << rd->start_pos + e << ",length=" << e - b << tail << "\n";
                else if (rd->align.compare("reverse") == 0)
                    reads_final << ">read=" << read_num << "," << rd->align << ",position=" << rd->end_pos - e << "-"
                                << rd->end_pos - b << ",length=" << e - b << tail << "\n";
            } else {
                reads_final << ">read=" << read_num << "," << rd->name << ",pos_on_original_read=" << b << "-" << e << "\n";
            }
            reads_final << rd->bases.substr(b, e - b) << "\n";
        }
    }
}

int main(int argc, char *argv[])
{
    algoParams params;
    params.initParams();
    int option;
    while ((option = getopt(argc, argv, "r:e:m:l:i:p:f:v:o:")) != -1) {          // the reference's flags (main.cpp:28-59)
        switch (option) {
        case 'r': params.reso = atoi(optarg); break;
        case 'e': params.est_cov = atoi(optarg); break;
        case 'm': params.cov_mul = std::stod(optarg); break;
        case 'l': params.read_length = atoi(optarg); break;
        case 'p': params.repeat_length = atoi(optarg); params.interval_length = atoi(optarg); break;
        case 'f': params.flanking_length = atoi(optarg); break;
        case 'v': params.overlap_length = atoi(optarg);   /* falls through, as in the reference */
        case 'o': params.outputfilename = optarg; break;
        default: return 1;
        }
    }
    if (argc < optind + 2 || params.est_cov <= 0) return 1;
    params.printParams();
    auto tStart = std::chrono::system_clock::now();
    std::cout << "INFO, main(), started timer\n";
    break_long_reads_hip(argv[optind], argv[optind + 1], params);
    std::chrono::duration<double> wct = std::chrono::system_clock::now() - tStart;
    std::cout << "INFO, main(), program completed after " << wct.count() << " seconds\n";
    return 0;
}
