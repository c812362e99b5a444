// raft_main.cpp -- `raft [options] <input-reads.fa> <in.paf>`: the reference's command line (main.cpp:21-87,
// chop.hpp:331-373) in front of the MI355X engine.  Host code only tokenises text, resolves names and formats the
// four output files; everything between "records are integers" and "fragment bounds exist" runs on the GPU through
// the C ABI of include/raft_hip.h.  There is no CPU fallback: without libraft_hip.so / a gfx950 device it fails.
//
// Kept on purpose (they change file names or exit codes): -p sets repeat_length AND
// interval_length (main.cpp:44-47); -v falls through into -o (main.cpp:51-55); -i is accepted by getopt but has
// no case, so it prints the usage and exits 1 (main.cpp:56-57); PREFIX.reads.fasta is created before the inputs
// are validated (chop.hpp:333-349); messages go to stdout.
#include "../../include/raft_hip.h"
#include "../../include/raft_host.h"

#include <unistd.h>

#include <chrono>
#include <thread>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <fstream>
#include <future>
#include <iostream>
#include <ctime>
#include <string>
#include <vector>

namespace {

struct Params {            // param.hpp:18-31
    int reso = 50, est_cov = 0;
    double cov_mul = 1.5;
    int repeat_length = 10000, interval_length = 10000, read_length = 20000, overlap_length = 500, flanking_length = 1000;
    std::string prefix = "raft";
};

[[noreturn]] void print_help(const Params &p) // main.cpp:7-19
{
    std::cout << "Usage: raft [options] <input-reads.fa> <in.paf>\n";
    std::cout << "  -r NUM     resolution of coverage " << p.reso << "\n";
    std::cout << "  -e NUM     estimated coverage " << "\n";
    std::cout << "  -m NUM     coverage multiplier " << p.cov_mul << "\n";
    std::cout << "  -l NUM     read_length " << p.read_length << "\n";
    std::cout << "  -v NUM     overlap_length " << p.overlap_length << "\n";
    std::cout << "  -p NUM     repeat_length " << p.repeat_length << "\n";
    std::cout << "  -f NUM     flanking_length " << p.flanking_length << "\n";
    std::cout << "  -o FILE    prefix of output files " << p.prefix << "\n";
    std::cout.flush();
    exit(1);
}

bool missing_or_empty(const char *fn) // chop.hpp:326-329,336-349
{
    std::ifstream f(fn);
    return !f || f.peek() == std::ifstream::traits_type::eof();
}

std::thread *g_background[4] = {nullptr, nullptr, nullptr, nullptr};   // helpers that must be finished before the process exits

[[noreturn]] void die(const std::string &msg)
{
    for (std::thread *t : g_background)
        if (t && t->joinable()) t->join();
    std::cout << msg << "\n";
    std::cout.flush();
    exit(1);
}

} // namespace

int main(int argc, char *argv[])
{
    Params p;
    int option;
    while ((option = getopt(argc, argv, "r:e:m:l:i:p:f:v:o:")) != -1) {
        switch (option) {
        case 'r': p.reso = atoi(optarg); break;
        case 'e': p.est_cov = atoi(optarg); break;
        case 'm': p.cov_mul = std::stod(optarg); break;
        case 'l': p.read_length = atoi(optarg); break;
        case 'p': p.repeat_length = atoi(optarg); p.interval_length = atoi(optarg); break;
        case 'f': p.flanking_length = atoi(optarg); break;
        case 'v': p.overlap_length = atoi(optarg); // no break in the reference: -v also sets the prefix (main.cpp:51-55)
                  /* fall through */
        case 'o': p.prefix = optarg; break;
        default: print_help(p);
        }
    }
    if (argc < optind + 2) print_help(p);
    if (p.est_cov <= 0) {
        std::cout << "ERROR, main(), estimated coverage must be set properly\n";
        print_help(p);
    }
    // param.hpp:33-43
    std::cout << "INFO, printParams(), reso = " << p.reso << "\n";
    std::cout << "INFO, printParams(), est_cov = " << p.est_cov << "\n";
    std::cout << "INFO, printParams(), cov_mul = " << p.cov_mul << "\n";
    std::cout << "INFO, printParams(), repeat_length = " << p.repeat_length << "\n";
    std::cout << "INFO, printParams(), interval_length = " << p.interval_length << "\n";
    std::cout << "INFO, printParams(), read_length = " << p.read_length << "\n";
    std::cout << "INFO, printParams(), overlap_length = " << p.overlap_length << "\n";
    std::cout << "INFO, printParams(), flanking_length = " << p.flanking_length << "\n";

    const auto t_start = std::chrono::system_clock::now();
    std::cout << "INFO, main(), started timer\n";

    const char *reads_fn = argv[optind], *paf_fn = argv[optind + 1];
    const std::string fasta_out = p.prefix + ".reads.fasta";
    { std::ofstream touch(fasta_out); }               // chop.hpp:333: created before any validation
    if (missing_or_empty(reads_fn)) die(std::string("ERROR, break_long_reads(), ") + reads_fn + " input file either does not exist or is empty");
    if (missing_or_empty(paf_fn)) die(std::string("ERROR, break_long_reads(), ") + paf_fn + " input file either does not exist or is empty");

    raft_hip_params hp{};
    hp.reso = p.reso; hp.est_cov = p.est_cov; hp.cov_mul = p.cov_mul; hp.repeat_length = p.repeat_length;
    hp.interval_length = p.interval_length; hp.read_length = p.read_length; hp.overlap_length = p.overlap_length;
    hp.flanking_length = p.flanking_length; hp.symmetric_mode = -1;
    // stage clock on stderr when RAFT_TIMING is set (stdout stays the reference's)
    const bool timing = getenv("RAFT_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    const auto t_main = t_prev;
    auto stage = [&](const char *what) {
        const auto now = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "TIMING %-16s %8.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
        t_prev = now;
    };
    if (timing) {
        // what the stage clock cannot see from inside (1.4 s of a 4.8 s run appeared in no stage): how old the process was
        // when main() got here -- the loader mapping libamdhip64 and friends -- and, at the end, the moment main() leaves, so that
        // whoever started the process can tell what the teardown behind _exit took (tools/cli_big.py prints both)
        double age = -1.0;
        if (FILE *f = fopen("/proc/self/stat", "r")) {
            char buf[1024];
            const size_t n = fread(buf, 1, sizeof buf - 1, f);
            fclose(f);
            buf[n] = 0;
            if (const char *q = strrchr(buf, ')')) {           // (fields behind the command name: state is field 3, starttime field 22)
                unsigned long long start = 0;
                int field = 2;
                for (const char *t = q + 1; *t && field < 22; ++t)
                    if (*t == ' ') { ++field; if (field == 22) start = strtoull(t + 1, nullptr, 10); }
                double up = 0.0;
                if (FILE *u = fopen("/proc/uptime", "r")) { if (fscanf(u, "%lf", &up) != 1) up = 0.0; fclose(u); }
                if (start && up > 0.0) age = up - (double)start / (double)sysconf(_SC_CLK_TCK);
            }
        }
        fprintf(stderr, "TIMING %-16s %8.3f s\n", "process->main", age);
    }

    // The device contexts come up (runtime init, first allocations) while the host tokenises the inputs.
    // RAFT_DEVICES=0,1,...: the GPUs of the node that share the job (reads shard across them, host-routed, no collective;
    // SURVEY.md §8e); RAFT_DEVICE=n: a single one; default: device 0.  A device may be named twice.
    std::vector<int> devices;
    if (const char *e = getenv("RAFT_DEVICES")) {
        for (const char *q = e; *q;) {
            char *end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q) break;
            devices.push_back((int)v);
            q = (*end == ',') ? end + 1 : end;
            if (*end != ',' && *end != '\0') break;
        }
    }
    if (devices.empty()) { const char *e = getenv("RAFT_DEVICE"); devices.push_back(e ? atoi(e) : 0); }
    // RAFT_RANKS=N: the PRE-SPLIT job (BASELINE configs[3]): the record stream is cut into N contiguous slices, rank r
    // -- a context on device r modulo the devices named -- holds slice r, and ONE exchange step routes every interval to the rank
    // that owns its read (raft_hip_run_presplit_local).  Outputs are the single-rank run's, byte for byte.
    const int ranks = getenv("RAFT_RANKS") ? std::max(1, std::min(64, atoi(getenv("RAFT_RANKS")))) : 0;
    if (ranks > 0) {
        const std::vector<int> named = devices;
        devices.clear();
        for (int r = 0; r < ranks; ++r) devices.push_back(named[(size_t)r % named.size()]);
    }
    std::vector<raft_hip_ctx *> ctxs(devices.size(), nullptr);
    std::vector<int> create_rc(devices.size(), RAFT_HIP_OK);
    std::promise<void> devices_up_p;
    std::shared_future<void> devices_up = devices_up_p.get_future().share();
    std::thread bring_up([&] {
        std::vector<std::thread> th;
        // (created and warmed up: the engine's code goes to the device, the pipeline's lanes come up -- 70-80 ms that would
        // otherwise sit inside the first job)
        auto up = [&](size_t d) {
            create_rc[d] = raft_hip_create(devices[d], &hp, &ctxs[d]);
            if (create_rc[d] == RAFT_HIP_OK && !getenv("RAFT_NO_WARM_UP")) create_rc[d] = raft_hip_warm_up(ctxs[d]);
        };
        for (size_t d = 1; d < devices.size(); ++d) th.emplace_back([&, d] { up(d); });
        up(0);
        for (auto &t : th) t.join();
        devices_up_p.set_value();
    });
    g_background[0] = &bring_up;
    // Arrays that cross PCIe are page-locked once the runtime is up (raft_hip_host_register: the link's 53 GB/s instead of
    // the runtime's staging of pageable memory); not worth the calls for small inputs.  A failed registration is not an
    // error: the copy then takes the pageable path.
    const bool no_pin = getenv("RAFT_NO_PIN") != nullptr;
    auto pin = [&](const void *ptr, size_t bytes) {
        if (!no_pin && ptr && bytes >= (size_t)(64u << 10)) (void)raft_hip_host_register(const_cast<void *>(ptr), bytes);
    };

    // The overlaps file's bytes need nothing of the reads: they are read -- inflated, for a .gz -- beside the loading of
    // the reads (on gz inputs, the reference's own quick-start shape, the two inflations are most of the run).
    raft_host_text *paf_text = nullptr;
    int paf_text_rc = RAFT_HOST_OK;
    std::thread paf_reader([&] { paf_text_rc = raft_host_text_read(paf_fn, &paf_text); });
    g_background[2] = &paf_reader;

    raft_host_reads *reads = nullptr;
    int rc = raft_host_reads_load(reads_fn, &reads);
    stage("reads_load");
    if (rc == RAFT_HOST_ERR_DUP_NAME) die("ERROR, loadFASTA(), two reads share a name");
    if (rc != RAFT_HOST_OK) die(std::string("ERROR, loadFASTA(), cannot read ") + reads_fn);
    const int32_t n_reads = raft_host_reads_count(reads);
    std::cout.flush();
    if (n_reads > 0) fprintf(stdout, "Real Reads %d \n", raft_host_reads_real(reads)); // chop.hpp:105

    // Host arrays for everything that comes back, sized by the bounds of raft_hip.h (from the read lengths alone).
    // Coverage returns in its transfer encoding (a byte per window + the windows at or above 255): a quarter of the
    // int32 array's bytes over PCIe, and the formatter below reads it as it is.  They are allocated and page-locked beside
    // the tokenising of the overlaps (pinning untouched pages costs their first touch: 0.1 s for the 2 GB of a human set).
    const int32_t *rl = raft_host_reads_lengths(reads);
    int64_t n_win = 0, rep_cap = 0, frag_cap = 0;
    std::vector<int64_t> cov_off((size_t)n_reads + 1), rep_off((size_t)n_reads + 1), frag_off((size_t)n_reads + 1);
    std::unique_ptr<uint8_t[]> cov8;
    std::unique_ptr<int32_t[]> rep_s, rep_e, fb, fe;
    std::vector<int64_t> exc_i;
    std::vector<int32_t> exc_v;
    std::vector<int32_t> cov_anchor;          // the four-bit step encoding's block anchors (cov_width = RAFT_HIP_COV_DELTA4)
    int64_t exc_cap0 = 0;
    std::thread out_prep([&] {
        const int64_t minw = std::max<int64_t>(((int64_t)p.repeat_length + p.reso - 1) / p.reso, 1);
        int64_t sum_len = 0;
        for (int32_t i = 0; i < n_reads; ++i) { n_win += ((int64_t)rl[i] + p.reso - 1) / p.reso; sum_len += rl[i]; }
        rep_cap = (n_win + n_reads) / (minw + 1);
        frag_cap = sum_len / p.interval_length + 2 * (int64_t)n_reads;
        cov8.reset(new uint8_t[((size_t)n_win + 1) * 2]);                         // (not value-initialised: no zero fill)
        rep_s.reset(new int32_t[(size_t)rep_cap + 1]); rep_e.reset(new int32_t[(size_t)rep_cap + 1]);
        fb.reset(new int32_t[(size_t)frag_cap + 1]); fe.reset(new int32_t[(size_t)frag_cap + 1]);
        exc_cap0 = std::max<int64_t>(1 << 16, n_win / 64);
        exc_i.resize((size_t)exc_cap0); exc_v.resize((size_t)exc_cap0);
        cov_anchor.resize((size_t)n_win / 1024 + 2);
        devices_up.wait();
        for (size_t d = 0; d < devices.size(); ++d) if (create_rc[d] != RAFT_HIP_OK) return;
        pin(exc_i.data(), exc_i.size() * 8); pin(exc_v.data(), exc_v.size() * 4);
        pin(cov_anchor.data(), cov_anchor.size() * 4);
        pin(cov8.get(), ((size_t)n_win + 1) * (p.est_cov >= 40 ? 2 : 1));
        pin(fb.get(), ((size_t)frag_cap + 1) * 4); pin(fe.get(), ((size_t)frag_cap + 1) * 4);
        pin(rep_s.get(), ((size_t)rep_cap + 1) * 4); pin(rep_e.get(), ((size_t)rep_cap + 1) * 4);
        pin(cov_off.data(), cov_off.size() * 8); pin(frag_off.data(), frag_off.size() * 8); pin(rep_off.data(), rep_off.size() * 8);
        pin(rl, (size_t)n_reads * 4);
        // the job's device buffers, sized from what is known by now: the reads' lengths, and the record count to within a few
        // per cent from the size of the overlaps file (a PAF line of hifiasm's has ~63 bytes; .gz: ~4x that when inflated)
        if (!getenv("RAFT_NO_WARM_UP")) {
            std::ifstream pf(paf_fn, std::ios::binary | std::ios::ate);
            const std::string pn(paf_fn);
            const bool gz = pn.size() > 3 && pn.compare(pn.size() - 3, 3, ".gz") == 0;
            const int64_t est = pf ? (int64_t)pf.tellg() * (gz ? 4 : 1) / 60 : 0;
            for (size_t d = 0; d < devices.size(); ++d)
                (void)raft_hip_reserve(ctxs[d], n_reads, rl, est, (int32_t)devices.size(),
                                       getenv("RAFT_NO_DELTA4") ? (p.est_cov >= 40 ? 2 : 1) : RAFT_HIP_COV_DELTA4);   // (what a hifiasm-shaped PAF will use)
        }
    });
    g_background[3] = &out_prep;

    raft_host_paf *paf = nullptr;
    char bad[256] = {0};
    paf_reader.join();
    stage("paf_read (rest)");
    rc = paf_text_rc;
    if (rc == RAFT_HOST_OK) rc = raft_host_paf_parse(paf_text, reads, &paf, bad, sizeof bad);
    if (raft_host_paf_count(paf) < (1 << 22)) { raft_host_text_free(paf_text); paf_text = nullptr; }   // (a big file's GBs of touched pages are left to the process exit: unmapping them here -- or beside
                                                                                   // the pass, where it holds the address space's lock against the page-locking -- cost 0.3 s of a 4 s run)
    if (rc == RAFT_HOST_ERR_UNKNOWN_NAME) die(std::string("ERROR, create_pileup(), read ") + bad + " of the overlaps file is not in the reads file");
    if (rc != RAFT_HOST_OK) die(std::string("ERROR, create_pileup(), cannot read ") + paf_fn);
    const int64_t n_rec = raft_host_paf_count(paf);
    stage("paf_load");
    bring_up.join();
    for (size_t d = 0; d < devices.size(); ++d)
        if (create_rc[d] != RAFT_HIP_OK)
            die(std::string("ERROR, raft_hip_create(), device ") + std::to_string(devices[d]) + ": " + raft_hip_strerror(create_rc[d]));
    raft_hip_ctx *ctx = ctxs[0];
    stage("device_wait");

    // The tokeniser already knows whether the PAF is symmetric (chop.hpp:171-184, found while the lines were in
    // registers): the engine is told, so it neither scans for the mirror of record 0 nor -- for a symmetric PAF -- is
    // handed the target columns at all (half of the upload).
    hp.symmetric_mode = ranks > 0 ? -1 : (raft_host_paf_symmetric(paf) ? 1 : 0);   // (a pre-split job finds the flag across its ranks)
    out_prep.join();                                  // (its raft_hip_reserve reads the contexts' parameters: done before they change)
    rc = raft_hip_set_params(ctx, &hp);
    if (rc != RAFT_HIP_OK) die(std::string("ERROR, raft_hip_set_params(), ") + raft_hip_strerror(rc));
    const bool sym = hp.symmetric_mode == 1 || (ranks > 0 && raft_host_paf_symmetric(paf));

    // hifiasm writes its PAF grouped by query (reference README.md:36-38): a symmetric stream of at most four runs sorted by
    // read id is handed over in its grouped form -- per run, where every read's records begin -- and the query column stays
    // on the host (a third of the upload); any other stream goes up as it is.
    // Round 4: by default neither is prepared here any more.  The engine's host pipeline derives both from the plain columns of a
    // symmetric stream itself, chunk by chunk, beside the uploads of the chunks before (raft_hip.h "derived input"): the two
    // stages below were a second and a third pass over the columns between the parse and the engine (19 ms for 4.4e7 records).
    // RAFT_CLI_PREPARE=1 keeps them (A/B, and the grouped / window-record entry points through the CLI).
    std::unique_ptr<int64_t[]> rec_off;
    int32_t n_runs = 0;
    const bool prepare = getenv("RAFT_CLI_PREPARE") != nullptr && ranks == 0;
    if (prepare && sym && n_rec > 0 && !getenv("RAFT_NO_GROUPED")) {
        rec_off.reset(new int64_t[(size_t)4 * ((size_t)n_reads + 1)]);
        if (raft_host_group_offsets(n_reads, n_rec, raft_host_paf_column(paf, 0), 4, &n_runs, rec_off.get()) != RAFT_HOST_OK) n_runs = 0;
        if (n_runs == 0 && devices.size() == 1 && n_rec < ((int64_t)1 << 29)) {
            // more than four runs (a PAF concatenated from more than two pairs of files): up to sixteen are still grouped input
            // for one device -- merged into one run on the device instead of falling to the counting sort
            rec_off.reset(new int64_t[(size_t)16 * ((size_t)n_reads + 1)]);
            if (raft_host_group_offsets(n_reads, n_rec, raft_host_paf_column(paf, 0), 16, &n_runs, rec_off.get()) != RAFT_HOST_OK) n_runs = 0;
        }
    }
    stage("group_offsets");
    // ... and its records as window records: what profileCoverage uses of an interval is the windows it touches
    // (repeat.hpp:69-72), two 16-bit indices where reads stay below 65,535 windows -- one word per record goes up instead
    // of two.  They are written over the query column, which grouped input no longer needs.
    const uint32_t *win = nullptr;
    if (n_runs > 0 && p.reso <= 32767 && !getenv("RAFT_NO_WINDOWS")) {
        uint32_t *w = reinterpret_cast<uint32_t *>(const_cast<int32_t *>(raft_host_paf_column(paf, 0)));
        int64_t bad = -1;
        if (raft_host_pack_windows(n_rec, raft_host_paf_column(paf, 1), raft_host_paf_column(paf, 2), p.reso, w, &bad) == RAFT_HOST_OK) win = w;
        // (a negative coordinate or a window index beyond 16 bits: the coordinate columns go up, and the engine reports the former)
    }
    stage("pack_windows");
    if (n_runs > 0) pin(rec_off.get(), (size_t)n_runs * ((size_t)n_reads + 1) * 8);
    if (win) pin(const_cast<uint32_t *>(win), (size_t)n_rec * 4);
    else for (int k = n_runs > 0 ? 1 : 0; k < (sym ? 3 : 6); ++k) pin(raft_host_paf_column(paf, k), (size_t)n_rec * 4);
    stage("page-lock");
    // one byte per window unless the expected coverage lets repeats pile up beyond it (from 40x on: two), and two in any
    // case when the first attempt meets more windows at or above 255 than the exception list holds
    // ... but grouped input (whose chunks the pipelines can cut where they like) brings the coverage back as four-bit steps: the
    // step from one window to the next is the pileup's own difference array, within +-7 for all but a few windows in a
    // thousand whatever the depth -- half of a byte per window, a quarter of two
    // (handing over the plain columns, the CLI does not know the stream's shape exactly; 8 k samples tell a handful of sorted runs
    // -- which the engine cuts into chunks wherever it likes -- from a shuffled stream, whose routed chunks end where the host's
    // buckets do and keep the byte encodings)
    bool few_runs = false;
    if (sym && !prepare && n_rec > 0) {
        const int32_t *q = raft_host_paf_column(paf, 0);
        const int64_t S = std::min<int64_t>(n_rec, 8192);
        int descents = 0;
        int64_t prev = 0;
        for (int64_t i = 1; i < S; ++i) {
            const int64_t pos = S > 1 ? i * (n_rec - 1) / (S - 1) : 0;
            if (q[pos] < q[prev]) ++descents;
            prev = pos;
        }
        few_runs = descents < 4;
    }
    int cov_width = ((n_runs > 0 || few_runs) && !getenv("RAFT_NO_DELTA4") && ranks == 0) ? RAFT_HIP_COV_DELTA4 : (p.est_cov >= 40 ? 2 : 1);
    raft_hip_summary s{};
    int64_t n_exc = 0;
    const char *chunks_env = getenv("RAFT_CHUNKS");   // 0 / unset: the engine decides (one piece for small inputs)
    for (int64_t exc_cap = exc_cap0, attempt = 0; attempt < 3; ++attempt) {
        if ((int64_t)exc_i.size() != exc_cap) {          // (a retry with more room: the first size was made and page-locked beside the tokenising)
            if (!no_pin) { (void)raft_hip_host_unregister(exc_i.data()); (void)raft_hip_host_unregister(exc_v.data()); }
            exc_i.resize((size_t)exc_cap); exc_v.resize((size_t)exc_cap);
            pin(exc_i.data(), exc_i.size() * 8); pin(exc_v.data(), exc_v.size() * 4);
        }
        raft_hip_host_outputs ho{};
        ho.cov_offset = cov_off.data(); ho.cov8 = cov8.get(); ho.cov8_cap = n_win; ho.cov_width = cov_width;
        ho.cov_anchor = cov_anchor.data(); ho.anchor_cap = (int64_t)cov_anchor.size();
        ho.exc_index = exc_i.data(); ho.exc_value = exc_v.data(); ho.exc_cap = exc_cap;
        ho.rep_offset = rep_off.data(); ho.rep_s = rep_s.get(); ho.rep_e = rep_e.get(); ho.rep_cap = rep_cap;
        ho.frag_offset = frag_off.data(); ho.frag_begin = fb.get(); ho.frag_end = fe.get(); ho.frag_cap = frag_cap;
        // upload, pass and download of consecutive read ranges overlap, on every device named (one piece for small inputs)
        if (ranks > 0)
            rc = raft_hip_run_presplit_local(ctxs.data(), (int32_t)ctxs.size(), n_reads, rl, n_rec, raft_host_paf_column(paf, 0), raft_host_paf_column(paf, 1),
                                             raft_host_paf_column(paf, 2), raft_host_paf_column(paf, 3), raft_host_paf_column(paf, 4), raft_host_paf_column(paf, 5), &ho, &s);
        else if (win)
            rc = raft_hip_run_multi_windows(ctxs.data(), (int32_t)ctxs.size(), n_reads, rl, n_rec, n_runs, rec_off.get(), win,
                                            chunks_env ? atoi(chunks_env) : 0, &ho, &s);
        else if (n_runs > 0)
            rc = raft_hip_run_multi_grouped(ctxs.data(), (int32_t)ctxs.size(), n_reads, rl, n_rec, n_runs, rec_off.get(),
                                            raft_host_paf_column(paf, 1), raft_host_paf_column(paf, 2), chunks_env ? atoi(chunks_env) : 0, &ho, &s);
        else
            rc = raft_hip_run_multi(ctxs.data(), (int32_t)ctxs.size(), n_reads, rl, n_rec, raft_host_paf_column(paf, 0),
                                    raft_host_paf_column(paf, 1), raft_host_paf_column(paf, 2), sym ? nullptr : raft_host_paf_column(paf, 3),
                                    sym ? nullptr : raft_host_paf_column(paf, 4), sym ? nullptr : raft_host_paf_column(paf, 5),
                                    chunks_env ? atoi(chunks_env) : 0, &ho, &s);
        n_exc = ho.n_exc;
        if (rc != RAFT_HIP_ERR_TOO_LARGE || attempt == 2 || n_exc <= exc_cap) break;
        // more windows at or above the limit than the list holds (n_exc says how many): two bytes per window when a byte
        // leaves more than one window in 16 on the list, else room for exactly those
        if (cov_width == RAFT_HIP_COV_DELTA4 && n_exc > n_win / 8) cov_width = 2;      // (steps that mostly do not fit: not a coverage profile)
        else if (cov_width == 1 && n_exc > n_win / 16) cov_width = 2;
        else exc_cap = n_exc;
    }
    if (rc != RAFT_HIP_OK) {
        std::string m = std::string("ERROR, raft_hip, ") + raft_hip_strerror(rc);
        if (s.error_index >= 0) m += " (index " + std::to_string(s.error_index) + ")";
        const char *d = raft_hip_last_error(ctx);
        if (d && *d) m += std::string(" [") + d + "]";
        die(m);
    }
    stage("engine+fetch");
    if (timing) fprintf(stderr, "TIMING devices_used %d input %s\n", s.n_devices_used, ranks > 0 ? "pre-split slices (one exchange step)" : win ? "windows" : (n_runs > 0 ? "grouped" : (sym ? "columns (offsets and window records derived by the engine)" : "columns")));
    if (timing) fprintf(stderr, "TIMING coverage_encoding %s\n", cov_width == RAFT_HIP_COV_DELTA4 ? "delta4" : (cov_width == 2 ? "uint16" : "uint8"));
    fprintf(stdout, "INFO, Symmetric overlaps %d \n", s.symmetric);            // chop.hpp:189-190
    fprintf(stdout, "INFO, length of alignments  %d()\n", (int)s.n_records);
    fprintf(stdout, "high_cov %d\n", s.high_cov);                              // repeat.hpp:91

    // the four output files are independent: the FASTA is written beside the coverage/repeat tables
    int fasta_rc = RAFT_HOST_OK;
    std::thread fasta_writer([&] { fasta_rc = raft_host_write_fasta(fasta_out.c_str(), reads, frag_off.data(), fb.get(), fe.get()); });
    g_background[1] = &fasta_writer;
    const int cov_rc = cov_width == RAFT_HIP_COV_DELTA4
        ? raft_host_write_coverage_d4((p.prefix + ".coverage.txt").c_str(), n_reads, p.reso, cov_off.data(), cov8.get(), cov_anchor.data(), n_exc,
                                      exc_i.data(), exc_v.data())
        : raft_host_write_coverage_packed_w(cov_width, (p.prefix + ".coverage.txt").c_str(), n_reads, p.reso, cov_off.data(), cov8.get(), n_exc,
                                            exc_i.data(), exc_v.data());
    if (cov_rc != RAFT_HOST_OK ||
        raft_host_write_repeats((p.prefix + ".long_repeats.txt").c_str(), (p.prefix + ".long_repeats.bed").c_str(), reads,
                                rep_off.data(), rep_s.get(), rep_e.get()) != RAFT_HOST_OK) {
        die("ERROR, repeat_annotate(), cannot write output files");
    }
    stage("write_tables");
    // repeat.hpp:173-178 (total_windows is an int in the reference; identical below 2^31 windows)
    const double cpw = (double)s.total_coverage / (double)s.total_windows;
    fprintf(stdout, "coverage per window is %f \n", cpw);
    fprintf(stdout, "coverage per window/average coverage is %f \n", cpw / p.est_cov);
    fprintf(stdout, "fraction_of_repeat_length %f \n", (double)s.total_repeat_length / (double)s.total_read_length);

    fasta_writer.join();
    if (fasta_rc != RAFT_HOST_OK) die("ERROR, break_reads(), cannot write " + fasta_out);
    stage("write_fasta");
    fflush(stdout);

    const std::chrono::duration<double> wct = std::chrono::system_clock::now() - t_start;
    std::cout << "INFO, main(), program completed after " << wct.count() << " seconds\n";
    std::cout.flush();
    fprintf(stdout, "INFO, %s(), CMD:", __func__);   // main.cpp:81-84
    for (int i = 0; i < argc; ++i) fprintf(stdout, " %s", argv[i]);
    fflush(stdout);
    std::cout << "\n";
    stage("stdout");
    if (timing) {
        timespec ts{};
        clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "TIMING %-16s %8.3f s\n", "main() total", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count());
        fprintf(stderr, "TIMING %-16s %lld.%03ld\n", "leaving-at", (long long)ts.tv_sec, ts.tv_nsec / 1000000);
    }
    // Everything is written and closed.  Unmapping a GB of reads, freeing the device buffers and tearing the HIP runtime
    // down cost 0.2-0.3 s of a run that takes a second: leave that to the kernel's process exit.
    const bool out_ok = fflush(stdout) == 0 && !ferror(stdout) && std::cout.good();
    fflush(stderr);
    if (!out_ok) _exit(1);                           // a failed stdout write (closed pipe, full disk) is not a success
    // The fast exit skips atexit handlers and static destructors, which tools that finalise at exit rely on (rocprofv3 /
    // rocprofiler-sdk traces, gcov / llvm-cov counters, sanitizer reports): it is off whenever such a tool shows in
    // the environment, and RAFT_CLEAN_EXIT=1 turns it off by hand.
    auto has = [](const char *v) { const char *e = getenv(v); return e && *e; };
    const char *preload = getenv("LD_PRELOAD");
    const bool tooling = has("RAFT_CLEAN_EXIT") || has("ROCP_TOOL_LIBRARIES") || has("ROCPROFILER_REGISTER_FORCE_LOAD") ||
                         has("HSA_TOOLS_LIB") || has("ROCP_TOOL_LIB") || has("LLVM_PROFILE_FILE") || has("GCOV_PREFIX") ||
                         has("ASAN_OPTIONS") || has("LSAN_OPTIONS") || has("UBSAN_OPTIONS") ||
                         (preload && (strstr(preload, "rocprof") || strstr(preload, "asan") || strstr(preload, "tsan")));
    if (!tooling) _exit(0);
    if (paf_text) raft_host_text_free(paf_text);     // (the clean exit gives everything back: leak checkers see none)
    raft_host_paf_free(paf);
    raft_host_reads_free(reads);
    for (raft_hip_ctx *c : ctxs) raft_hip_destroy(c);
    return 0;
}
