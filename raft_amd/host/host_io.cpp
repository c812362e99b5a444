// host_io.cpp -- text layer of the MI355X RAFT engine: FASTA/FASTQ + PAF readers and the writers of
// PREFIX.coverage.txt / .long_repeats.txt / .long_repeats.bed / .reads.fasta (include/raft_host.h).
//
// Everything here is written from the behaviour of the reference's readers and writers (file:line in
// raft_host.h); kseq.h / paf.hpp are not copied.  Tokenisation rules kept exactly:
//   * a sequence record starts at the next '>' or '@'; its name runs to the first isspace() byte, the
//     rest of the header line is dropped; sequence lines are concatenated until a line that STARTS with
//     '>', '@' or '+'; empty lines are skipped; one trailing '\r' per line is dropped (only when the
//     accumulated sequence is longer than one byte -- kseq's `l > 1` test);
//   * after '+', the rest of that line is skipped and quality lines are consumed until they are at
//     least as long as the sequence; a length mismatch or EOF ends the file (records so far are kept);
//   * a PAF line is cut at '\n', loses one trailing '\r' (when longer than one byte), is split on TAB
//     only and is skipped when it has fewer than 10 fields; numeric fields go through strtol.
#include "../../include/raft_host.h"

#include <emmintrin.h>
#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// Worker threads of the host layer: RAFT_HOST_THREADS (at most kMaxThreads = 128); default: the hardware threads, at most 16 --
// the loaders and writers are bound by memory and by the page cache, not by arithmetic, and on the 256-thread MI355X box the
// 10 GB set ran 3.7 s with 16 workers against 4.1 / 4.7 / 4.4 s with 32 / 64 / 128 (profiles/r05_cli_thread_sweep.txt).  1 = everything
// inline on the calling thread (the reference's own behaviour; used by tests to cross-check the parallel paths).
constexpr int kMaxThreads = 128, kDefaultThreads = 16;
std::atomic<int> g_threads{0};   // 0 = not chosen yet

int host_threads()
{
    int n = g_threads.load(std::memory_order_relaxed);
    if (n > 0) return n;
    const char *e = getenv("RAFT_HOST_THREADS");
    int v = e ? atoi(e) : std::min((int)std::thread::hardware_concurrency(), kDefaultThreads);
    n = std::min(std::max(v, 1), kMaxThreads);
    g_threads.store(n, std::memory_order_relaxed);
    return n;
}

// ... and the workers of the text FORMATTERS (coverage.txt, the fragment FASTA: write_ordered below), which turn integers into
// digits and copy bases into blocks -- arithmetic and private buffers, where more workers do help: RAFT_FORMAT_THREADS, default the
// hardware threads, at most 64 (write_tables 0.8 s with 128 workers, 1.5-1.7 s with 16 on the same box).
int format_threads()
{
    if (host_threads() <= 1) return 1;           // (one thread means one thread everywhere)
    static const int n = [] {
        const char *e = getenv("RAFT_FORMAT_THREADS");
        const int v = e ? atoi(e) : std::min((int)std::thread::hardware_concurrency(), 64);
        return std::min(std::max(v, 1), kMaxThreads);
    }();
    return std::max(n, host_threads());
}

template <class F> void parallel_for(int n_tasks, F fn)   // fn(task) for task in [0, n_tasks), one thread per task
{
    if (n_tasks <= 1) { for (int t = 0; t < n_tasks; ++t) fn(t); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)n_tasks - 1);
    for (int t = 1; t < n_tasks; ++t) th.emplace_back([&fn, t] { fn(t); });
    fn(0);
    for (auto &x : th) x.join();
}

// ---- buffered byte stream over gz/plain files --------------------------------------------------
// With more than one host thread the file is inflated / read by a helper thread that keeps a few 4 MB blocks ahead of
// the parser (gz inputs -- the reference's own quick-start shape, chop.hpp:93, paf.hpp:29 -- are bound by zlib's
// single-stream inflate; the parser's work now hides beside it instead of adding to it).  One thread: read in place.
class Stream {
public:
    explicit Stream(const char *path) : f_(gzopen(path, "rb"))
    {
        if (!f_) return;
        gzbuffer(f_, 1 << 20);
        async_ = host_threads() > 1;
        if (async_) {
            for (auto &b : ring_) b.data.resize(kBlock);
            producer_ = std::thread([this] { produce(); });
        } else ring_[0].data.resize(1 << 20);
    }
    ~Stream()
    {
        if (async_) {
            { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
            cv_.notify_all();
            if (producer_.joinable()) producer_.join();
        }
        if (f_) gzclose(f_);
    }
    Stream(const Stream &) = delete;
    Stream &operator=(const Stream &) = delete;
    bool ok() const { return f_ != nullptr; }
    int getc()
    {
        if (b_ >= e_ && !fill()) return -1;
        return cur_[b_++];
    }
    // Appends bytes up to (not including) the first byte for which is_delim holds and consumes that byte.
    // Returns false only when the stream was already exhausted (nothing at all could be looked at).
    template <class Pred> bool get_until(Pred is_delim, std::string &out, int *delim)
    {
        bool gotany = false;
        if (delim) *delim = 0;
        for (;;) {
            if (b_ >= e_ && !fill()) break;
            gotany = true;
            size_t i = b_;
            while (i < e_ && !is_delim(cur_[i])) ++i;
            out.append(reinterpret_cast<const char *>(&cur_[b_]), i - b_);
            b_ = i + 1;
            if (i < e_) { if (delim) *delim = cur_[i]; break; }
            b_ = e_;
        }
        return gotany;
    }
    bool get_line(std::string &out) // appends; drops one trailing '\r' when the result is longer than one byte
    {
        const bool any = get_until([](unsigned char c) { return c == '\n'; }, out, nullptr);
        if (any && out.size() > 1 && out.back() == '\r') out.pop_back();
        return any;
    }
    // Bulk form for callers that want the whole rest of the stream: hands out the blocks as they come.
    // Returns the number of bytes placed at *p (valid until the next call), 0 at the end.
    size_t next_block(const unsigned char **p)
    {
        if (b_ >= e_ && !fill()) return 0;
        *p = cur_ + b_;
        const size_t n = e_ - b_;
        b_ = e_;
        return n;
    }

private:
    static constexpr size_t kBlock = 4u << 20;
    static constexpr int kRing = 4;
    struct Block { std::vector<unsigned char> data; size_t n = 0; bool full = false; };
    void produce()
    {
        for (int w = 0;; w = (w + 1) % kRing) {
            Block &blk = ring_[w];
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return !blk.full || stop_; });
                if (stop_) return;
            }
            const int n = gzread(f_, blk.data.data(), (unsigned)kBlock);
            {
                std::lock_guard<std::mutex> g(mu_);
                blk.n = n > 0 ? (size_t)n : 0;
                blk.full = true;
                if (n <= 0) done_ = true;
            }
            cv_.notify_all();
            if (n <= 0) return;
        }
    }
    bool fill()
    {
        if (eof_ || !f_) return false;
        if (!async_) {
            const int n = gzread(f_, ring_[0].data.data(), (unsigned)ring_[0].data.size());
            if (n <= 0) { eof_ = true; b_ = e_ = 0; return false; }
            cur_ = ring_[0].data.data(); b_ = 0; e_ = (size_t)n;
            return true;
        }
        if (have_) {                                   // give the block just consumed back to the producer
            { std::lock_guard<std::mutex> g(mu_); ring_[r_].full = false; }
            cv_.notify_all();
            r_ = (r_ + 1) % kRing;
            have_ = false;
        }
        Block &blk = ring_[r_];
        {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return blk.full; });
        }
        if (blk.n == 0) { eof_ = true; b_ = e_ = 0; return false; }
        cur_ = blk.data.data(); b_ = 0; e_ = blk.n; have_ = true;
        return true;
    }
    gzFile f_;
    Block ring_[kRing];
    const unsigned char *cur_ = nullptr;
    size_t b_ = 0, e_ = 0;
    bool eof_ = false, async_ = false, have_ = false;
    int r_ = 0;
    std::thread producer_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false, done_ = false;
};

// ---- BGZF (blocked gzip: bgzip, htslib, samtools) inflated by all workers --------------------------------------------
// A plain .gz is ONE deflate stream and inflates on one thread (the Stream above: the floor of a gz run).  A BGZF file is a
// series of gzip members of at most 64 KiB, each carrying its compressed size in an extra field ('B','C', RFC 1952 / SAM spec
// 4.1): the members are found by hopping from header to header and inflate independently.  Returns false -- and leaves the
// single-stream reader to it -- unless the WHOLE file is well-formed BGZF; `out` gets one spare byte behind the data.
bool inflate_bgzf_parallel(const char *path, std::unique_ptr<char[]> &out, size_t &out_n)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 28) { close(fd); return false; }
    const size_t n = (size_t)st.st_size;
    void *map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return false;
    struct Unmap { void *p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, n};
    const unsigned char *d = static_cast<const unsigned char *>(map);
    struct Member { size_t at, data, dlen, isize, dst; };
    std::vector<Member> mem;
    size_t total = 0;
    for (size_t p = 0; p < n;) {
        if (n - p < 18 || d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8 || !(d[p + 3] & 4)) return false;
        if (d[p + 3] & ~4) return false;                                   // (name / comment / header crc: not what bgzip writes)
        const size_t xlen = (size_t)d[p + 10] | ((size_t)d[p + 11] << 8);
        if (p + 12 + xlen > n) return false;
        size_t bsize = 0;
        for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const size_t slen = (size_t)d[q + 2] | ((size_t)d[q + 3] << 8);
            if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2 && q + 6 <= p + 12 + xlen) bsize = ((size_t)d[q + 4] | ((size_t)d[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || p + bsize > n) return false;
        const size_t isize = (size_t)d[p + bsize - 4] | ((size_t)d[p + bsize - 3] << 8) | ((size_t)d[p + bsize - 2] << 16) | ((size_t)d[p + bsize - 1] << 24);
        if (isize > 65536) return false;
        mem.push_back(Member{p, p + 12 + xlen, bsize - 12 - xlen - 8, isize, total});
        total += isize;
        p += bsize;
    }
    if (mem.empty()) return false;
    std::unique_ptr<char[]> buf(new char[total + 1]);
    const int T = std::max(1, host_threads());
    std::vector<char> bad((size_t)T, 0);
    parallel_for(T, [&](int t) {
        const size_t a = mem.size() * (size_t)t / (size_t)T, b = mem.size() * ((size_t)t + 1) / (size_t)T;
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { bad[(size_t)t] = 1; return; }
        for (size_t i = a; i < b && !bad[(size_t)t]; ++i) {
            const Member &m = mem[i];
            if (m.isize == 0) continue;                                    // (the empty member that ends a BGZF file)
            inflateReset(&z);
            z.next_in = const_cast<unsigned char *>(d + m.data); z.avail_in = (unsigned)m.dlen;
            z.next_out = reinterpret_cast<unsigned char *>(buf.get() + m.dst); z.avail_out = (unsigned)m.isize;
            const int rc = inflate(&z, Z_FINISH);
            if (rc != Z_STREAM_END || z.avail_out != 0) { bad[(size_t)t] = 1; break; }
            const unsigned long crc = crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const unsigned char *>(buf.get() + m.dst), (unsigned)m.isize);
            const unsigned char *c = d + m.at + (m.data - m.at) + m.dlen;
            if (crc != ((unsigned long)c[0] | ((unsigned long)c[1] << 8) | ((unsigned long)c[2] << 16) | ((unsigned long)c[3] << 24))) bad[(size_t)t] = 1;
        }
        inflateEnd(&z);
    });
    for (char b : bad) if (b) return false;
    out.swap(buf);
    out_n = total;
    return true;
}

// ---- name table: open addressing over (offset, length) into one arena ---------------------------
// Read names -> ids.  Open addressing over 16-byte slots that hold what a lookup needs next to each other: 32 bits of the
// hash, the id, and where the name lies in the arena.  A PAF line looks its target name up in a table of 10^5..10^7 names -- a miss
// in every cache level that matters -- so the dependent loads are the cost: slot, then the name's bytes (compared in full: the
// hash only decides where to look).  (Until round 6: slot -> length -> offset -> bytes, four misses and a bytewise hash,
// 160 of the tokeniser's 230 ns per record.)
class NameTable {
public:
    int32_t find(const char *s, size_t n) const { return find_hashed(hash(s, n), s, n); }
    // a lookup in three steps, so that a caller with several names in hand can have their cache misses overlap:
    // hash() + prefetch_slot(), then prefetch_name(), then find_hashed()
    void prefetch_slot(unsigned long long hv) const { if (!slots_.empty()) __builtin_prefetch(&slots_[(size_t)hv & (slots_.size() - 1)]); }
    void prefetch_name(unsigned long long hv) const
    {
        if (slots_.empty()) return;
        const Slot &e = slots_[(size_t)hv & (slots_.size() - 1)];
        if (e.id >= 0) __builtin_prefetch(arena_.data() + e.off);
    }
    int32_t find_hashed(unsigned long long hv, const char *s, size_t n) const
    {
        if (slots_.empty()) return -1;
        const uint32_t tag = (uint32_t)(hv >> 32);
        size_t h = (size_t)hv & (slots_.size() - 1);
        for (;;) {
            const Slot &e = slots_[h];
            if (e.id < 0) return -1;
            if (e.tag == tag && e.len == n && memcmp(arena_.data() + e.off, s, n) == 0) return e.id;
            h = (h + 1) & (slots_.size() - 1);
        }
    }
    // returns the new id, or -1 - existing_id when the name is already present
    int32_t add(const char *s, size_t n)
    {
        if ((off_.size() + 1) * 2 > slots_.size()) grow();
        const int32_t have = find(s, n);
        if (have >= 0) return -1 - have;
        const int32_t id = (int32_t)off_.size();
        off_.push_back(arena_.size()); len_.push_back(n);
        arena_.append(s, n); arena_.push_back('\0');
        place(id);
        return id;
    }
    const char *name(int32_t id) const { return arena_.data() + off_[id]; }
    size_t name_len(int32_t id) const { return len_[id]; }
    size_t size() const { return off_.size(); }

    // eight bytes at a time (multiply, fold), the tail zero-extended: any 64-bit mixing does, equal strings hash equal
    static unsigned long long hash(const char *s, size_t n)
    {
        unsigned long long h = 0x9e3779b97f4a7c15ull ^ (unsigned long long)n;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            unsigned long long w;
            memcpy(&w, s + i, 8);
            h = (h ^ w) * 0xff51afd7ed558ccdull;
            h ^= h >> 32;
        }
        if (i < n) {
            unsigned long long w = 0;
            memcpy(&w, s + i, n - i);
            h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
            h ^= h >> 32;
        }
        h *= 0xff51afd7ed558ccdull;
        return h ^ (h >> 29);
    }
private:
    struct Slot { uint32_t tag; int32_t id; uint64_t off : 40, len : 24; };
    static_assert(sizeof(Slot) == 16, "one slot, one aligned 16 bytes");
    void place(int32_t id)
    {
        const unsigned long long hv = hash(arena_.data() + off_[id], len_[id]);
        size_t h = (size_t)hv & (slots_.size() - 1);
        while (slots_[h].id >= 0) h = (h + 1) & (slots_.size() - 1);
        slots_[h].tag = (uint32_t)(hv >> 32); slots_[h].id = id; slots_[h].off = off_[id]; slots_[h].len = len_[id];
    }
    void grow()
    {
        const size_t n = slots_.empty() ? 1024 : slots_.size() * 2;
        Slot empty{};
        empty.id = -1;
        slots_.assign(n, empty);
        for (int32_t id = 0; id < (int32_t)off_.size(); ++id) place(id);
    }
    std::string arena_;
    std::vector<size_t> off_, len_;
    std::vector<Slot> slots_;
};

// chop.hpp:101: ^read=[0-9]+,[a-z]+,position=[0-9]+-[0-9]+,length=[0-9]+,(.*)  (whole-name match)
bool looks_simulated(const std::string &s)
{
    size_t i = 0;
    auto lit = [&](const char *t) { const size_t n = strlen(t); if (s.compare(i, n, t) != 0) return false; i += n; return true; };
    auto digits = [&]() { const size_t b = i; while (i < s.size() && isdigit((unsigned char)s[i])) ++i; return i > b; };
    auto lower = [&]() { const size_t b = i; while (i < s.size() && s[i] >= 'a' && s[i] <= 'z') ++i; return i > b; };
    return lit("read=") && digits() && lit(",") && lower() && lit(",position=") && digits() && lit("-") && digits() &&
           lit(",length=") && digits() && lit(",");
}

// ---- fast integer formatting ---------------------------------------------------------------------
class Out {
public:
    explicit Out(const char *path) : f_(fopen(path, "wb")) { buf_.reserve(kCap + 64); }
    ~Out() { close(); }
    bool ok() const { return f_ != nullptr && !err_; }
    void ch(char c) { buf_.push_back(c); maybe_flush(); }
    void str(const char *s, size_t n)
    {
        if (n > kCap) { flush(); if (f_ && fwrite(s, 1, n, f_) != n) err_ = true; return; }
        buf_.append(s, n); maybe_flush();
    }
    void str(const char *s) { str(s, strlen(s)); }
    void num(long long v)
    {
        char t[24];
        int n = 0;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) t[n++] = '-';
        while (n) buf_.push_back(t[--n]);
        maybe_flush();
    }
    bool close()
    {
        flush();
        if (f_) { if (fclose(f_) != 0) err_ = true; f_ = nullptr; }
        return !err_;
    }

private:
    static constexpr size_t kCap = 4u << 20;
    void maybe_flush() { if (buf_.size() >= kCap) flush(); }
    void flush()
    {
        if (f_ && !buf_.empty() && fwrite(buf_.data(), 1, buf_.size(), f_) != buf_.size()) err_ = true;
        buf_.clear();
    }
    FILE *f_;
    std::string buf_;
    bool err_ = false;
};

// Appends v in decimal.
inline void put_num(std::string &b, long long v)
{
    char t[24];
    int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) t[n++] = '-';
    while (n) b.push_back(t[--n]);
}

// Writes items 0..n-1 to `path` in order.  Items are grouped into blocks of about `block` weight units; up to
// host_threads() blocks are formatted concurrently (fmt(i, buffer) appends item i) and then written in order, so
// the bytes are exactly those of a sequential writer.  (Writing the blocks concurrently with pwrite at their offsets
// was measured slower: writes to one file serialise on its inode.)
template <class W, class F>
int write_ordered(const char *path, long long n, long long block, W weight, F fmt)
{
    FILE *f = fopen(path, "wb");
    if (!f) return RAFT_HOST_ERR_IO;
    const int T = format_threads();
    bool ok = true;
    long long i = 0;
    std::vector<std::string> buf((size_t)T);
    while (i < n && ok) {
        std::vector<long long> cut{i};
        for (int t = 0; t < T && cut.back() < n; ++t) {
            long long j = cut.back(), acc = 0;
            while (j < n && acc < block) acc += weight(j++);
            cut.push_back(j);
        }
        const int nb = (int)cut.size() - 1;
        parallel_for(nb, [&](int t) {
            buf[(size_t)t].clear();
            for (long long k = cut[(size_t)t]; k < cut[(size_t)t + 1]; ++k) fmt(k, buf[(size_t)t]);
        });
        for (int t = 0; t < nb && ok; ++t)
            if (!buf[(size_t)t].empty() && fwrite(buf[(size_t)t].data(), 1, buf[(size_t)t].size(), f) != buf[(size_t)t].size()) ok = false;
        i = cut.back();
    }
    if (fclose(f) != 0) ok = false;
    return ok ? RAFT_HOST_OK : RAFT_HOST_ERR_IO;
}

} // namespace

struct raft_host_reads {
    bool allow_dup = false;           // split_naive keeps every record, repeated names included
    std::vector<std::string> dup_names;   // (allow_dup) names in file order; `names` is not used then
    NameTable names;
    std::vector<int32_t> lens;
    std::vector<size_t> base_off;
    std::string bases;                // streaming reader: all sequences, concatenated
    std::unique_ptr<char[]> raw_bases;   // mapped-file reader: the same, allocated without being touched (a GB-sized
                                      // zero fill on one thread cost as much as the parallel copy that follows)
    void *map_ptr = nullptr;          // ... or, when every sequence of the file is one line, the mapped file itself (base_off: byte positions in it)
    size_t map_n = 0;
    ~raft_host_reads() { if (map_ptr) munmap(map_ptr, map_n); }
    const char *base_ptr() const { return map_ptr ? static_cast<const char *>(map_ptr) : (raw_bases ? raw_bases.get() : bases.data()); }
    int real_reads = 1;
    // simulated-read mode (chop.hpp:116-121): parsed from every name
    std::vector<int32_t> start_pos, end_pos;
    std::vector<std::string> align, chr;
};

struct raft_host_text {                  // a file's bytes in memory (inflated if it was gz), newline-terminated
    std::unique_ptr<char[]> buf;
    size_t n = 0;
};

struct raft_host_paf {
    std::unique_ptr<int32_t[]> col[6];   // allocated untouched: the workers' copies are the first writes
    size_t n = 0;
    int symmetric = 0;                   // chop.hpp:175-184: some record after the first mirrors the first
};

namespace {

// Name-derived state of one read, in file order: mode decision on the first read (chop.hpp:99-106), id assignment
// (chop.hpp:108), and the simulated-read fields (chop.hpp:25-70, 116-121).
int add_read_meta(raft_host_reads *R, const std::string &name)
{
    if (R->allow_dup) { R->dup_names.push_back(name); return RAFT_HOST_OK; }
    if (R->names.size() == 0 && looks_simulated(name)) R->real_reads = 0;
    const int32_t id = R->names.add(name.data(), name.size());
    if (id < 0) return RAFT_HOST_ERR_DUP_NAME;
    if (!R->real_reads) {
        // position=<start>-<end>, second field = orientation, last field = contig
        const size_t c1 = name.find(',');
        size_t eq = c1 == std::string::npos ? std::string::npos : name.find('=', c1);
        int32_t sp = 0, ep = 0;
        if (eq != std::string::npos) sp = atoi(name.c_str() + eq + 1);
        const size_t dash = name.find('-');
        if (dash != std::string::npos) ep = atoi(name.c_str() + dash + 1);
        R->start_pos.push_back(sp); R->end_pos.push_back(ep);
        std::string al;
        if (c1 != std::string::npos) { const size_t c2 = name.find(',', c1 + 1); al = name.substr(c1 + 1, c2 == std::string::npos ? std::string::npos : c2 - c1 - 1); }
        R->align.push_back(al);
        const size_t lc = name.rfind(',');
        R->chr.push_back(lc == std::string::npos ? std::string() : name.substr(lc + 1));
    }
    return RAFT_HOST_OK;
}

// Fast path for the common input: an uncompressed FASTA whose first byte is '>' and which holds no '\r' and no line
// starting with '@' or '+'.  On such a file kseq's state machine (chop.hpp:88-131 via kseq.h) reduces to "a record
// starts at every line that begins with '>'", so the file is mapped, cut into byte ranges, and scanned by
// host_threads() workers.  Returns -1 when the file is not of that shape (the streaming reader then handles it),
// otherwise a RAFT_HOST_* code.  Results are identical to the streaming reader's.
int load_plain_fasta_parallel(const char *path, raft_host_reads *R)
{
    const int T = host_threads();
    if (T <= 1) return -1;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    unsigned char magic[2] = {0, 0};
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0 || pread(fd, magic, 2, 0) != 2) { close(fd); return -1; }
    size_t n = (size_t)st.st_size;
    void *map = nullptr;
    std::unique_ptr<char[]> heap;                    // a BGZF file's bytes, inflated by all workers (else: the mapped file)
    if (magic[0] == 0x1f && magic[1] == 0x8b) {
        close(fd);
        if (!inflate_bgzf_parallel(path, heap, n) || n == 0) return -1;      // (a plain .gz: the streaming reader)
    } else {
        map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (map == MAP_FAILED) return -1;
    }
    const char *d = map ? static_cast<const char *>(map) : heap.get();
    struct Unmap { void *p; size_t n; ~Unmap() { if (p) munmap(p, n); } } unmap{map, n};
    if (d[0] != '>') return -1;

    // The ONE pass over the file's bytes (round 5; three before: record starts, newlines per record, copy): every worker walks
    // the lines of its byte range -- the '\n' scan brings a line in, the '\r' check runs over it while it is in cache -- and
    // notes the record starts together with how many other lines follow each of them inside the range.  A record's sequence
    // length is then its span minus its newlines, known without looking at the bases again; and when every sequence is ONE
    // line (hifiasm's corrected reads, any FASTA written one line per record) the bases are not copied at all: the mapping
    // stays and every read points into it.
    struct Range { std::vector<size_t> starts; std::vector<uint32_t> lines; size_t lead = 0; char bad = 0; };
    std::vector<Range> rg((size_t)T);
    parallel_for(T, [&](int t) {
        const size_t lo = n * (size_t)t / (size_t)T, hi = n * ((size_t)t + 1) / (size_t)T;
        if (lo >= hi) return;
        Range &g = rg[(size_t)t];
        auto line_start = [&](size_t q) {
            const char c = d[q];
            if (c == '>') { g.starts.push_back(q); g.lines.push_back(0); }
            else {
                if (c == '@' || c == '+') g.bad = 1;
                if (g.starts.empty()) ++g.lead; else ++g.lines.back();
            }
        };
        if (lo == 0) line_start(0);
        // a line start q belongs to the range holding q; its '\n' is at q-1 >= lo-1
        size_t p = lo == 0 ? 0 : lo - 1;
        while (p < hi - 1 && !g.bad) {
            const char *nl = static_cast<const char *>(memchr(d + p, '\n', hi - 1 - p));
            const size_t le = nl ? (size_t)(nl - d) : hi - 1;
            if (memchr(d + p, '\r', le - p)) { g.bad = 1; break; }
            if (!nl) break;
            line_start(le + 1);
            p = le + 1;
        }
        if (!g.bad && hi == n && n >= 1 && d[n - 1] == '\r') g.bad = 1;   // (the file's last byte is scanned by nobody else)
    });
    for (const Range &g : rg) if (g.bad) return -1;
    std::vector<size_t> rec;
    std::vector<uint32_t> seq_lines;             // lines of the record other than its header line
    for (int t = 0; t < T; ++t) {
        const Range &g = rg[(size_t)t];
        if (g.lead) { if (seq_lines.empty()) return -1; seq_lines.back() += (uint32_t)g.lead; }   // (lines of a record that began in an earlier range)
        rec.insert(rec.end(), g.starts.begin(), g.starts.end());
        seq_lines.insert(seq_lines.end(), g.lines.begin(), g.lines.end());
    }
    size_t data_end = n;
    if (!rec.empty() && rec.back() + 1 == n) { data_end = rec.back(); rec.pop_back(); seq_lines.pop_back(); }   // bare '>' as the last byte: kseq finds no name and stops
    const size_t n_rec = rec.size();
    if (n_rec > 0x7fffffffull) return RAFT_HOST_ERR_ARG;

    // name span, first sequence byte, sequence length of every record: a look at the header line only
    std::vector<size_t> name_end(n_rec), seq_begin(n_rec);
    R->lens.assign(n_rec, 0);
    auto rec_range = [&](int t, size_t &a, size_t &b) { a = n_rec * (size_t)t / (size_t)T; b = n_rec * ((size_t)t + 1) / (size_t)T; };
    std::vector<char> multi((size_t)T, 0);
    parallel_for(T, [&](int t) {
        size_t a, b;
        rec_range(t, a, b);
        for (size_t i = a; i < b; ++i) {
            const size_t end = i + 1 < n_rec ? rec[i + 1] : data_end;
            size_t q = rec[i] + 1;
            while (q < end && !isspace((unsigned char)d[q])) ++q;
            name_end[i] = q;
            if (q < end && d[q] != '\n') {                          // comment: skipped to the end of the line
                const char *nl = static_cast<const char *>(memchr(d + q, '\n', end - q));
                q = nl ? (size_t)(nl - d) : end;
            }
            const size_t sb = q < end ? q + 1 : end;
            seq_begin[i] = sb;
            // newlines inside [sb, end): one between two of its lines, one behind the last line unless the file ends without
            size_t newlines = 0;
            if (sb < end) {
                const size_t lines = seq_lines[i];                   // (>= 1: sb itself is a line start that is not a record start)
                newlines = (lines ? lines - 1 : 0) + (d[end - 1] == '\n' ? 1 : 0);
                if (lines > 1) multi[(size_t)t] = 1;
            }
            R->lens[i] = (int32_t)(end - sb - newlines);
        }
    });
    bool any_multi = false;
    for (char m : multi) any_multi = any_multi || m;
    R->base_off.resize(n_rec);
    if (!any_multi) {
        // every sequence is one line: the reads ARE the mapping
        for (size_t i = 0; i < n_rec; ++i) R->base_off[i] = seq_begin[i];
        if (map) { R->map_ptr = map; R->map_n = n; unmap.p = nullptr; }   // (kept until raft_host_reads_free)
        else R->raw_bases.swap(heap);                                 // (the inflated file itself)
    } else {
        size_t total = 0;
        for (size_t i = 0; i < n_rec; ++i) { R->base_off[i] = total; total += (size_t)R->lens[i]; }
        R->raw_bases.reset(new char[total ? total : 1]);
        // bases, lines joined
        parallel_for(T, [&](int t) {
            size_t a, b;
            rec_range(t, a, b);
            for (size_t i = a; i < b; ++i) {
                const size_t end = i + 1 < n_rec ? rec[i + 1] : data_end;
                char *dst = R->raw_bases.get() + R->base_off[i];
                for (size_t p = seq_begin[i]; p < end;) {
                    const char *nl = static_cast<const char *>(memchr(d + p, '\n', end - p));
                    const size_t le = nl ? (size_t)(nl - d) : end;
                    memcpy(dst, d + p, le - p);
                    dst += le - p;
                    p = le + 1;
                }
            }
        });
    }

    // names, in file order (ids are FASTA positions)
    std::string name;
    for (size_t i = 0; i < n_rec; ++i) {
        name.assign(d + rec[i] + 1, name_end[i] - rec[i] - 1);
        const int rc = add_read_meta(R, name);
        if (rc != RAFT_HOST_OK) return rc;
    }
    return RAFT_HOST_OK;
}

} // namespace

// codes/exceptions (raft_hip_fetch_packed[_w]) -> the int32 coverage array
template <class T>
static int unpack_coverage_t(int64_t n_bins, const T *code, int64_t n_exc, const int64_t *exc_index, const int32_t *exc_value, int32_t *cov)
{
    constexpr int32_t kEscape = sizeof(T) == 1 ? 255 : 65535;
    if (n_bins < 0 || n_exc < 0 || (n_bins && (!code || !cov)) || (n_exc && (!exc_index || !exc_value))) return RAFT_HOST_ERR_ARG;
    const int T_ = n_bins < (1 << 22) ? 1 : host_threads();
    parallel_for(T_, [&](int t) {
        const int64_t lo = n_bins * t / T_, hi = n_bins * (t + 1) / T_;
        for (int64_t i = lo; i < hi; ++i) cov[i] = code[i];
    });
    for (int64_t k = 0; k < n_exc; ++k) {
        if (exc_index[k] < 0 || exc_index[k] >= n_bins || code[exc_index[k]] != kEscape) return RAFT_HOST_ERR_ARG;
        cov[exc_index[k]] = exc_value[k];
    }
    return RAFT_HOST_OK;
}

// repeat.hpp:105-108 from the packed form: the escape code stands for the next entry of the (ascending) exception list
template <class T>
static int write_coverage_packed_t(const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset, const T *code,
                                   int64_t n_exc, const int64_t *exc_index, const int32_t *exc_value)
{
    constexpr long long kEscape = sizeof(T) == 1 ? 255 : 65535;
    return write_ordered(path, n_reads, 1 << 20,
                         [&](long long i) { return (long long)(cov_offset[i + 1] - cov_offset[i]) + 4; },
                         [&](long long i, std::string &o) {
                             o.append("read ", 5); put_num(o, i); o.push_back(' ');
                             const int64_t b = cov_offset[i], e = cov_offset[i + 1];
                             const int64_t *x = n_exc ? std::lower_bound(exc_index, exc_index + n_exc, b) : exc_index;
                             for (int64_t j = b; j < e; ++j) {
                                 long long v = code[j];
                                 if (v == kEscape) { v = exc_value[x - exc_index]; ++x; }
                                 put_num(o, (long long)(j - b) * reso); o.push_back(','); put_num(o, v); o.push_back(' ');
                             }
                             o.push_back('\n');
                         });
}

extern "C" {

int raft_host_set_threads(int n)
{
    if (n < 0) return RAFT_HOST_ERR_ARG;
    g_threads.store(n > kMaxThreads ? kMaxThreads : n, std::memory_order_relaxed);   // 0: back to RAFT_HOST_THREADS / hardware default
    return RAFT_HOST_OK;
}

int raft_host_get_threads(void) { return host_threads(); }

static int load_reads(const char *path, raft_host_reads **out, bool allow_dup);

int raft_host_reads_load(const char *path, raft_host_reads **out) { return load_reads(path, out, false); }

static int load_reads(const char *path, raft_host_reads **out, bool allow_dup)
{
    if (!path || !out) return RAFT_HOST_ERR_ARG;
    *out = nullptr;
    {
        raft_host_reads *P = new raft_host_reads();
        P->allow_dup = allow_dup;
        const int prc = load_plain_fasta_parallel(path, P);
        if (prc == RAFT_HOST_OK) { *out = P; return RAFT_HOST_OK; }
        delete P;
        if (prc > 0) return prc;
    }
    Stream in(path);
    if (!in.ok()) return RAFT_HOST_ERR_OPEN;
    raft_host_reads *R = new raft_host_reads();
    R->allow_dup = allow_dup;
    int last = 0; // header byte already consumed by the previous record
    std::string name, seq, junk;
    int rc = RAFT_HOST_OK;
    for (;;) {
        int c;
        if (last == 0) {
            while ((c = in.getc()) >= 0 && c != '>' && c != '@') {}
            if (c < 0) break;
        }
        name.clear();
        int delim = 0;
        if (!in.get_until([](unsigned char ch) { return isspace(ch) != 0; }, name, &delim)) break;
        if (delim != '\n') { junk.clear(); in.get_line(junk); } // comment
        seq.clear();
        while ((c = in.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back((char)c);
            in.get_line(seq);
        }
        last = (c == '>' || c == '@') ? c : 0;
        if (c == '+') {
            while ((c = in.getc()) >= 0 && c != '\n') {}
            if (c < 0) break;                       // no quality string: record and the rest are dropped
            std::string qual;
            // kseq.h:290: one quality line is read in any case (also for an empty sequence), more while it is shorter
            // than the sequence -- so a quality line that starts with '@' or '>' is never taken for a header
            while (in.get_line(qual) && qual.size() < seq.size()) {}
            last = 0;
            if (qual.size() != seq.size()) break;   // kseq_read returns -2: loadFASTA's loop ends
        }
        rc = add_read_meta(R, name);
        if (rc != RAFT_HOST_OK) break;
        R->lens.push_back((int32_t)seq.size());
        R->base_off.push_back(R->bases.size());
        R->bases.append(seq);
    }
    if (rc != RAFT_HOST_OK) { delete R; return rc; }
    *out = R;
    return RAFT_HOST_OK;
}

void raft_host_reads_free(raft_host_reads *r) { delete r; }
int32_t raft_host_reads_count(const raft_host_reads *r) { return r ? (int32_t)r->lens.size() : 0; }
const int32_t *raft_host_reads_lengths(const raft_host_reads *r) { return r ? r->lens.data() : nullptr; }
const char *raft_host_reads_name(const raft_host_reads *r, int32_t i) { return r->names.name(i); }
const char *raft_host_reads_bases(const raft_host_reads *r, int32_t i) { return r->base_ptr() + r->base_off[i]; }
int raft_host_reads_real(const raft_host_reads *r) { return r ? r->real_reads : 1; }

// The bytes of a PAF file -- needs nothing of the reads, so a caller may fetch them (and inflate a .gz) on a thread of
// its own while the reads are still being loaded (raft_main.cpp does: on gz inputs the two inflations are the run time).
int raft_host_text_read(const char *path, raft_host_text **out)
{
    if (!path || !out) return RAFT_HOST_ERR_ARG;
    *out = nullptr;
    std::unique_ptr<char[]> data_buf;    // the file's bytes plus one '\n'; allocated untouched (no serial zero fill)
    size_t data_n = 0;
    bool have = false;
    {   // an uncompressed regular file is read by all workers at once (one pread per slice of the page cache)
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return RAFT_HOST_ERR_OPEN;
        struct stat st;
        unsigned char magic[2] = {0, 0};
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 2 && pread(fd, magic, 2, 0) == 2 &&
            !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            const size_t n = (size_t)st.st_size;
            data_buf.reset(new char[n + 1]);
            char *const data_p = data_buf.get();
            const int T = host_threads();
            std::vector<char> ok((size_t)T, 1);
            parallel_for(T, [&](int t) {
                size_t lo = n * (size_t)t / (size_t)T;
                const size_t hi = n * ((size_t)t + 1) / (size_t)T;
                while (lo < hi) {
                    const ssize_t got = pread(fd, data_p + lo, hi - lo, (off_t)lo);
                    if (got <= 0) { ok[(size_t)t] = 0; break; }
                    lo += (size_t)got;
                }
            });
            have = true;
            for (char o : ok) if (!o) have = false;
            if (have) { data_p[n] = '\n'; data_n = n + 1; }   // a last line without newline is still a line
        }
        close(fd);
    }
    if (!have) {                                         // blocked gzip: every worker inflates its share of the members
        size_t got = 0;
        if (host_threads() > 1 && inflate_bgzf_parallel(path, data_buf, got) && got > 0) {
            data_buf[got] = '\n';                       // a last line without newline is still a line
            data_n = got + 1;
            have = true;
        }
    }
    if (!have) {
        // gz (or not a regular file): inflated by the stream's helper thread while this thread gathers the blocks
        Stream in(path);
        if (!in.ok()) return RAFT_HOST_ERR_OPEN;
        size_t used = 0, cap = 64u << 20;
        std::unique_ptr<char[]> grow(new char[cap]);
        const unsigned char *blk = nullptr;
        for (size_t n; (n = in.next_block(&blk)) != 0;) {
            if (used + n + 1 > cap) {
                while (used + n + 1 > cap) cap *= 2;
                std::unique_ptr<char[]> bigger(new char[cap]);
                memcpy(bigger.get(), grow.get(), used);
                grow.swap(bigger);
            }
            memcpy(grow.get() + used, blk, n);
            used += n;
        }
        if (used) {
            grow[used] = '\n';                          // a last line without newline is still a line
            data_buf.swap(grow);
            data_n = used + 1;
        }
    }
    raft_host_text *T = new raft_host_text();
    T->buf.swap(data_buf);
    T->n = data_n;
    *out = T;
    return RAFT_HOST_OK;
}

void raft_host_text_free(raft_host_text *t) { delete t; }

int raft_host_paf_load(const char *path, const raft_host_reads *reads, raft_host_paf **out, char *err_name, int err_cap)
{
    if (!path || !reads || !out) return RAFT_HOST_ERR_ARG;
    *out = nullptr;
    raft_host_text *text = nullptr;
    int rc = raft_host_text_read(path, &text);
    if (rc == RAFT_HOST_OK) rc = raft_host_paf_parse(text, reads, out, err_name, err_cap);
    raft_host_text_free(text);
    return rc;
}

int raft_host_paf_parse(raft_host_text *text, const raft_host_reads *reads, raft_host_paf **out, char *err_name, int err_cap)
{
    if (!text || !reads || !out) return RAFT_HOST_ERR_ARG;
    *out = nullptr;
    const size_t data_n = text->n;
    char *const data = text->buf.get();     // (tokenised in place: the text is consumed)
    // Lines are independent: the buffer is cut at newlines into one chunk per thread, each chunk is tokenised into
    // its own columns (paf.hpp:50-87 rules), and the chunks are concatenated in file order.
    struct Chunk { size_t lines = 0, n = 0; size_t err_pos = (size_t)-1; std::string err_name; bool mirror = false; };
    const size_t total = data_n;
    auto num = [](const char *s) -> int32_t {
        // paf.hpp:64-75 -> chop.hpp:157-160: strtol, then uint32, then int.  Plain runs of up to 18 digits (every
        // real PAF field) take the short loop -- the value is the same, without glibc's locale and range machinery;
        // anything else (blanks, signs, overflow) goes to strtol itself.
        unsigned long long v = 0;
        int n = 0;
        while (s[n] >= '0' && s[n] <= '9' && n < 19) { v = v * 10 + (unsigned)(s[n] - '0'); ++n; }
        if (n == 0 || n == 19) return (int32_t)(uint32_t)strtol(s, nullptr, 10);
        return (int32_t)(uint32_t)v;       // trailing garbage ends the number, as in strtol
    };
    // Record 0 (the first accepted line), read ahead without touching the buffer: every worker compares its records
    // with it while they are in registers, which is the reference's symmetric-PAF detection (chop.hpp:171-184) at no
    // extra pass over the columns.  The engine is then told the flag and neither detects nor uploads target columns.
    bool have0 = false;
    size_t rec0_pos = 0;
    int32_t r0[6] = {0, 0, 0, 0, 0, 0};
    for (size_t pos = 0; pos < total && !have0;) {
        const char *line = data + pos;
        const char *nl = (const char *)memchr(line, '\n', total - pos);
        if (!nl) break;
        size_t len = (size_t)(nl - line);
        const size_t line_pos = pos;
        pos += len + 1;
        if (len > 1 && line[len - 1] == '\r') --len;
        std::string f[9];
        int nf = 0;
        size_t b = 0;
        for (size_t i = 0; i <= len; ++i) {
            if (i < len && line[i] != '\t') continue;
            if (nf < 9) f[nf].assign(line + b, i - b);
            ++nf;
            b = i + 1;
        }
        if (nf < 10) continue;
        const int32_t a = reads->names.find(f[0].data(), f[0].size()), t = reads->names.find(f[5].data(), f[5].size());
        if (a < 0 || t < 0) break;                       // the tokenising pass reports the unknown name
        r0[0] = a; r0[1] = num(f[2].c_str()); r0[2] = num(f[3].c_str()); r0[3] = t; r0[4] = num(f[7].c_str()); r0[5] = num(f[8].c_str());
        rec0_pos = line_pos;
        have0 = true;
    }
    int T = host_threads();
    if (total < (1u << 20)) T = 1;
    std::vector<size_t> cut((size_t)T + 1, total);
    cut[0] = 0;
    for (int t = 1; t < T; ++t) {
        size_t p0 = total / (size_t)T * (size_t)t;
        if (p0 < cut[t - 1]) p0 = cut[t - 1];
        const char *nl = p0 < total ? (const char *)memchr(data + p0, '\n', total - p0) : nullptr;
        cut[t] = nl ? (size_t)(nl - data) + 1 : total;
    }
    std::vector<Chunk> chunks((size_t)T);
    // The columns are allocated once, for as many records as there are lines, and every worker writes its records where they
    // belong: a pass that only counts newlines tells where that is.  (Until round 6 every worker filled vectors of its own, which
    // were then copied together: twice the pages touched, 24 more bytes moved per record.)  Lines that are not records -- fewer
    // than ten fields, paf.hpp:84-85 -- leave gaps, closed afterwards; a PAF has none.
    parallel_for(T, [&](int t) {
        size_t c = 0, i = cut[t];
        const size_t end = cut[t + 1];
        const __m128i vn = _mm_set1_epi8('\n');
        for (; i + 16 <= end; i += 16)
            c += (size_t)__builtin_popcount((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(data + i)), vn)));
        for (; i < end; ++i) c += data[i] == '\n';
        chunks[(size_t)t].lines = c;
    });
    std::vector<size_t> off((size_t)T + 1, 0);
    for (int t = 0; t < T; ++t) off[(size_t)t + 1] = off[(size_t)t] + chunks[(size_t)t].lines;
    std::unique_ptr<raft_host_paf> P(new raft_host_paf());
    for (int k = 0; k < 6; ++k) P->col[k].reset(new int32_t[off[(size_t)T] ? off[(size_t)T] : 1]);
    parallel_for(T, [&](int t) {
        Chunk &C = chunks[(size_t)t];
        int32_t *const out_col[6] = {P->col[0].get() + off[(size_t)t], P->col[1].get() + off[(size_t)t], P->col[2].get() + off[(size_t)t],
                                     P->col[3].get() + off[(size_t)t], P->col[4].get() + off[(size_t)t], P->col[5].get() + off[(size_t)t]};
        const char *lastq = nullptr, *lastt = nullptr;
        size_t lastq_n = 0, lastt_n = 0;
        int32_t lastq_id = -1, lastt_id = -1;
        auto resolve = [&](char *s, size_t n, const char *&cs, size_t &cn, int32_t &cid) -> int32_t {
            if (cs && cn == n && memcmp(cs, s, n) == 0) return cid;
            const int32_t id = reads->names.find(s, n);
            if (id >= 0) { cs = s; cn = n; cid = id; }
            return id;
        };
        const size_t pos0 = cut[t];
        const size_t end = cut[t + 1];
        // One pass over the bytes finds tabs and newlines sixteen at a time (SSE2: every x86-64 has it); a line is handed on when
        // its newline turns up, as the positions of its first ten tabs.  Split on TAB only (paf.hpp:56-58); nothing is written
        // into the text.  (Until round 6: memchr for the newline, then a byte loop for the tabs -- 80 of the tokeniser's 230 ns
        // per record.)
        auto field_num = [&](size_t b, size_t e) -> int32_t {
            // one to eight digits and nothing else -- every coordinate of a real PAF: eight bytes at once, the bytes before the
            // field replaced by '0'
            const size_t L = e - b;
            if (L - 1 < 8 && e >= 8) {
                unsigned long long w;
                memcpy(&w, data + e - 8, 8);
                const unsigned long long low = L == 8 ? 0ull : ((1ull << ((8 - L) * 8)) - 1ull);
                w = (w & ~low) | (0x3030303030303030ull & low);
                const unsigned long long d = w - 0x3030303030303030ull;
                if (((d | (d + 0x7676767676767676ull) | w) & 0x8080808080808080ull) == 0ull) {
                    unsigned long long x = (d * 2561ull) >> 8;
                    x = ((x & 0x00ff00ff00ff00ffull) * 6553601ull) >> 16;
                    return (int32_t)(uint32_t)(((x & 0x0000ffff0000ffffull) * 42949672960001ull) >> 32);
                }
            }
            unsigned long long v = 0;
            size_t i = b;
            while (i < e && i - b < 19 && data[i] >= '0' && data[i] <= '9') { v = v * 10 + (unsigned)(data[i] - '0'); ++i; }
            if (i == b || i - b == 19) return num(std::string(data + b, e - b).c_str());       // blanks, signs, overflow: strtol's rules
            return (int32_t)(uint32_t)v;       // trailing garbage ends the number, as in strtol
        };
        size_t line_start = pos0, tab[10];
        int nt = 0;
        bool stop = false;
        // Lines wait in groups of eight for their target names: the hashes first (each fetching its slot of the name table), then the
        // slots (each fetching its name's bytes), then the lines in file order -- two cache misses per group where there were two per
        // line.  The query name is the line before's nearly always (a PAF grouped by query).
        struct Pending { size_t ls, tab[9]; unsigned long long hv; };
        constexpr int kGroup = 8;
        Pending pend[kGroup];
        int np = 0;
        auto flush = [&]() {
            for (int k = 0; k < np; ++k) {
                Pending &L = pend[k];
                L.hv = NameTable::hash(data + L.tab[4] + 1, L.tab[5] - L.tab[4] - 1);
                reads->names.prefetch_slot(L.hv);
            }
            for (int k = 0; k < np; ++k) reads->names.prefetch_name(pend[k].hv);
            for (int k = 0; k < np && !stop; ++k) {
                const Pending &L = pend[k];
                char *const q = data + L.ls, *const tn = data + L.tab[4] + 1;
                const size_t qn = L.tab[0] - L.ls, tnn = L.tab[5] - L.tab[4] - 1;
                const int32_t a = resolve(q, qn, lastq, lastq_n, lastq_id);
                int32_t b = lastt_id;
                if (!(lastt && lastt_n == tnn && memcmp(lastt, tn, tnn) == 0)) {
                    b = reads->names.find_hashed(L.hv, tn, tnn);
                    if (b >= 0) { lastt = tn; lastt_n = tnn; lastt_id = b; }
                }
                if (a < 0 || b < 0) { C.err_pos = L.ls; C.err_name = a < 0 ? std::string(q, qn) : std::string(tn, tnn); stop = true; break; }
                const int32_t v_qs = field_num(L.tab[1] + 1, L.tab[2]), v_qe = field_num(L.tab[2] + 1, L.tab[3]);
                const int32_t v_ts = field_num(L.tab[6] + 1, L.tab[7]), v_te = field_num(L.tab[7] + 1, L.tab[8]);
                const size_t at = C.n++;
                out_col[0][at] = a; out_col[1][at] = v_qs; out_col[2][at] = v_qe;
                out_col[3][at] = b; out_col[4][at] = v_ts; out_col[5][at] = v_te;
                if (have0 && a == r0[3] && b == r0[0] && v_ts == r0[1] && v_te == r0[2] && v_qs == r0[4] && v_qe == r0[5] && L.ls != rec0_pos)
                    C.mirror = true;
            }
            np = 0;
        };
        auto line_done = [&](size_t nl) {
            const size_t ls = line_start;
            const int ntabs = nt;
            line_start = nl + 1; nt = 0;
            if (ntabs + 1 < 10) return;                   // paf.hpp:84-85: silently skipped
            Pending &L = pend[np++];
            L.ls = ls;
            for (int k = 0; k < 9; ++k) L.tab[k] = tab[k];
            if (np == kGroup) flush();
        };
        auto delim = [&](size_t p) {
            if (data[p] == '\t') { if (nt < 10) tab[nt] = p; ++nt; }
            else line_done(p);
        };
        size_t i = pos0;
        const __m128i vt = _mm_set1_epi8('\t'), vn = _mm_set1_epi8('\n');
        for (; i + 16 <= end && !stop; i += 16) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(data + i));
            unsigned m = (unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(v, vt), _mm_cmpeq_epi8(v, vn)));
            while (m && !stop) {
                delim(i + (size_t)__builtin_ctz(m));
                m &= m - 1;
            }
        }
        for (; i < end && !stop; ++i)
            if (data[i] == '\t' || data[i] == '\n') delim(i);
        if (!stop) flush();
    });
    int rc = RAFT_HOST_OK;
    for (const Chunk &C : chunks)                         // the first offending line in file order wins
        if (C.err_pos != (size_t)-1) {
            rc = RAFT_HOST_ERR_UNKNOWN_NAME;
            if (err_name && err_cap > 0) snprintf(err_name, (size_t)err_cap, "%s", C.err_name.c_str());
            break;
        }
    if (rc != RAFT_HOST_OK) return rc;
    for (const Chunk &C : chunks) if (C.mirror) P->symmetric = 1;
    size_t n_rec = 0;
    for (int t = 0; t < T; ++t) {                         // (gaps: lines that were no records)
        const Chunk &C = chunks[(size_t)t];
        if (n_rec != off[(size_t)t] && C.n)
            for (int k = 0; k < 6; ++k) memmove(P->col[k].get() + n_rec, P->col[k].get() + off[(size_t)t], C.n * sizeof(int32_t));
        n_rec += C.n;
    }
    P->n = n_rec;
    *out = P.release();
    return RAFT_HOST_OK;
}

// The grouped form of a tokenised query column (include/raft_hip.h raft_hip_run_device_grouped): hifiasm writes its PAF
// grouped by query (reference README.md:36-38), so the column is a handful of runs sorted by read id.  Pass 1 finds the
// places where the id steps back (and ids outside [0, n_reads)); pass 2 turns every run into its per-read offsets by a
// linear merge of "read r" against the run, a range of reads per worker.
int raft_host_group_offsets(int32_t n_reads, int64_t n_rec, const int32_t *qid, int32_t max_runs, int32_t *n_runs, int64_t *rec_offset)
{
    if (!n_runs || n_reads < 0 || n_rec < 0 || max_runs < 1 || (n_rec > 0 && !qid) || !rec_offset) return RAFT_HOST_ERR_ARG;
    *n_runs = 0;
    int T = host_threads();
    if (n_rec < (1 << 20)) T = 1;
    std::vector<std::vector<int64_t>> desc((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    parallel_for(T, [&](int t) {
        const int64_t lo = n_rec * t / T, hi = n_rec * (t + 1) / T;
        int32_t prev = lo > 0 ? qid[lo - 1] : INT32_MIN;
        for (int64_t i = lo; i < hi; ++i) {
            const int32_t q = qid[i];
            if (q < prev && (int)desc[(size_t)t].size() <= max_runs) desc[(size_t)t].push_back(i);
            if ((uint32_t)q >= (uint32_t)n_reads) bad[(size_t)t] = 1;
            prev = q;
        }
    });
    std::vector<int64_t> start{0};
    for (int t = 0; t < T; ++t) {
        if (bad[(size_t)t]) return RAFT_HOST_OK;          // an id outside the reads: not grouped (the plain pass reports it)
        for (int64_t d : desc[(size_t)t]) start.push_back(d);
    }
    if ((int64_t)start.size() > max_runs) return RAFT_HOST_OK;   // more runs than the engine takes: not grouped
    const int R = (int)start.size();
    start.push_back(n_rec);
    const int64_t stride = (int64_t)n_reads + 1;
    const int Tr = n_reads < (1 << 16) ? 1 : T;
    parallel_for(Tr, [&](int t) {
        const int64_t r_lo = stride * t / Tr, r_hi = stride * (t + 1) / Tr;     // entries [r_lo, r_hi) of every run
        for (int k = 0; k < R; ++k) {
            const int64_t a = start[(size_t)k], b = start[(size_t)k + 1];
            int64_t pos = std::lower_bound(qid + a, qid + b, (int32_t)std::min<int64_t>(r_lo, INT32_MAX)) - qid;
            int64_t *o = rec_offset + (int64_t)k * stride;
            for (int64_t r = r_lo; r < r_hi; ++r) {
                while (pos < b && qid[pos] < r) ++pos;
                o[r] = pos;
            }
        }
    });
    *n_runs = R;
    return RAFT_HOST_OK;
}

int raft_host_pack_windows(int64_t n_rec, const int32_t *qs, const int32_t *qe, int32_t reso, uint32_t *win, int64_t *bad_index)
{
    if (bad_index) *bad_index = -1;
    if (n_rec < 0 || reso < 1 || reso > 32767 || (n_rec > 0 && (!qs || !qe || !win))) return RAFT_HOST_ERR_ARG;
    int T = host_threads();
    if (n_rec < (1 << 18)) T = 1;
    std::vector<int64_t> neg((size_t)T, INT64_MAX), far((size_t)T, INT64_MAX);
    const uint32_t d = (uint32_t)reso;
    parallel_for(T, [&](int t) {
        const int64_t lo = n_rec * t / T, hi = n_rec * (t + 1) / T;
        int64_t first_neg = INT64_MAX, first_far = INT64_MAX;
        for (int64_t i = lo; i < hi; ++i) {
            const int32_t s = qs[i], e = qe[i];
            if ((s | e) < 0) { first_neg = std::min(first_neg, i); win[i] = 0; continue; }
            const uint32_t first = (uint32_t)s / d, last1 = e > 0 ? (uint32_t)(e - 1) / d + 1u : 0u;
            uint32_t w = 0;
            if (last1 > first) {
                if (last1 > 65535u) first_far = std::min(first_far, i);
                w = first | (last1 << 16);
            }
            win[i] = w;
        }
        neg[(size_t)t] = first_neg; far[(size_t)t] = first_far;
    });
    int64_t n0 = INT64_MAX, f0 = INT64_MAX;
    for (int t = 0; t < T; ++t) { n0 = std::min(n0, neg[(size_t)t]); f0 = std::min(f0, far[(size_t)t]); }
    if (n0 != INT64_MAX) { if (bad_index) *bad_index = n0; return RAFT_HOST_ERR_COORD; }
    if (f0 != INT64_MAX) { if (bad_index) *bad_index = f0; return RAFT_HOST_ERR_RANGE; }
    return RAFT_HOST_OK;
}

void raft_host_paf_free(raft_host_paf *p) { delete p; }
int64_t raft_host_paf_count(const raft_host_paf *p) { return p ? (int64_t)p->n : 0; }
const int32_t *raft_host_paf_column(const raft_host_paf *p, int k) { return (p && k >= 0 && k < 6) ? p->col[k].get() : nullptr; }
int raft_host_paf_symmetric(const raft_host_paf *p) { return p ? p->symmetric : 0; }

// ---- the four-bit step encoding of the coverage array (include/raft_hip.h "delta4") ----
namespace {
struct D4Cursor {                 // walks the encoding forward from a block anchor
    const uint8_t *nib; const int32_t *anchor; const int64_t *exc_index; const int32_t *exc_value; int64_t n_exc;
    int64_t w = 0, x = 0;         // next window; next exception
    long long v = 0;
    void seek(int64_t to)         // position before window `to`: v = cov[to - 1]
    {
        const int64_t k = to >> 10;
        w = k << 10; v = anchor[k];
        x = n_exc ? std::lower_bound(exc_index, exc_index + n_exc, w) - exc_index : 0;
        while (w < to) next();
    }
    inline long long next()
    {
        const unsigned c = (nib[w >> 1] >> (4 * (w & 1))) & 15u;
        if (c) v += (long long)c - 8;
        else { v = (x < n_exc && exc_index[x] == w) ? exc_value[x] : -1; ++x; }     // (-1: an escape without its entry -- checked by the unpacker)
        ++w;
        return v;
    }
};
}

int raft_host_unpack_coverage_d4(int64_t n_bins, const uint8_t *nib, const int32_t *anchor, int64_t n_exc, const int64_t *exc_index,
                                 const int32_t *exc_value, int32_t *cov)
{
    if (n_bins < 0 || n_exc < 0 || (n_bins && (!nib || !anchor || !cov)) || (n_exc && (!exc_index || !exc_value))) return RAFT_HOST_ERR_ARG;
    for (int64_t k = 0; k < n_exc; ++k)
        if (exc_index[k] < 0 || exc_index[k] >= n_bins || (k && exc_index[k] <= exc_index[k - 1]) ||
            ((nib[exc_index[k] >> 1] >> (4 * (exc_index[k] & 1))) & 15u) != 0u) return RAFT_HOST_ERR_ARG;
    const int64_t n_blocks = (n_bins + 1023) >> 10;
    const int T_ = n_bins < (1 << 22) ? 1 : host_threads();
    std::vector<int> bad((size_t)T_, 0);
    parallel_for(T_, [&](int t) {
        const int64_t b0 = n_blocks * t / T_, b1 = n_blocks * (t + 1) / T_;
        D4Cursor c{nib, anchor, exc_index, exc_value, n_exc};
        for (int64_t b = b0; b < b1; ++b) {
            c.seek(b << 10);
            const int64_t hi = std::min<int64_t>(n_bins, (b + 1) << 10);
            while (c.w < hi) { const int64_t at = c.w; const long long v = c.next(); if (v < 0) bad[(size_t)t] = 1; cov[at] = (int32_t)v; }
        }
    });
    for (int b : bad) if (b) return RAFT_HOST_ERR_ARG;
    return RAFT_HOST_OK;
}

int raft_host_write_coverage_d4(const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset, const uint8_t *nib,
                                const int32_t *anchor, int64_t n_exc, const int64_t *exc_index, const int32_t *exc_value)
{
    return write_ordered(path, n_reads, 1 << 20,
                         [&](long long i) { return (long long)(cov_offset[i + 1] - cov_offset[i]) + 4; },
                         [&](long long i, std::string &o) {
                             o.append("read ", 5); put_num(o, i); o.push_back(' ');
                             const int64_t b = cov_offset[i], e = cov_offset[i + 1];
                             D4Cursor c{nib, anchor, exc_index, exc_value, n_exc};
                             if (e > b) c.seek(b);
                             for (int64_t j = b; j < e; ++j) {
                                 const long long v = c.next();
                                 put_num(o, (long long)(j - b) * reso); o.push_back(','); put_num(o, v); o.push_back(' ');
                             }
                             o.push_back('\n');
                         });
}

int raft_host_unpack_coverage_w(int32_t width, int64_t n_bins, const void *cov_packed, int64_t n_exc, const int64_t *exc_index,
                                const int32_t *exc_value, int32_t *cov)
{
    if (width == 1) return unpack_coverage_t(n_bins, static_cast<const uint8_t *>(cov_packed), n_exc, exc_index, exc_value, cov);
    if (width == 2) return unpack_coverage_t(n_bins, static_cast<const uint16_t *>(cov_packed), n_exc, exc_index, exc_value, cov);
    return RAFT_HOST_ERR_ARG;
}

int raft_host_write_coverage_packed_w(int32_t width, const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset,
                                      const void *cov_packed, int64_t n_exc, const int64_t *exc_index, const int32_t *exc_value)
{
    if (width == 1) return write_coverage_packed_t(path, n_reads, reso, cov_offset, static_cast<const uint8_t *>(cov_packed), n_exc, exc_index, exc_value);
    if (width == 2) return write_coverage_packed_t(path, n_reads, reso, cov_offset, static_cast<const uint16_t *>(cov_packed), n_exc, exc_index, exc_value);
    return RAFT_HOST_ERR_ARG;
}

int raft_host_unpack_coverage(int64_t n_bins, const uint8_t *cov8, int64_t n_exc, const int64_t *exc_index,
                              const int32_t *exc_value, int32_t *cov)
{
    return raft_host_unpack_coverage_w(1, n_bins, cov8, n_exc, exc_index, exc_value, cov);
}

int raft_host_write_coverage_packed(const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset,
                                    const uint8_t *cov8, int64_t n_exc, const int64_t *exc_index, const int32_t *exc_value)
{
    return raft_host_write_coverage_packed_w(1, path, n_reads, reso, cov_offset, cov8, n_exc, exc_index, exc_value);
}

// repeat.hpp:105-108: "read <i> " then "<pos>,<cov> " per window, then newline
int raft_host_write_coverage(const char *path, int32_t n_reads, int32_t reso, const int64_t *cov_offset, const int32_t *cov)
{
    return write_ordered(path, n_reads, 1 << 20,
                         [&](long long i) { return (long long)(cov_offset[i + 1] - cov_offset[i]) + 4; },
                         [&](long long i, std::string &o) {
                             o.append("read ", 5); put_num(o, i); o.push_back(' ');
                             const int64_t b = cov_offset[i], e = cov_offset[i + 1];
                             for (int64_t j = b; j < e; ++j) { put_num(o, (long long)(j - b) * reso); o.push_back(','); put_num(o, cov[j]); o.push_back(' '); }
                             o.push_back('\n');
                         });
}

// repeat.hpp:180-203: every read gets a line "read <i>, " + "<s>,<e>    " per repeat; .bed only in simulated mode
int raft_host_write_repeats(const char *txt_path, const char *bed_path, const raft_host_reads *reads,
                            const int64_t *rep_offset, const int32_t *rep_s, const int32_t *rep_e)
{
    Out t(txt_path), b(bed_path);
    if (!t.ok() || !b.ok()) return RAFT_HOST_ERR_IO;
    const int32_t n = (int32_t)reads->lens.size();
    for (int32_t i = 0; i < n; ++i) {
        t.str("read ", 5); t.num(i); t.str(", ", 2);
        for (int64_t j = rep_offset[i]; j < rep_offset[i + 1]; ++j) {
            t.num(rep_s[j]); t.ch(','); t.num(rep_e[j]); t.str("    ", 4);
            if (!reads->real_reads) {
                const std::string &al = reads->align[i];
                if (al == "forward") {
                    b.str(reads->chr[i].data(), reads->chr[i].size()); b.ch('\t'); b.num(reads->start_pos[i] + rep_s[j]);
                    b.ch('\t'); b.num(reads->start_pos[i] + rep_e[j]); b.ch('\n');
                } else if (al == "reverse") {
                    b.str(reads->chr[i].data(), reads->chr[i].size()); b.ch('\t'); b.num(reads->end_pos[i] - rep_e[j]);
                    b.ch('\t'); b.num(reads->end_pos[i] - rep_s[j]); b.ch('\n');
                }
            }
        }
        t.ch('\n');
    }
    const bool ok1 = t.close(), ok2 = b.close();
    return (ok1 && ok2) ? RAFT_HOST_OK : RAFT_HOST_ERR_IO;
}

// chop.hpp:250-322: one FASTA record per fragment, read_num counts from 1 over the whole file
int raft_host_write_fasta(const char *path, const raft_host_reads *reads, const int64_t *frag_offset,
                          const int32_t *frag_begin, const int32_t *frag_end)
{
    const int32_t n = (int32_t)reads->lens.size();
    // the header line of fragment f of read i (empty: chop.hpp:293-311 writes none for that orientation)
    auto one_header = [&](long long i, int64_t f, std::string &o) {
        const char *name = reads->names.name((int32_t)i);
        const size_t name_n = reads->names.name_len((int32_t)i);
        const int64_t f0 = frag_offset[i], f1 = frag_offset[i + 1];
        const bool whole = (f1 - f0) == 1;             // kept in one piece (chop.hpp:250-267)
        const long long read_num = f + 1;
        const int32_t b = frag_begin[f], e = frag_end[f];
        if (reads->real_reads) {
            o.append(">read=", 6); put_num(o, read_num); o.push_back(','); o.append(name, name_n);
            o.append(",pos_on_original_read=", 22); put_num(o, b); o.push_back('-'); put_num(o, e); o.push_back('\n');
        } else {
            // tail = name.substr(name.find_last_of(','))  -> ",<contig>"
            const std::string nm(name, name_n);
            const size_t lc = nm.find_last_of(',');
            const std::string tail = lc == std::string::npos ? std::string() : nm.substr(lc);
            const std::string &al = reads->align[(size_t)i];
            const int32_t sp = reads->start_pos[(size_t)i], ep = reads->end_pos[(size_t)i];
            bool header = true;
            long long p0 = 0, p1 = 0, ln = 0;
            if (whole) { p0 = sp; p1 = ep; ln = reads->lens[(size_t)i]; }
            else if (al == "forward") { p0 = (long long)sp + b; p1 = (long long)sp + e; ln = e - b; }
            else if (al == "reverse") { p0 = (long long)ep - e; p1 = (long long)ep - b; ln = e - b; }
            else header = false;                     // chop.hpp:293-311 writes no header for other orientations
            if (header) {
                o.append(">read=", 6); put_num(o, read_num); o.push_back(','); o.append(al); o.append(",position=", 10);
                put_num(o, p0); o.push_back('-'); put_num(o, p1); o.append(",length=", 8); put_num(o, ln); o.append(tail); o.push_back('\n');
            }
        }
    };
    auto one_read = [&](long long i, std::string &o) {
        const char *seq = reads->base_ptr() + reads->base_off[(size_t)i];
        for (int64_t f = frag_offset[i]; f < frag_offset[i + 1]; ++f) {
            one_header(i, f, o);
            o.append(seq + frag_begin[f], (size_t)(frag_end[f] - frag_begin[f]));
            o.push_back('\n');
        }
    };
    // (Measured and dropped, round 5: the file sized first, created at that size, mapped, and every worker formatting its blocks
    // straight into the mapping -- one copy of every base instead of two.  On the box's tmpfs the workers' page faults serialise
    // on the file's mapping and drag the table writers that run beside them down with them: write_fasta 1.36 -> 3.0 s,
    // write_tables 0.76 -> 6.9 s (profiles/r05_cli_s500k_mapped_writer.txt).  The single write() per block stays.)
    return write_ordered(path, n, 8 << 20, [&](long long i) { return (long long)reads->lens[(size_t)i] + 64; }, one_read);
}

// split_naive.cpp:10-44: every read is cut into consecutive pieces of split_len bases, no overlaps, written as
// ">name_k\n<piece>\n" with k counting from 1; a read without bases writes nothing.
int raft_host_split_naive(const char *in_path, const char *out_path, int32_t split_len, int32_t *n_reads_out)
{
    if (!in_path || !out_path || split_len <= 0) return RAFT_HOST_ERR_ARG;
    raft_host_reads *R = nullptr;
    const int rc = load_reads(in_path, &R, true);
    if (rc != RAFT_HOST_OK) return rc;
    const long long n = (long long)R->lens.size();
    if (n_reads_out) *n_reads_out = (int32_t)n;
    const int wrc = write_ordered(out_path, n, 8 << 20, [&](long long i) { return (long long)R->lens[(size_t)i] + 64; },
                                  [&](long long i, std::string &o) {
                                      const std::string &name = R->dup_names[(size_t)i];
                                      const char *seq = R->base_ptr() + R->base_off[(size_t)i];
                                      const long long len = R->lens[(size_t)i];
                                      long long piece = 1;
                                      for (long long b = 0; b < len; b += split_len, ++piece) {
                                          o.push_back('>'); o.append(name); o.push_back('_'); put_num(o, piece); o.push_back('\n');
                                          o.append(seq + b, (size_t)std::min<long long>(split_len, len - b));
                                          o.push_back('\n');
                                      }
                                  });
    delete R;
    return wrc;
}

} // extern "C"
