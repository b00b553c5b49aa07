// Native proto3 scanner/emitter for the solver boundary + match-graph construction.
//   MatchingFile  types.proto:3-28   -> lfr::Graph           (replaces solve.cc:426-481)
//   SolutionFile  types.proto:30-46  <- positions            (replaces solve.cc:644-679)
// No libprotobuf: only varint / fixed32 / length-delimited fields occur in these messages.
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <atomic>
#include <memory>
#include <new>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <set>
#include <thread>
#include <cstdlib>
#include <algorithm>

#include "lfr_internal.hpp"

namespace lfr {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------- graph building
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

int32_t Graph::intern_image(const std::string &name, float fact) {
    auto it = image_index.find(name);
    if (it != image_index.end()) return it->second;      // first fact wins (solve.cc:449,451)
    const int32_t idx = (int32_t)image_names.size();
    image_index.emplace(name, idx);
    image_names.push_back(name);
    image_fact.push_back(fact);
    return idx;
}

uint32_t Graph::find_or_create_node(int32_t image, uint32_t feature) {
    if ((uint64_t)(hcount + 1) * 2 > hkeys.size()) {      // grow / first use
        const uint64_t ncap = hkeys.empty() ? (1u << 16) : hkeys.size() * 2;
        std::vector<uint64_t> nk(ncap);
        std::vector<int64_t> nv(ncap, -1);
        const uint64_t nmask = ncap - 1;
        for (uint64_t i = 0; i < hkeys.size(); ++i)
            if (hvals[i] >= 0) {
                uint64_t h = mix64(hkeys[i]) & nmask;
                while (nv[h] >= 0) h = (h + 1) & nmask;
                nk[h] = hkeys[i]; nv[h] = hvals[i];
            }
        hkeys.swap(nk); hvals.swap(nv); hmask = nmask;
    }
    const uint64_t key = ((uint64_t)(uint32_t)image << 32) | feature;
    uint64_t h = mix64(key) & hmask;
    for (;;) {
        if (hvals[h] < 0) {
            hkeys[h] = key; hvals[h] = (int64_t)node_image.size(); ++hcount;
            node_image.push_back(image); node_feat.push_back(feature);
            return (uint32_t)hvals[h];
        }
        if (hkeys[h] == key) return (uint32_t)hvals[h];
        h = (h + 1) & hmask;
    }
}

void Graph::add_match(int32_t img1, int32_t img2, uint32_t f1, uint32_t f2, float sim, const float *d1, int n1,
                      const float *d2, int n2) {
    const uint32_t a = find_or_create_node(img1, f1);     // node1 before node2 (solve.cc:474-475)
    const uint32_t b = find_or_create_node(img2, f2);
    m_node1.push_back(a); m_node2.push_back(b); m_sim.push_back(sim);
    const size_t o = m_disp1.size();
    m_disp1.resize(o + 18); m_disp2.resize(o + 18);            // zero filled
    if (n1 > 0) memcpy(&m_disp1[o], d1, sizeof(float) * 2 * (size_t)n1);
    if (n2 > 0) memcpy(&m_disp2[o], d2, sizeof(float) * 2 * (size_t)n2);
}

void Graph::finish() {
    std::vector<uint64_t>().swap(hkeys);
    std::vector<int64_t>().swap(hvals);
    // nodes grouped by image, in node order (lfr_apply_displacements; built here so that the handle is
    // read-only - and therefore thread safe - afterwards)
    const size_t ni = image_names.size();
    img_off.assign(ni + 1, 0);
    for (int32_t im : node_image) ++img_off[im + 1];
    for (size_t i = 0; i < ni; ++i) img_off[i + 1] += img_off[i];
    img_nodes.resize(node_image.size());
    std::vector<int64_t> cur(img_off.begin(), img_off.end() - 1);
    for (size_t n = 0; n < node_image.size(); ++n) img_nodes[cur[node_image[n]]++] = (uint32_t)n;
    // exponent range of the similarities (see sims_sum_exactly); integer compares on the exponent fields, ~1 ms per 10^7 matches
    {
        uint32_t emin = 0xffu, emax = 0u;
        bool odd = false;
        const float *sp = m_sim.data();
        const size_t M = m_sim.size();
        for (size_t m = 0; m < M; ++m) {
            uint32_t bits; memcpy(&bits, sp + m, 4);
            const uint32_t e = (bits >> 23) & 0xffu;
            if (e == 0xffu) odd = true;                      // inf / nan
            else if (e != 0u || (bits & 0x7fffffu)) { emin = std::min(emin, e); emax = std::max(emax, e); }   // (zeros add nothing)
        }
        sims_sum_exactly = !odd && (emax < emin || emax - emin <= 10u);
        // ... and the NUMBER of terms of one sum: n float32 values whose exponents span s binades add exactly in fp64 while
        // 24 + s + log2(n) <= 53.  A root score sums one node's matches (<= max degree), a meta-edge weight the matches between two
        // tracks (<= #images nodes per track, each with <= max degree matches).  The reference keeps duplicated matches
        // (solve.cc:476-478), so the degree is unbounded in principle: count it.
        if (sims_sum_exactly && M > 0) {
            std::vector<uint32_t> deg(node_image.size(), 0);
            for (size_t m = 0; m < M; ++m) { ++deg[m_node1[m]]; ++deg[m_node2[m]]; }
            uint64_t dmax = 0;
            for (uint32_t d : deg) dmax = std::max<uint64_t>(dmax, d);
            const uint32_t spread = emax < emin ? 0u : emax - emin;
            const uint64_t terms = std::min<uint64_t>(M, dmax * std::max<uint64_t>(1, image_names.size()));   // (no sum has more terms than there are matches)
            sims_sum_exactly = terms <= ((uint64_t)1 << (29u - spread));
            if (!sims_sum_exactly && (getenv("LFR_VERBOSE") || getenv("LFR_TIMING")))
                fprintf(stderr, "lfr: similarity sums may round (up to %llu terms over %u binades): the graph stage will run on the host\n", (unsigned long long)terms, spread);
        } else if (!sims_sum_exactly && (getenv("LFR_VERBOSE") || getenv("LFR_TIMING"))) {
            fprintf(stderr, "lfr: similarities %s: the graph stage will run on the host\n", odd ? "hold inf / nan" : "span more than 10 binades");
        }
    }
}

// ---------------------------------------------------------------------------- wire primitives
struct Cursor {
    const uint8_t *p, *end;
    bool ok = true;
    bool done() const { return p >= end; }
    uint64_t varint() {
        uint64_t r = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) { ok = false; return 0; }
            const uint8_t b = *p++;
            r |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return r;
        }
        ok = false;
        return 0;
    }
    float f32() {
        if (end - p < 4) { ok = false; return 0.f; }
        float v; memcpy(&v, p, 4); p += 4; return v;
    }
    Cursor sub() {
        const uint64_t n = varint();
        if (!ok || (uint64_t)(end - p) < n) { ok = false; return Cursor{p, p}; }
        Cursor c{p, p + n}; p += n; return c;
    }
    // an unknown field (protobuf skips them; so does the reference's generated parser).  A group (wire type 3) runs to the END_GROUP
    // tag (wire type 4) of the same field number, nested fields skipped on the way; a stray END_GROUP or wire types 6/7 are malformed.
    void skip(int wt, int field, int depth = 0) {
        switch (wt) {
            case 0: varint(); break;
            case 1: if (end - p < 8) ok = false; else p += 8; break;
            case 2: sub(); break;
            case 3:
                if (depth >= 64) { ok = false; break; }
                for (;;) {
                    if (p >= end) { ok = false; break; }
                    const uint64_t key = varint();
                    if (!ok) break;
                    const int f = (int)(key >> 3), w = (int)(key & 7);
                    if (f == 0) { ok = false; break; }
                    if (w == 4) { if (f != field) ok = false; break; }
                    skip(w, f, depth + 1);
                    if (!ok) break;
                }
                break;
            case 5: if (end - p < 4) ok = false; else p += 4; break;
            default: ok = false;      // stray END_GROUP / invalid wire types
        }
    }
};

static bool parse_displacements(Cursor c, float *out, int &n) {   // one Displacement message
    float di = 0.f, dj = 0.f;
    while (!c.done() && c.ok) {
        const uint64_t key = c.varint();
        const int field = (int)(key >> 3), wt = (int)(key & 7);
        if (field == 0) return false;
        if (field == 1 && wt == 5) di = c.f32();
        else if (field == 2 && wt == 5) dj = c.f32();
        else c.skip(wt, field);
    }
    if (!c.ok) return false;
    if (n >= 9) { n = 10; return true; }        // > 9 grid points: the reference overruns its buffer
    out[2 * n] = di; out[2 * n + 1] = dj; ++n;
    return true;
}

// ---- parallel scanner -----------------------------------------------------------------------
// Pass A (sequential): split the top level into ImagePair cursors.  Pass B (one thread per chunk
// of pairs, balanced by bytes): decode pairs and matches into thread-local SoA buffers.  Pass C
// (sequential, order-dependent): banned-image filter, image interning (first fact wins), node
// numbering in order of first appearance (solve.cc:444-451,474-475).  Pass D (parallel): move the
// similarities and flow grids into the graph's arrays.
// Large scanner buffers ask for transparent huge pages (this image runs THP in `madvise` mode): 1.3 GB of decode buffers are 340 k
// first-touch faults in 4-KB pages, taken by 32 threads against the address-space lock that the HIP runtime's start-up on the side
// thread holds for its own mappings - the CLI's parse ran 2x slower than the same parse in a process without a GPU context.
template <class T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < ((size_t)4 << 20)) { if (void *p = malloc(bytes ? bytes : 1)) return (T *)p; throw std::bad_alloc(); }
        void *p = nullptr;
        const size_t len = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
        if (posix_memalign(&p, (size_t)2 << 20, len) != 0 || !p) throw std::bad_alloc();
        (void)madvise(p, len, MADV_HUGEPAGE);
        return (T *)p;
    }
    void deallocate(T *p, size_t) { free(p); }
    template <class U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <class T> using HugeVec = std::vector<T, HugeAlloc<T>>;

struct MatchBuf {
    HugeVec<uint32_t> f1, f2;
    HugeVec<float> sim, d1, d2;          // 18 floats per match, zero padded
    int rc = LFR_OK;
};
struct PairRec {
    const char *name1 = nullptr, *name2 = nullptr;
    uint32_t len1 = 0, len2 = 0;
    float fact1 = 0.f, fact2 = 0.f;
    int32_t buf = 0;                     // which MatchBuf
    int64_t first = 0, count = 0;        // its matches inside that buffer
    bool overflow = false;               // a match with more than 9 grid displacements: an error unless the pair is banned (the
                                         // reference `continue`s past banned pairs without reading their matches, solve.cc:444-446)
};

static int parse_match(Cursor c, MatchBuf &out, bool &overflow) {
    uint32_t f1 = 0, f2 = 0; float sim = 0.f;
    float d1[18], d2[18]; int n1 = 0, n2 = 0;
    memset(d1, 0, sizeof d1); memset(d2, 0, sizeof d2);
    while (!c.done() && c.ok) {
        const uint64_t key = c.varint();
        const int field = (int)(key >> 3), wt = (int)(key & 7);
        if (field == 0) return LFR_ERR_PARSE;
        if (field == 1 && wt == 0) f1 = (uint32_t)c.varint();
        else if (field == 2 && wt == 0) f2 = (uint32_t)c.varint();
        else if (field == 3 && wt == 5) sim = c.f32();
        else if (field == 4 && wt == 2) { if (!parse_displacements(c.sub(), d1, n1)) return LFR_ERR_PARSE; }
        else if (field == 5 && wt == 2) { if (!parse_displacements(c.sub(), d2, n2)) return LFR_ERR_PARSE; }
        else c.skip(wt, field);
    }
    if (!c.ok) return LFR_ERR_PARSE;
    if (n1 > 9 || n2 > 9) overflow = true;       // (the first 9 are kept; the caller rejects the file if the pair survives the banned filter)
    out.f1.push_back(f1); out.f2.push_back(f2); out.sim.push_back(sim);
    out.d1.insert(out.d1.end(), d1, d1 + 18); out.d2.insert(out.d2.end(), d2, d2 + 18);
    return LFR_OK;
}

static int parse_pair(Cursor c, PairRec &rec, MatchBuf &out) {
    rec.first = (int64_t)out.sim.size();
    while (!c.done() && c.ok) {
        const uint64_t key = c.varint();
        const int field = (int)(key >> 3), wt = (int)(key & 7);
        if (field == 0) return LFR_ERR_PARSE;
        if (field == 1 && wt == 2) { Cursor s = c.sub(); rec.name1 = (const char *)s.p; rec.len1 = (uint32_t)(s.end - s.p); }
        else if (field == 2 && wt == 5) rec.fact1 = c.f32();
        else if (field == 3 && wt == 2) { Cursor s = c.sub(); rec.name2 = (const char *)s.p; rec.len2 = (uint32_t)(s.end - s.p); }
        else if (field == 4 && wt == 5) rec.fact2 = c.f32();
        else if (field == 5 && wt == 2) {
            Cursor m = c.sub();
            if (!c.ok) return LFR_ERR_PARSE;
            const int rc = parse_match(m, out, rec.overflow);
            if (rc != LFR_OK) return rc;
        } else c.skip(wt, field);
    }
    if (!c.ok) return LFR_ERR_PARSE;
    rec.count = (int64_t)out.sim.size() - rec.first;
    return LFR_OK;
}

static int scan_threads() {
    if (const char *e = getenv("LFR_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return std::min(v, 256); }
    const unsigned h = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(h ? h : 1u, 32u));
}

static double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int parse_matching_buffer(const uint8_t *data, size_t size, Graph &g, const std::set<std::string> &banned) {
    const bool verbose = getenv("LFR_VERBOSE") != nullptr;
    const double t_a = wall_ms();
    // pass A
    std::vector<Cursor> pairs;
    {
        Cursor c{data, data + size};
        while (!c.done() && c.ok) {
            const uint64_t key = c.varint();
            const int field = (int)(key >> 3), wt = (int)(key & 7);
            if (!c.ok || field == 0) return LFR_ERR_PARSE;
            if (field == 1 && wt == 2) {
                Cursor pc = c.sub();
                if (!c.ok) return LFR_ERR_PARSE;
                pairs.push_back(pc);
            } else c.skip(wt, field);
        }
        if (!c.ok) return LFR_ERR_PARSE;
    }
    const int64_t P = (int64_t)pairs.size();
    if (P == 0) return LFR_OK;
    const double t_b = wall_ms();
    // pass B
    int T = (int)std::max<int64_t>(1, std::min<int64_t>(scan_threads(), (int64_t)(size >> 22) + 1));
    std::vector<int64_t> cuts(T + 1, P);
    cuts[0] = 0;
    for (int t = 1; t < T; ++t) {           // balance by bytes: first pair starting at/after the byte target
        const uint8_t *target = data + size / T * t;
        int64_t lo = cuts[t - 1], hi = P;
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (pairs[mid].p < target) lo = mid + 1; else hi = mid; }
        cuts[t] = lo;
    }
    std::vector<MatchBuf> bufs(T);
    std::vector<PairRec> recs(P);
    {
        auto work = [&](int t) {
            MatchBuf &b = bufs[t];
            for (int64_t i = cuts[t]; i < cuts[t + 1]; ++i) {
                recs[i].buf = t;
                const int rc = parse_pair(pairs[i], recs[i], b);
                if (rc != LFR_OK) { b.rc = rc; return; }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (auto &b : bufs) if (b.rc != LFR_OK) {
        if (b.rc == LFR_ERR_UNSUPPORTED)
            set_error("match with more than 9 grid displacements (the reference overflows flow_array, solve.cc:460-472)");
        return b.rc;
    }
    const double t_c = wall_ms();
    // pass C
    struct Job { int32_t buf; int64_t src, dst, n; };
    std::vector<Job> jobs;
    int64_t M = g.n_matches();
    {   // one (pinned) allocation per array and file instead of geometric growth
        int64_t add = 0;
        for (int64_t i = 0; i < P; ++i) add += recs[i].count;
        g.m_node1.reserve((size_t)(M + add)); g.m_node2.reserve((size_t)(M + add));
    }
    for (int64_t i = 0; i < P; ++i) {
        const PairRec &r = recs[i];
        const std::string name1(r.name1 ? r.name1 : "", r.len1), name2(r.name2 ? r.name2 : "", r.len2);
        if (banned.count(name1) || banned.count(name2)) continue;        // solve.cc:444-446
        if (r.overflow) {
            set_error("match with more than 9 grid displacements (the reference overflows flow_array, solve.cc:460-472)");
            return LFR_ERR_UNSUPPORTED;
        }
        const int32_t i1 = g.intern_image(name1, r.fact1);                // solve.cc:448-451
        const int32_t i2 = g.intern_image(name2, r.fact2);
        const MatchBuf &b = bufs[r.buf];
        for (int64_t m = r.first; m < r.first + r.count; ++m) {
            const uint32_t a = g.find_or_create_node(i1, b.f1[m]);        // node1 before node2 (solve.cc:474-475)
            const uint32_t c2 = g.find_or_create_node(i2, b.f2[m]);
            g.m_node1.push_back(a); g.m_node2.push_back(c2);
        }
        if (r.count) {
            if (!jobs.empty() && jobs.back().buf == r.buf && jobs.back().src + jobs.back().n == r.first) jobs.back().n += r.count;
            else jobs.push_back(Job{r.buf, r.first, M, r.count});
            M += r.count;
        }
    }
    const double t_d = wall_ms();
    // pass D
    g.m_sim.resize(M); g.m_disp1.resize(18 * M); g.m_disp2.resize(18 * M);
    const double t_d2 = wall_ms();
    {
        const int TJ = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, jobs.size()));
        auto work = [&](int t) {
            for (size_t j = t; j < jobs.size(); j += TJ) {
                const Job &jb = jobs[j];
                const MatchBuf &b = bufs[jb.buf];
                memcpy(&g.m_sim[jb.dst], &b.sim[jb.src], sizeof(float) * jb.n);
                memcpy(&g.m_disp1[18 * jb.dst], &b.d1[18 * jb.src], sizeof(float) * 18 * jb.n);
                memcpy(&g.m_disp2[18 * jb.dst], &b.d2[18 * jb.src], sizeof(float) * 18 * jb.n);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < TJ; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    if (verbose)
        fprintf(stderr, "lfr scanner: %zu bytes, %lld pairs, %d threads: split %.1f ms, decode %.1f ms, intern+number %.1f ms, alloc %.1f ms, move %.1f ms\n",
                size, (long long)P, T, t_b - t_a, t_c - t_b, t_d - t_c, t_d2 - t_d, wall_ms() - t_d2);
    return LFR_OK;
}


// ---- whole-input parallel scanner (all files of a matches set at once) ------------------------------------
// Every order-dependent rule of solve.cc:438-478 is a "first appearance" rule (image list, first fact wins, node
// ids: node1 before node2, match by match, file by file), and first appearance = minimum position, which is
// order-free to compute:
//   0  mmap all files, fault the pages in from T threads (the page-cache walk is the slowest sequential step otherwise)
//   A  per file: split the top level into ImagePair cursors            (sequential per file, files in parallel)
//   B  chunks of pairs (balanced by bytes): decode into thread-local SoA buffers, distinct image names per chunk in
//      order of local first appearance
//   C  merge the chunk name lists in chunk order -> global image ids, first facts, banned filter (a few thousand names)
//   D  per pair: global match offset (prefix sum over the kept pairs)
//   E  nodes: a concurrent hash map (image, feature) -> smallest position (2 * match + side), filled from all threads
//      with CAS / atomic-min; occupied slots sorted by position = node ids in order of first appearance; one more
//      parallel pass writes the endpoints of every match.  Meanwhile another thread allocates the (pinned) arrays
//      of similarities and flows and moves the decoded values in.
namespace {

struct NodeSlot { std::atomic<uint64_t> key; std::atomic<uint64_t> val; };      // key = (image << 32 | feature) + 1, 0 = empty

struct MappedFile {
    const uint8_t *data = nullptr;
    size_t size = 0;
    int fd = -1;
    ~MappedFile() {
        if (data && size) munmap((void *)data, size);
        if (fd >= 0) close(fd);
    }
};

template <class F>
void run_threads(int T, F f) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(f, t);
    f(0);
    for (auto &x : th) x.join();
}

}  // namespace

static int parse_all(const std::vector<std::string> &paths, Graph &g, const std::set<std::string> &banned) {
    const bool verbose = getenv("LFR_VERBOSE") != nullptr;
    const double t0 = wall_ms();
    const int T = scan_threads();
    // ---- 0: map + prefault
    std::vector<MappedFile> files(paths.size());
    size_t total_bytes = 0;
    for (size_t i = 0; i < paths.size(); ++i) {
        MappedFile &mf = files[i];
        mf.fd = open(paths[i].c_str(), O_RDONLY);
        if (mf.fd < 0) { set_error("cannot open %s", paths[i].c_str()); return LFR_ERR_IO; }
        struct stat st;
        if (fstat(mf.fd, &st) != 0) { set_error("cannot stat %s", paths[i].c_str()); return LFR_ERR_IO; }
        mf.size = (size_t)st.st_size;
        if (mf.size) {
            void *m = mmap(nullptr, mf.size, PROT_READ, MAP_PRIVATE, mf.fd, 0);
            if (m == MAP_FAILED) { mf.size = 0; set_error("cannot mmap %s", paths[i].c_str()); return LFR_ERR_IO; }
            mf.data = (const uint8_t *)m;
        }
        total_bytes += mf.size;
    }
    {
        std::atomic<size_t> cursor{0};
        const size_t kStep = (size_t)8 << 20;
        std::vector<std::pair<const uint8_t *, size_t>> spans;
        for (auto &mf : files) for (size_t o = 0; o < mf.size; o += kStep) spans.push_back({mf.data + o, std::min(kStep, mf.size - o)});
        std::atomic<uint64_t> sink{0};                 // (keeps the page-touching loads alive)
        run_threads((int)std::max<size_t>(1, std::min<size_t>((size_t)T, spans.size())), [&](int) {
            uint64_t acc = 0;
            for (;;) {
                const size_t k = cursor.fetch_add(1);
                if (k >= spans.size()) break;
                for (size_t o = 0; o < spans[k].second; o += 4096) acc += spans[k].first[o];
            }
            sink.fetch_add(acc, std::memory_order_relaxed);
        });
    }
    const double t_a = wall_ms();
    // ---- A: pair cursors.  The top level is a chain of (tag 0x0A, length, ImagePair) records: finding record k needs
    // record k-1's length, one dependent cache miss per pair (846 k of them in a 605 MB file = 0.2 s).  So a big file is
    // cut into segments; each segment's thread GUESSES a record start at/after its cut (a position from which several
    // records in a row parse as ImagePairs whose first field is a string), walks on from there, and the walks are
    // then chained: a segment's walk must end exactly where the next one began, otherwise (a wrong guess, or a file
    // that is not a clean chain) that stretch is re-walked sequentially.  The result is the sequential split, always.
    std::vector<std::vector<Cursor>> file_pairs(files.size());
    std::vector<int> file_rc(files.size(), LFR_OK);
    auto walk = [](const uint8_t *base, const uint8_t *from, const uint8_t *until, const uint8_t *end, std::vector<Cursor> &out,
                   const uint8_t *&stopped_at) -> int {
        // records that START before `until`; stopped_at = start of the first record not taken
        Cursor c{from, end};
        (void)base;
        while (!c.done() && c.p < until) {
            const uint64_t key = c.varint();
            const int field = (int)(key >> 3), wt = (int)(key & 7);
            if (!c.ok || field == 0) return LFR_ERR_PARSE;
            if (field == 1 && wt == 2) {
                Cursor pc = c.sub();
                if (!c.ok) return LFR_ERR_PARSE;
                out.push_back(pc);
            } else { c.skip(wt, field); if (!c.ok) return LFR_ERR_PARSE; }
        }
        stopped_at = c.p;
        return LFR_OK;
    };
    auto plausible_start = [](const uint8_t *p, const uint8_t *end) -> bool {
        // kCheck records in a row: tag 0x0A, a length that fits, and inside: tag 0x0A + a short string (image_name1)
        constexpr int kCheck = 6;
        Cursor c{p, end};
        for (int k = 0; k < kCheck; ++k) {
            if (c.done()) return k > 0;
            if (*c.p != 0x0A) return false;
            ++c.p;
            Cursor pc = c.sub();
            if (!c.ok) return false;
            if (pc.p < pc.end) {
                if (*pc.p != 0x0A) return false;
                ++pc.p;
                const uint64_t len = pc.varint();
                if (!pc.ok || len > 4096 || (uint64_t)(pc.end - pc.p) < len) return false;
            }
        }
        return true;
    };
    for (size_t i = 0; i < files.size(); ++i) {
        const MappedFile &mf = files[i];
        const uint8_t *b = mf.data, *e = mf.data + mf.size;
        size_t seg_bytes = (size_t)8 << 20;                                                         // >= 8 MB per segment
        if (const char *sb = getenv("LFR_SCANNER_SEGMENT_BYTES")) seg_bytes = (size_t)std::max(64ll, atoll(sb));   // (tests: tiny segments)
        const int S = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, mf.size / seg_bytes));
        auto &out = file_pairs[i];
        if (S <= 1) {
            const uint8_t *stop = b;
            out.reserve(mf.size / 600 + 16);
            file_rc[i] = mf.size ? walk(b, b, e, e, out, stop) : LFR_OK;
            continue;
        }
        std::vector<const uint8_t *> start(S, nullptr), stop(S, nullptr);
        std::vector<std::vector<Cursor>> seg(S);
        std::vector<int> rcs(S, LFR_OK);
        run_threads(S, [&](int t) {
            const uint8_t *cut = b + mf.size / S * t, *next_cut = t + 1 < S ? b + mf.size / S * (t + 1) : e;
            const uint8_t *p = cut;
            if (t > 0) {
                const uint8_t *limit = std::min(next_cut, cut + ((size_t)4 << 20));
                while (p < limit && !plausible_start(p, e)) ++p;
                if (p >= limit) { start[t] = nullptr; return; }          // no guess: the chain step re-walks this stretch
            }
            start[t] = p;
            seg[t].reserve((size_t)(next_cut - cut) / 600 + 16);
            rcs[t] = walk(b, p, next_cut, e, seg[t], stop[t]);
        });
        // chain
        const uint8_t *pos = b;
        int rc = LFR_OK;
        for (int t = 0; t < S && rc == LFR_OK; ++t) {
            const uint8_t *next_cut = t + 1 < S ? b + mf.size / S * (t + 1) : e;
            if (start[t] == pos && rcs[t] == LFR_OK) {
                out.insert(out.end(), seg[t].begin(), seg[t].end());
                pos = stop[t];
            } else if (pos < next_cut) {                               // wrong / missing guess: the sequential truth for this stretch
                rc = walk(b, pos, next_cut, e, out, pos);
            }                                                          // (else: the previous walk already ran past this segment)
        }
        if (rc == LFR_OK && pos != e) rc = pos < e ? walk(b, pos, e, e, out, pos) : LFR_ERR_PARSE;
        file_rc[i] = rc;
    }
    for (int rc : file_rc) if (rc != LFR_OK) return rc;
    std::vector<Cursor> pairs;
    {
        size_t n = 0;
        for (auto &v : file_pairs) n += v.size();
        pairs.reserve(n);
        for (auto &v : file_pairs) { pairs.insert(pairs.end(), v.begin(), v.end()); std::vector<Cursor>().swap(v); }
    }
    const int64_t P = (int64_t)pairs.size();
    if (P == 0) return LFR_OK;
    const double t_b = wall_ms();
    // ---- B: decode (chunks balanced by bytes) + per-chunk distinct names
    const int TB = (int)std::max<int64_t>(1, std::min<int64_t>(T, (int64_t)(total_bytes >> 22) + 1));
    std::vector<int64_t> cuts(TB + 1, P);
    cuts[0] = 0;
    {
        std::vector<size_t> pre(P + 1, 0);
        for (int64_t i = 0; i < P; ++i) pre[i + 1] = pre[i] + (size_t)(pairs[i].end - pairs[i].p);
        for (int t = 1; t < TB; ++t) {
            const size_t target = pre[P] / TB * t;
            cuts[t] = std::max<int64_t>(cuts[t - 1], (int64_t)(std::lower_bound(pre.begin(), pre.end(), target) - pre.begin()));
            cuts[t] = std::min<int64_t>(cuts[t], P);
        }
    }
    struct ChunkNames {
        std::vector<std::string> names;          // distinct, in order of local first appearance
        std::vector<float> facts;                // fact at that first appearance
        std::unordered_map<std::string, int32_t> index;
    };
    std::vector<MatchBuf> bufs(TB);
    HugeVec<PairRec> recs(P);
    std::vector<ChunkNames> cnames(TB);
    std::vector<int32_t> loc1(P), loc2(P);       // chunk-local image ids of the pair
    run_threads(TB, [&](int t) {
        MatchBuf &b = bufs[t];
        ChunkNames &cn = cnames[t];
        {   // one allocation per array instead of geometric growth (a match is >= ~170 bytes on the wire with its two 9-point grids;
            // re-growing 5 x ~20 MB vectors per thread cost a third of the decode)
            size_t bytes = 0;
            for (int64_t i = cuts[t]; i < cuts[t + 1]; ++i) bytes += (size_t)(pairs[i].end - pairs[i].p);
            const size_t est = bytes / 170 + 64;
            b.f1.reserve(est); b.f2.reserve(est); b.sim.reserve(est); b.d1.reserve(18 * est); b.d2.reserve(18 * est);
        }
        const char *last1 = nullptr; uint32_t last1_len = 0; int32_t last1_id = -1;
        for (int64_t i = cuts[t]; i < cuts[t + 1]; ++i) {
            recs[i].buf = t;
            const int rc = parse_pair(pairs[i], recs[i], b);
            if (rc != LFR_OK) { b.rc = rc; return; }
            const PairRec &r = recs[i];
            auto intern = [&](const char *nm, uint32_t len, float fact) -> int32_t {
                std::string key(nm ? nm : "", len);
                auto it = cn.index.find(key);
                if (it != cn.index.end()) return it->second;
                const int32_t id = (int32_t)cn.names.size();
                cn.index.emplace(key, id); cn.names.push_back(std::move(key)); cn.facts.push_back(fact);
                return id;
            };
            // (pairs usually come grouped by their first image: one string hash per pair instead of two)
            if (last1_id >= 0 && r.len1 == last1_len && (r.len1 == 0 || memcmp(r.name1, last1, r.len1) == 0)) loc1[i] = last1_id;
            else { loc1[i] = intern(r.name1, r.len1, r.fact1); last1 = r.name1; last1_len = r.len1; last1_id = loc1[i]; }
            loc2[i] = intern(r.name2, r.len2, r.fact2);
        }
    });
    for (auto &b : bufs) if (b.rc != LFR_OK) {
        if (b.rc == LFR_ERR_UNSUPPORTED)
            set_error("match with more than 9 grid displacements (the reference overflows flow_array, solve.cc:460-472)");
        return b.rc;
    }
    const double t_c = wall_ms();
    // ---- C: global image ids in order of first appearance over the KEPT pairs (solve.cc:444-451)
    // A name first seen in a banned pair must not enter the list there, and "first appearance" inside a chunk must skip
    // banned pairs too: mark banned names per chunk, then walk every chunk's kept pairs for names not yet interned.
    std::vector<std::vector<uint8_t>> cbanned(TB);
    for (int t = 0; t < TB; ++t) {
        cbanned[t].resize(cnames[t].names.size());
        for (size_t k = 0; k < cnames[t].names.size(); ++k) cbanned[t][k] = banned.count(cnames[t].names[k]) ? 1 : 0;
    }
    std::vector<std::vector<int32_t>> cglobal(TB);
    const size_t images_before = g.image_names.size();
    (void)images_before;
    for (int t = 0; t < TB; ++t) {
        cglobal[t].assign(cnames[t].names.size(), -1);
        size_t pending = cnames[t].names.size();
        for (int64_t i = cuts[t]; i < cuts[t + 1] && pending > 0; ++i) {
            const int32_t a = loc1[i], b = loc2[i];
            if (cbanned[t][a] || cbanned[t][b]) continue;
            // first fact wins = the fact carried by the first KEPT pair that names the image (solve.cc:449,451)
            if (cglobal[t][a] < 0) { cglobal[t][a] = g.intern_image(cnames[t].names[a], recs[i].fact1); --pending; }
            if (cglobal[t][b] < 0) { cglobal[t][b] = g.intern_image(cnames[t].names[b], recs[i].fact2); --pending; }
        }
    }
    // ---- D: global match offsets of the kept pairs
    std::vector<int64_t> moff(P + 1, 0);
    const int64_t M0 = g.n_matches();
    {
        int64_t acc = M0;
        for (int t = 0; t < TB; ++t)
            for (int64_t i = cuts[t]; i < cuts[t + 1]; ++i) {
                moff[i] = acc;
                if (!(cbanned[t][loc1[i]] || cbanned[t][loc2[i]])) {
                    if (recs[i].overflow) {
                        set_error("match with more than 9 grid displacements (the reference overflows flow_array, solve.cc:460-472)");
                        return LFR_ERR_UNSUPPORTED;
                    }
                    acc += recs[i].count;
                } else recs[i].count = -recs[i].count - 1;          // banned: remember it (count < 0)
            }
        moff[P] = acc;
    }
    const int64_t M = moff[P];
    const double t_d = wall_ms();
    // ---- E: node numbering (concurrent first-position map) beside the allocation + move of similarities and flows
    std::thread mover([&] {
        // ingest straight to a device: the flows go to HBM once, right away - pinning 144 B per match for that (hipHostMalloc of 360 MB
        // for config 4: tens of ms under the address-space lock, and as long again when the process is torn down) buys nothing
        static const bool pin_anyway = [] { const char *e = getenv("LFR_PIN_FLOWS"); return e && e[0] == '1'; }();
        if (g.prefetch_device >= 0 && M0 == 0 && !pin_anyway) { g.m_disp1.allow_pinning(false); g.m_disp2.allow_pinning(false); }
        g.m_sim.resize((size_t)M); g.m_disp1.resize((size_t)18 * M); g.m_disp2.resize((size_t)18 * M);
        std::atomic<int64_t> next{0};
        run_threads(TB, [&](int) {
            for (;;) {
                const int64_t lo = next.fetch_add(4096);
                if (lo >= P) break;
                const int64_t hi = std::min<int64_t>(P, lo + 4096);
                for (int64_t i = lo; i < hi; ++i) {
                    const PairRec &r = recs[i];
                    if (r.count <= 0) continue;
                    const MatchBuf &b = bufs[r.buf];
                    memcpy(&g.m_sim[moff[i]], &b.sim[r.first], sizeof(float) * r.count);
                    memcpy(&g.m_disp1[18 * moff[i]], &b.d1[18 * r.first], sizeof(float) * 18 * r.count);
                    memcpy(&g.m_disp2[18 * moff[i]], &b.d2[18 * r.first], sizeof(float) * 18 * r.count);
                }
            }
        });
        // ingest straight to a device: 144 of the 164 bytes per match are final now; they travel while the nodes are numbered
        if (g.prefetch_device >= 0 && M0 == 0) prestage_flows(g, g.prefetch_device, 2 * M);      // (M0 == 0: no nodes yet; this thread must not look at g.node_image, the main thread is resizing it)
    });
    const int64_t n_new = M - M0;
    const int64_t N0 = g.n_nodes();
    // (image, feature) -> slot.  Feature indices are dense small integers per image in real inputs (the running index of the
    // extractor's keypoint list), so the slot is img_off[image] + feature in a table of sum(max feature + 1) entries - 7 MB for
    // config 4, cache resident - instead of a 120 MB open-addressing table whose every touch and lookup misses (round 2: 100 of
    // the scanner's 270 ms).  Inputs with sparse feature indices (the table would exceed 4 slots per endpoint) keep the hash table.
    const size_t n_img = g.image_names.size();
    std::vector<uint64_t> img_off(n_img + 1, 0);
    bool dense = true;
    {
        std::vector<std::vector<uint32_t>> tmax(TB, std::vector<uint32_t>());
        std::vector<uint32_t> gmax(n_img, 0);
        std::vector<uint8_t> seen(n_img, 0);
        run_threads(TB, [&](int t) {
            std::vector<uint32_t> &mx = tmax[t];
            mx.assign(2 * n_img, 0);                                     // [2 i] = max feature, [2 i + 1] = seen
            for (int64_t i = cuts[t]; i < cuts[t + 1]; ++i) {
                const PairRec &r = recs[i];
                if (r.count <= 0) continue;
                const size_t a1 = (size_t)cglobal[t][loc1[i]], a2 = (size_t)cglobal[t][loc2[i]];
                const MatchBuf &bb = bufs[r.buf];
                uint32_t m1 = mx[2 * a1], m2 = mx[2 * a2];
                for (int64_t k = 0; k < r.count; ++k) { m1 = std::max(m1, bb.f1[r.first + k]); m2 = std::max(m2, bb.f2[r.first + k]); }
                mx[2 * a1] = std::max(mx[2 * a1], m1); mx[2 * a1 + 1] = 1;
                mx[2 * a2] = std::max(mx[2 * a2], m2); mx[2 * a2 + 1] = 1;
            }
        });
        for (int t = 0; t < TB; ++t)
            for (size_t i = 0; i < n_img; ++i) if (tmax[t][2 * i + 1]) { gmax[i] = std::max(gmax[i], tmax[t][2 * i]); seen[i] = 1; }
        for (int64_t n = 0; n < N0; ++n) { const size_t im = (size_t)g.node_image[n]; gmax[im] = std::max(gmax[im], g.node_feat[n]); seen[im] = 1; }
        const uint64_t budget = 4 * (uint64_t)(2 * n_new + N0) + ((uint64_t)1 << 20);
        for (size_t i = 0; i < n_img; ++i) {
            img_off[i + 1] = img_off[i] + (seen[i] ? (uint64_t)gmax[i] + 1 : 0);
            if (img_off[i + 1] > budget) { dense = false; break; }
        }
        if (const char *e = getenv("LFR_SCANNER_DENSE_NODES")) dense = dense && e[0] != '0';       // (tests: force the hash table)
    }
    // existing nodes of the graph (an earlier call on the same handle): seed the map with their ids
    uint64_t cap = 1024;
    if (dense) cap = std::max<uint64_t>(img_off[n_img], 1);
    else while (cap < (uint64_t)(2 * n_new + g.n_nodes()) * 3 / 2 + 16) cap <<= 1;     // load <= 2/3 even if every endpoint is a new node
    std::unique_ptr<NodeSlot[]> slots(new NodeSlot[cap]);
    const uint64_t mask = cap - 1;
    run_threads(TB, [&](int t) {
        const uint64_t lo = cap / TB * t, hi = t == TB - 1 ? cap : cap / TB * (t + 1);
        for (uint64_t k = lo; k < hi; ++k) { slots[k].key.store(0, std::memory_order_relaxed); slots[k].val.store(~0ull, std::memory_order_relaxed); }
    });
    auto dense_slot = [&](uint64_t key) -> uint64_t { return img_off[(size_t)(key >> 32)] + (uint32_t)key; };
    auto touch = [&](uint64_t key, uint64_t pos) {
        if (dense) {
            std::atomic<uint64_t> &val = slots[dense_slot(key)].val;
            uint64_t v = val.load(std::memory_order_relaxed);
            while (pos < v && !val.compare_exchange_weak(v, pos, std::memory_order_relaxed)) {}
            return;
        }
        uint64_t h = mix64(key) & mask;
        for (;;) {
            uint64_t cur = slots[h].key.load(std::memory_order_acquire);
            if (cur == 0) {
                uint64_t expected = 0;
                if (slots[h].key.compare_exchange_strong(expected, key + 1, std::memory_order_acq_rel)) cur = key + 1;
                else cur = expected;
            }
            if (cur == key + 1) {
                uint64_t v = slots[h].val.load(std::memory_order_relaxed);
                while (pos < v && !slots[h].val.compare_exchange_weak(v, pos, std::memory_order_relaxed)) {}
                return;
            }
            h = (h + 1) & mask;
        }
    };
    for (int64_t n = 0; n < N0; ++n) touch(((uint64_t)(uint32_t)g.node_image[n] << 32) | g.node_feat[n], (uint64_t)n);   // positions below every new one
    const uint64_t pos_base = (uint64_t)N0;
    run_threads(TB, [&](int t) {
        for (int64_t i = cuts[t]; i < cuts[t + 1]; ++i) {
            const PairRec &r = recs[i];
            if (r.count <= 0) continue;
            const uint64_t i1 = (uint64_t)(uint32_t)cglobal[t][loc1[i]] << 32, i2 = (uint64_t)(uint32_t)cglobal[t][loc2[i]] << 32;
            const MatchBuf &b = bufs[r.buf];
            for (int64_t k = 0; k < r.count; ++k) {
                const uint64_t pos = pos_base + 2 * (uint64_t)(moff[i] - M0 + k);
                touch(i1 | b.f1[r.first + k], pos);                      // node1 before node2 (solve.cc:474-475)
                touch(i2 | b.f2[r.first + k], pos + 1);
            }
        }
    });
    // occupied slots in order of first position = node ids.  Positions are distinct integers below N0 + 2 * n_new: every
    // slot drops its index at its position, a prefix count over the positions ranks them - no sort.
    const uint64_t n_pos = (uint64_t)N0 + 2 * (uint64_t)n_new;
    HugeVec<uint32_t> slot_at(n_pos, 0);                   // slot index + 1 (the table has < 2^32 slots: checked below)
    if (cap >= ((uint64_t)1 << 32)) { mover.join(); set_error("matches file too large for the node table"); return LFR_ERR_UNSUPPORTED; }
    run_threads(TB, [&](int t) {
        const uint64_t lo = cap / TB * t, hi = t == TB - 1 ? cap : cap / TB * (t + 1);
        for (uint64_t k = lo; k < hi; ++k) {
            const uint64_t v = slots[k].val.load(std::memory_order_relaxed);
            if (dense ? v != ~0ull : slots[k].key.load(std::memory_order_relaxed) != 0) slot_at[v] = (uint32_t)k + 1;
        }
    });
    std::vector<int64_t> chunk_count(TB + 1, 0);
    run_threads(TB, [&](int t) {
        int64_t c = 0;
        for (uint64_t q = n_pos * t / TB; q < n_pos * (t + 1) / TB; ++q) c += slot_at[q] != 0;
        chunk_count[t + 1] = c;
    });
    for (int t = 0; t < TB; ++t) chunk_count[t + 1] += chunk_count[t];
    const int64_t N = chunk_count[TB];
    if (N >= ((int64_t)1 << 32)) { mover.join(); set_error("more than 2^32 nodes"); return LFR_ERR_UNSUPPORTED; }
    g.node_image.resize(N); g.node_feat.resize(N);
    run_threads(TB, [&](int t) {
        int64_t n = chunk_count[t];
        for (uint64_t q = n_pos * t / TB; q < n_pos * (t + 1) / TB; ++q) {
            if (!slot_at[q]) continue;
            const uint64_t k = slot_at[q] - 1;
            NodeSlot &sl = slots[k];
            uint64_t key;
            if (dense) {
                const size_t im = (size_t)(std::upper_bound(img_off.begin(), img_off.end(), k) - img_off.begin()) - 1;
                key = ((uint64_t)im << 32) | (uint32_t)(k - img_off[im]);
            } else key = sl.key.load(std::memory_order_relaxed) - 1;
            sl.val.store((uint64_t)n, std::memory_order_relaxed);                        // position -> node id
            if (n >= N0) { g.node_image[n] = (int32_t)(key >> 32); g.node_feat[n] = (uint32_t)key; }
            ++n;
        }
    });
    auto lookup = [&](uint64_t key) -> uint32_t {
        if (dense) return (uint32_t)slots[dense_slot(key)].val.load(std::memory_order_relaxed);
        uint64_t h = mix64(key) & mask;
        while (slots[h].key.load(std::memory_order_relaxed) != key + 1) h = (h + 1) & mask;
        return (uint32_t)slots[h].val.load(std::memory_order_relaxed);
    };
    g.m_node1.resize((size_t)M); g.m_node2.resize((size_t)M);
    run_threads(TB, [&](int t) {
        for (int64_t i = cuts[t]; i < cuts[t + 1]; ++i) {
            const PairRec &r = recs[i];
            if (r.count <= 0) continue;
            const uint64_t i1 = (uint64_t)(uint32_t)cglobal[t][loc1[i]] << 32, i2 = (uint64_t)(uint32_t)cglobal[t][loc2[i]] << 32;
            const MatchBuf &b = bufs[r.buf];
            for (int64_t k = 0; k < r.count; ++k) {
                g.m_node1[moff[i] + k] = lookup(i1 | b.f1[r.first + k]);
                g.m_node2[moff[i] + k] = lookup(i2 | b.f2[r.first + k]);
            }
        }
    });
    const double t_e = wall_ms();
    mover.join();
    // Do NOT give ~1.3 GB of mappings back to the OS now: unmapping large ranges right before GPU work stalls the device's
    // queues (measured on this stack: the first kernels after the parse started 25-40 ms late - the whole "Total time" of a
    // one-shot run - and on time when the ranges stay mapped; the address-space notifier of the GPU driver is the likely
    // cause).  The decode buffers, the node table and the file mappings ride along with the graph and go when it goes
    // (lfr_graph_free; a one-shot caller simply exits).  LFR_FREE_INGEST_EARLY=1 restores the immediate release.
    {
        const char *early = getenv("LFR_FREE_INGEST_EARLY");
        if (!(early && early[0] == '1')) {
            struct Keep {
                std::vector<MatchBuf> bufs; HugeVec<PairRec> recs; HugeVec<uint32_t> slot_at;
                std::unique_ptr<NodeSlot[]> slots; std::vector<MappedFile> files;
            };
            auto keep = std::make_shared<Keep>();
            keep->bufs = std::move(bufs); keep->recs = std::move(recs); keep->slot_at = std::move(slot_at);
            keep->slots = std::move(slots); keep->files = std::move(files);
            g.ingest_keepalive.push_back(keep);
        }
    }
    if (verbose)
        fprintf(stderr, "lfr scanner: %zu bytes in %zu file(s), %lld pairs, %d threads: map+prefault %.1f ms, split %.1f ms, decode %.1f ms, "
                        "images+offsets %.1f ms, nodes %.1f ms (+%.1f ms waiting for the flow arrays)\n",
                total_bytes, files.size(), (long long)P, TB, t_a - t0, t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, wall_ms() - t_e);
    return LFR_OK;
}

static int parse_matching_path(const char *path, Graph &g, const std::set<std::string> &banned) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { set_error("cannot open %s", path); return LFR_ERR_IO; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); set_error("cannot stat %s", path); return LFR_ERR_IO; }
    int rc = LFR_OK;
    if (st.st_size > 0) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); set_error("cannot mmap %s", path); return LFR_ERR_IO; }
        madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        rc = parse_matching_buffer((const uint8_t *)m, (size_t)st.st_size, g, banned);
        munmap(m, (size_t)st.st_size);
    }
    close(fd);
    if (rc == LFR_ERR_PARSE) set_error("Failed to parse proto object.");
    return rc;
}

// ---------------------------------------------------------------------------- emit
struct Out {
    std::vector<uint8_t> buf;
    void varint(uint64_t v) { while (v >= 0x80) { buf.push_back((uint8_t)(v | 0x80)); v >>= 7; } buf.push_back((uint8_t)v); }
    void tag(int field, int wt) { varint(((uint64_t)field << 3) | wt); }
    void f32(int field, float v) {           // proto3: zero-valued scalars are omitted (bitwise zero)
        uint32_t bits; memcpy(&bits, &v, 4);
        if (!bits) return;
        tag(field, 5);
        const size_t o = buf.size(); buf.resize(o + 4); memcpy(&buf[o], &bits, 4);
    }
    void u32(int field, uint32_t v) { if (v) { tag(field, 0); varint(v); } }
    void bytes(int field, const void *p, size_t n) { tag(field, 2); varint(n); const size_t o = buf.size(); buf.resize(o + n); if (n) memcpy(&buf[o], p, n); }
    void str(int field, const std::string &s) { if (!s.empty()) bytes(field, s.data(), s.size()); }
};

static bool write_file(const char *path, const std::vector<const std::vector<uint8_t> *> &chunks) {
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    bool ok = true;
    for (auto *c : chunks) if (!c->empty() && fwrite(c->data(), 1, c->size(), f) != c->size()) ok = false;
    if (fclose(f) != 0) ok = false;
    return ok;
}

}  // namespace lfr

using namespace lfr;

extern "C" {

int lfr_version(void) { return LFR_VERSION; }
const char *lfr_last_error(void) { return g_error; }

static int graph_from_files(const char *const *paths, int n_paths, const char *const *banned, int n_banned, int device,
                            lfr_graph **out) {
    if (!out || (n_paths > 0 && !paths)) { set_error("bad argument"); return LFR_ERR_ARG; }
    std::set<std::string> ban;
    for (int i = 0; i < n_banned; ++i) ban.insert(banned[i]);
    lfr_graph *h = new lfr_graph();
    h->g.prefetch_device = device;
    const char *seq = getenv("LFR_SCANNER_SEQUENTIAL");
    if (seq && seq[0] == '1') {                        // the file-by-file scanner with the sequential numbering pass (cross-check)
        for (int i = 0; i < n_paths; ++i) {
            const int rc = parse_matching_path(paths[i], h->g, ban);
            if (rc != LFR_OK) { delete h; *out = nullptr; return rc; }
        }
    } else {
        std::vector<std::string> all;
        for (int i = 0; i < n_paths; ++i) all.push_back(paths[i]);
        const int rc = parse_all(all, h->g, ban);
        if (rc == LFR_ERR_PARSE) set_error("Failed to parse proto object.");
        if (rc != LFR_OK) { delete h; *out = nullptr; return rc; }
    }
    h->g.finish();
    *out = h;
    if (device >= 0 && h->g.n_nodes() > 0) {             // the rest of the graph follows the flows (asynchronous)
        const int rc = graph_make_resident(h->g, device);
        if (rc != LFR_OK) { delete h; *out = nullptr; return rc; }
    }
    return LFR_OK;
}

int lfr_graph_from_files(const char *const *paths, int n_paths, const char *const *banned, int n_banned,
                         lfr_graph **out) {
    return graph_from_files(paths, n_paths, banned, n_banned, -1, out);
}

static int graph_from_matches_file(const char *path, const char *const *banned, int n_banned, int device, lfr_graph **out);

int lfr_graph_from_matches_file(const char *path, const char *const *banned, int n_banned, lfr_graph **out) {
    return graph_from_matches_file(path, banned, n_banned, -1, out);
}

int lfr_graph_from_matches_file_device(const char *path, const char *const *banned, int n_banned, int device, lfr_graph **out) {
    if (device < 0) { set_error("bad device ordinal %d", device); return LFR_ERR_ARG; }
    return graph_from_matches_file(path, banned, n_banned, device, out);
}

static int graph_from_matches_file(const char *path, const char *const *banned, int n_banned, int device, lfr_graph **out) {
    if (!path || !out) { set_error("bad argument"); return LFR_ERR_ARG; }
    std::vector<std::string> files;
    if (access(path, R_OK) == 0) files.push_back(path);      // solve.cc:416-424
    else
        for (size_t part = 0;; ++part) {
            const std::string f = std::string(path) + ".part." + std::to_string(part);
            if (access(f.c_str(), R_OK) != 0) break;
            files.push_back(f);
        }
    std::vector<const char *> ptrs;
    for (auto &f : files) ptrs.push_back(f.c_str());
    return graph_from_files(ptrs.data(), (int)ptrs.size(), banned, n_banned, device, out);
}

int lfr_graph_from_arrays(int32_t n_images, const char *const *image_names, const float *image_facts,
                          int64_t n_pairs, const int32_t *pair_img1, const int32_t *pair_img2,
                          const int64_t *pair_off, const uint32_t *feat1, const uint32_t *feat2,
                          const float *sim, const float *disp1, const float *disp2,
                          const char *const *banned, int n_banned, lfr_graph **out) {
    if (!out || n_images < 0 || n_pairs < 0) { set_error("bad argument"); return LFR_ERR_ARG; }
    std::set<std::string> ban;
    for (int i = 0; i < n_banned; ++i) ban.insert(banned[i]);
    lfr_graph *h = new lfr_graph();
    Graph &g = h->g;
    const int64_t M = n_pairs ? pair_off[n_pairs] : 0;
    g.m_node1.reserve(M); g.m_node2.reserve(M); g.m_sim.reserve(M);
    g.m_disp1.reserve(18 * M); g.m_disp2.reserve(18 * M);
    for (int64_t p = 0; p < n_pairs; ++p) {
        const int32_t a = pair_img1[p], b = pair_img2[p];
        if (a < 0 || a >= n_images || b < 0 || b >= n_images) { delete h; set_error("image index out of range"); return LFR_ERR_ARG; }
        const std::string na = image_names[a], nb = image_names[b];
        if (ban.count(na) || ban.count(nb)) continue;
        const int32_t i1 = g.intern_image(na, image_facts[a]);
        const int32_t i2 = g.intern_image(nb, image_facts[b]);
        for (int64_t m = pair_off[p]; m < pair_off[p + 1]; ++m)
            g.add_match(i1, i2, feat1[m], feat2[m], sim[m], disp1 + 18 * m, 9, disp2 + 18 * m, 9);
    }
    g.finish();
    *out = h;
    return LFR_OK;
}

int lfr_graph_from_arrays_device_flows(int32_t n_images, const char *const *image_names, const float *image_facts,
                                       int64_t n_pairs, const int32_t *pair_img1, const int32_t *pair_img2,
                                       const int64_t *pair_off, const uint32_t *feat1, const uint32_t *feat2,
                                       const float *sim, const void *disp1_device, const void *disp2_device, int device,
                                       const char *const *banned, int n_banned, lfr_graph **out) {
    if (!out || n_images < 0 || n_pairs < 0 || !disp1_device || !disp2_device || device < 0) { set_error("bad argument"); return LFR_ERR_ARG; }
    std::set<std::string> ban;
    for (int i = 0; i < n_banned; ++i) ban.insert(banned[i]);
    lfr_graph *h = new lfr_graph();
    Graph &g = h->g;
    const int64_t M = n_pairs ? pair_off[n_pairs] : 0;
    g.m_node1.reserve(M); g.m_node2.reserve(M); g.m_sim.reserve(M); g.m_flow_row.reserve(M);
    for (int64_t p = 0; p < n_pairs; ++p) {
        const int32_t a = pair_img1[p], b = pair_img2[p];
        if (a < 0 || a >= n_images || b < 0 || b >= n_images) { delete h; set_error("image index out of range"); return LFR_ERR_ARG; }
        const std::string na = image_names[a], nb = image_names[b];
        if (ban.count(na) || ban.count(nb)) continue;
        const int32_t i1 = g.intern_image(na, image_facts[a]);
        const int32_t i2 = g.intern_image(nb, image_facts[b]);
        for (int64_t m = pair_off[p]; m < pair_off[p + 1]; ++m) {
            const uint32_t n1 = g.find_or_create_node(i1, feat1[m]);      // node1 before node2 (solve.cc:474-475)
            const uint32_t n2 = g.find_or_create_node(i2, feat2[m]);
            g.m_node1.push_back(n1); g.m_node2.push_back(n2); g.m_sim.push_back(sim[m]);
            g.m_flow_row.push_back((uint32_t)m);
        }
    }
    g.dev_disp1 = (const float *)disp1_device; g.dev_disp2 = (const float *)disp2_device; g.dev_flows_device = device;
    g.finish();
    *out = h;
    return LFR_OK;
}

void lfr_graph_free(lfr_graph *g) { delete g; }
int64_t lfr_graph_num_nodes(const lfr_graph *g) { return g ? g->g.n_nodes() : 0; }
int64_t lfr_graph_num_edges(const lfr_graph *g) { return g ? 2 * g->g.n_matches() : 0; }
int32_t lfr_graph_num_images(const lfr_graph *g) { return g ? (int32_t)g->g.image_names.size() : 0; }
int lfr_graph_get_nodes(const lfr_graph *g, int32_t *node_image, uint32_t *node_feature) {
    if (!g) return LFR_ERR_ARG;
    const int64_t n = g->g.n_nodes();
    if (node_image && n) memcpy(node_image, g->g.node_image.data(), sizeof(int32_t) * n);
    if (node_feature && n) memcpy(node_feature, g->g.node_feat.data(), sizeof(uint32_t) * n);
    return LFR_OK;
}
const char *lfr_graph_image_name(const lfr_graph *g, int32_t image) {
    if (!g || image < 0 || image >= (int32_t)g->g.image_names.size()) return nullptr;
    return g->g.image_names[image].c_str();
}
float lfr_graph_image_fact(const lfr_graph *g, int32_t image) {
    if (!g || image < 0 || image >= (int32_t)g->g.image_fact.size()) return 0.f;
    return g->g.image_fact[image];
}

int lfr_write_matching_file(const char *path, int32_t n_images, const char *const *image_names,
                            const float *image_facts, int64_t n_pairs, const int32_t *pair_img1,
                            const int32_t *pair_img2, const int64_t *pair_off, const uint32_t *feat1,
                            const uint32_t *feat2, const float *sim, const float *disp1, const float *disp2) {
    if (!path) { set_error("bad argument"); return LFR_ERR_ARG; }
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot open %s for writing", path); return LFR_ERR_IO; }
    Out pair, match, disp, head;
    bool ok = true;
    for (int64_t p = 0; p < n_pairs && ok; ++p) {
        const int32_t a = pair_img1[p], b = pair_img2[p];
        if (a < 0 || a >= n_images || b < 0 || b >= n_images) { fclose(f); set_error("image index out of range"); return LFR_ERR_ARG; }
        pair.buf.clear();
        pair.str(1, image_names[a]); pair.f32(2, image_facts[a]);
        pair.str(3, image_names[b]); pair.f32(4, image_facts[b]);
        for (int64_t m = pair_off[p]; m < pair_off[p + 1]; ++m) {
            match.buf.clear();
            match.u32(1, feat1[m]); match.u32(2, feat2[m]); match.f32(3, sim[m]);
            for (int which = 0; which < 2; ++which) {
                const float *d = (which == 0 ? disp1 : disp2) + 18 * m;
                for (int k = 0; k < 9; ++k) {
                    disp.buf.clear();
                    disp.f32(1, d[2 * k]); disp.f32(2, d[2 * k + 1]);
                    match.bytes(which == 0 ? 4 : 5, disp.buf.data(), disp.buf.size());
                }
            }
            pair.bytes(5, match.buf.data(), match.buf.size());
        }
        head.buf.clear();
        head.tag(1, 2); head.varint(pair.buf.size());
        ok = fwrite(head.buf.data(), 1, head.buf.size(), f) == head.buf.size() &&
             (pair.buf.empty() || fwrite(pair.buf.data(), 1, pair.buf.size(), f) == pair.buf.size());
    }
    if (fclose(f) != 0) ok = false;
    if (!ok) { set_error("Failed to write proto object."); return LFR_ERR_IO; }
    return LFR_OK;
}

int lfr_apply_displacements(const lfr_graph *gh, const double *positions, const char *image_name, float *keypoints,
                            int64_t num_features, int64_t stride) {
    if (!gh || !image_name || (!keypoints && num_features > 0) || num_features < 0 || stride < 2 || (!positions && gh->g.n_nodes() > 0)) {
        set_error("bad argument"); return LFR_ERR_ARG;
    }
    const Graph &g = gh->g;
    const auto it = g.image_index.find(image_name);
    if (it != g.image_index.end()) {
        const int32_t im = it->second;
        const float fact = g.image_fact[im];
        for (int64_t k = g.img_off[im]; k < g.img_off[im + 1]; ++k) {
            const uint32_t n = g.img_nodes[k];
            const uint32_t f = g.node_feat[n];
            if ((int64_t)f >= num_features) { set_error("feature_idx %u out of range (%lld keypoints)", f, (long long)num_features); return LFR_ERR_ARG; }
            // float32 arithmetic of colmap_utils.py:129-136: displacements[f] = [dj, di]; *= fact; keypoints += displacements * 16
            const float dj = (float)positions[2 * n + 1], di = (float)positions[2 * n];
            float *kp = keypoints + (size_t)f * stride;
            volatile float tx = dj * fact, ty = di * fact;      // volatile: keep every product rounded to float32
            volatile float ux = tx * 16.0f, uy = ty * 16.0f;
            kp[0] = kp[0] + ux; kp[1] = kp[1] + uy;
        }
    }
    for (int64_t f = 0; f < num_features; ++f) {                 // colmap_utils.py:137
        float *kp = keypoints + (size_t)f * stride;
        kp[0] = kp[0] + 0.5f; kp[1] = kp[1] + 0.5f;
    }
    return LFR_OK;
}

int lfr_write_solution(const lfr_graph *gh, const double *positions, const char *path, int64_t *n_outside) {
    if (!gh || !path || (!positions && gh->g.n_nodes() > 0)) { set_error("bad argument"); return LFR_ERR_ARG; }
    const Graph &g = gh->g;
    const int64_t n = g.n_nodes();
    // images in order of first node (solve.cc:647-659); every node is emitted (solve.cc:661-664)
    std::vector<int32_t> slot(g.image_names.size(), -1);
    std::vector<int32_t> order;
    for (int64_t i = 0; i < n; ++i)
        if (slot[g.node_image[i]] < 0) { slot[g.node_image[i]] = (int32_t)order.size(); order.push_back(g.node_image[i]); }
    std::vector<Out> bodies(order.size());
    for (size_t k = 0; k < order.size(); ++k) {
        bodies[k].str(1, g.image_names[order[k]]);
        bodies[k].f32(2, g.image_fact[order[k]]);
    }
    int64_t outside = 0;
    Out d;
    for (int64_t i = 0; i < n; ++i) {
        const double di = positions[2 * i], dj = positions[2 * i + 1];
        d.buf.clear();
        d.u32(1, g.node_feat[i]); d.f32(2, (float)di); d.f32(3, (float)dj);
        bodies[slot[g.node_image[i]]].bytes(3, d.buf.data(), d.buf.size());
        if (std::fabs(dj) > 0.5 || std::fabs(di) > 0.5) ++outside;     // solve.cc:666-668
    }
    if (n_outside) *n_outside = outside;
    std::vector<Out> heads(order.size());
    std::vector<const std::vector<uint8_t> *> chunks;
    for (size_t k = 0; k < order.size(); ++k) {
        heads[k].tag(1, 2); heads[k].varint(bodies[k].buf.size());
        chunks.push_back(&heads[k].buf); chunks.push_back(&bodies[k].buf);
    }
    if (!write_file(path, chunks)) { set_error("Failed to write proto object."); return LFR_ERR_IO; }
    return LFR_OK;
}

}  // extern "C"
