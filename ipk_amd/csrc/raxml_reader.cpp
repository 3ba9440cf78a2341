// raxml_reader.cpp -- RAxML-ng `.raxml.ancestralProbs` loader: the input
// edge of the hot path, host side, multi-threaded.
//
// Restates raxmlng_reader (ipk/src/ar.cpp:144-270):
//   * build_index (:150-188): skip the header line; a node's block starts where the first column
//     (text before the first TAB) changes; the byte offset of that line is remembered per label
//     (a label that re-appears later overwrites its offset, like the reference's map assignment).
//   * read_node (:200-270): from the node's offset read rows `Node \t Site \t State \t p_1 .. p_sigma`
//     until the label changes; empty lines and lines starting with '.' are comments (:217,
//     single_and_empty_line_comment<'.'>); fields are trimmed of spaces (trim_chars<' '>); AA columns
//     are permuted from RAxML-ng order a,r,n,d,c,q,e,g,h,i,l,k,m,f,p,s,t,w,y,v to IPK order
//     r,h,k,d,e,s,t,n,q,c,g,p,a,i,l,m,f,w,y,v (:227-234); every value goes through log10 in float
//     (:257-259).
// Text -> float conversion: the reference uses the un-vendored "strasser" fast-cpp-csv-parser, whose
// parse_float accumulates digits in the target type (x = x*10 + d; fraction: pos /= 10, x += d*pos;
// exponent by repeated multiplication) -- NOT a correctly rounded strtof.  That algorithm is restated
// here in float so the matrices match what IPK would compute; it is an assumption about un-vendored
// code (parity unpinned).
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ipkgpu.h"

struct ipkgpu_ar {
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;
    uint32_t sigma = 0;
    std::vector<std::string> labels;             // in order of first appearance
    std::vector<size_t> offsets;                 // block start of each label (last block wins, as in the reference)
    std::unordered_map<std::string, uint32_t> index;
    uint32_t sites = 0;                          // rows of the first node
    std::string err;
};

static thread_local std::string g_ar_err;

namespace {

// fast-cpp-csv-parser parse_float<float>, restated (see header comment)
bool parse_float_csv(const char* col, const char* end, float& out)
{
    while (col < end && *col == ' ') ++col;                      // trim_chars<' '>
    while (end > col && end[-1] == ' ') --end;
    bool neg = false;
    if (col < end && *col == '-') { neg = true; ++col; }
    else if (col < end && *col == '+') ++col;
    float x = 0;
    while (col < end && '0' <= *col && *col <= '9') { const int y = *col - '0'; x *= 10; x += y; ++col; }
    if (col < end && (*col == '.' || *col == ',')) {
        ++col;
        float pos = 1;
        while (col < end && '0' <= *col && *col <= '9') { pos /= 10; const int y = *col - '0'; ++col; x += y * pos; }
    }
    if (col < end && (*col == 'e' || *col == 'E')) {
        ++col;
        bool eneg = false;
        if (col < end && *col == '-') { eneg = true; ++col; }
        else if (col < end && *col == '+') ++col;
        if (col >= end) return false;
        long e = 0;
        while (col < end && '0' <= *col && *col <= '9') { e = e * 10 + (*col - '0'); if (e > 100000) e = 100000; ++col; }
        if (col != end) return false;
        if (e != 0) {
            float base = eneg ? 0.1f : 10.0f;
            while (e != 1) {
                if ((e & 1) == 0) { base = base * base; e >>= 1; }
                else { x *= base; --e; }
            }
            x *= base;
        }
    } else if (col != end) {
        return false;                                            // error::no_digit
    }
    out = neg ? -x : x;
    return true;
}

// The same arithmetic for the common shape of a field -- digits '.' digits, nothing else -- without a division and a
// multiplication per digit: parse_float's `pos /= 10` runs through a fixed sequence of floats (1, 0.1f, fl(0.1f / 10), ...)
// and `y * pos` is one of ten floats per position, so both come from tables built once with exactly those operations; what is
// left per digit is the float addition, in the same order.  Returns the first character after the number (the caller checks
// that it ends the field), or nullptr when the field is not of this shape (the general parser then decides).
struct FracTable {
    static constexpr int N = 48;
    float prod[N][10];
    FracTable()
    {
        float pos = 1;
        for (int k = 0; k < N; ++k) {
            pos /= 10;                                                 // parse_float: pos /= 10
            for (int d = 0; d < 10; ++d) prod[k][d] = (float)d * pos;   //              x += y * pos   (y = digit, int -> float exact)
        }
    }
};
static const FracTable g_frac;

inline const char* parse_plain_decimal(const char* col, float& out)
{
    float x = 0;
    const char* c = col;
    while ((unsigned)(*c - '0') <= 9u) { x *= 10; x += (float)(*c - '0'); ++c; }
    if (*c == '.') {
        ++c;
        int k = 0;
        while ((unsigned)(*c - '0') <= 9u) {
            if (k >= FracTable::N) return nullptr;
            x += g_frac.prod[k][*c - '0'];
            ++c; ++k;
        }
    }
    if (c == col) return nullptr;
    out = x;
    return c;
}

// Worker threads when the caller does not say: the cores this process may run on (affinity mask), at most 16 -- a GPU's
// share of a multi-GPU host; IPKGPU_THREADS overrides.
inline uint32_t default_threads()
{
    if (const char* e = getenv("IPKGPU_THREADS")) { const long v = atol(e); if (v > 0) return (uint32_t)std::min<long>(v, 256); }
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<uint32_t>(n, (uint32_t)std::max(1, CPU_COUNT(&set)));
    return std::min<uint32_t>(n, 16u);
}

const int AA_FROM_RAXML[20] = {1, 8, 11, 3, 6, 15, 16, 2, 5, 4, 7, 14, 0, 9, 10, 12, 13, 17, 18, 19};

inline const char* line_end(const char* p, const char* end)
{
    const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
    return q ? q : end;
}

// Parses the block starting at `off` for `label`; writes sites x sigma log10 floats when out != nullptr.
// Returns the number of rows, or -1 on a malformed row.
long read_block(const ipkgpu_ar* ar, size_t off, const std::string& label, float* out, uint32_t max_sites)
{
    const char* p = ar->data + off;
    const char* end = ar->data + ar->size;
    const uint32_t sigma = ar->sigma;
    long rows = 0;
    float col[20];
    // The file ends in a line feed in every RAxML-ng output; the fast path below scans without bounds checks up to the last
    // line feed and leaves a final unterminated line to the general path.
    const char* last_lf = end;
    while (last_lf > ar->data && last_lf[-1] != '\n') --last_lf;     // one past the last '\n' (ar->data if there is none)
    while (p < end) {
        // ---- fast path: `label TAB digits TAB state TAB v TAB v ... v LF`, every value a plain decimal -- one left-to-right
        //      scan, no library calls; anything else (spaces to trim, exponents, CR LF, comments, a short row) falls through
        //      to the general parser of the same line, so the two never disagree on what a well-formed row means ----------
        if (p + label.size() + 1 < last_lf && memcmp(p, label.data(), label.size()) == 0 && p[label.size()] == '\t') {
            const char* f = p + label.size() + 1;
            while ((unsigned)(*f - '0') <= 9u) ++f;                              // Site
            if (*f == '\t' && f[1] != '\t' && f[1] != '\n' && f[2] == '\t') {     // State: one character
                f += 3;
                bool ok = true;
                for (uint32_t i = 0; i < sigma && ok; ++i) {
                    const char* e = parse_plain_decimal(f, col[i]);
                    ok = e && *e == (i + 1 < sigma ? '\t' : '\n');
                    f = e ? e + 1 : f;
                }
                if (ok) {
                    if (out) {
                        if ((uint32_t)rows >= max_sites) return -1;
                        float* o = out + (size_t)rows * sigma;
                        if (sigma == 20) for (int i = 0; i < 20; ++i) o[i] = log10f(col[AA_FROM_RAXML[i]]);
                        else for (uint32_t i = 0; i < sigma; ++i) o[i] = log10f(col[i]);
                    }
                    ++rows;
                    p = f;                                                       // one past the row's line feed
                    continue;
                }
            }
        }
        const char* le = line_end(p, end);
        const char* q = le;
        if (q > p && q[-1] == '\r') --q;
        if (q == p || *p == '.') { p = le + 1; continue; }           // empty / comment line
        const char* t = (const char*)memchr(p, '\t', (size_t)(q - p));
        const size_t ll = t ? (size_t)(t - p) : (size_t)(q - p);
        // the label field is trimmed like every other field
        const char* a = p; const char* b = p + ll;
        while (a < b && *a == ' ') ++a;
        while (b > a && b[-1] == ' ') --b;
        if ((size_t)(b - a) != label.size() || memcmp(a, label.data(), label.size()) != 0) break;   // next node
        // skip Site, State
        const char* f = t ? t + 1 : q;
        for (int s = 0; s < 2; ++s) {
            const char* n = (const char*)memchr(f, '\t', (size_t)(q - f));
            if (!n) return -1;
            f = n + 1;
        }
        for (uint32_t i = 0; i < sigma; ++i) {
            // fast path: a plain decimal that ends exactly at the field's end (a TAB, or the line's end for the last column);
            // the line ends in '\n' (or the file's last byte is consumed by the general path), so the scan cannot run away
            const char* e = (q < end) ? parse_plain_decimal(f, col[i]) : nullptr;
            if (e && e <= q && ((i + 1 < sigma) ? (*e == '\t') : (e == q))) { f = e + 1; continue; }
            const char* n = (i + 1 < sigma) ? (const char*)memchr(f, '\t', (size_t)(q - f)) : q;
            if (!n) return -1;                                       // too few columns
            if (!parse_float_csv(f, n, col[i])) return -1;
            f = n + 1;
        }
        if (out) {
            if ((uint32_t)rows >= max_sites) return -1;
            float* o = out + (size_t)rows * sigma;
            if (sigma == 20) for (int i = 0; i < 20; ++i) o[i] = log10f(col[AA_FROM_RAXML[i]]);
            else for (uint32_t i = 0; i < sigma; ++i) o[i] = log10f(col[i]);
        }
        ++rows;
        p = le + 1;
    }
    return rows;
}

}  // namespace

extern "C" {

const char* ipkgpu_ar_last_error(void) { return g_ar_err.c_str(); }

int ipkgpu_ar_open(const char* path, uint32_t sigma, ipkgpu_ar** out)
{
    if (!out) return IPKGPU_ERR_INVALID;
    *out = nullptr;
    if (!path || (sigma != 4 && sigma != 20)) { g_ar_err = "bad path or alphabet size (4 or 20)"; return IPKGPU_ERR_INVALID; }
    ipkgpu_ar* ar = new ipkgpu_ar();
    ar->sigma = sigma;
    ar->fd = open(path, O_RDONLY);
    struct stat st;
    if (ar->fd < 0 || fstat(ar->fd, &st) != 0) {
        g_ar_err = std::string("cannot open ") + path;
        if (ar->fd >= 0) close(ar->fd);
        delete ar;
        return IPKGPU_ERR_INVALID;
    }
    ar->size = (size_t)st.st_size;
    if (ar->size) {
        void* m = mmap(nullptr, ar->size, PROT_READ, MAP_PRIVATE, ar->fd, 0);
        if (m == MAP_FAILED) { g_ar_err = "mmap failed"; close(ar->fd); delete ar; return IPKGPU_ERR_NOMEM; }
        ar->data = (const char*)m;
    }
    // build_index, ar.cpp:150-188: a block starts where the first field changes.  The scan is split over the host's cores at
    // line boundaries (a 1.5 GB file: 0.4 s on one core); every range lists the lines whose label differs from the line before
    // it IN the range (its first line always), and the ranges are stitched in file order with the label carried across -- the
    // same sequence of (label, offset) events as one left-to-right pass.
    const char* p = ar->data;
    const char* end = ar->data + ar->size;
    if (p < end) p = line_end(p, end) + 1;                           // skip the header
    struct Ev { const char* at; size_t len; };
    const size_t body = p < end ? (size_t)(end - p) : 0;
    uint32_t nth = (uint32_t)std::min<size_t>(default_threads(), body / (8u << 20) + 1);
    std::vector<std::vector<Ev>> evs(nth);
    auto scan = [&](uint32_t t) {
        const char* a = p + body * t / nth;
        const char* b = p + body * (t + 1) / nth;
        if (t > 0) { a = line_end(a - 1, end); a = a < end ? a + 1 : end; }        // first line starting at or after the cut
        if (t + 1 < nth) { b = line_end(b - 1, end); b = b < end ? b + 1 : end; }
        const char* cur = nullptr; size_t cur_len = 0;
        for (const char* q = a; q < b;) {
            const char* le = line_end(q, end);
            const char* tb = (const char*)memchr(q, '\t', (size_t)(le - q));
            const size_t ll = tb ? (size_t)(tb - q) : (size_t)(le - q);
            if (!cur || ll != cur_len || memcmp(q, cur, ll) != 0) { evs[t].push_back({q, ll}); cur = q; cur_len = ll; }
            q = le + 1;
        }
    };
    {
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < nth; ++t) th.emplace_back(scan, t);
        scan(0);
        for (auto& x : th) x.join();
    }
    std::string current;
    bool have = false;
    for (uint32_t t = 0; t < nth; ++t)
        for (const Ev& e : evs[t]) {
            if (have && e.len == current.size() && memcmp(e.at, current.data(), e.len) == 0) continue;   // a range's first line continuing the block
            current.assign(e.at, e.len);
            have = true;
            if (!current.empty() && current[0] != '.') {             // comment / empty lines never name a node
                auto it = ar->index.find(current);
                if (it == ar->index.end()) {
                    ar->index.emplace(current, (uint32_t)ar->labels.size());
                    ar->labels.push_back(current);
                    ar->offsets.push_back((size_t)(e.at - ar->data));
                } else {
                    ar->offsets[it->second] = (size_t)(e.at - ar->data);
                }
            }
        }
    if (!ar->labels.empty()) {
        const long rows = read_block(ar, ar->offsets[0], ar->labels[0], nullptr, 0);
        if (rows < 0) { g_ar_err = "malformed row in the block of node " + ar->labels[0]; ipkgpu_ar_close(ar); return IPKGPU_ERR_INVALID; }
        ar->sites = (uint32_t)rows;
    }
    *out = ar;
    return IPKGPU_OK;
}

void ipkgpu_ar_close(ipkgpu_ar* ar)
{
    if (!ar) return;
    if (ar->data) munmap((void*)ar->data, ar->size);
    if (ar->fd >= 0) close(ar->fd);
    delete ar;
}

uint32_t ipkgpu_ar_num_nodes(const ipkgpu_ar* ar) { return ar ? (uint32_t)ar->labels.size() : 0; }
uint32_t ipkgpu_ar_sites(const ipkgpu_ar* ar) { return ar ? ar->sites : 0; }
const char* ipkgpu_ar_node_label(const ipkgpu_ar* ar, uint32_t i) { return (ar && i < ar->labels.size()) ? ar->labels[i].c_str() : nullptr; }
int64_t ipkgpu_ar_find(const ipkgpu_ar* ar, const char* label)
{
    if (!ar || !label) return -1;
    auto it = ar->index.find(label);
    return it == ar->index.end() ? -1 : (int64_t)it->second;
}

int ipkgpu_ar_read_nodes(ipkgpu_ar* ar, const uint32_t* node_idx, uint32_t n, float* out, uint32_t n_threads)
{
    if (!ar || !node_idx || !out) { g_ar_err = "null argument"; return IPKGPU_ERR_INVALID; }
    for (uint32_t i = 0; i < n; ++i)
        if (node_idx[i] >= ar->labels.size()) { g_ar_err = "node index out of range"; return IPKGPU_ERR_INVALID; }
    if (n_threads == 0) n_threads = default_threads();
    n_threads = std::min<uint32_t>(n_threads, std::max(1u, n));
    std::atomic<uint32_t> next{0};
    std::atomic<int> bad{-1};
    const size_t stride = (size_t)ar->sites * ar->sigma;
    auto work = [&]() {
        for (;;) {
            const uint32_t i = next.fetch_add(1);
            if (i >= n) break;
            const uint32_t node = node_idx[i];
            const long rows = read_block(ar, ar->offsets[node], ar->labels[node], out + (size_t)i * stride, ar->sites);
            if (rows != (long)ar->sites) { int exp = -1; bad.compare_exchange_strong(exp, (int)i); }
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < n_threads; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (bad.load() >= 0) {
        // ar.cpp:264-267 "Could not read the AR matrix for the node"; also raised here when a node's block has a
        // different number of sites than the first node (the reference would build a ragged matrix set)
        g_ar_err = "Could not read the AR matrix for the node " + ar->labels[node_idx[bad.load()]] +
                   " (malformed row or a site count different from the first node's)";
        return IPKGPU_ERR_INVALID;
    }
    return IPKGPU_OK;
}

}  // extern "C"
