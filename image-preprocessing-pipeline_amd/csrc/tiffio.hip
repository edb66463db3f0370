// TIFF series I/O of the block pipeline (include/mi_tiffio.h): host code only -- a box of a folder of slices in, one file per z slice
// out, one slice per task on all cores, no interpreter lock anywhere near the codec.
//
// What it replaces: load_bl_tif.cpp (box reads through libtiff, one handle per thread) and save_bl_tif.cpp (slices written on all
// cores: Adobe deflate, ZIPQUALITY 1, predictor 1, strips -- save_bl_tif.cpp:336-346).  No libtiff here (its headers are not in the
// image): the files the pipeline meets are strips of deflate or raw samples, which is a header of a dozen tags and a zlib stream per
// strip.  The deflate codec is libdeflate when libdeflate.so.0 can be loaded (2-3 x zlib at level 1, same format), else zlib.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mi_internal.h"
#include "mi_tiffio.h"

namespace {

using mi::fail;

// ------------------------------------------------------------------------------------------------ deflate: libdeflate or zlib
struct Deflate {
    void* so = nullptr;
    void* (*alloc_c)(int) = nullptr;
    size_t (*zcomp)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*zbound)(void*, size_t) = nullptr;
    void (*free_c)(void*) = nullptr;
    void* (*alloc_d)() = nullptr;
    int (*zdecomp)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_d)(void*) = nullptr;
    bool ok = false;
    Deflate() {
        if (std::getenv("MI_TIFF_ZLIB")) return;  // (tests: the zlib route on a host that has libdeflate)
        so = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!so) return;
        alloc_c = reinterpret_cast<void* (*)(int)>(dlsym(so, "libdeflate_alloc_compressor"));
        zcomp = reinterpret_cast<size_t (*)(void*, const void*, size_t, void*, size_t)>(dlsym(so, "libdeflate_zlib_compress"));
        zbound = reinterpret_cast<size_t (*)(void*, size_t)>(dlsym(so, "libdeflate_zlib_compress_bound"));
        free_c = reinterpret_cast<void (*)(void*)>(dlsym(so, "libdeflate_free_compressor"));
        alloc_d = reinterpret_cast<void* (*)()>(dlsym(so, "libdeflate_alloc_decompressor"));
        zdecomp = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(so, "libdeflate_zlib_decompress"));
        free_d = reinterpret_cast<void (*)(void*)>(dlsym(so, "libdeflate_free_decompressor"));
        ok = alloc_c && zcomp && zbound && free_c && alloc_d && zdecomp && free_d;
    }
};
const Deflate& codec() {
    static const Deflate* d = new Deflate;  // (never unloaded)
    return *d;
}

// one thread's compressor / decompressor
struct Zip {
    void* c = nullptr;
    int level;
    explicit Zip(int lv) : level(lv) {
        if (codec().ok) c = codec().alloc_c(lv);
    }
    ~Zip() {
        if (c) codec().free_c(c);
    }
    size_t bound(size_t n) const { return c ? codec().zbound(c, n) : compressBound((uLong)n); }
    // 0: failure
    size_t pack(const void* in, size_t n, void* out, size_t cap) const {
        if (c) return codec().zcomp(c, in, n, out, cap);
        uLongf len = (uLongf)cap;
        return compress2(static_cast<Bytef*>(out), &len, static_cast<const Bytef*>(in), (uLong)n, level) == Z_OK ? (size_t)len : 0;
    }
};
struct Unzip {
    void* d = nullptr;
    Unzip() {
        if (codec().ok) d = codec().alloc_d();
    }
    ~Unzip() {
        if (d) codec().free_d(d);
    }
    // exactly n bytes expected (a strip may be followed by padding in the file, never be shorter)
    bool unpack(const void* in, size_t in_n, void* out, size_t n) const {
        if (d) {
            size_t got = 0;
            const int r = codec().zdecomp(d, in, in_n, out, n, &got);
            return r == 0 && got == n;
        }
        uLongf len = (uLongf)n;
        return uncompress(static_cast<Bytef*>(out), &len, static_cast<const Bytef*>(in), (uLong)in_n) == Z_OK && (size_t)len == n;
    }
};

// the cores this process may keep busy: its affinity mask capped by the cgroup CPU quota (a 16-CPU container on a 256-core host has
// 256 cores in its mask; threads beyond the quota only get every thread of the group stopped until the next period)
int host_cores() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    n = std::max(1, n);
    long long quota = -1, period = 0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
        std::fclose(f);
    } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
        std::fclose(g);
        if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(h, "%lld", &period) != 1) period = 0;
            std::fclose(h);
        }
    }
    if (quota > 0 && period > 0) n = (int)std::max<long long>(1, std::min<long long>(n, (quota + period - 1) / period));
    return n;
}
int thread_count(int asked, int jobs) { return std::max(1, std::min(asked > 0 ? asked : host_cores(), jobs)); }

// runs job(k, thread) for k < jobs on nt threads; the first error message wins
template <class F>
int run_jobs(int jobs, int nt, F&& job) {
    std::atomic<int> next{0};
    std::mutex mu;
    std::string err;
    auto worker = [&](int t) {
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= jobs) return;
            {
                std::lock_guard<std::mutex> g(mu);
                if (!err.empty()) return;
            }
            std::string e = job(k, t);
            if (!e.empty()) {
                std::lock_guard<std::mutex> g(mu);
                if (err.empty()) err = std::move(e);
                return;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& x : th) x.join();
    if (!err.empty()) return fail(MI_ERR_INVALID, "%s", err.c_str());
    return MI_OK;
}

// ------------------------------------------------------------------------------------------------ reading
struct TiffInfo {
    uint32_t nx = 0, ny = 0, bits = 0, comp = 1, spp = 1, rps = 0, fmt = 1, predictor = 1, planar = 1, photometric = 1;
    bool tiled = false;
    std::vector<uint64_t> off, cnt;
    int dtype() const {
        if (spp != 1) return 0;
        if (bits == 8 && fmt == 1) return 1;
        if (bits == 16 && fmt == 1) return 2;
        if (bits == 32 && fmt == 3) return 4;
        return 0;
    }
    bool fast() const {
        return dtype() != 0 && !tiled && planar == 1 && (comp == 1 || comp == 8 || comp == 32946) &&
               (predictor == 1 || (predictor == 2 && fmt == 1)) && photometric == 1 && !off.empty() && off.size() == cnt.size() && rps > 0 &&
               off.size() == (size_t)((ny + rps - 1) / rps);
    }
};

bool pread_all(int fd, void* buf, size_t n, uint64_t at) {
    char* p = static_cast<char*>(buf);
    while (n) {
        const ssize_t r = pread(fd, p, n, (off_t)at);
        if (r <= 0) {
            if (r < 0 && errno == EINTR) continue;
            return false;
        }
        p += r;
        at += (uint64_t)r;
        n -= (size_t)r;
    }
    return true;
}

// "" or why the file is not a classic little-endian TIFF this reader parses (then *other = true: a TIFF, but for the general reader)
std::string parse(int fd, TiffInfo& ti, bool* other) {
    *other = false;
    unsigned char h[8];
    if (!pread_all(fd, h, 8, 0)) return "shorter than a TIFF header";
    const bool le = h[0] == 'I' && h[1] == 'I', be = h[0] == 'M' && h[1] == 'M';
    if (!le && !be) return "not a TIFF";
    if (be) { *other = true; return "big-endian TIFF"; }
    const uint16_t magic = (uint16_t)(h[2] | h[3] << 8);
    if (magic == 43) { *other = true; return "BigTIFF"; }
    if (magic != 42) return "not a TIFF";
    const uint32_t ifd = (uint32_t)h[4] | (uint32_t)h[5] << 8 | (uint32_t)h[6] << 16 | (uint32_t)h[7] << 24;
    unsigned char nb[2];
    if (!pread_all(fd, nb, 2, ifd)) return "truncated before its directory";
    const int n = nb[0] | nb[1] << 8;
    std::vector<unsigned char> e((size_t)n * 12);
    if (n == 0 || !pread_all(fd, e.data(), e.size(), (uint64_t)ifd + 2)) return "truncated directory";
    auto u16 = [](const unsigned char* p) { return (uint32_t)(p[0] | p[1] << 8); };
    auto u32 = [](const unsigned char* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; };
    std::string err;
    auto values = [&](const unsigned char* ent, std::vector<uint64_t>& out) {
        const uint32_t type = u16(ent + 2), count = u32(ent + 4);
        const size_t sz = type == 3 ? 2 : type == 4 ? 4 : type == 1 ? 1 : 0;
        if (!sz || count == 0 || count > (1u << 24)) { err = "directory entry of a type this reader does not take"; return; }
        std::vector<unsigned char> buf((size_t)count * sz);
        if (buf.size() <= 4) std::memcpy(buf.data(), ent + 8, buf.size());
        else if (!pread_all(fd, buf.data(), buf.size(), u32(ent + 8))) { err = "truncated directory values"; return; }
        out.resize(count);
        for (uint32_t i = 0; i < count; ++i) out[i] = sz == 2 ? u16(buf.data() + 2 * i) : sz == 4 ? u32(buf.data() + 4 * i) : buf[i];
    };
    for (int i = 0; i < n && err.empty(); ++i) {
        const unsigned char* ent = e.data() + (size_t)i * 12;
        const uint32_t tag = u16(ent);
        std::vector<uint64_t> v;
        switch (tag) {
            case 256: values(ent, v); if (!v.empty()) ti.nx = (uint32_t)v[0]; break;
            case 257: values(ent, v); if (!v.empty()) ti.ny = (uint32_t)v[0]; break;
            case 258: values(ent, v); if (!v.empty()) ti.bits = (uint32_t)v[0]; if (v.size() > 1) ti.spp = (uint32_t)v.size(); break;
            case 259: values(ent, v); if (!v.empty()) ti.comp = (uint32_t)v[0]; break;
            case 262: values(ent, v); if (!v.empty()) ti.photometric = (uint32_t)v[0]; break;
            case 273: values(ent, ti.off); break;
            case 277: values(ent, v); if (!v.empty()) ti.spp = (uint32_t)v[0]; break;
            case 278: values(ent, v); if (!v.empty()) ti.rps = (uint32_t)v[0]; break;
            case 279: values(ent, ti.cnt); break;
            case 284: values(ent, v); if (!v.empty()) ti.planar = (uint32_t)v[0]; break;
            case 317: values(ent, v); if (!v.empty()) ti.predictor = (uint32_t)v[0]; break;
            case 322: case 323: case 324: case 325: ti.tiled = true; break;
            case 339: values(ent, v); if (!v.empty()) ti.fmt = (uint32_t)v[0]; break;
            default: break;
        }
    }
    if (!err.empty()) { *other = true; return err; }
    if (ti.nx == 0 || ti.ny == 0) return "no image extents";
    if (ti.rps == 0 || ti.rps > ti.ny) ti.rps = ti.ny;  // (RowsPerStrip defaults to the whole image)
    return "";
}

struct Fd {
    int fd;
    explicit Fd(const char* p) : fd(open(p, O_RDONLY | O_CLOEXEC)) {}
    ~Fd() {
        if (fd >= 0) close(fd);
    }
};

// rows [y0, y1), columns [x0, x1) of one file into dst (row pitch x1 - x0 samples); "" or the error
std::string read_slice(const char* path, int nx, int ny, int dtype, int y0, int y1, int x0, int x1, char* dst, Unzip& uz,
                       std::vector<unsigned char>& comp, std::vector<unsigned char>& raw) {
    Fd f(path);
    if (f.fd < 0) return std::string(path) + ": " + std::strerror(errno);
    TiffInfo ti;
    bool other = false;
    const std::string e = parse(f.fd, ti, &other);
    if (!e.empty()) return std::string(path) + ": " + e;
    if (!ti.fast()) return std::string(path) + ": not a file this reader decodes (strips of raw or deflate samples, one sample per pixel)";
    if ((int)ti.nx != nx || (int)ti.ny != ny || ti.dtype() != dtype) return std::string(path) + ": slice shape / type differs from the first slice";
    const size_t bps = (size_t)(dtype == 4 ? 4 : dtype), rowb = (size_t)nx * bps, outb = (size_t)(x1 - x0) * bps;
    struct stat st;
    if (fstat(f.fd, &st) != 0) return std::string(path) + ": " + std::strerror(errno);
    for (uint32_t s = (uint32_t)y0 / ti.rps; s <= (uint32_t)(y1 - 1) / ti.rps; ++s) {
        if (ti.off[s] > (uint64_t)st.st_size || ti.cnt[s] > (uint64_t)st.st_size - ti.off[s]) return std::string(path) + ": a strip lies outside the file";
        const uint32_t r0 = s * ti.rps, r1 = std::min<uint32_t>(ti.ny, r0 + ti.rps);
        const size_t want = (size_t)(r1 - r0) * rowb;
        const unsigned char* rows = nullptr;
        if (ti.comp == 1) {
            if (ti.cnt[s] < want) return std::string(path) + ": strip shorter than its rows";
            raw.resize(want);
            if (!pread_all(f.fd, raw.data(), want, ti.off[s])) return std::string(path) + ": truncated strip";
            rows = raw.data();
        } else {
            comp.resize((size_t)ti.cnt[s]);
            if (!pread_all(f.fd, comp.data(), comp.size(), ti.off[s])) return std::string(path) + ": truncated strip";
            raw.resize(want);
            if (!uz.unpack(comp.data(), comp.size(), raw.data(), want)) return std::string(path) + ": a strip does not inflate to its rows";
            rows = raw.data();
        }
        if (ti.predictor == 2) {  // horizontal differencing, per row (TIFF 6.0 section 14); samples are little-endian like the host
            unsigned char* w = raw.data();
            for (uint32_t r = 0; r < r1 - r0; ++r) {
                unsigned char* p = w + (size_t)r * rowb;
                if (bps == 1) for (int x = 1; x < nx; ++x) p[x] = (unsigned char)(p[x] + p[x - 1]);
                else if (bps == 2) { uint16_t* q = reinterpret_cast<uint16_t*>(p); for (int x = 1; x < nx; ++x) q[x] = (uint16_t)(q[x] + q[x - 1]); }
                else { uint32_t* q = reinterpret_cast<uint32_t*>(p); for (int x = 1; x < nx; ++x) q[x] += q[x - 1]; }
            }
        }
        const uint32_t a = std::max<uint32_t>(r0, (uint32_t)y0), b = std::min<uint32_t>(r1, (uint32_t)y1);
        for (uint32_t y = a; y < b; ++y)
            std::memcpy(dst + (size_t)(y - (uint32_t)y0) * outb, rows + (size_t)(y - r0) * rowb + (size_t)x0 * bps, outb);
    }
    return "";
}

// ------------------------------------------------------------------------------------------------ writing
void put16(std::vector<unsigned char>& b, uint32_t v) { b.push_back((unsigned char)(v & 255)); b.push_back((unsigned char)(v >> 8 & 255)); }
void put32(std::vector<unsigned char>& b, uint32_t v) { put16(b, v & 0xffff); put16(b, v >> 16); }

bool write_all(int fd, const void* buf, size_t n) {
    const char* p = static_cast<const char*>(buf);
    while (n) {
        const ssize_t r = write(fd, p, n);
        if (r <= 0) {
            if (r < 0 && errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}

// strip tables (when there is more than one strip), directory, header; written under a temporary name and renamed
std::string finish_file(const char* path, std::vector<unsigned char>& file, const std::vector<uint32_t>& off, const std::vector<uint32_t>& cnt, int nx,
                        int ny, int dtype, int compression, uint32_t rps, int predictor = 1) {
    const size_t bps = (size_t)(dtype == 4 ? 4 : dtype);
    const uint32_t ns = (uint32_t)off.size();
    uint32_t off_tab = 0, cnt_tab = 0;
    if (ns > 1) {
        off_tab = (uint32_t)file.size();
        for (uint32_t v : off) put32(file, v);
        cnt_tab = (uint32_t)file.size();
        for (uint32_t v : cnt) put32(file, v);
    }
    const uint32_t ifd = (uint32_t)file.size();
    struct Ent { uint16_t tag, type; uint32_t count, value; };
    const Ent ents[] = {
        {256, 4, 1, (uint32_t)nx}, {257, 4, 1, (uint32_t)ny}, {258, 3, 1, (uint32_t)(8 * bps)}, {259, 3, 1, compression ? 8u : 1u},
        {262, 3, 1, 1u}, {273, 4, ns, ns > 1 ? off_tab : off[0]}, {277, 3, 1, 1u}, {278, 4, 1, rps}, {279, 4, ns, ns > 1 ? cnt_tab : cnt[0]},
        {284, 3, 1, 1u}, {317, 3, 1, (uint32_t)predictor}, {339, 3, 1, dtype == 4 ? 3u : 1u},
    };
    const int n_ent = (int)(sizeof ents / sizeof ents[0]) - (predictor == 1 ? 1 : 0);   // (the tag is left out when nothing is predicted)
    put16(file, (uint32_t)n_ent);
    for (const Ent& e : ents) {
        if (e.tag == 317 && predictor == 1) continue;
        put16(file, e.tag); put16(file, e.type); put32(file, e.count); put32(file, e.value);
    }
    put32(file, 0);  // no further directory
    file[0] = 'I'; file[1] = 'I'; file[2] = 42; file[3] = 0;
    file[4] = (unsigned char)(ifd & 255); file[5] = (unsigned char)(ifd >> 8 & 255); file[6] = (unsigned char)(ifd >> 16 & 255); file[7] = (unsigned char)(ifd >> 24);
    const std::string tmp = std::string(path) + ".tmp";
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) return tmp + ": " + std::strerror(errno);
    const bool ok = write_all(fd, file.data(), file.size());
    const int ce = close(fd);
    if (!ok || ce != 0) { unlink(tmp.c_str()); return tmp + ": write failed: " + std::strerror(errno); }
    if (rename(tmp.c_str(), path) != 0) { unlink(tmp.c_str()); return std::string(path) + ": " + std::strerror(errno); }
    return "";
}

// header | strips | directory.  Strips of ~1 MiB of rows: a reader that wants a few rows inflates a few strips.
std::string write_slice(const char* path, const char* data, int nx, int ny, int dtype, int compression, Zip& zp, std::vector<unsigned char>& file) {
    const size_t bps = (size_t)(dtype == 4 ? 4 : dtype), rowb = (size_t)nx * bps;
    const uint32_t rps = (uint32_t)std::max<size_t>(1, std::min<size_t>((size_t)ny, ((size_t)1 << 20) / std::max<size_t>(1, rowb)));
    const uint32_t ns = ((uint32_t)ny + rps - 1) / rps;
    std::vector<uint32_t> off(ns), cnt(ns);
    file.clear();
    file.reserve((size_t)ny * rowb / (compression ? 2 : 1) + 4096);
    file.resize(8);
    for (uint32_t s = 0; s < ns; ++s) {
        const uint32_t r0 = s * rps, r1 = std::min<uint32_t>((uint32_t)ny, r0 + rps);
        const size_t n = (size_t)(r1 - r0) * rowb, at = file.size();
        if (compression) {
            const size_t cap = zp.bound(n);
            file.resize(at + cap);
            const size_t got = zp.pack(data + (size_t)r0 * rowb, n, file.data() + at, cap);
            if (!got) return std::string(path) + ": deflate failed";
            file.resize(at + got);
            cnt[s] = (uint32_t)got;
        } else {
            file.insert(file.end(), reinterpret_cast<const unsigned char*>(data) + (size_t)r0 * rowb,
                        reinterpret_cast<const unsigned char*>(data) + (size_t)r0 * rowb + n);
            cnt[s] = (uint32_t)n;
        }
        off[s] = (uint32_t)at;
        if (file.size() & 1) file.push_back(0);  // (word alignment of what follows)
        if (file.size() > 0xfff00000ull) return std::string(path) + ": a slice of more than 4 GB needs BigTIFF";
    }
    return finish_file(path, file, off, cnt, nx, ny, dtype, compression, rps);
}

// ------------------------------------------------------------------------------------------------ deflate on the device
// The writer above spends the host's cores on deflate: 2.7 GB/s on the 16 CPUs of a one-GPU container, while the slab it writes was
// computed on the device in a fraction of that time.  mi_tiff_write_series_device compresses where the samples are: every strip
// becomes ONE dynamic-Huffman deflate block without string matching -- microscope samples are noise on a smooth field: what deflate
// saves on them at level 1 is the entropy coding, matches are rare -- in three steps:
//   k_strip_hist    per strip: histogram of its bytes and the two sums of its Adler-32;
//   host            per strip: length-limited Huffman code (15 bits) of the 256 literals + end-of-block, the block header
//                   (RFC 1951 3.2.7: every code length sent plainly through the code-length code, no run symbols);
//   k_strip_encode  per strip: a work-group walks over tiles of 8 KB; a lane takes 32 bytes, the lanes' bit counts are scanned, every
//                   lane ORs its codes into the tile's bit image in LDS, the image goes out as whole words.
// The host then only frames the streams (zlib header, header bits, Adler-32) and writes the files.  Any inflater reads the result;
// a strip whose bytes are all equally likely costs 130 bytes more than stored.
struct HuffCode {
    uint8_t len[288];
    uint16_t code[288];  // bit-reversed: emitted least significant bit first
};

// code lengths of at most `maxlen` bits for the symbols with f[i] > 0 (n <= 288, at least two of them used)
void huff_lengths(const uint32_t* f, int n, int maxlen, uint8_t* len) {
    struct Node { uint64_t w; int l, r; };
    std::vector<Node> nodes;
    std::vector<int> used;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (f[i]) used.push_back(i);
    }
    if (used.size() == 1) { len[used[0]] = 1; return; }
    std::sort(used.begin(), used.end(), [&](int a, int b) { return f[a] != f[b] ? f[a] < f[b] : a < b; });
    const int m = (int)used.size();
    nodes.reserve(2 * (size_t)m);
    for (int i = 0; i < m; ++i) nodes.push_back({f[used[(size_t)i]], -1, -1});
    // two queues: leaves in ascending order, internal nodes in creation order (ascending too)
    int a = 0, b = m;
    auto take = [&]() {
        int pick;
        if (a < m && (b >= (int)nodes.size() || nodes[(size_t)a].w <= nodes[(size_t)b].w)) pick = a++;
        else pick = b++;
        return pick;
    };
    for (int k = 0; k < m - 1; ++k) {
        const int x = take(), y = take();
        nodes.push_back({nodes[(size_t)x].w + nodes[(size_t)y].w, x, y});
    }
    std::vector<int> depth(nodes.size(), 0);
    for (int i = (int)nodes.size() - 1; i >= m; --i) {
        depth[(size_t)nodes[(size_t)i].l] = depth[(size_t)i] + 1;
        depth[(size_t)nodes[(size_t)i].r] = depth[(size_t)i] + 1;
    }
    // the number of codes of every length, limited to maxlen (the usual Kraft repair: a code moves down from the deepest occupied
    // level above, its two children take the place), then handed out by frequency: the rarest symbols get the longest codes
    int cnt[64] = {0};
    for (int i = 0; i < m; ++i) cnt[std::min(depth[(size_t)i], 63)]++;
    for (int i = maxlen + 1; i < 64; ++i) { cnt[maxlen] += cnt[i]; cnt[i] = 0; }
    uint64_t total = 0;
    for (int i = maxlen; i > 0; --i) total += (uint64_t)cnt[i] << (maxlen - i);
    while (total != (1ull << maxlen)) {
        cnt[maxlen]--;
        for (int i = maxlen - 1; i > 0; --i)
            if (cnt[i]) { cnt[i]--; cnt[i + 1] += 2; break; }
        total--;
    }
    int at = 0;
    for (int l = maxlen; l > 0; --l)
        for (int c = 0; c < cnt[l]; ++c) len[used[(size_t)at++]] = (uint8_t)l;
}

// canonical codes of RFC 1951 3.2.2, each reversed over its own length
void huff_codes(const uint8_t* len, int n, uint16_t* code) {
    int bl[16] = {0}, next[16] = {0};
    for (int i = 0; i < n; ++i) bl[len[i]]++;
    bl[0] = 0;
    int c = 0;
    for (int b = 1; b <= 15; ++b) { c = (c + bl[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; ++i) {
        code[i] = 0;
        if (!len[i]) continue;
        const int v = next[len[i]]++;
        int r = 0;
        for (int k = 0; k < len[i]; ++k) r |= ((v >> k) & 1) << (len[i] - 1 - k);
        code[i] = (uint16_t)r;
    }
}

struct BitWriter {
    std::vector<unsigned char> b;
    uint64_t n = 0;
    void put(uint32_t v, int bits) {
        for (int k = 0; k < bits; ++k, ++n) {
            if ((n & 7) == 0) b.push_back(0);
            b.back() |= (unsigned char)(((v >> k) & 1u) << (n & 7));
        }
    }
};

// the block header of one strip (bits before the first literal) and the table the encode kernel takes: [s] = len << 16 | code
void strip_code(const uint32_t* hist, BitWriter& hdr, uint32_t* table /* 257 */, uint64_t* payload_bits) {
    uint32_t f[257];
    for (int i = 0; i < 256; ++i) f[i] = hist[i];
    f[256] = 1;  // end of block
    HuffCode lit;
    huff_lengths(f, 257, 15, lit.len);
    huff_codes(lit.len, 257, lit.code);
    // 257 literal / length code lengths + two distance codes of one bit each (none is ever used; two, because some inflaters want a
    // complete distance code) sent one by one through the code-length code
    uint8_t all[259];
    for (int i = 0; i < 257; ++i) all[i] = lit.len[i];
    all[257] = all[258] = 1;
    uint32_t cf[19] = {0};
    for (uint8_t v : all) cf[v]++;
    uint8_t cl[19];
    uint16_t cc[19];
    huff_lengths(cf, 19, 7, cl);
    huff_codes(cl, 19, cc);
    hdr.put(1, 1);   // BFINAL
    hdr.put(2, 2);   // BTYPE: dynamic Huffman
    hdr.put(0, 5);   // HLIT: 257 codes
    hdr.put(1, 5);   // HDIST: 2 codes
    hdr.put(15, 4);  // HCLEN: all 19 code-length code lengths
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int i = 0; i < 19; ++i) hdr.put(cl[order[i]], 3);
    for (uint8_t v : all) hdr.put(cc[v], cl[v]);
    uint64_t bits = lit.len[256];
    for (int i = 0; i < 256; ++i) bits += (uint64_t)hist[i] * lit.len[i];
    *payload_bits = bits;
    for (int i = 0; i < 257; ++i) table[i] = (uint32_t)lit.len[i] << 16 | lit.code[i];
}

constexpr int kEncThreads = 256, kEncBytes = 32, kTileBytes = kEncThreads * kEncBytes;   // 8 KB of samples per tile
constexpr int kTileWords = kTileBytes * 15 / 8 / 4 + 4;                                     // its bit image at 15 bits per byte

struct StripGeom {
    size_t slice_bytes, rowb;
    uint32_t rps, ns, ny, bps;
    __host__ __device__ size_t start(uint32_t strip) const { return (size_t)(strip / ns) * slice_bytes + (size_t)(strip % ns) * rps * rowb; }
    __host__ __device__ uint32_t bytes(uint32_t strip) const {
        const uint32_t r0 = (strip % ns) * rps, r1 = r0 + rps < ny ? r0 + rps : ny;
        return (uint32_t)((r1 - r0) * rowb);
    }
};

// NB bytes of a strip from byte j0 on (cnt of them exist), as stored: plain, or -- pred -- every 8 / 16-bit sample minus its left
// neighbour in the row (TIFF 6.0 section 14, horizontal differencing; the first sample of a row stays).  j0 is a multiple of NB.
template <int NB>
__device__ __forceinline__ void strip_bytes(const unsigned char* __restrict__ p, uint32_t j0, uint32_t cnt, bool aligned, uint32_t bps, uint32_t rowb,
                                            bool pred, unsigned char (&b)[NB]) {
    if (aligned && cnt == (uint32_t)NB) {
#pragma unroll
        for (int q = 0; q < NB / 16; ++q) *reinterpret_cast<uint4*>(b + 16 * q) = *reinterpret_cast<const uint4*>(p + j0 + 16 * q);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)NB; ++k) b[k] = k < cnt ? p[j0 + k] : 0;
    }
    if (!pred || cnt == 0) return;
    uint32_t col = j0 % rowb;   // byte column of the chunk's first byte
    if (bps == 2) {
        uint32_t prev = col ? (uint32_t)p[j0 - 2] | (uint32_t)p[j0 - 1] << 8 : 0;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)NB / 2; ++k) {
            const uint32_t v = (uint32_t)b[2 * k] | (uint32_t)b[2 * k + 1] << 8;
            const uint32_t d = (v - (col ? prev : 0u)) & 0xffffu;
            b[2 * k] = (unsigned char)(d & 255);
            b[2 * k + 1] = (unsigned char)(d >> 8);
            prev = v;
            col += 2;
            if (col >= rowb) col -= rowb;
        }
    } else {
        uint32_t prev = col ? (uint32_t)p[j0 - 1] : 0;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)NB; ++k) {
            const uint32_t v = b[k];
            b[k] = (unsigned char)((v - (col ? prev : 0u)) & 255);
            prev = v;
            col += 1;
            if (col >= rowb) col -= rowb;
        }
    }
}

// per strip: hist[strip][v][256] and sums[strip][v][2] for v = 0 (the bytes as they are) and -- `both` -- v = 1 (horizontally differenced)
__global__ __launch_bounds__(256) void k_strip_hist(const unsigned char* __restrict__ base, StripGeom g, int both, uint32_t* __restrict__ hist,
                                                     unsigned long long* __restrict__ sums) {
    __shared__ uint32_t h[2][2][256];
    __shared__ unsigned long long ssum[4];
    const uint32_t strip = blockIdx.x, n = g.bytes(strip), tid = threadIdx.x;
    const unsigned char* p = base + g.start(strip);
    for (int i = tid; i < 1024; i += 256) (&h[0][0][0])[i] = 0;
    if (tid < 4) ssum[tid] = 0;
    __syncthreads();
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    for (int v = 0; v < (both ? 2 : 1); ++v) {
        unsigned long long s1 = 0, s2 = 0;
        uint32_t* mine = h[v][tid & 1];
        for (uint32_t i = tid * 16; i < n; i += 256 * 16) {
            unsigned char b[16];
            const uint32_t m = n - i < 16 ? n - i : 16;
            strip_bytes<16>(p, i, m, aligned, g.bps, (uint32_t)g.rowb, v == 1, b);
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) {
                if (k < m) {
                    atomicAdd(&mine[b[k]], 1u);
                    s1 += b[k];
                    s2 += (unsigned long long)(n - (i + k)) * b[k];
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_down(s1, o, 64);
            s2 += __shfl_down(s2, o, 64);
        }
        if ((tid & 63) == 0) {
            atomicAdd(&ssum[2 * v], s1);
            atomicAdd(&ssum[2 * v + 1], s2);
        }
    }
    __syncthreads();
    hist[(size_t)strip * 512 + tid] = h[0][0][tid] + h[0][1][tid];
    hist[(size_t)strip * 512 + 256 + tid] = h[1][0][tid] + h[1][1][tid];
    if (tid < 4) sums[(size_t)strip * 4 + tid] = ssum[tid];
}

// the literals of a strip as bits, from bit hbits[strip] of its output slot on (the host puts the block header in front)
__global__ __launch_bounds__(kEncThreads) void k_strip_encode(const unsigned char* __restrict__ base, StripGeom g, const uint32_t* __restrict__ tables,
                                                              const uint32_t* __restrict__ hbits, unsigned char* __restrict__ out, size_t cap) {
    __shared__ uint32_t tab[257];
    __shared__ uint32_t img[kTileWords];
    __shared__ uint32_t wsum[kEncThreads / 64];
    const uint32_t strip = blockIdx.x, n = g.bytes(strip), tid = threadIdx.x;
    const unsigned char* p = base + g.start(strip);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (size_t)strip * cap);
    for (int i = tid; i < 257; i += kEncThreads) tab[i] = tables[(size_t)strip * 257 + i];
    for (int i = tid; i < kTileWords; i += kEncThreads) img[i] = 0;
    __syncthreads();
    const bool pred = (hbits[strip] >> 31) != 0;                    // (top bit: the strip is stored horizontally differenced)
    unsigned long long bitpos = hbits[strip] & 0x7fffffffu;          // where the next tile's first bit goes in the strip's stream
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    for (uint32_t t0 = 0; t0 < n; t0 += kTileBytes) {
        const uint32_t j0 = t0 + tid * kEncBytes;
        const uint32_t cnt = j0 < n ? (n - j0 < (uint32_t)kEncBytes ? n - j0 : (uint32_t)kEncBytes) : 0;
        unsigned char b[kEncBytes];
        strip_bytes<kEncBytes>(p, j0 < n ? j0 : 0, cnt, aligned, g.bps, (uint32_t)g.rowb, pred, b);
        const bool last = cnt > 0 && j0 + cnt == n;   // this lane appends the end-of-block symbol
        uint32_t mybits = 0;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)kEncBytes; ++k) mybits += k < cnt ? tab[b[k]] >> 16 : 0;
        if (last) mybits += tab[256] >> 16;
        // exclusive scan of the lanes' bit counts over the work-group
        uint32_t incl = mybits;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if ((int)(tid & 63) >= o) incl += v;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, tile_bits = 0;
        for (int w = 0; w < kEncThreads / 64; ++w) {
            if (w < (int)(tid >> 6)) before += wsum[w];
            tile_bits += wsum[w];
        }
        const uint32_t rel = (uint32_t)(bitpos & 31);   // the image's word 0 is the stream's word bitpos / 32
        uint32_t pos = rel + before + incl - mybits;
        uint32_t w = pos >> 5;
        unsigned long long acc = 0;
        uint32_t nb = pos & 31;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)kEncBytes; ++k) {
            if (k < cnt) {
                const uint32_t c = tab[b[k]];
                acc |= (unsigned long long)(c & 0xffffu) << nb;
                nb += c >> 16;
                if (nb >= 32) {
                    atomicOr(&img[w++], (uint32_t)acc);
                    acc >>= 32;
                    nb -= 32;
                }
            }
        }
        if (last) {
            const uint32_t c = tab[256];
            acc |= (unsigned long long)(c & 0xffffu) << nb;
            nb += c >> 16;
            if (nb >= 32) {
                atomicOr(&img[w++], (uint32_t)acc);
                acc >>= 32;
                nb -= 32;
            }
        }
        if (nb > 0) atomicOr(&img[w], (uint32_t)acc);
        __syncthreads();
        // whole words out; the last, partial one stays behind as word 0 of the next tile's image
        const uint32_t end = rel + tile_bits, full = end >> 5;
        uint32_t* d = dst + (bitpos >> 5);
        for (uint32_t i = tid; i < full; i += kEncThreads) d[i] = img[i];
        const uint32_t carry = img[full];
        __syncthreads();
        for (uint32_t i = tid; i <= full; i += kEncThreads) img[i] = 0;
        __syncthreads();
        if (tid == 0) img[0] = carry;
        bitpos += tile_bits;
        __syncthreads();
    }
    if (tid == 0) dst[bitpos >> 5] = img[0];
}

// frames the device's streams into files: per strip zlib header | block header bits OR literals | Adler-32
struct PackedStrip {
    std::vector<unsigned char> hdr;  // the block header, its last byte partly filled
    uint64_t hbits = 0, payload_bits = 0;
    uint32_t adler = 1;
};

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        if (hipMalloc(&p, n) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return fail(MI_ERR_NOMEM, "mi_tiff_write_series_device: hipMalloc(%zu bytes) failed", n); }
        return MI_OK;
    }
};
struct PinMem {
    void* p = nullptr;
    ~PinMem() { if (p) (void)hipHostFree(p); }
    int alloc(size_t n) {
        if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return fail(MI_ERR_NOMEM, "mi_tiff_write_series_device: %zu bytes of pinned memory", n); }
        return MI_OK;
    }
};

}  // namespace

extern "C" const char* mi_tiff_codec(void) { return codec().ok ? "libdeflate" : "zlib"; }

extern "C" int mi_tiff_info(const char* path, int* nx, int* ny, int* dtype, int* fast) {
    MI_REQUIRE(path && nx && ny && dtype && fast, "mi_tiff_info: null pointer");
    Fd f(path);
    if (f.fd < 0) return fail(MI_ERR_INVALID, "%s: %s", path, std::strerror(errno));
    TiffInfo ti;
    bool other = false;
    const std::string e = parse(f.fd, ti, &other);
    if (!e.empty() && !other) return fail(MI_ERR_INVALID, "%s: %s", path, e.c_str());
    *nx = (int)ti.nx;
    *ny = (int)ti.ny;
    *dtype = e.empty() ? ti.dtype() : 0;
    *fast = e.empty() && ti.fast() ? 1 : 0;
    return MI_OK;
}

extern "C" int mi_tiff_read_box(const char* const* paths, int n, int nx, int ny, int dtype, int y0, int y1, int x0, int x1, void* out,
                                int n_threads) {
    MI_REQUIRE(paths && out, "mi_tiff_read_box: null pointer");
    MI_REQUIRE(n >= 0 && nx > 0 && ny > 0 && (dtype == 1 || dtype == 2 || dtype == 4), "mi_tiff_read_box: invalid extents or sample type");
    MI_REQUIRE(0 <= y0 && y0 < y1 && y1 <= ny && 0 <= x0 && x0 < x1 && x1 <= nx, "mi_tiff_read_box: box [%d, %d) x [%d, %d) outside %d x %d", y0, y1,
               x0, x1, ny, nx);
    if (n == 0) return MI_OK;
    const int nt = thread_count(n_threads, n);
    const size_t slice_bytes = (size_t)(y1 - y0) * (size_t)(x1 - x0) * (size_t)(dtype == 4 ? 4 : dtype);
    std::vector<Unzip> uz((size_t)nt);
    std::vector<std::vector<unsigned char>> comp((size_t)nt), raw((size_t)nt);
    return run_jobs(n, nt, [&](int k, int t) {
        return read_slice(paths[k], nx, ny, dtype, y0, y1, x0, x1, static_cast<char*>(out) + (size_t)k * slice_bytes, uz[(size_t)t], comp[(size_t)t],
                          raw[(size_t)t]);
    });
}

extern "C" int mi_tiff_write_series(const char* const* paths, int nz, const void* vol, int dtype, int nx, int ny, int compression, int level,
                                    int n_threads, int* written) {
    MI_REQUIRE(paths && vol, "mi_tiff_write_series: null pointer");
    MI_REQUIRE(nz >= 0 && nx > 0 && ny > 0 && (dtype == 1 || dtype == 2 || dtype == 4), "mi_tiff_write_series: invalid extents or sample type");
    MI_REQUIRE((compression == 0 || compression == 1) && level >= 1 && level <= 9, "mi_tiff_write_series: compression 0 / 1, level 1 .. 9");
    if (written) *written = 0;
    if (nz == 0) return MI_OK;
    const int nt = thread_count(n_threads, nz);
    const size_t slice_bytes = (size_t)nx * ny * (size_t)(dtype == 4 ? 4 : dtype);
    std::vector<Zip*> zp((size_t)nt, nullptr);
    std::vector<std::vector<unsigned char>> buf((size_t)nt);
    std::atomic<int> made{0};
    const int rc = run_jobs(nz, nt, [&](int k, int t) -> std::string {
        struct stat st;
        if (stat(paths[k], &st) == 0) return "";  // LsDeconv.m:1120-1132: slices that exist are kept
        if (!zp[(size_t)t]) zp[(size_t)t] = new Zip(level);
        std::string e = write_slice(paths[k], static_cast<const char*>(vol) + (size_t)k * slice_bytes, nx, ny, dtype, compression, *zp[(size_t)t],
                                    buf[(size_t)t]);
        if (e.empty()) made.fetch_add(1);
        return e;
    });
    for (Zip* z : zp) delete z;
    if (written) *written = made.load();
    return rc;
}

extern "C" int mi_tiff_write_series_device(int dev, void* stream, const char* const* paths, int nz, const void* vol, int dtype, int nx, int ny,
                                           int n_threads, int* written) {
    MI_TRY(mi::use_device(dev));
    MI_REQUIRE(paths && vol, "mi_tiff_write_series_device: null pointer");
    MI_REQUIRE(nz >= 0 && nx > 0 && ny > 0 && (dtype == 1 || dtype == 2 || dtype == 4), "mi_tiff_write_series_device: invalid extents or sample type");
    if (written) *written = 0;
    if (nz == 0) return MI_OK;
    hipStream_t s = mi::as_stream(stream);
    StripGeom g;
    const size_t bps = (size_t)(dtype == 4 ? 4 : dtype);
    g.rowb = (size_t)nx * bps;
    g.slice_bytes = g.rowb * (size_t)ny;
    g.ny = (uint32_t)ny;
    g.bps = (uint32_t)bps;
    g.rps = (uint32_t)std::max<size_t>(1, std::min<size_t>((size_t)ny, ((size_t)1 << 20) / std::max<size_t>(1, g.rowb)));
    g.ns = ((uint32_t)ny + g.rps - 1) / g.rps;
    MI_REQUIRE(g.slice_bytes < 0xfff00000ull, "mi_tiff_write_series_device: a slice of more than 4 GB needs BigTIFF");
    const size_t strip_max = (size_t)g.rps * g.rowb;
    const size_t cap = ((strip_max + strip_max / 32 + 4096) + 15) & ~(size_t)15;   // a strip's stream: at most ~8.01 bits per byte + header
    // slices per batch: ~512 MB of samples at a time
    const int per_batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)nz, ((size_t)512 << 20) / std::max<size_t>(1, g.slice_bytes)));
    const size_t bs = (size_t)per_batch * g.ns;   // strips per batch
    DevMem d_hist, d_sums, d_tab, d_hbits, d_out;
    PinMem h_hist, h_sums, h_tab, h_hbits, h_out;
    MI_TRY(d_hist.alloc(bs * 512 * 4));
    MI_TRY(d_sums.alloc(bs * 32));
    MI_TRY(d_tab.alloc(bs * 257 * 4));
    MI_TRY(d_hbits.alloc(bs * 4));
    MI_TRY(d_out.alloc(bs * cap));
    MI_TRY(h_hist.alloc(bs * 512 * 4));
    MI_TRY(h_sums.alloc(bs * 32));
    MI_TRY(h_tab.alloc(bs * 257 * 4));
    MI_TRY(h_hbits.alloc(bs * 4));
    MI_TRY(h_out.alloc(bs * cap));
    const int nt = thread_count(n_threads, std::max(1, (int)bs));
    std::atomic<int> made{0};
    std::vector<PackedStrip> ps(bs);
    std::vector<int> pred_of;
    std::vector<std::vector<unsigned char>> filebuf((size_t)nt);
    for (int z0 = 0; z0 < nz; z0 += per_batch) {
        const int nb = std::min(per_batch, nz - z0);
        const uint32_t nstrip = (uint32_t)nb * g.ns;
        const unsigned char* base = static_cast<const unsigned char*>(vol) + (size_t)z0 * g.slice_bytes;
        // integer samples: the strips are also counted horizontally differenced (TIFF predictor 2); a slice is stored that way when
        // that makes it smaller -- smooth content: entropy coding without string matching then beats deflate level 1, noise: never
        const int both = dtype != 4 ? 1 : 0;
        hipLaunchKernelGGL(k_strip_hist, dim3(nstrip), dim3(256), 0, s, base, g, both, static_cast<uint32_t*>(d_hist.p),
                           static_cast<unsigned long long*>(d_sums.p));
        MI_TRY(mi::launch_check("k_strip_hist"));
        MI_HIP(hipMemcpyAsync(h_hist.p, d_hist.p, (size_t)nstrip * 512 * 4, hipMemcpyDeviceToHost, s));
        MI_HIP(hipMemcpyAsync(h_sums.p, d_sums.p, (size_t)nstrip * 32, hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        // codes and block headers, a slice per task (its strips share the choice of the predictor)
        pred_of.assign((size_t)nb, 0);
        MI_TRY(run_jobs(nb, std::min(nt, nb), [&](int k, int) -> std::string {
            std::vector<uint32_t> tab2((size_t)g.ns * 257);
            std::vector<PackedStrip> alt(g.ns);
            uint64_t bits[2] = {0, 0};
            for (int v = 0; v < (both ? 2 : 1); ++v) {
                for (uint32_t si = 0; si < g.ns; ++si) {
                    const size_t strip = (size_t)k * g.ns + si;
                    PackedStrip& q = v ? alt[si] : ps[strip];
                    BitWriter bw;
                    strip_code(static_cast<const uint32_t*>(h_hist.p) + strip * 512 + (size_t)v * 256, bw,
                               v ? tab2.data() + (size_t)si * 257 : static_cast<uint32_t*>(h_tab.p) + strip * 257, &q.payload_bits);
                    q.hbits = bw.n;
                    q.hdr.swap(bw.b);
                    const unsigned long long* sm = static_cast<const unsigned long long*>(h_sums.p) + strip * 4 + (size_t)v * 2;
                    const uint64_t n = g.bytes((uint32_t)strip);
                    q.adler = (uint32_t)(((n + sm[1]) % 65521ull) << 16 | ((1ull + sm[0]) % 65521ull));
                    bits[v] += q.hbits + q.payload_bits;
                }
            }
            const bool use_pred = both && bits[1] < bits[0];
            pred_of[(size_t)k] = use_pred ? 1 : 0;
            for (uint32_t si = 0; si < g.ns; ++si) {
                const size_t strip = (size_t)k * g.ns + si;
                if (use_pred) {
                    ps[strip] = std::move(alt[si]);
                    std::memcpy(static_cast<uint32_t*>(h_tab.p) + strip * 257, tab2.data() + (size_t)si * 257, 257 * 4);
                }
                const PackedStrip& q = ps[strip];
                static_cast<uint32_t*>(h_hbits.p)[strip] = (uint32_t)q.hbits | (use_pred ? 0x80000000u : 0u);
                if ((q.hbits + q.payload_bits + 7) / 8 + 8 > cap) return "a strip's deflate stream does not fit its slot";
            }
            return "";
        }));
        MI_HIP(hipMemcpyAsync(d_tab.p, h_tab.p, (size_t)nstrip * 257 * 4, hipMemcpyHostToDevice, s));
        MI_HIP(hipMemcpyAsync(d_hbits.p, h_hbits.p, (size_t)nstrip * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_strip_encode, dim3(nstrip), dim3(kEncThreads), 0, s, base, g, static_cast<const uint32_t*>(d_tab.p),
                           static_cast<const uint32_t*>(d_hbits.p), static_cast<unsigned char*>(d_out.p), cap);
        MI_TRY(mi::launch_check("k_strip_encode"));
        MI_HIP(hipMemcpyAsync(h_out.p, d_out.p, (size_t)nstrip * cap, hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        // files, a slice per task
        MI_TRY(run_jobs(nb, std::min(nt, nb), [&](int k, int t) -> std::string {
            const char* path = paths[z0 + k];
            struct stat st;
            if (stat(path, &st) == 0) return "";   // LsDeconv.m:1120-1132: slices that exist are kept
            std::vector<unsigned char>& file = filebuf[(size_t)t];
            file.clear();
            file.resize(8);
            std::vector<uint32_t> off(g.ns), cnt(g.ns);
            for (uint32_t si = 0; si < g.ns; ++si) {
                const size_t strip = (size_t)k * g.ns + si;
                const PackedStrip& q = ps[strip];
                const size_t nbytes = (size_t)((q.hbits + q.payload_bits + 7) / 8), at = file.size();
                file.resize(at + 2 + nbytes + 4);
                unsigned char* w = file.data() + at;
                w[0] = 0x78;
                w[1] = 0x01;
                std::memcpy(w + 2, static_cast<const unsigned char*>(h_out.p) + strip * cap, nbytes);
                // (the kernel wrote from the word of its first bit on, zeros below that bit; the header's whole bytes are assigned, its
                //  last, partly filled byte is OR-ed onto the first literals)
                for (size_t i = 0; i < q.hdr.size(); ++i) {
                    if (i < q.hbits / 8) w[2 + i] = q.hdr[i];
                    else w[2 + i] |= q.hdr[i];
                }
                w[2 + nbytes + 0] = (unsigned char)(q.adler >> 24);
                w[2 + nbytes + 1] = (unsigned char)(q.adler >> 16);
                w[2 + nbytes + 2] = (unsigned char)(q.adler >> 8);
                w[2 + nbytes + 3] = (unsigned char)q.adler;
                off[si] = (uint32_t)at;
                cnt[si] = (uint32_t)(2 + nbytes + 4);
                if (file.size() & 1) file.push_back(0);
            }
            std::string e = finish_file(path, file, off, cnt, nx, ny, dtype, 1, g.rps, pred_of[(size_t)k] ? 2 : 1);
            if (e.empty()) made.fetch_add(1);
            return e;
        }));
    }
    if (written) *written = made.load();
    return MI_OK;
}
