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
    for (uint32_t s = (uint32_t)y0 / ti.rps; s <= (uint32_t)(y1 - 1) / ti.rps; ++s) {
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
    // strip tables (when there is more than one strip), then the directory
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
        {284, 3, 1, 1u}, {339, 3, 1, dtype == 4 ? 3u : 1u},
    };
    put16(file, (uint32_t)(sizeof ents / sizeof ents[0]));
    for (const Ent& e : ents) { put16(file, e.tag); put16(file, e.type); put32(file, e.count); put32(file, e.value); }
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
