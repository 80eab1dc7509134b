// BAL text loader, behaviour of /root/reference/src/BalDataloader.cpp:9-163:
// every scalar is parsed as float (:32-33,:44-45,:55-56) and widened to double only in
// initialize (:131-157), which perturbs the state with Marsaglia-polar normals drawn from the
// C library's rand() (unseeded => glibc seed-1 stream), six per camera then three per point.
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <sys/stat.h>
#include <thread>
#include <tuple>
#include <vector>

#include "bundleadjust/BalDataloader.h"

// Fast ingest.  The reference reads the file with serial `ifstream >>`
// (src/BalDataloader.cpp:21-58): minutes for the 1.4 GB Final-13682 file against a solve of seconds.
// Here the file is read in one piece, cut into byte ranges at whitespace, and the ranges are tokenised by
// worker threads with std::from_chars; a token's position in the stream tells which field it is, so the
// result is identical to the serial parse (every scalar rounded to float exactly as `>> float` does).
namespace {

inline bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

struct Range { const char *b, *e; long long first_token; };

long long count_tokens(const char *b, const char *e) {
    long long n = 0;
    bool in = false;
    for (const char *p = b; p < e; ++p) {
        const bool w = is_ws(*p);
        if (!w && !in) ++n;
        in = !w;
    }
    return n;
}

bool parse_token(const char *b, const char *e, long long &out) {
    if (b < e && *b == '+') ++b;   // from_chars takes no leading plus sign, `>>` does
    auto r = std::from_chars(b, e, out);
    return r.ec == std::errc() && r.ptr == e;
}

// strtof on a copy of the token: the slow, always-right path
bool parse_float_slow(const char *b, const char *e, float &out) {
    char tmp[128];
    const size_t n = (size_t) (e - b);
    if (n == 0 || n >= sizeof(tmp)) return false;
    memcpy(tmp, b, n);
    tmp[n] = 0;
    char *endp = nullptr;
    out = strtof(tmp, &endp);
    return endp == tmp + n;
}

// Decimal text -> float, rounded exactly as strtof / `ifstream >> float` round it.  (libstdc++ 11's
// from_chars<float> goes through strtod under a temporary locale, which serialises the worker threads.)
// Fast path: at most 19 significant digits (< 2^64) and |exponent| <= 27 are exact in x87 extended precision
// (64-bit significand; 5^27 < 2^64), so one multiplication or division gives the decimal correctly rounded
// to 64 bits; that rounds to the same float as the decimal itself unless it lies within its own rounding
// error of a float rounding boundary -- those tokens, and everything unusual, go to strtof.
static_assert(sizeof(long double) >= 10 && __LDBL_MANT_DIG__ == 64, "x87 extended precision expected");
bool parse_token(const char *b, const char *e, float &out) {
    static const long double p10[28] = {1e0L, 1e1L, 1e2L, 1e3L, 1e4L, 1e5L, 1e6L, 1e7L, 1e8L, 1e9L, 1e10L, 1e11L, 1e12L, 1e13L,
                                        1e14L, 1e15L, 1e16L, 1e17L, 1e18L, 1e19L, 1e20L, 1e21L, 1e22L, 1e23L, 1e24L, 1e25L,
                                        1e26L, 1e27L};
    const char *p = b;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; ++p; }
    unsigned long long m = 0;
    int nd = 0, frac = 0;
    bool any = false, dot = false;
    for (; p < e; ++p) {
        const char c = *p;
        if (c >= '0' && c <= '9') {
            any = true;
            if (m == 0 && c == '0') { if (dot) ++frac; continue; }   // leading zeros carry no digits
            if (nd >= 19) return parse_float_slow(b, e, out);
            m = m * 10 + (unsigned) (c - '0');
            ++nd;
            if (dot) ++frac;
        } else if (c == '.' && !dot) dot = true;
        else break;
    }
    if (!any) return parse_float_slow(b, e, out);   // nan, inf, garbage
    int ex = 0;
    if (p < e && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < e && (*p == '+' || *p == '-')) { eneg = *p == '-'; ++p; }
        if (p >= e) return parse_float_slow(b, e, out);   // "1e": strtof decides, as for every token off the fast path
        for (; p < e && *p >= '0' && *p <= '9'; ++p) { ex = ex * 10 + (*p - '0'); if (ex > 100000) return parse_float_slow(b, e, out); }
        if (eneg) ex = -ex;
    }
    if (p != e) return parse_float_slow(b, e, out);
    if (m == 0) { out = neg ? -0.0f : 0.0f; return true; }
    const int e10 = ex - frac;
    if (e10 < -27 || e10 > 27) return parse_float_slow(b, e, out);
    const long double d = e10 < 0 ? (long double) m / p10[-e10] : (long double) m * p10[e10];
    if (!(d >= 1e-30L && d <= 1e30L)) return parse_float_slow(b, e, out);   // near float under/overflow: let strtof decide
    unsigned long long mant;
    memcpy(&mant, &d, 8);                                    // x87: explicit 64-bit significand in the low 8 bytes
    const unsigned long long low = mant & ((1ULL << 40) - 1);   // the 40 significand bits a float drops
    const unsigned long long half = 1ULL << 39;
    if (low >= half - 2 && low <= half + 2) return parse_float_slow(b, e, out);
    out = (float) (neg ? -d : d);
    return true;
}

// Binary cache of the parsed arrays, opt-in with BA_LOADER_CACHE=1: <path>.mibacache holds
// the float-rounded values exactly as the text parse produced them, keyed by the text file's size and mtime.
struct CacheHeader {
    char magic[8];
    long long src_size, src_mtime_ns;
    int n_cam, n_pt, n_obs, pad;
};
const char kCacheMagic[8] = {'M', 'I', 'B', 'A', 'B', 'A', 'L', '1'};

bool cache_key(const std::string &path, CacheHeader &h) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return false;
    memcpy(h.magic, kCacheMagic, 8);
    h.src_size = (long long) st.st_size;
    h.src_mtime_ns = (long long) st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec;
    return true;
}

}  // namespace

// false: no usable cache (missing, stale, short): the caller parses the text
static bool load_cache(BalDataloader &d, const std::string &path) {
    CacheHeader want{}, got{};
    if (!cache_key(path, want)) return false;
    FILE *fp = fopen((path + ".mibacache").c_str(), "rb");
    if (!fp) return false;
    bool ok = fread(&got, sizeof(got), 1, fp) == 1 && memcmp(got.magic, kCacheMagic, 8) == 0 && got.src_size == want.src_size &&
              got.src_mtime_ns == want.src_mtime_ns && got.n_cam >= 0 && got.n_pt >= 0 && got.n_obs >= 0;
    if (ok) {
        std::vector<int> oc((size_t) got.n_obs), op((size_t) got.n_obs);
        d.observations.resize((size_t) got.n_obs);
        d.cameras.resize((size_t) got.n_cam);
        d.points.resize((size_t) got.n_pt);
        static_assert(sizeof(std::pair<float, float>) == 8 && sizeof(Camera) == 36 && sizeof(std::tuple<float, float, float>) == 12,
                      "cache layout");
        auto rd = [&](void *dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, fp) == bytes; };
        ok = rd(oc.data(), oc.size() * sizeof(int)) && rd(op.data(), op.size() * sizeof(int)) &&
             rd(d.observations.data(), d.observations.size() * 8) && rd(d.cameras.data(), d.cameras.size() * sizeof(Camera)) &&
             rd(d.points.data(), d.points.size() * 12);
        if (ok) {
            d.obs_cam.resize(oc.size());
            d.obs_point.resize(op.size());
            for (size_t i = 0; i < oc.size(); ++i) { d.obs_cam[i] = (size_t) (long long) oc[i]; d.obs_point[i] = (size_t) (long long) op[i]; }
            d.num_camera = got.n_cam; d.num_points = got.n_pt; d.num_observations = got.n_obs;
        }
    }
    fclose(fp);
    return ok;
}

static void store_cache(const BalDataloader &d, const std::string &path) {
    CacheHeader h{};
    if (!cache_key(path, h)) return;
    h.n_cam = d.num_camera; h.n_pt = d.num_points; h.n_obs = d.num_observations;
    std::vector<int> oc(d.obs_cam.size()), op(d.obs_point.size());
    for (size_t i = 0; i < oc.size(); ++i) { oc[i] = (int) (long long) d.obs_cam[i]; op[i] = (int) (long long) d.obs_point[i]; }
    const std::string tmp = path + ".mibacache.tmp";
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (!fp) return;   // read-only dataset directory: no cache, no error
    auto wr = [&](const void *src, size_t bytes) { return bytes == 0 || fwrite(src, 1, bytes, fp) == bytes; };
    const bool ok = wr(&h, sizeof(h)) && wr(oc.data(), oc.size() * sizeof(int)) && wr(op.data(), op.size() * sizeof(int)) &&
                    wr(d.observations.data(), d.observations.size() * 8) && wr(d.cameras.data(), d.cameras.size() * sizeof(Camera)) &&
                    wr(d.points.data(), d.points.size() * 12);
    if (fclose(fp) != 0 || !ok || rename(tmp.c_str(), (path + ".mibacache").c_str()) != 0) remove(tmp.c_str());
}

BalDataloader::BalDataloader(std::string path) {
    // <num_cameras> <num_points> <num_observations>
    // <camera_index> <point_index> <x> <y>        (num_observations lines)
    // 9 scalars per camera: R3, t3, f, k1, k2      (one per line)
    // 3 scalars per point
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (getenv("BA_LOADER_DEBUG"))
            fprintf(stderr, "[BalDataloader] %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    const char *ce = getenv("BA_LOADER_CACHE");
    const bool use_cache = ce && atoi(ce) != 0;
    if (use_cache && load_cache(*this, path)) { lap("binary cache read"); return; }
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { throw std::invalid_argument("File could not be opened"); }
    std::vector<char> buf;
    {
        fseek(fp, 0, SEEK_END);
        const long sz = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t) sz : 0);
        const size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), fp);
        fclose(fp);
        buf.resize(got);
    }
    lap("file read");
    const char *p = buf.data(), *end = buf.data() + buf.size();
    // header, serially
    long long hdr[3] = {-1, -1, -1};
    for (int k = 0; k < 3; ++k) {
        while (p < end && is_ws(*p)) ++p;
        const char *b = p;
        while (p < end && !is_ws(*p)) ++p;
        if (b == p || !parse_token(b, p, hdr[k]) || hdr[k] < 0 || hdr[k] > 2000000000LL)
            throw std::invalid_argument("Not a BAL file (bad header)");
    }
    num_camera = (int) hdr[0];
    num_points = (int) hdr[1];
    num_observations = (int) hdr[2];
    const long long n_obs_tok = 4LL * num_observations, n_cam_tok = 9LL * num_camera, n_pt_tok = 3LL * num_points;
    const long long need = n_obs_tok + n_cam_tok + n_pt_tok;

    obs_cam.resize(num_observations);
    obs_point.resize(num_observations);
    observations.resize(num_observations);
    cameras.resize(num_camera);
    points.resize(num_points);

    lap("arrays sized");
    // byte ranges that start and end between tokens
    unsigned n_thr = std::thread::hardware_concurrency();
    if (const char *e = getenv("BA_LOADER_THREADS")) n_thr = (unsigned) atoi(e);
    if (n_thr < 1) n_thr = 1;
    if (n_thr > 32) n_thr = 32;
    const size_t body = (size_t) (end - p);
    if (body < (1u << 20)) n_thr = 1;
    std::vector<Range> rg(n_thr);
    {
        const char *b = p;
        for (unsigned t = 0; t < n_thr; ++t) {
            const char *e = (t + 1 == n_thr) ? end : p + body * (t + 1) / n_thr;
            if (e < b) e = b;
            while (e < end && !is_ws(*e)) ++e;   // do not cut a token
            rg[t] = Range{b, e, 0};
            b = e;
        }
    }
    std::vector<long long> cnt(n_thr, 0);
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < n_thr; ++t) th.emplace_back([&, t] { cnt[t] = count_tokens(rg[t].b, rg[t].e); });
        cnt[0] = count_tokens(rg[0].b, rg[0].e);
        for (auto &x : th) x.join();
    }
    lap("tokens counted");
    long long total = 0;
    for (unsigned t = 0; t < n_thr; ++t) { rg[t].first_token = total; total += cnt[t]; }
    if (total < need) { throw std::invalid_argument("BAL file is truncated"); }

    std::vector<int> bad(n_thr, 0);
    auto work = [&](unsigned t) {
        long long g = rg[t].first_token;
        const char *q = rg[t].b, *e = rg[t].e;
        while (q < e && g < need) {
            while (q < e && is_ws(*q)) ++q;
            if (q >= e) break;
            const char *b = q;
            while (q < e && !is_ws(*q)) ++q;
            bool ok;
            if (g < n_obs_tok) {
                const long long o = g >> 2;
                const int k = (int) (g & 3);
                if (k < 2) {
                    long long v = 0;
                    ok = parse_token(b, q, v);
                    if (k == 0) obs_cam[o] = (size_t) v; else obs_point[o] = (size_t) v;   // -1 wraps, as in the reference
                } else {
                    float v = 0.f;
                    ok = parse_token(b, q, v);
                    if (k == 2) observations[o].first = v; else observations[o].second = v;
                }
            } else if (g < n_obs_tok + n_cam_tok) {
                const long long c = (g - n_obs_tok) / 9;
                const int k = (int) ((g - n_obs_tok) % 9);
                float v = 0.f;
                ok = parse_token(b, q, v);
                Camera &cam = cameras[c];
                if (k < 3) cam.R[k] = v; else if (k < 6) cam.t[k - 3] = v; else if (k == 6) cam.f = v; else if (k == 7) cam.k1 = v; else cam.k2 = v;
            } else {
                const long long i = (g - n_obs_tok - n_cam_tok) / 3;
                const int k = (int) ((g - n_obs_tok - n_cam_tok) % 3);
                float v = 0.f;
                ok = parse_token(b, q, v);
                if (k == 0) std::get<0>(points[i]) = v; else if (k == 1) std::get<1>(points[i]) = v; else std::get<2>(points[i]) = v;
            }
            if (!ok) bad[t] = 1;
            ++g;
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < n_thr; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    lap("parsed");
    for (unsigned t = 0; t < n_thr; ++t)
        if (bad[t]) { throw std::invalid_argument("Not a BAL file (unparsable number)"); }
    if (use_cache) { store_cache(*this, path); lap("binary cache written"); }
}

BalDataloader::~BalDataloader() {}

std::vector<cv::Point2f> BalDataloader::getObservations() const {
    // same bytes: two floats per observation in both element types
    static_assert(sizeof(cv::Point2f) == 2 * sizeof(float) && sizeof(std::pair<float, float>) == 2 * sizeof(float), "layout");
    std::vector<cv::Point2f> obs(observations.size());
    if (!obs.empty()) memcpy(static_cast<void *>(obs.data()), static_cast<const void *>(observations.data()), obs.size() * sizeof(cv::Point2f));
    return obs;
}

int BalDataloader::getObsCam(int index) const { return (int) obs_cam[index]; }
int BalDataloader::getObsPoint(int index) const { return (int) obs_point[index]; }
int BalDataloader::getNumPoints() const { return num_points; }
int BalDataloader::getNumObservations() const { return num_observations; }
int BalDataloader::getNumFrames() const { return num_camera; }
bool BalDataloader::isColorAvailable() const { return false; }
bool BalDataloader::isDepthAvailable() const { return false; }
cv::Mat BalDataloader::getColor(int) const { return cv::Mat(); }
cv::Mat BalDataloader::getDepth(int) const { return cv::Mat(); }

namespace {
// uniform in [0,1]
inline double RandDouble() { return static_cast<double>(rand()) / RAND_MAX; }

// Marsaglia polar method; the second variate is discarded as in the reference
inline double RandNormal() {
    double x1, x2, w;
    do {
        x1 = 2.0 * RandDouble() - 1.0;
        x2 = 2.0 * RandDouble() - 1.0;
        w = x1 * x1 + x2 * x2;
    } while (w >= 1.0 || w == 0.0);
    w = sqrt((-2.0 * log(w)) / w);
    return x1 * w;
}
}  // namespace

void BalDataloader::initialize(double *R, double *T, double *intrinsics, double *X) {
    for (int i = 0; i < num_camera; ++i) {
        const Camera &cam = cameras[i];
        for (int k = 0; k < 3; ++k) R[3 * i + k] = cam.R[k] + RandNormal() * sigmaCam;
        for (int k = 0; k < 3; ++k) T[3 * i + k] = cam.t[k] + RandNormal() * sigmaCam;
        intrinsics[6 * i + 0] = cam.f;  // fx
        intrinsics[6 * i + 1] = cam.f;  // fy
        intrinsics[6 * i + 2] = 0;      // ox
        intrinsics[6 * i + 3] = 0;      // oy
        intrinsics[6 * i + 4] = cam.k1;
        intrinsics[6 * i + 5] = cam.k2;
    }
    for (int i = 0; i < num_points; ++i) {
        const auto &pt = points[i];
        X[3 * i + 0] = std::get<0>(pt) + RandNormal() * sigmaX;
        X[3 * i + 1] = std::get<1>(pt) + RandNormal() * sigmaX;
        X[3 * i + 2] = std::get<2>(pt) + RandNormal() * sigmaX;
    }
}

Eigen::Vector3i BalDataloader::getPointColor(int) const {
    Eigen::Vector3i color;
    color.setConstant(255);
    return color;
}

const float *BalDataloader::getEstimatedPose(int) const {
    static const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    return identity;
}
