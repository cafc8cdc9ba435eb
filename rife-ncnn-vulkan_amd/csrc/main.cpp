// rife-hip — the reference's command line (src/main.cpp:102-121 usage, 442-917 main) on top of `class RIFE` (rife.h):
//
//     rife-hip -0 in0.png -1 in1.png -o out.png [options]
//     rife-hip -i indir -o outdir [options]
//
// Same flags, defaults, validation order and messages, frame / timestep schedule (src/main.cpp:705-731) and three-stage
// pipeline (load -> proc -> save over bounded queues, one RIFE per -g id, -j load:proc[,proc..]:save; src/main.cpp:248-436,
// 819-904), re-hosted on std::thread.  Codecs: PNG (8-bit, non-interlaced) through zlib, binary PPM, and WebP through the
// system libwebp (its stable simple API, declared below because the image ships the library without headers; lossless encoding
// like src/webp_image.h:66-68), and baseline JPEG through jpeg_codec.h (quality 100 on output like src/main.cpp:215).
// Host glue only: every pixel of arithmetic happens in librife_hip.so.
#include <dirent.h>
#include <getopt.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rife_hip.h"
#include "jpeg_codec.h"
#include "rife.h"

#ifdef RIFE_HIP_WITH_WEBP
extern "C" {      // libwebp simple API (webp/decode.h, webp/encode.h)
uint8_t* WebPDecodeRGB(const uint8_t* data, size_t data_size, int* width, int* height);
size_t WebPEncodeLosslessRGB(const uint8_t* rgb, int width, int height, int stride, uint8_t** output);
void WebPFree(void* ptr);
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// image files: PNG (zlib) and PPM (P6)
// ---------------------------------------------------------------------------------------------------------------
static bool read_file(const std::string& path, std::vector<unsigned char>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

static uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// PNG (every colour type, bit depth and interlace mode stb_image reads) -> tightly packed 8-bit RGB
static bool decode_png(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgb) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 33 || memcmp(d.data(), sig, 8)) return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, pal;
    while (pos + 12 <= d.size()) {
        const uint32_t len = be32(&d[pos]);
        const unsigned char* typ = &d[pos + 4];
        if (pos + 12 + (size_t)len > d.size()) return false;
        const unsigned char* body = &d[pos + 8];
        if (!memcmp(typ, "IHDR", 4)) {
            if (len != 13 || pos != 8) return false;          // IHDR is 13 bytes and the first chunk (a short one would be read past its end)
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        }
        else if (!memcmp(typ, "PLTE", 4)) pal.assign(body, body + len);
        else if (!memcmp(typ, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(typ, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    // every colour type / bit depth / interlace mode stb_image reads in the reference (1-, 2-, 4-, 8-, 16-bit; Adam7); like stb with
    // 3 requested channels: 16-bit samples keep their high byte, low-depth grey is scaled to 0..255, alpha (and tRNS) is dropped
    if (w <= 0 || h <= 0 || interlace > 1) return false;
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || (ctype == 3 && pal.empty())) return false;
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) || (ctype == 3 && depth == 16)) return false;
    const int bpp = depth * ch, fb = std::max(1, bpp / 8);               // bits per pixel, bytes per filter unit
    auto row_bytes = [&](int pw) { return ((size_t)pw * bpp + 7) / 8; };
    // pass geometry: one pass for non-interlaced files, the seven Adam7 passes otherwise
    static const int X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = interlace ? 7 : 1;
    size_t total = 0;
    for (int p = 0; p < npass; p++) {
        const int pw = interlace ? (w - X0[p] + DX[p] - 1) / DX[p] : w, ph = interlace ? (h - Y0[p] + DY[p] - 1) / DY[p] : h;
        if (pw > 0 && ph > 0) total += (row_bytes(pw) + 1) * ph;
    }
    // a header that promises more pixels than the compressed data can hold (deflate expands at most ~1032 : 1) or than any frame has
    // (2^28 pixels = 16K x 16K) is refused before anything of that size is allocated
    if ((size_t)w * h > ((size_t)1 << 28) || total > idat.size() * 1100 + 65536) return false;
    std::vector<unsigned char> raw(total);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    rgb.assign((size_t)w * h * 3, 0);
    const int gscale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    size_t off = 0;
    for (int p = 0; p < npass; p++) {
        const int pw = interlace ? (w - X0[p] + DX[p] - 1) / DX[p] : w, ph = interlace ? (h - Y0[p] + DY[p] - 1) / DY[p] : h;
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = row_bytes(pw);
        std::vector<unsigned char> img(stride * ph);
        for (int y = 0; y < ph; y++) {
            const unsigned char* src = &raw[off + (stride + 1) * y];
            unsigned char* cur = &img[stride * y];
            const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
            const int ft = src[0];
            if (ft > 4) return false;
            const size_t F = (size_t)fb, head = std::min(F, stride);
            // the first pixel of a row has no left neighbour; after it one tight loop per filter type (this is 1/4 of the decode time)
            for (size_t x = 0; x < head; x++) {
                const int b = up ? up[x] : 0;
                cur[x] = (unsigned char)(src[1 + x] + (ft == 2 ? b : ft == 3 ? b >> 1 : ft == 4 ? paeth(0, b, 0) : 0));
            }
            const unsigned char* in = src + 1;
            switch (ft) {
                case 0: std::memcpy(cur + head, in + head, stride - head); break;
                case 1: for (size_t x = head; x < stride; x++) cur[x] = (unsigned char)(in[x] + cur[x - F]); break;
                case 2: if (up) { for (size_t x = head; x < stride; x++) cur[x] = (unsigned char)(in[x] + up[x]); } else std::memcpy(cur + head, in + head, stride - head); break;
                case 3: for (size_t x = head; x < stride; x++) cur[x] = (unsigned char)(in[x] + ((cur[x - F] + (up ? up[x] : 0)) >> 1)); break;
                default: for (size_t x = head; x < stride; x++) cur[x] = (unsigned char)(in[x] + paeth(cur[x - F], up ? up[x] : 0, up ? up[x - F] : 0)); break;
            }
        }
        off += (stride + 1) * ph;
        if (!interlace && depth == 8 && ctype == 2) { std::memcpy(rgb.data(), img.data(), img.size()); continue; }      // 8-bit RGB rows are the output rows
        for (int y = 0; y < ph; y++)
            for (int x = 0; x < pw; x++) {
                const unsigned char* row = &img[stride * y];
                auto sample = [&](int c) -> int {          // channel c of pixel x as 8 bits (index for palette images)
                    if (depth == 8) return row[(size_t)x * ch + c];
                    if (depth == 16) return row[((size_t)x * ch + c) * 2];
                    const int per = 8 / depth, sh = (per - 1 - x % per) * depth;
                    return (row[x / per] >> sh) & ((1 << depth) - 1);
                };
                const int ox = interlace ? X0[p] + x * DX[p] : x, oy = interlace ? Y0[p] + y * DY[p] : y;
                unsigned char* o = &rgb[((size_t)oy * w + ox) * 3];
                if (ctype == 2 || ctype == 6) { o[0] = (unsigned char)sample(0); o[1] = (unsigned char)sample(1); o[2] = (unsigned char)sample(2); }
                else if (ctype == 0 || ctype == 4) { o[0] = o[1] = o[2] = (unsigned char)(sample(0) * (depth < 8 ? gscale : 1)); }
                else {
                    const size_t k = (size_t)sample(0) * 3;
                    if (k + 2 >= pal.size()) return false;
                    o[0] = pal[k]; o[1] = pal[k + 1]; o[2] = pal[k + 2];
                }
            }
    }
    return true;
}

// BMP as stb_image reads it in the reference: uncompressed 24- / 32-bit (BI_RGB, or BI_BITFIELDS with the standard masks) and 8-bit palette
static bool decode_bmp(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgb) {
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') return false;
    auto le32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
    const uint32_t dataoff = le32(10), hsz = le32(14);
    if (hsz < 40 || 14 + (size_t)hsz > d.size()) return false;
    const int bw = (int)le32(18), bh = (int)le32(22), bits = d[28] | (d[29] << 8);
    const uint32_t comp = le32(30);
    if (bw <= 0 || bh == 0 || (d[26] | (d[27] << 8)) != 1) return false;
    if (!((bits == 24 && comp == 0) || (bits == 32 && (comp == 0 || comp == 3)) || (bits == 8 && comp == 0))) return false;
    const bool flip = bh > 0;                                 // positive height = bottom-up rows
    w = bw; h = bh > 0 ? bh : -bh;
    const size_t stride = ((size_t)w * bits / 8 + 3) & ~(size_t)3;
    if ((size_t)dataoff + stride * h > d.size()) return false;
    const unsigned char* pal = &d[14 + hsz];
    uint32_t ncol = le32(46);
    if (bits == 8) { if (!ncol) ncol = 256; if (14 + (size_t)hsz + 4 * (size_t)ncol > d.size()) return false; }
    int rs = 16, gs = 8, bs = 0;                              // 32-bit BI_RGB is B, G, R, X
    if (bits == 32 && comp == 3) {
        if (hsz < 52 && 14 + (size_t)hsz + 12 > d.size()) return false;
        const uint32_t rm = le32(54), gm = le32(58), bm = le32(62);
        auto shift_of = [](uint32_t m) { int sft = 0; while (m && !(m & 1)) { m >>= 1; sft++; } return m == 0xff ? sft : -1; };
        rs = shift_of(rm); gs = shift_of(gm); bs = shift_of(bm);
        if (rs < 0 || gs < 0 || bs < 0) return false;
    }
    rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++) {
        const unsigned char* row = &d[dataoff + stride * (size_t)(flip ? h - 1 - y : y)];
        unsigned char* o = &rgb[(size_t)y * w * 3];
        for (int x = 0; x < w; x++, o += 3) {
            if (bits == 24) { o[0] = row[3 * x + 2]; o[1] = row[3 * x + 1]; o[2] = row[3 * x]; }
            else if (bits == 32) {
                const uint32_t v = (uint32_t)row[4 * x] | ((uint32_t)row[4 * x + 1] << 8) | ((uint32_t)row[4 * x + 2] << 16) | ((uint32_t)row[4 * x + 3] << 24);
                o[0] = (unsigned char)(v >> rs); o[1] = (unsigned char)(v >> gs); o[2] = (unsigned char)(v >> bs);
            } else {
                if (row[x] >= ncol) return false;
                const unsigned char* e = pal + 4 * (size_t)row[x];
                o[0] = e[2]; o[1] = e[1]; o[2] = e[0];
            }
        }
    }
    return true;
}

static void put_chunk(std::vector<unsigned char>& out, const char* typ, const unsigned char* body, size_t len) {
    const unsigned char l[4] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len};
    out.insert(out.end(), l, l + 4);
    const size_t start = out.size();
    out.insert(out.end(), typ, typ + 4);
    if (len) out.insert(out.end(), body, body + len);
    const uint32_t crc = (uint32_t)crc32(0L, &out[start], (uInt)(len + 4));
    const unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
    out.insert(out.end(), c, c + 4);
}

// Helper threads one encode_png() call may use besides its own (set in main() from the core count and the -j settings).
static int g_png_helpers = 0;

// One band of rows: filter "up" (cheap and effective on video frames; the first row of the image has no row above) and raw deflate.
// Every band but the last ends with a sync flush, i.e. on a byte boundary, so the bands concatenate into one valid deflate stream -
// the bands share no history, which costs a little ratio and lets them run on separate cores.
static bool png_deflate_band(const unsigned char* rgb, int w, int y0, int y1, bool last, std::vector<unsigned char>& comp, uLong& adler) {
    const size_t stride = (size_t)w * 3;
    std::vector<unsigned char> raw((stride + 1) * (size_t)(y1 - y0));
    for (int y = y0; y < y1; y++) {
        unsigned char* dst = &raw[(stride + 1) * (size_t)(y - y0)];
        const unsigned char* cur = rgb + stride * y;
        const unsigned char* up = y ? rgb + stride * (y - 1) : nullptr;
        dst[0] = up ? 2 : 0;
        if (up) for (size_t x = 0; x < stride; x++) dst[1 + x] = (unsigned char)(cur[x] - up[x]);
        else std::memcpy(dst + 1, cur, stride);
    }
    adler = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
    z_stream z;
    std::memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 3, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    comp.resize(deflateBound(&z, (uLong)raw.size()) + 16);
    z.next_in = raw.data(); z.avail_in = (uInt)raw.size();
    z.next_out = comp.data(); z.avail_out = (uInt)comp.size();
    const int rc = deflate(&z, last ? Z_FINISH : Z_SYNC_FLUSH);
    const bool ok = last ? rc == Z_STREAM_END : (rc == Z_OK && z.avail_in == 0 && z.avail_out != 0);
    comp.resize(comp.size() - z.avail_out);
    deflateEnd(&z);
    return ok;
}

static bool encode_png(const std::string& path, int w, int h, const unsigned char* rgb) {
    const size_t stride = (size_t)w * 3;
    // bands of about 1 MB of pixels, at most 64 (a band must stay below the 4 GB zlib counts in any case)
    int nband = (int)std::min<size_t>(64, std::max<size_t>(1, stride * h >> 20));
    nband = std::min(nband, h);
    std::vector<std::vector<unsigned char>> comp(nband);
    std::vector<uLong> adler(nband);
    std::vector<size_t> rawlen(nband);
    std::vector<char> good(nband, 0);
    std::atomic<int> next(0);
    auto work = [&]() {
        for (int b; (b = next.fetch_add(1)) < nband;) {
            const int y0 = (int)((long long)h * b / nband), y1 = (int)((long long)h * (b + 1) / nband);
            rawlen[b] = (stride + 1) * (size_t)(y1 - y0);
            good[b] = png_deflate_band(rgb, w, y0, y1, b == nband - 1, comp[b], adler[b]);
        }
    };
    std::vector<std::thread> helpers;
    for (int i = 0; i < std::min(g_png_helpers, nband - 1); i++) helpers.emplace_back(work);
    work();
    for (auto& t : helpers) t.join();
    std::vector<unsigned char> zs = {0x78, 0x5e};                       // zlib header: deflate, 32 KB window, "fast" level hint
    uLong ad = adler32(0L, Z_NULL, 0);
    for (int b = 0; b < nband; b++) {
        if (!good[b]) return false;
        zs.insert(zs.end(), comp[b].begin(), comp[b].end());
        ad = adler32_combine(ad, adler[b], (z_off_t)rawlen[b]);
    }
    for (int i = 3; i >= 0; i--) zs.push_back((unsigned char)(ad >> (8 * i)));
    const unsigned char* const comp_data = zs.data();
    const size_t clen = zs.size();
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w,
                              (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h, 8, 2, 0, 0, 0};
    put_chunk(out, "IHDR", ihdr, 13);
    put_chunk(out, "IDAT", comp_data, clen);
    put_chunk(out, "IEND", nullptr, 0);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

static bool decode_ppm(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgb) {
    if (d.size() < 7 || d[0] != 'P' || (d[1] != '6' && d[1] != '5')) return false;      // P6 = rgb, P5 = grey (stb_image's pnm reader takes both)
    const bool grey = d[1] == '5';
    size_t pos = 2;
    int vals[3], n = 0;
    while (n < 3 && pos < d.size()) {
        while (pos < d.size() && (isspace(d[pos]) || d[pos] == '#')) { if (d[pos] == '#') while (pos < d.size() && d[pos] != '\n') pos++; else pos++; }
        int v = 0; bool any = false;
        while (pos < d.size() && isdigit(d[pos])) { v = v * 10 + (d[pos] - '0'); pos++; any = true; }
        if (!any) return false;
        vals[n++] = v;
    }
    if (n != 3 || vals[2] != 255 || pos >= d.size()) return false;
    pos++;                                       // the single whitespace after maxval
    w = vals[0]; h = vals[1];
    if (w <= 0 || h <= 0 || d.size() - pos < (size_t)w * h * (grey ? 1 : 3)) return false;
    if (!grey) { rgb.assign(d.begin() + pos, d.begin() + pos + (size_t)w * h * 3); return true; }
    rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = d[pos + i];
    return true;
}

static bool encode_ppm(const std::string& path, int w, int h, const unsigned char* rgb) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    const bool ok = fwrite(rgb, 1, (size_t)w * h * 3, f) == (size_t)w * h * 3;
    fclose(f);
    return ok;
}

static bool decode_webp(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgb) {
#ifdef RIFE_HIP_WITH_WEBP
    if (d.size() < 12 || memcmp(d.data(), "RIFF", 4) || memcmp(d.data() + 8, "WEBP", 4)) return false;
    uint8_t* px = WebPDecodeRGB(d.data(), d.size(), &w, &h);
    if (!px) return false;
    rgb.assign(px, px + (size_t)w * h * 3);
    WebPFree(px);
    return true;
#else
    (void)d; (void)w; (void)h; (void)rgb;
    return false;
#endif
}

static bool encode_webp(const std::string& path, int w, int h, const unsigned char* rgb) {
#ifdef RIFE_HIP_WITH_WEBP
    uint8_t* out = nullptr;
    const size_t n = WebPEncodeLosslessRGB(rgb, w, h, w * 3, &out);
    if (!n || !out) return false;
    FILE* f = fopen(path.c_str(), "wb");
    const bool ok = f && fwrite(out, 1, n, f) == n;
    if (f) fclose(f);
    WebPFree(out);
    return ok;
#else
    (void)path; (void)w; (void)h; (void)rgb;
    return false;
#endif
}

static bool encode_image(const std::string& path, int w, int h, const unsigned char* rgb);

static std::string ext_of(const std::string& p) {
    const size_t dot = p.rfind('.');
    if (dot == std::string::npos || p.find('/', dot) != std::string::npos) return "";
    std::string e = p.substr(dot + 1);
    std::transform(e.begin(), e.end(), e.begin(), ::tolower);
    return e;
}

static bool decode_image(const std::string& path, int& w, int& h, std::vector<unsigned char>& rgb) {
    std::vector<unsigned char> d;
    if (!read_file(path, d)) return false;
    if (decode_png(d, w, h, rgb) || decode_ppm(d, w, h, rgb) || decode_webp(d, w, h, rgb) || decode_bmp(d, w, h, rgb)) return true;
    std::string why;
    if (jpeg::decode(d, w, h, rgb, &why)) return true;
    if (d.size() > 2 && d[0] == 0xFF && d[1] == 0xD8) fprintf(stderr, "%s: %s\n", path.c_str(), why.c_str());
    return false;
}

static bool encode_image(const std::string& path, int w, int h, const unsigned char* rgb) {
    const std::string e = ext_of(path);
    if (e == "ppm") return encode_ppm(path, w, h, rgb);
    if (e == "webp") return encode_webp(path, w, h, rgb);
    if (e == "jpg" || e == "jpeg") return jpeg::encode(path, w, h, rgb, g_png_helpers);      // same spare cores as the PNG writer
    return encode_png(path, w, h, rgb);
}

// ---------------------------------------------------------------------------------------------------------------
// tasks and queues (src/main.cpp:231-295)
// ---------------------------------------------------------------------------------------------------------------
// A decoded frame plus its copies in device memory, one per engine that has needed it so far (uploaded by the first proc thread
// that does).  The reference uploads both frames in every process() call (src/rife.cpp:2490-2530); in a sequence each frame serves
// two pairs and, with -n, several timesteps per pair, so the upload happens once here.
struct SharedFrame {
    int w = 0, h = 0;
    std::vector<unsigned char> px;
    const rife_hip_frame* on(const RIFE* r) {
        std::lock_guard<std::mutex> g(mu);
        for (auto& e : resident) if (e.first == r) return e.second;
        rife_hip_frame* f = r->upload(ncnn::Mat(w, h, (void*)px.data(), (size_t)3, 3));
        if (f) resident.emplace_back(r, f);
        return f;
    }
    ~SharedFrame() { for (auto& e : resident) RIFE::release(e.second); }
private:
    std::mutex mu;
    std::vector<std::pair<const RIFE*, rife_hip_frame*>> resident;
};

struct Task {
    int id = 0;
    float timestep = 0.5f;
    std::string in0path, in1path, outpath;
    std::shared_ptr<SharedFrame> fr0, fr1;                            // decoded frames are shared between the tasks that use them
    std::vector<unsigned char> out;
    int w = 0, h = 0;
};

// Decoded-frame cache: in directory mode consecutive tasks use the same files (output i takes frames sx, sx + 1; with the default
// -n every input frame is needed by about four tasks), and decoding is the slowest stage of the whole program.  The reference
// decodes per task (src/main.cpp:315-334); this keeps the last few decoded frames.
class FrameCache {
public:
    typedef std::shared_ptr<SharedFrame> Frame;
    template <class Decode>
    bool get(const std::string& path, Frame& f, Decode decode) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto& e : entries) if (e.first == path) { f = e.second; return true; }
        }
        Frame n = std::make_shared<SharedFrame>();
        if (!decode(path, n->w, n->h, n->px)) return false;
        std::lock_guard<std::mutex> g(mu);
        for (auto& e : entries) if (e.first == path) { f = e.second; return true; }     // another loader decoded it meanwhile: keep one copy
        entries.emplace_back(path, n);
        if (entries.size() > 6) entries.erase(entries.begin());
        f = n;
        return true;
    }
    void clear() { std::lock_guard<std::mutex> g(mu); entries.clear(); }
private:
    std::mutex mu;
    std::vector<std::pair<std::string, Frame>> entries;
};

class TaskQueue {
public:
    void put(Task&& t) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.size() < 8; });          // bounded at 8 like the reference (main.cpp:260)
        q.push(std::move(t));
        cv.notify_all();
    }
    Task get() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q.empty(); });
        Task t = std::move(q.front());
        q.pop();
        cv.notify_all();
        return t;
    }
private:
    std::mutex mu;
    std::condition_variable cv;
    std::queue<Task> q;
};

static void print_usage() {
    fprintf(stderr, "Usage: rife-hip -0 infile -1 infile1 -o outfile [options]...\n");
    fprintf(stderr, "       rife-hip -i indir -o outdir [options]...\n\n");
    fprintf(stderr, "  -h                   show this help\n");
    fprintf(stderr, "  -v                   verbose output\n");
    fprintf(stderr, "  -0 input0-path       input image0 path (jpg/png/webp/bmp/pnm)\n");
    fprintf(stderr, "  -1 input1-path       input image1 path (jpg/png/webp/bmp/pnm)\n");
    fprintf(stderr, "  -i input-path        input image directory (jpg/png/webp/bmp/pnm)\n");
    fprintf(stderr, "  -o output-path       output image path (jpg/png/webp/ppm) or directory\n");
    fprintf(stderr, "  -n num-frame         target frame count (default=N*2)\n");
    fprintf(stderr, "  -s time-step         time step (0~1, default=0.5)\n");
    fprintf(stderr, "  -m model-path        rife model path (default=rife-v2.3)\n");
    fprintf(stderr, "  -g gpu-id            gpu device to use (default=0) can be 0,1,2 for multi-gpu\n");
    fprintf(stderr, "  -j load:proc:save    thread count for load/proc/save (default=1:2:2) can be 1:2,2,2:2 for multi-gpu\n");
    fprintf(stderr, "  -x                   enable spatial tta mode\n");
    fprintf(stderr, "  -z                   enable temporal tta mode\n");
    fprintf(stderr, "  -u                   enable UHD mode\n");
    fprintf(stderr, "  -f pattern-format    output image filename pattern format (%%08d.jpg/png/webp/ppm, default=ext/%%08d.png)\n");
}

static bool is_dir(const std::string& p) { struct stat s; return stat(p.c_str(), &s) == 0 && S_ISDIR(s.st_mode); }

static bool list_directory(const std::string& d, std::vector<std::string>& names) {
    DIR* dir = opendir(d.c_str());
    if (!dir) return false;
    while (struct dirent* e = readdir(dir)) {
        const std::string full = d + "/" + e->d_name;
        struct stat s;
        if (stat(full.c_str(), &s) == 0 && S_ISREG(s.st_mode)) names.push_back(e->d_name);
    }
    closedir(dir);
    std::sort(names.begin(), names.end());
    return true;
}

static std::vector<int> parse_int_list(const std::string& s) {
    std::vector<int> v;
    size_t pos = 0;
    while (pos <= s.size()) {
        const size_t c = s.find(',', pos);
        const std::string tok = s.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
        v.push_back(atoi(tok.c_str()));
        if (c == std::string::npos) break;
        pos = c + 1;
    }
    return v;
}

int main(int argc, char** argv) {
    // codec self-test without a GPU:  rife-hip --transcode in.(png|ppm) out.(png|ppm)
    if (argc == 4 && std::string(argv[1]) == "--transcode") {
        int w = 0, h = 0;
        std::vector<unsigned char> rgb;
        g_png_helpers = std::max(0, std::min(15, (int)std::thread::hardware_concurrency() - 1));
        if (!decode_image(argv[2], w, h, rgb)) { fprintf(stderr, "decode image %s failed\n", argv[2]); return 1; }
        const bool ok = encode_image(argv[3], w, h, rgb.data());
        if (!ok) { fprintf(stderr, "encode image %s failed\n", argv[3]); return 1; }
        return 0;
    }
    std::string input0, input1, inputpath, outputpath, model = "rife-v2.3", pattern_format = "%08d.png";
    int numframe = 0;
    float timestep = 0.5f;
    std::vector<int> gpuid, jobs_proc;
    int jobs_load = 1, jobs_save = 2;
    bool verbose = false, tta = false, tta_temporal = false, uhd = false;

    int opt;
    while ((opt = getopt(argc, argv, "0:1:i:o:n:s:m:g:j:f:vxzuh")) != -1) {
        switch (opt) {
            case '0': input0 = optarg; break;
            case '1': input1 = optarg; break;
            case 'i': inputpath = optarg; break;
            case 'o': outputpath = optarg; break;
            case 'n': numframe = atoi(optarg); break;
            case 's': timestep = (float)atof(optarg); break;
            case 'm': model = optarg; break;
            case 'g': gpuid = parse_int_list(optarg); break;
            case 'j': {
                const std::string a = optarg;
                const size_t c1 = a.find(':'), c2 = a.rfind(':');
                if (c1 == std::string::npos || c1 == c2) { fprintf(stderr, "invalid thread count argument\n"); return -1; }
                jobs_load = atoi(a.substr(0, c1).c_str());
                jobs_proc = parse_int_list(a.substr(c1 + 1, c2 - c1 - 1));
                jobs_save = atoi(a.substr(c2 + 1).c_str());
                break;
            }
            case 'f': pattern_format = optarg; break;
            case 'v': verbose = true; break;
            case 'x': tta = true; break;
            case 'z': tta_temporal = true; break;
            case 'u': uhd = true; break;
            case 'h':
            default: print_usage(); return -1;
        }
    }

    // ---- validation, in the reference's order (src/main.cpp:575-689) ----
    if (((input0.empty() || input1.empty()) && inputpath.empty()) || outputpath.empty()) { print_usage(); return -1; }
    if (inputpath.empty() && (timestep <= 0.f || timestep >= 1.f)) { fprintf(stderr, "invalid timestep argument, must be 0~1\n"); return -1; }
    if (!inputpath.empty() && numframe < 0) { fprintf(stderr, "invalid numframe argument, must not be negative\n"); return -1; }
    if (jobs_load < 1 || jobs_save < 1) { fprintf(stderr, "invalid thread count argument\n"); return -1; }
    if (!jobs_proc.empty() && jobs_proc.size() != (gpuid.empty() ? 1 : gpuid.size())) { fprintf(stderr, "invalid jobs_proc thread count argument\n"); return -1; }
    for (int j : jobs_proc) if (j < 1) { fprintf(stderr, "invalid jobs_proc thread count argument\n"); return -1; }

    std::string pattern = pattern_format, format;
    {
        const size_t dot = pattern_format.rfind('.');
        if (dot != std::string::npos) { pattern = pattern_format.substr(0, dot); format = pattern_format.substr(dot + 1); }
        else { pattern = "%08d"; format = pattern_format; }
        if (pattern.empty()) pattern = "%08d";
    }
    if (!is_dir(outputpath)) {
        const std::string e = ext_of(outputpath);
        if (e == "png") format = "png";
        else if (e == "ppm") format = "ppm";
        else if (e == "webp") format = "webp";
        else if (e == "jpg" || e == "jpeg") format = "jpg";
        else { fprintf(stderr, "invalid outputpath extension type\n"); return -1; }
    }
    if (format != "png" && format != "ppm" && format != "webp" && format != "jpg") { fprintf(stderr, "invalid format argument\n"); return -1; }
#ifndef RIFE_HIP_WITH_WEBP
    if (format == "webp") { fprintf(stderr, "this rife-hip was built without libwebp\n"); return -1; }
#endif

    bool rife_v2 = false, rife_v4 = false;      // family from the directory name (src/main.cpp:658-683)
    if (model.find("rife-v2") != std::string::npos || model.find("rife-v3") != std::string::npos) rife_v2 = true;
    else if (model.find("rife-v4") != std::string::npos) rife_v4 = true;
    else if (model.find("rife") == std::string::npos) { fprintf(stderr, "unknown model dir type\n"); return -1; }
    if (!rife_v4 && (numframe != 0 || timestep != 0.5f)) { fprintf(stderr, "only rife-v4 model support custom numframe and timestep\n"); return -1; }

    // ---- task list (src/main.cpp:692-766) ----
    std::vector<Task> tasks;
    if (!inputpath.empty() && is_dir(inputpath) && is_dir(outputpath)) {
        std::vector<std::string> names;
        if (!list_directory(inputpath, names) || names.size() < 2) return -1;
        const int count = (int)names.size();
        if (numframe == 0) numframe = count * 2;
        const double scale = (double)count / numframe;                          // double product rounded to float, like src/main.cpp:713-718
        for (int i = 0; i < numframe; i++) {
            float fx = (float)(i * scale);
            int sx = (int)std::floor(fx);
            fx -= sx;
            if (sx < 0) { sx = 0; fx = 0.f; }
            if (sx >= count - 1) { sx = count - 2; fx = 1.f; }
            char name[512];
            snprintf(name, sizeof name, pattern.c_str(), i + 1);               // ffmpeg numbering starts at 1
            Task t;
            t.id = i; t.timestep = fx;
            t.in0path = inputpath + "/" + names[sx]; t.in1path = inputpath + "/" + names[sx + 1];
            t.outpath = outputpath + "/" + name + "." + format;
            tasks.push_back(std::move(t));
        }
    } else if (inputpath.empty() && !is_dir(input0) && !is_dir(input1) && !is_dir(outputpath)) {
        Task t;
        t.timestep = timestep; t.in0path = input0; t.in1path = input1; t.outpath = outputpath;
        tasks.push_back(std::move(t));
    } else {
        fprintf(stderr, "input0path, input1path and outputpath must be file at the same time\n");
        fprintf(stderr, "inputpath and outputpath must be directory at the same time\n");
        return -1;
    }

    // RIFE_HIP_CLI_TIMING=1: one summary line on stderr that separates start-up (HIP init + model load) from the pipeline
    const bool timing = getenv("RIFE_HIP_CLI_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    // ---- devices (src/main.cpp:774-828) ----
    if (gpuid.empty()) gpuid.push_back(0);
    if (jobs_proc.empty()) jobs_proc.assign(gpuid.size(), 2);
    {   // cores the load / proc threads do not need are lent to the PNG encoders (band-parallel deflate), at most 15 helpers per save thread
        int busy = jobs_load;
        for (int j : jobs_proc) busy += j;
        const int spare = (int)std::thread::hardware_concurrency() - busy;
        g_png_helpers = std::max(0, std::min(15, spare / jobs_save - 1));
    }
    const int ndev = rife_hip_device_count();
    for (int g : gpuid) if (g < 0 || g >= ndev) { fprintf(stderr, "invalid gpu device\n"); return -1; }
    std::vector<RIFE*> rife;
    for (int g : gpuid) {
        RIFE* r = new RIFE(g, tta, tta_temporal, uhd, 1, rife_v2, rife_v4);
        if (r->load(model) != 0) { fprintf(stderr, "loading %s failed: %s\n", model.c_str(), rife_hip_last_error()); return -1; }
        rife.push_back(r);
    }

    const auto tp1 = std::chrono::steady_clock::now();
    // ---- load -> proc -> save (src/main.cpp:309-436, 830-904) ----
    TaskQueue toproc, tosave;
    FrameCache cache;
    std::mutex next_mu;
    size_t next_task = 0;
    auto load = [&]() {
        for (;;) {
            size_t k;
            { std::lock_guard<std::mutex> g(next_mu); if (next_task >= tasks.size()) return; k = next_task++; }
            Task t = std::move(tasks[k]);
            FrameCache::Frame f0, f1;
            if (!cache.get(t.in0path, f0, decode_image) || !cache.get(t.in1path, f1, decode_image)) { fprintf(stderr, "decode image %s or %s failed\n", t.in0path.c_str(), t.in1path.c_str()); continue; }
            if (f1->w != f0->w || f1->h != f0->h) { fprintf(stderr, "%s and %s differ in size\n", t.in0path.c_str(), t.in1path.c_str()); continue; }
            t.w = f0->w; t.h = f0->h; t.fr0 = f0; t.fr1 = f1;
            toproc.put(std::move(t));
        }
    };
    std::vector<std::atomic<int>> replica_tasks(rife.size());               // tasks every replica took off the shared queue (timing line; tests/test_cli.py)
    for (auto& n : replica_tasks) n = 0;
    auto proc = [&](RIFE* r) {
        size_t me = 0;
        while (me < rife.size() && rife[me] != r) me++;
        for (;;) {
            Task t = toproc.get();
            if (t.id == -233) return;                                          // end marker, like the reference
            replica_tasks[me]++;
            if (t.timestep == 0.f || t.timestep == 1.f) t.out = (t.timestep == 0.f ? t.fr0 : t.fr1)->px;      // rife.cpp:2470-2480: an input frame, unchanged
            else {
                const rife_hip_frame* d0 = t.fr0->on(r);
                const rife_hip_frame* d1 = t.fr1->on(r);
                t.out.resize((size_t)t.w * t.h * 3);
                ncnn::Mat out(t.w, t.h, (void*)t.out.data(), (size_t)3, 3);
                if (!d0 || !d1 || r->process(d0, d1, t.timestep, out) != 0) { fprintf(stderr, "process %s failed: %s\n", t.outpath.c_str(), rife_hip_last_error()); continue; }
            }
            t.fr0.reset(); t.fr1.reset();                                      // the save stage needs only the output
            tosave.put(std::move(t));
        }
    };
    auto save = [&]() {
        for (;;) {
            Task t = tosave.get();
            if (t.id == -233) return;
            const bool ok = encode_image(t.outpath, t.w, t.h, t.out.data());
            if (!ok) fprintf(stderr, "encode image %s failed\n", t.outpath.c_str());
            else if (verbose) fprintf(stderr, "%s %s %f -> %s done\n", t.in0path.c_str(), t.in1path.c_str(), t.timestep, t.outpath.c_str());
        }
    };
    std::vector<std::thread> loaders, procs, savers;
    for (int i = 0; i < jobs_load; i++) loaders.emplace_back(load);
    for (size_t d = 0; d < gpuid.size(); d++) for (int i = 0; i < jobs_proc[d]; i++) procs.emplace_back(proc, rife[d]);
    for (int i = 0; i < jobs_save; i++) savers.emplace_back(save);
    for (auto& t : loaders) t.join();
    for (size_t i = 0; i < procs.size(); i++) { Task e; e.id = -233; toproc.put(std::move(e)); }
    for (auto& t : procs) t.join();
    for (size_t i = 0; i < savers.size(); i++) { Task e; e.id = -233; tosave.put(std::move(e)); }
    for (auto& t : savers) t.join();
    if (timing) {
        const auto tp2 = std::chrono::steady_clock::now();
        const double a = std::chrono::duration<double>(tp1 - tp0).count(), b = std::chrono::duration<double>(tp2 - tp1).count();
        fprintf(stderr, "timing: devices + model load %.3f s, pipeline %.3f s for %zu frames = %.1f frames/s\n", a, b, tasks.size(), tasks.size() / b);
        for (size_t d = 0; d < rife.size(); d++) fprintf(stderr, "timing: replica %zu (gpu %d) took %d task(s)\n", d, gpuid[d], replica_tasks[d].load());
    }
    cache.clear();                                                             // resident frames go before their engines
    for (RIFE* r : rife) delete r;
    return 0;
}
