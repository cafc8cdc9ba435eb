// Baseline JPEG for the rife-hip command line (host glue): the reference reads jpg through stb_image and writes
// it through stb_image_write at quality 100 (src/main.cpp:123-229, 215); neither library nor libjpeg headers exist in this image,
// so this is a from-scratch codec of the subset that matters for frame sequences:
//   decode: baseline sequential DCT (SOF0 / SOF1 with 8-bit samples), Huffman, 1 or 3 components, any sampling factors up to 2x2
//           (4:4:4, 4:2:2, 4:2:0, ...), restart intervals, JFIF YCbCr -> RGB; progressive DCT (SOF2: spectral selection and successive
//           approximation) and sequential files with one scan per component as well; arithmetic / lossless / 12-bit files are refused;
//   encode: baseline, 4:4:4, all-ones quantisation tables (what "quality 100" means), standard Huffman tables.
// Floating-point IDCT / FDCT (separable, exact to rounding), so decoded pixels agree with libjpeg's to within +-1..2 levels.
#pragma once
#include <atomic>
#include <thread>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace jpeg {

static const int ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    // canonical code tables: for each length 1..16 the first code, first symbol index and count
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    uint16_t lut[512];        // the next 9 bits -> (length << 8) | symbol for codes of up to 9 bits, 0 = longer code
    bool ok = false;
    void build(const uint8_t* counts, const uint8_t* symbols, int nsym) {
        std::memcpy(vals, symbols, (size_t)nsym);
        std::memset(lut, 0, sizeof lut);
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k; mincode[l] = code;
            for (int i = 0; i < counts[l - 1]; i++, k++, code++)
                if (l <= 9 && k < nsym)
                    for (int f = 0; f < (1 << (9 - l)); f++) {
                        const int idx = (code << (9 - l)) | f;
                        if (idx < 512) lut[idx] = (uint16_t)((l << 8) | vals[k]);
                    }
            maxcode[l] = counts[l - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        ok = true;
    }
};

// Entropy-coded segment reader: a 64-bit window refilled bytewise (0xFF 0x00 unstuffed); at a marker it stops consuming (`p` stays on the
// 0xFF) and feeds zero bits until the caller has dealt with the marker.
struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int n = 0; bool marker = false;
    void fill() {
        while (n <= 56) {
            int b = 0;
            if (p < end && !marker) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0) p++;               // stuffed zero
                    else { marker = true; b = 0; p--; }        // a marker: zeros from here on
                }
            }
            acc |= (uint64_t)b << (56 - n);
            n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return (int)(acc >> (64 - k)); }                 // 1 <= k <= 16
    void skip(int k) { acc <<= k; n -= k; }
    int bit() { const int v = peek(1); skip(1); return v; }
    int bits(int k) { if (k == 0) return 0; const int v = peek(k); skip(k); return v; }
    void align() { const int r = n & 7; acc <<= r; n -= r; }                              // drop the pad bits before a marker
};

static inline int decode_symbol(BitReader& br, const Huff& h) {
    const int look = br.peek(16);
    const int e = h.lut[look >> 7];
    if (e) { br.skip(e >> 8); return e & 0xff; }
    for (int l = 10; l <= 16; l++) {
        const int code = look >> (16 - l);
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) { br.skip(l); return h.vals[h.valptr[l] + code - h.mincode[l]]; }
    }
    return -1;
}

static inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

// 8x8 inverse DCT, level shift + clamp to u8: the AAN factorisation libjpeg's float decoder uses (5 multiplies per 1-D pass; the per-
// frequency scale factors and the 1/8 are folded into the dequantisation table, `idct_scale`), columns whose AC terms are all zero are a
// copy.  Agrees with the textbook separable IDCT to 1e-3 of a level on coefficients of +-1000.
static inline float idct_scale(int natural_index) {
    static const float AAN[8] = {1.0f, 1.387039845f, 1.306562965f, 1.175875602f, 1.0f, 0.785694958f, 0.541196100f, 0.275899379f};
    return AAN[natural_index >> 3] * AAN[natural_index & 7] * 0.125f;
}
static inline void idct_1d(const float* i0, int is, float* o0, int os) {
    float tmp0 = i0[0 * is], tmp1 = i0[2 * is], tmp2 = i0[4 * is], tmp3 = i0[6 * is];
    float tmp10 = tmp0 + tmp2, tmp11 = tmp0 - tmp2;
    float tmp13 = tmp1 + tmp3, tmp12 = (tmp1 - tmp3) * 1.414213562f - tmp13;
    tmp0 = tmp10 + tmp13; tmp3 = tmp10 - tmp13; tmp1 = tmp11 + tmp12; tmp2 = tmp11 - tmp12;
    float tmp4 = i0[1 * is], tmp5 = i0[3 * is], tmp6 = i0[5 * is], tmp7 = i0[7 * is];
    const float z13 = tmp6 + tmp5, z10 = tmp6 - tmp5, z11 = tmp4 + tmp7, z12 = tmp4 - tmp7;
    tmp7 = z11 + z13;
    tmp11 = (z11 - z13) * 1.414213562f;
    const float z5 = (z10 + z12) * 1.847759065f;
    tmp10 = 1.082392200f * z12 - z5;
    tmp12 = -2.613125930f * z10 + z5;
    tmp6 = tmp12 - tmp7; tmp5 = tmp11 - tmp6; tmp4 = tmp10 + tmp5;
    o0[0 * os] = tmp0 + tmp7; o0[7 * os] = tmp0 - tmp7;
    o0[1 * os] = tmp1 + tmp6; o0[6 * os] = tmp1 - tmp6;
    o0[2 * os] = tmp2 + tmp5; o0[5 * os] = tmp2 - tmp5;
    o0[4 * os] = tmp3 + tmp4; o0[3 * os] = tmp3 - tmp4;
}
static inline void idct8x8(const float* in /* coefficients x idct_scale, natural order */, uint8_t* out, int stride) {
    float t[64];
    bool dc_only = true;
    for (int u = 0; u < 8; u++) {
        if (in[8 + u] == 0.f && in[16 + u] == 0.f && in[24 + u] == 0.f && in[32 + u] == 0.f && in[40 + u] == 0.f && in[48 + u] == 0.f && in[56 + u] == 0.f) {
            for (int v = 0; v < 8; v++) t[v * 8 + u] = in[u];
            if (u && in[u] != 0.f) dc_only = false;
        } else { idct_1d(in + u, 8, t + u, 8); dc_only = false; }
    }
    if (dc_only) {
        const float r = in[0] + 128.5f;                    // round half up (same as lround wherever the clamp does not decide)
        const uint8_t px = (uint8_t)(r < 0.f ? 0 : r >= 255.f ? 255 : (int)r);
        for (int y = 0; y < 8; y++) std::memset(out + y * stride, px, 8);
        return;
    }
    for (int v = 0; v < 8; v++) {
        float row[8];
        idct_1d(t + v * 8, 1, row, 1);
        for (int x = 0; x < 8; x++) {
            const float r = row[x] + 128.5f;
            out[v * stride + x] = (uint8_t)(r < 0.f ? 0 : r >= 255.f ? 255 : (int)r);
        }
    }
}

inline bool decode(const std::vector<unsigned char>& d, int& w, int& h, std::vector<unsigned char>& rgb, std::string* why = nullptr) {
    auto fail = [&](const char* m) { if (why) *why = m; return false; };
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail("not a JPEG");
    float qt[4][64]; bool have_qt[4] = {false, false, false, false};      // natural order, pre-multiplied by idct_scale
    Huff dc[4], ac[4];
    struct Comp {
        int id, hs, vs, tq, td, ta; int bw, bh; std::vector<uint8_t> plane; int pred;
        // multi-scan files (progressive, or sequential with one scan per component): all coefficients are kept until the last scan
        std::vector<int16_t> coef; int nbx = 0, nby = 0;      // block grid padded to whole MCUs
        int cbw = 0, cbh = 0;                                 // blocks that cover the component itself (the grid of a non-interleaved scan)
    } comp[3];
    int ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    bool have_sof = false, progressive = false, multi = false;
    // chroma upsampling and colour conversion of the decoded planes (shared by the single-scan and the multi-scan path)
    auto finish = [&]() -> bool {
        // chroma upsampling: libjpeg's "fancy" triangle filters for the two common layouts (h2v1 = 4:2:2, h2v2 = 4:2:0; the same
        // 3:1 weights stb_image uses), replication for anything else; then JFIF YCbCr -> RGB
        std::vector<uint8_t> full[3];
        for (int c = 0; c < ncomp; c++) {
            full[c].resize((size_t)w * h);
            const Comp& C = comp[c];
            const int cw = (w * C.hs + hmax - 1) / hmax, chh = (h * C.vs + vmax - 1) / vmax;      // valid samples of this component
            auto S = [&](int yy, int xx) -> int { return C.plane[(size_t)std::min(std::max(yy, 0), chh - 1) * C.bw + std::min(std::max(xx, 0), cw - 1)]; };
            const bool h2 = hmax == 2 && C.hs == 1, v2 = vmax == 2 && C.vs == 1;
            if (C.hs == hmax && C.vs == vmax) {
                for (int y = 0; y < h; y++) std::memcpy(&full[c][(size_t)y * w], &C.plane[(size_t)y * C.bw], (size_t)w);
            } else if (h2 && !v2 && C.vs == vmax) {                                                 // h2v1
                for (int y = 0; y < h; y++) {
                    const uint8_t* p = &C.plane[(size_t)std::min(y, chh - 1) * C.bw];
                    uint8_t* o = &full[c][(size_t)y * w];
                    for (int x = 0; x < w; x++) {
                        const int i = x >> 1;
                        o[x] = (uint8_t)((cw == 1 || x == 0 || x == 2 * cw - 1) ? p[std::min(i, cw - 1)] : (x & 1) ? (p[i] * 3 + p[i + 1] + 2) >> 2 : (p[i] * 3 + p[i - 1] + 1) >> 2);
                    }
                }
            } else if (h2 && v2) {                                                                  // h2v2
                std::vector<int> cs(cw);                                                            // 3 * nearer row + farther row, per chroma column
                for (int y = 0; y < h; y++) {
                    const int r = std::min(y >> 1, chh - 1), rn = std::min(std::max((y & 1) ? r + 1 : r - 1, 0), chh - 1);      // edges replicate
                    const uint8_t* p0 = &C.plane[(size_t)r * C.bw];
                    const uint8_t* p1 = &C.plane[(size_t)rn * C.bw];
                    for (int i = 0; i < cw; i++) cs[i] = p0[i] * 3 + p1[i];
                    uint8_t* o = &full[c][(size_t)y * w];
                    for (int x = 0; x < w; x++) {
                        const int i = x >> 1;
                        int v;
                        if (cw == 1 || x == 0) v = (cs[0] * 4 + 8) >> 4;
                        else if (x == 2 * cw - 1) v = (cs[cw - 1] * 4 + 7) >> 4;
                        else if (x & 1) v = (cs[i] * 3 + cs[i + 1] + 7) >> 4;
                        else v = (cs[i] * 3 + cs[i - 1] + 8) >> 4;
                        o[x] = (uint8_t)v;
                    }
                }
            } else {
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) full[c][(size_t)y * w + x] = (uint8_t)S(y * C.vs / vmax, x * C.hs / hmax);
            }
        }
        rgb.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            unsigned char* o = &rgb[i * 3];
            if (ncomp == 1) { o[0] = o[1] = o[2] = full[0][i]; continue; }
            // JFIF YCbCr -> RGB in 16-bit fixed point (libjpeg's constants); the arithmetic shift floors, + 32768 rounds
            const int Y = full[0][i], cb = (int)full[1][i] - 128, cr = (int)full[2][i] - 128;
            const int ri = Y + ((91881 * cr + 32768) >> 16), gi = Y - ((22554 * cb + 46802 * cr + 32768) >> 16), bi = Y + ((116130 * cb + 32768) >> 16);
            o[0] = (uint8_t)(ri < 0 ? 0 : ri > 255 ? 255 : ri); o[1] = (uint8_t)(gi < 0 ? 0 : gi > 255 ? 255 : gi); o[2] = (uint8_t)(bi < 0 ? 0 : bi > 255 ? 255 : bi);
        }
        return true;
    };
    // ---- multi-scan files: coefficient store, one decoder for every kind of scan, reconstruction at the end ----
    auto setup_geometry = [&]() {
        const int mx = (w + 8 * hmax - 1) / (8 * hmax), my = (h + 8 * vmax - 1) / (8 * vmax);
        for (int c = 0; c < ncomp; c++) {
            Comp& C = comp[c];
            C.nbx = mx * C.hs; C.nby = my * C.vs;
            C.bw = C.nbx * 8; C.bh = C.nby * 8;
            C.cbw = ((w * C.hs + hmax - 1) / hmax + 7) / 8; C.cbh = ((h * C.vs + vmax - 1) / vmax + 7) / 8;
        }
    };
    // Sequential scans are the case Ss = 0, Se = 63, Ah = Al = 0 of the progressive ones (an EOB is an EOB run of one block).
    auto decode_scan = [&](const int* sc, int ns, int Ss, int Se, int Ah, int Al, const uint8_t* start) -> bool {
        BitReader br{start, d.data() + d.size()};
        int eobrun = 0, count = 0;
        for (int k = 0; k < ns; k++) comp[sc[k]].pred = 0;
        const int p1 = 1 << Al, m1 = -(1 << Al);
        auto block = [&](Comp& C, int bx, int by) -> bool {
            int16_t* q = &C.coef[((size_t)by * C.nbx + bx) * 64];
            int k = Ss;
            if (Ss == 0) {
                if (Ah == 0) {
                    const int t = decode_symbol(br, dc[C.td]);
                    if (t < 0 || t > 15) return false;
                    C.pred += t ? extend(br.bits(t), t) : 0;
                    q[0] = (int16_t)(C.pred * p1);
                } else if (br.bit()) q[0] = (int16_t)(q[0] | p1);
                k = 1;
            }
            if (Se == 0) return true;
            if (Ah == 0) {                                     // first pass over these coefficients
                if (eobrun > 0) { eobrun--; return true; }
                for (; k <= Se; k++) {
                    const int rs = decode_symbol(br, ac[C.ta]);
                    if (rs < 0) return false;
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                        if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.bits(r); break; }
                        k += 15;
                    } else {
                        k += r;
                        if (k > 63) return false;
                        q[ZIGZAG[k]] = (int16_t)(extend(br.bits(sz), sz) * p1);
                    }
                }
                return true;
            }
            // refinement pass: one more bit for the coefficients that are already non-zero, new +-1 coefficients in between
            if (eobrun == 0) {
                for (; k <= Se; k++) {
                    const int rs = decode_symbol(br, ac[C.ta]);
                    if (rs < 0) return false;
                    int r = rs >> 4, sz = rs & 15, val = 0;
                    if (sz == 0) {
                        if (r < 15) { eobrun = 1 << r; if (r) eobrun += br.bits(r); break; }      // counts this block too
                    } else val = br.bit() ? p1 : m1;
                    for (; k <= Se; k++) {
                        int16_t& cf = q[ZIGZAG[k]];
                        if (cf != 0) {
                            if (br.bit() && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
                        } else if (--r < 0) break;
                    }
                    if (val && k <= Se) q[ZIGZAG[k]] = (int16_t)val;
                }
            }
            if (eobrun > 0) {
                for (; k <= Se; k++) {
                    int16_t& cf = q[ZIGZAG[k]];
                    if (cf != 0 && br.bit() && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
                }
                eobrun--;
            }
            return true;
        };
        auto at_restart = [&]() {
            if (!(restart && count && count % restart == 0)) return;
            br.align();
            while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
            if (br.p + 1 < br.end) br.p += 2;
            br.marker = false; br.n = 0; br.acc = 0;
            for (int k = 0; k < ns; k++) comp[sc[k]].pred = 0;
            eobrun = 0;
        };
        if (ns == 1) {                                         // non-interleaved: the component's own block grid, row by row
            Comp& C = comp[sc[0]];
            for (int by = 0; by < C.cbh; by++)
                for (int bx = 0; bx < C.cbw; bx++) { at_restart(); count++; if (!block(C, bx, by)) return false; }
            return true;
        }
        const int mx = comp[0].nbx / comp[0].hs, my = comp[0].nby / comp[0].vs;
        for (int yy = 0; yy < my; yy++)
            for (int xx = 0; xx < mx; xx++) {
                at_restart(); count++;
                for (int k = 0; k < ns; k++) {
                    Comp& C = comp[sc[k]];
                    for (int by = 0; by < C.vs; by++)
                        for (int bx = 0; bx < C.hs; bx++) if (!block(C, xx * C.hs + bx, yy * C.vs + by)) return false;
                }
            }
        return true;
    };
    auto reconstruct = [&]() -> bool {
        for (int c = 0; c < ncomp; c++) {
            Comp& C = comp[c];
            if (!have_qt[C.tq]) return fail("missing quantisation table");
            C.plane.assign((size_t)C.bw * C.bh, 0);
            for (int by = 0; by < C.nby; by++)
                for (int bx = 0; bx < C.nbx; bx++) {
                    const int16_t* q = &C.coef[((size_t)by * C.nbx + bx) * 64];
                    float blk[64];
                    for (int k = 0; k < 64; k++) blk[k] = (float)q[k] * qt[C.tq][k];
                    idct8x8(blk, &C.plane[(size_t)by * 8 * C.bw + bx * 8], C.bw);
                }
        }
        return finish();
    };
    size_t pos = 2;
    while (pos + 4 <= d.size()) {
        if (d[pos] != 0xFF) { pos++; continue; }
        const int m = d[pos + 1];
        if (m == 0xFF) { pos++; continue; }
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > d.size()) return fail("truncated");
        const size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
        if (len < 2 || pos + len > d.size()) return fail("truncated segment");
        const uint8_t* s = &d[pos + 2]; const size_t n = len - 2;
        if (m == 0xDB) {
            size_t i = 0;
            while (i < n) {
                const int pq = s[i] >> 4, tq = s[i] & 15; i++;
                if (tq > 3 || i + (pq ? 128 : 64) > n) return fail("bad DQT");
                for (int k = 0; k < 64; k++) { qt[tq][ZIGZAG[k]] = (float)(pq ? ((s[i] << 8) | s[i + 1]) : s[i]) * idct_scale(ZIGZAG[k]); i += pq ? 2 : 1; }
                have_qt[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t i = 0;
            while (i + 17 <= n) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                int total = 0; for (int k = 0; k < 16; k++) total += s[i + 1 + k];
                if (th > 3 || total > 256 || i + 17 + total > n) return fail("bad DHT");
                (tc ? ac[th] : dc[th]).build(&s[i + 1], &s[i + 17], total);
                i += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            // one frame header per image: a second SOF after the geometry / coefficient store have been set up would re-size w, h, ncomp and
            // the sampling factors under them (heap overflow in finish(): ADVICE r1)
            if (have_sof) return fail("duplicate SOF");
            progressive = m == 0xC2;
            if (n < 6 || s[0] != 8) return fail("only 8-bit samples are supported");
            h = (s[1] << 8) | s[2]; w = (s[3] << 8) | s[4]; ncomp = s[5];
            if ((ncomp != 1 && ncomp != 3) || n < 6 + 3 * (size_t)ncomp || w <= 0 || h <= 0) return fail("unsupported component count");
            // every 8x8 block costs at least a few bits of entropy-coded data: a header that promises far more blocks than the file can hold is corrupt
            if ((size_t)w * h > ((size_t)1 << 28) || (size_t)w * h / 64 > d.size() * 16 + 4096) return fail("image size does not fit the file");
            for (int c = 0; c < ncomp; c++) {
                comp[c].id = s[6 + 3 * c]; comp[c].hs = s[7 + 3 * c] >> 4; comp[c].vs = s[7 + 3 * c] & 15; comp[c].tq = s[8 + 3 * c];
                if (comp[c].hs < 1 || comp[c].hs > 2 || comp[c].vs < 1 || comp[c].vs > 2 || comp[c].tq > 3) return fail("unsupported sampling factors");
                hmax = std::max(hmax, comp[c].hs); vmax = std::max(vmax, comp[c].vs);
            }
            // a single-component scan is never interleaved: its MCU is one block whatever the header's sampling factors say (like libjpeg / stb_image)
            if (ncomp == 1) { comp[0].hs = comp[0].vs = 1; hmax = vmax = 1; }
            have_sof = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return fail("lossless / hierarchical / arithmetic-coded JPEG is not supported");
        } else if (m == 0xDD) {
            if (n >= 2) restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof) return fail("SOS before SOF");
            if (n < 1 || s[0] < 1 || s[0] > ncomp || n < 1 + 2 * (size_t)s[0] + 3) return fail("bad SOS");
            if (progressive || multi || s[0] != ncomp) {
                // one of several scans: decode it into the coefficient store, then go on with the marker after its entropy-coded data
                const int ns = s[0];
                int sc[3];
                if (!multi) {
                    setup_geometry();
                    for (int c = 0; c < ncomp; c++) comp[c].coef.assign((size_t)comp[c].nbx * comp[c].nby * 64, 0);
                    multi = true;
                }
                for (int k = 0; k < ns; k++) {
                    int ci = -1;
                    for (int c = 0; c < ncomp; c++) if (comp[c].id == s[1 + 2 * k]) ci = c;
                    if (ci < 0) return fail("bad scan component");
                    comp[ci].td = s[2 + 2 * k] >> 4; comp[ci].ta = s[2 + 2 * k] & 15;
                    if (comp[ci].td > 3 || comp[ci].ta > 3) return fail("bad table selector");
                    sc[k] = ci;
                }
                const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
                if (Ss > Se || Se > 63 || Al > 13 || (Ss > 0 && ns != 1) || (!progressive && (Ss != 0 || Se != 63 || Ah || Al))) return fail("bad scan parameters");
                for (int k = 0; k < ns; k++) {
                    if (Ss == 0 && Ah == 0 && !dc[comp[sc[k]].td].ok) return fail("missing table");
                    if (Se > 0 && !ac[comp[sc[k]].ta].ok) return fail("missing table");
                }
                if (!decode_scan(sc, ns, Ss, Se, Ah, Al, &d[pos + len])) return fail("corrupt scan data");
                size_t q = pos + len;                        // next marker that is not a restart marker or a stuffed 0xFF
                while (q + 1 < d.size() && !(d[q] == 0xFF && d[q + 1] != 0x00 && d[q + 1] != 0xFF && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7))) q++;
                pos = q;
                continue;
            }
            for (int k = 0; k < ncomp; k++) {
                int ci = -1;
                for (int c = 0; c < ncomp; c++) if (comp[c].id == s[1 + 2 * k]) ci = c;
                if (ci < 0) return fail("bad scan component");
                comp[ci].td = s[2 + 2 * k] >> 4; comp[ci].ta = s[2 + 2 * k] & 15;
                if (comp[ci].td > 3 || comp[ci].ta > 3) return fail("bad table selector");
                if (!dc[comp[ci].td].ok || !ac[comp[ci].ta].ok || !have_qt[comp[ci].tq]) return fail("missing table");
            }
            const int mcuw = 8 * hmax, mcuh = 8 * vmax;
            const int mx = (w + mcuw - 1) / mcuw, my = (h + mcuh - 1) / mcuh;
            for (int c = 0; c < ncomp; c++) {
                comp[c].bw = mx * comp[c].hs * 8; comp[c].bh = my * comp[c].vs * 8;
                comp[c].plane.assign((size_t)comp[c].bw * comp[c].bh, 0);
                comp[c].pred = 0;
            }
            BitReader br{&d[pos + len], d.data() + d.size()};
            int count = 0;
            for (int yy = 0; yy < my; yy++)
                for (int xx = 0; xx < mx; xx++) {
                    if (restart && count && count % restart == 0) {
                        br.align();
                        // skip to and over the RSTn marker
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) br.p++;
                        if (br.p + 1 < br.end) br.p += 2;
                        br.marker = false; br.n = 0; br.acc = 0;
                        for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
                    }
                    count++;
                    for (int c = 0; c < ncomp; c++)
                        for (int by = 0; by < comp[c].vs; by++)
                            for (int bx = 0; bx < comp[c].hs; bx++) {
                                float blk[64];
                                for (int k = 0; k < 64; k++) blk[k] = 0.f;
                                const int t = decode_symbol(br, dc[comp[c].td]);
                                if (t < 0 || t > 11) return fail("bad DC code");
                                const int diff = t ? extend(br.bits(t), t) : 0;
                                comp[c].pred += diff;
                                blk[0] = (float)comp[c].pred * qt[comp[c].tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_symbol(br, ac[comp[c].ta]);
                                    if (rs < 0) return fail("bad AC code");
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) return fail("AC run past the block");
                                    blk[ZIGZAG[k]] = (float)extend(br.bits(sz), sz) * qt[comp[c].tq][ZIGZAG[k]];
                                    k++;
                                }
                                const int px = (xx * comp[c].hs + bx) * 8, py = (yy * comp[c].vs + by) * 8;
                                idct8x8(blk, &comp[c].plane[(size_t)py * comp[c].bw + px], comp[c].bw);
                            }
                }
            return finish();
        }
        pos += len;
    }
    if (multi) return reconstruct();
    return fail("no scan found");
}

// ------------------------------------------------------------------- encoder -------------------------------------------------------------------
// standard Huffman tables of ITU T.81 Annex K.3
static const uint8_t DC_L_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, DC_C_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_L_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, AC_C_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t AC_L_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1,
    0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39,
    0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8,
    0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t AC_C_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09,
    0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38,
    0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct EncTable { uint16_t code[256]; uint8_t len[256]; };
static inline void make_enc(const uint8_t* bits, const uint8_t* vals, EncTable& t) {
    std::memset(&t, 0, sizeof t);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++, k++) { t.code[vals[k]] = (uint16_t)code; t.len[vals[k]] = (uint8_t)l; code++; }
        code <<= 1;
    }
}

struct BitWriter {
    std::vector<unsigned char>& out; uint64_t acc = 0; int n = 0;      // n bits pending in the low end of acc (n < 32 between calls)
    void put(int code, int len) {                                      // len <= 26
        acc = (acc << len) | (uint64_t)((uint32_t)code & ((1u << len) - 1u)); n += len;
        while (n >= 8) { const uint8_t b = (uint8_t)(acc >> (n - 8)); out.push_back(b); if (b == 0xFF) out.push_back(0); n -= 8; }
    }
    void flush() { if (n) put(0x7F, 8 - n); }
};

// forward DCT, AAN factorisation (libjpeg's float method): out = true coefficients x 8 x AAN[u] x AAN[v]; `fdct_descale` undoes that.
// Agrees with the textbook separable DCT to 2e-4 on 8-bit samples.
static inline void fdct_1d(float* d, int s) {
    const float tmp0 = d[0] + d[7 * s], tmp7 = d[0] - d[7 * s], tmp1 = d[s] + d[6 * s], tmp6 = d[s] - d[6 * s];
    const float tmp2 = d[2 * s] + d[5 * s], tmp5 = d[2 * s] - d[5 * s], tmp3 = d[3 * s] + d[4 * s], tmp4 = d[3 * s] - d[4 * s];
    float tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    d[0] = tmp10 + tmp11; d[4 * s] = tmp10 - tmp11;
    const float z1 = (tmp12 + tmp13) * 0.707106781f;
    d[2 * s] = tmp13 + z1; d[6 * s] = tmp13 - z1;
    tmp10 = tmp4 + tmp5; tmp11 = tmp5 + tmp6; tmp12 = tmp6 + tmp7;
    const float z5 = (tmp10 - tmp12) * 0.382683433f;
    const float z2 = 0.541196100f * tmp10 + z5, z4 = 1.306562965f * tmp12 + z5, z3 = tmp11 * 0.707106781f;
    const float z11 = tmp7 + z3, z13 = tmp7 - z3;
    d[5 * s] = z13 + z2; d[3 * s] = z13 - z2; d[s] = z11 + z4; d[7 * s] = z11 - z4;
}
static inline void fdct8x8(float* blk /* in place */) {
    for (int y = 0; y < 8; y++) fdct_1d(blk + y * 8, 1);
    for (int x = 0; x < 8; x++) fdct_1d(blk + x, 8);
}
static inline float fdct_descale(int natural_index) { return 1.0f / (idct_scale(natural_index) * 64.0f); }      // 1 / (8 AAN[u] AAN[v])

static inline void seg(std::vector<unsigned char>& o, int marker, const std::vector<unsigned char>& body) {
    o.push_back(0xFF); o.push_back((unsigned char)marker);
    const size_t len = body.size() + 2;
    o.push_back((unsigned char)(len >> 8)); o.push_back((unsigned char)len);
    o.insert(o.end(), body.begin(), body.end());
}

// quality 100 (all-ones quantisation), 4:4:4, JFIF
inline bool encode(const std::string& path, int w, int h, const unsigned char* rgb, int helpers = 0) {
    std::vector<unsigned char> o = {0xFF, 0xD8};
    seg(o, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
    { std::vector<unsigned char> q(65, 1); q[0] = 0; seg(o, 0xDB, q); }
    seg(o, 0xC0, {8, (unsigned char)(h >> 8), (unsigned char)h, (unsigned char)(w >> 8), (unsigned char)w, 3, 1, 0x11, 0, 2, 0x11, 0, 3, 0x11, 0});
    auto dht = [&](int tc_th, const uint8_t* bits, const uint8_t* vals, int n) {
        std::vector<unsigned char> b; b.push_back((unsigned char)tc_th); b.insert(b.end(), bits, bits + 16); b.insert(b.end(), vals, vals + n); seg(o, 0xC4, b);
    };
    dht(0x00, DC_L_BITS, DC_VALS, 12); dht(0x10, AC_L_BITS, AC_L_VALS, 162); dht(0x01, DC_C_BITS, DC_VALS, 12); dht(0x11, AC_C_BITS, AC_C_VALS, 162);
    seg(o, 0xDA, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});
    EncTable dcl, acl, dcc, acc_;
    make_enc(DC_L_BITS, DC_VALS, dcl); make_enc(AC_L_BITS, AC_L_VALS, acl); make_enc(DC_C_BITS, DC_VALS, dcc); make_enc(AC_C_BITS, AC_C_VALS, acc_);
    // phase 1 (independent per block row, spread over `helpers` extra threads): colour conversion, DCT, rounding -> zig-zag int16 coefficients
    const int nbx = (w + 7) / 8, nby = (h + 7) / 8;
    std::vector<int16_t> coef((size_t)nbx * nby * 3 * 64);
    float descale[64];
    for (int k = 0; k < 64; k++) descale[k] = fdct_descale(ZIGZAG[k]);
    std::atomic<int> next_row(0);
    auto transform = [&]() {
        for (int byi; (byi = next_row.fetch_add(1)) < nby;) {
            const int by = byi * 8;
            for (int bxi = 0; bxi < nbx; bxi++) {
                const int bx = bxi * 8;
                float blk[3][64];
                for (int y = 0; y < 8; y++) {
                    const unsigned char* row = rgb + (size_t)std::min(by + y, h - 1) * w * 3;
                    for (int x = 0; x < 8; x++) {
                        const unsigned char* p = row + (size_t)std::min(bx + x, w - 1) * 3;
                        const float r = p[0], g = p[1], b = p[2];
                        blk[0][y * 8 + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.f;
                        blk[1][y * 8 + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                        blk[2][y * 8 + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
                    }
                }
                int16_t* q = &coef[((size_t)byi * nbx + bxi) * 3 * 64];
                for (int c = 0; c < 3; c++) {
                    fdct8x8(blk[c]);
                    for (int k = 0; k < 64; k++) {
                        const float f = blk[c][ZIGZAG[k]] * descale[k];
                        const int v = (int)(f < 0.f ? f - 0.5f : f + 0.5f);                 // round half away from zero
                        q[c * 64 + k] = (int16_t)(v < -1023 ? -1023 : v > 1023 ? 1023 : v);     // size categories <= 10 (AC) / 11 (DC diff)
                    }
                }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 0; i < std::min(helpers, nby - 1); i++) th.emplace_back(transform);
        transform();
        for (auto& t : th) t.join();
    }
    // phase 2: entropy coding, in block order
    o.reserve(o.size() + coef.size());
    BitWriter bw{o};
    int pred[3] = {0, 0, 0};
    auto ncat = [](int a) { return a ? 32 - __builtin_clz((unsigned)a) : 0; };
    for (size_t blk = 0; blk < (size_t)nbx * nby; blk++)
        for (int c = 0; c < 3; c++) {
            const int16_t* q = &coef[(blk * 3 + c) * 64];
            const EncTable& dct = c ? dcc : dcl; const EncTable& act = c ? acc_ : acl;
            const int diff = q[0] - pred[c]; pred[c] = q[0];
            const int t = ncat(diff < 0 ? -diff : diff);
            bw.put(dct.code[t], dct.len[t]);
            if (t) bw.put(diff < 0 ? diff - 1 : diff, t);
            int run = 0;
            for (int k = 1; k < 64; k++) {
                const int v = q[k];
                if (v == 0) { run++; continue; }
                while (run > 15) { bw.put(act.code[0xF0], act.len[0xF0]); run -= 16; }
                const int sz = ncat(v < 0 ? -v : v);
                bw.put((act.code[(run << 4) | sz] << sz) | ((v < 0 ? v - 1 : v) & ((1 << sz) - 1)), act.len[(run << 4) | sz] + sz);
                run = 0;
            }
            if (run) bw.put(act.code[0], act.len[0]);
        }
    bw.flush();
    o.push_back(0xFF); o.push_back(0xD9);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    fclose(f);
    return ok;
}

}  // namespace jpeg
