// Grayscale decoder for baseline / extended-sequential and progressive Huffman JPEG files, for the
// image loader (ref:src/image_loader.cpp:124 reads .jpg/.jpeg through cv::imread(IMREAD_GRAYSCALE)).
// libjpeg, which OpenCV wraps, delivers the luma plane of a YCbCr file when asked for
// grayscale: the chroma blocks are entropy-decoded (to stay in step with the bit stream) and
// dropped, the luma blocks go through the integer "slow" inverse DCT that libjpeg and
// libjpeg-turbo use by default, restated here so that the pixels are the same.
// A progressive file (SOF2) is read scan by scan into a coefficient array of the luma component -
// DC and AC bands, first passes and successive-approximation refinements as ITU-T T.81 annex G
// defines them; scans that carry no luma are stepped over - and transformed once the last scan
// is in (libjpeg smooths blocks only while coefficient bits are still missing, which a complete
// file does not leave).  Arithmetic-coded, lossless, 12-bit, CMYK and Adobe-RGB files are refused
// (the loader then reports "Failed to load image").
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include <visual_odometry/image_loader.hpp>

namespace visual_odometry::detail {

namespace {

constexpr int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present = false;
    uint8_t vals[256] = {};
    int mincode[17] = {}, maxcode[17] = {}, valptr[17] = {};
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;   // tables of the current scan
    int pred = 0;         // DC predictor
};

// Entropy-coded segment reader: a 0xFF data byte is followed by a stuffed 0x00; any other byte
// after 0xFF is a marker, which ends the segment (the reader then supplies zero bits).
struct BitReader {
    uint8_t const* p;
    uint8_t const* end;
    uint32_t acc = 0;
    int nbits = 0, nreal = 0;   // bits in the accumulator; how many of them came from the file
    bool at_marker = false, overrun = false;   // overrun: a bit was asked for that the segment does not have

    void fill() {
        while (nbits <= 24) {
            uint32_t b = 0;
            if (!at_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;
                    else { at_marker = true; b = 0; }   // (libjpeg pads a short segment with zero bits and
                                                        // carries on with a warning; here the file is refused)
                } else {
                    ++p;
                }
                if (!at_marker) nreal += 8;
            } else {
                at_marker = true;
            }
            acc |= b << (24 - nbits);
            nbits += 8;
        }
    }
    int bit() {
        if (nbits == 0) fill();
        int const r = (int)(acc >> 31);
        acc <<= 1;
        --nbits;
        if (nreal > 0) --nreal;
        else overrun = true;
        return r;
    }
    int bits(int n) {
        int v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | bit();
        return v;
    }
    void reset() { acc = 0; nbits = 0; nreal = 0; at_marker = false; }
};

int decode_symbol(BitReader& br, Huffman const& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// jidctint.c: 13-bit constants, two extra bits kept between the passes; 64-bit intermediates as
// libjpeg's JLONG on LP64 (legal streams stay far inside 32 bits)
constexpr int kConstBits = 13, kPass1Bits = 2;
constexpr int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633,
                  F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
inline int64_t descale(int64_t x, int n) { return (x + (1 << (n - 1))) >> n; }

inline void idct_1d(int64_t const in[8], int64_t out[8], int shift) {
    int64_t z2 = in[2], z3 = in[6];
    int64_t z1 = (z2 + z3) * F0_541;
    int64_t tmp2 = z1 + z3 * (-F1_847);
    int64_t tmp3 = z1 + z2 * F0_765;
    z2 = in[0];
    z3 = in[4];
    int64_t tmp0 = (z2 + z3) * (1 << kConstBits);
    int64_t tmp1 = (z2 - z3) * (1 << kConstBits);
    int64_t const tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int64_t z4 = tmp1 + tmp3;
    int64_t const z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298;
    tmp1 *= F2_053;
    tmp2 *= F3_072;
    tmp3 *= F1_501;
    z1 *= -F0_899;
    z2 *= -F2_562;
    z3 *= -F1_961;
    z4 *= -F0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, shift);
    out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift);
    out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift);
    out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift);
    out[4] = descale(tmp13 - tmp0, shift);
}

// coef: dequantised coefficients in natural order; dst: 8 rows of `stride` bytes
void idct_block(int32_t const coef[64], uint8_t* dst, size_t stride) {
    int64_t ws[64];
    for (int c = 0; c < 8; ++c) {   // pass 1: columns
        int64_t in[8], out[8];
        for (int r = 0; r < 8; ++r) in[r] = coef[r * 8 + c];
        idct_1d(in, out, kConstBits - kPass1Bits);
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = out[r];
    }
    for (int r = 0; r < 8; ++r) {   // pass 2: rows, level shift and clamp
        int64_t out[8];
        idct_1d(&ws[r * 8], out, kConstBits + kPass1Bits + 3);
        for (int c = 0; c < 8; ++c) {
            int64_t const v = out[c] + 128;
            dst[r * stride + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

struct Decoder {
    std::vector<uint8_t> const& f;
    size_t pos = 2;
    uint16_t quant[4][64] = {};
    bool have_quant[4] = {};
    Huffman dc[4], ac[4];
    Component comp[4];
    int ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, restart = 0;
    bool have_frame = false, adobe_rgb = false;
    std::vector<uint8_t> plane;   // luma, padded to whole MCUs
    size_t plane_w = 0, plane_h = 0;
    // progressive files: the luma coefficients (natural order, not yet dequantised) of every block
    // of the padded plane, the luma quantisation table as it stood at the first luma scan
    bool progressive = false, luma_quant_latched = false;
    std::vector<int32_t> coefs;
    uint16_t luma_quant[64] = {};

    explicit Decoder(std::vector<uint8_t> const& file) : f(file) {}

    bool read_dqt(size_t p, size_t e) {
        while (p < e) {
            int const pq = f[p] >> 4, tq = f[p] & 15;
            ++p;
            if (tq > 3 || pq > 1 || p + (size_t)64 * (pq + 1) > e) return false;
            for (int i = 0; i < 64; ++i) {
                quant[tq][kZigzag[i]] = pq ? (uint16_t)((f[p] << 8) | f[p + 1]) : f[p];
                p += pq + 1;
            }
            have_quant[tq] = true;
        }
        return true;
    }

    bool read_dht(size_t p, size_t e) {
        while (p < e) {
            int const tc = f[p] >> 4, th = f[p] & 15;
            ++p;
            if (tc > 1 || th > 3 || p + 16 > e) return false;
            Huffman& h = tc ? ac[th] : dc[th];
            int counts[17] = {}, total = 0;
            for (int l = 1; l <= 16; ++l) {
                counts[l] = f[p + l - 1];
                total += counts[l];
            }
            p += 16;
            if (total > 256 || p + (size_t)total > e) return false;
            std::memcpy(h.vals, &f[p], (size_t)total);
            p += (size_t)total;
            int code = 0, k = 0;
            for (int l = 1; l <= 16; ++l) {
                h.valptr[l] = k;
                h.mincode[l] = code;
                code += counts[l];
                k += counts[l];
                h.maxcode[l] = counts[l] ? code - 1 : -1;
                if (code > (1 << l)) return false;   // over-subscribed code
                code <<= 1;
            }
            h.present = true;
        }
        return true;
    }

    bool read_sof(size_t p, size_t e) {
        if (have_frame || e - p < 6 || f[p] != 8) return false;   // one frame, 8-bit samples
        height = (f[p + 1] << 8) | f[p + 2];
        width = (f[p + 3] << 8) | f[p + 4];
        ncomp = f[p + 5];
        if (width <= 0 || height <= 0 || (ncomp != 1 && ncomp != 3) || e - p < (size_t)6 + 3 * (size_t)ncomp) return false;
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.id = f[p + 6 + 3 * i];
            c.h = f[p + 7 + 3 * i] >> 4;
            c.v = f[p + 7 + 3 * i] & 15;
            c.tq = f[p + 8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        // the luma plane is delivered as decoded only if it has the frame's resolution
        if (comp[0].h != hmax || comp[0].v != vmax) return false;
        if (ncomp == 3 && (adobe_rgb || (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B'))) return false;
        size_t const mcux = ((size_t)width + 8 * (size_t)hmax - 1) / (8 * (size_t)hmax);
        size_t const mcuy = ((size_t)height + 8 * (size_t)vmax - 1) / (8 * (size_t)vmax);
        plane_w = mcux * 8 * (size_t)hmax;
        plane_h = mcuy * 8 * (size_t)vmax;
        // (an image cannot hold more blocks than the file has bytes times a generous factor: a block
        // costs at least two code bits)
        if (plane_w * plane_h / 64 > f.size() * 8) return false;
        plane.assign(plane_w * plane_h, 0);
        if (progressive) coefs.assign(plane_w * plane_h, 0);
        have_frame = true;
        return true;
    }

    bool decode_block(BitReader& br, Component& c, bool keep, size_t bx, size_t by) {
        Huffman const& hd = dc[c.td];
        Huffman const& ha = ac[c.ta];
        int32_t coef[64] = {};
        int s = decode_symbol(br, hd);
        if (s < 0 || s > 11) return false;
        if (s) c.pred += extend(br.bits(s), s);
        // a DC value of an 8-bit image fits 12 bits; a predictor that a hostile stream has driven
        // out of 16 would overflow the product below
        if (c.pred < -32768 || c.pred > 32767) return false;
        uint16_t const* q = quant[c.tq];
        coef[0] = c.pred * (int32_t)q[0];
        for (int k = 1; k < 64;) {
            int const rs = decode_symbol(br, ha);
            if (rs < 0) return false;
            int const r = rs >> 4;
            s = rs & 15;
            if (s == 0) {
                if (r != 15) break;   // end of block
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return false;
            int const nat = kZigzag[k];
            coef[nat] = extend(br.bits(s), s) * q[nat];
            ++k;
        }
        if (keep) idct_block(coef, &plane[by * 8 * plane_w + bx * 8], plane_w);
        return true;
    }

    // One scan; `p` is the first byte after the scan header.  Returns the position of the marker
    // that ends the entropy-coded data, or 0 on error.
    size_t decode_scan(size_t p, int ns, int const* index) {
        BitReader br{f.data() + p, f.data() + f.size()};   // (p may equal f.size(): a file that ends behind the scan header)
        for (int i = 0; i < ns; ++i) comp[index[i]].pred = 0;
        size_t mcux, mcuy;
        if (ns == 1) {   // non-interleaved: the component's own block grid
            Component const& c = comp[index[0]];
            size_t const cw = ((size_t)width * (size_t)c.h + (size_t)hmax - 1) / (size_t)hmax;
            size_t const ch = ((size_t)height * (size_t)c.v + (size_t)vmax - 1) / (size_t)vmax;
            mcux = (cw + 7) / 8;
            mcuy = (ch + 7) / 8;
        } else {
            mcux = plane_w / (8 * (size_t)hmax);
            mcuy = plane_h / (8 * (size_t)vmax);
        }
        int until_restart = restart, next_rst = 0;
        for (size_t my = 0; my < mcuy; ++my)
            for (size_t mx = 0; mx < mcux; ++mx) {
                if (restart && until_restart == 0) {
                    // the restart marker sits on a byte boundary behind the bits consumed so far
                    uint8_t const* q = br.p;
                    while (q + 2 < br.end && q[0] == 0xFF && q[1] == 0xFF) ++q;   // fill bytes before a marker
                    if (q + 1 >= br.end || q[0] != 0xFF || q[1] != 0xD0 + next_rst) return 0;
                    br.p = q + 2;
                    br.reset();
                    next_rst = (next_rst + 1) & 7;
                    until_restart = restart;
                    for (int i = 0; i < ns; ++i) comp[index[i]].pred = 0;
                }
                for (int i = 0; i < ns; ++i) {
                    Component& c = comp[index[i]];
                    bool const keep = index[i] == 0;
                    if (ns == 1) {
                        if (!decode_block(br, c, keep, mx, my)) return 0;
                    } else {
                        for (int v = 0; v < c.v; ++v)
                            for (int h = 0; h < c.h; ++h)
                                if (!decode_block(br, c, keep, mx * (size_t)c.h + (size_t)h, my * (size_t)c.v + (size_t)v)) return 0;
                    }
                }
                if (br.overrun) return 0;
                if (restart) --until_restart;
            }
        // the marker after the data: skip what is left of the last byte and any fill
        uint8_t const* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF)) ++q;
        return (size_t)(q - f.data());
    }

    // ---- progressive scans (ITU-T T.81 annex G; the procedures of libjpeg's jdphuff.c)
    struct Band {
        int ss, se, ah, al;
    };

    // end of the entropy-coded data that starts at p: the next marker that is not a restart marker
    size_t skip_scan(size_t p) const {
        while (p + 1 < f.size() && !(f[p] == 0xFF && f[p + 1] != 0x00 && f[p + 1] != 0xFF && !(f[p + 1] >= 0xD0 && f[p + 1] <= 0xD7)))
            ++p;
        return p + 1 < f.size() ? p : 0;
    }

    // a correction bit for a coefficient that is already non-zero
    static void refine_nonzero(BitReader& br, int32_t& c, int32_t p1) {
        if (br.bit() && (c & p1) == 0) c += c >= 0 ? p1 : -p1;
    }

    bool ac_first(BitReader& br, Huffman const& ha, int32_t* blk, Band const& b, int& eobrun) {
        if (eobrun > 0) {
            --eobrun;
            return true;
        }
        for (int k = b.ss; k <= b.se; ++k) {
            int const rs = decode_symbol(br, ha);
            if (rs < 0) return false;
            int const r = rs >> 4, s = rs & 15;
            if (s) {
                k += r;
                if (k > b.se) return false;
                blk[kZigzag[k]] = extend(br.bits(s), s) * (1 << b.al);
            } else if (r == 15) {
                k += 15;
            } else {
                eobrun = 1 << r;
                if (r) eobrun += br.bits(r);
                --eobrun;
                break;
            }
        }
        return true;
    }

    bool ac_refine(BitReader& br, Huffman const& ha, int32_t* blk, Band const& b, int& eobrun) {
        int32_t const p1 = 1 << b.al;
        int k = b.ss;
        if (eobrun == 0) {
            for (; k <= b.se; ++k) {
                int const rs = decode_symbol(br, ha);
                if (rs < 0) return false;
                int r = rs >> 4;
                int32_t val = 0;
                if (rs & 15) {
                    if ((rs & 15) != 1) return false;   // a newly non-zero coefficient has magnitude one here
                    val = br.bit() ? p1 : -p1;
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += br.bits(r);
                    break;
                }
                // step over r coefficients that are still zero; those already non-zero on the way take a
                // correction bit each
                for (; k <= b.se; ++k) {
                    int32_t& c = blk[kZigzag[k]];
                    if (c != 0) refine_nonzero(br, c, p1);
                    else if (--r < 0) break;
                }
                if (val) {
                    if (k > b.se) return false;
                    blk[kZigzag[k]] = val;
                }
            }
        }
        if (eobrun > 0) {   // the rest of the band: only correction bits
            for (; k <= b.se; ++k) {
                int32_t& c = blk[kZigzag[k]];
                if (c != 0) refine_nonzero(br, c, p1);
            }
            --eobrun;
        }
        return true;
    }

    // One progressive scan that carries the luma component (index[i] == 0 for one i).
    size_t decode_scan_progressive(size_t p, int ns, int const* index, Band const& b) {
        BitReader br{f.data() + p, f.data() + f.size()};
        for (int i = 0; i < ns; ++i) comp[index[i]].pred = 0;
        size_t const bw = plane_w / 8;   // blocks per row of the coefficient array
        size_t mcux, mcuy;
        if (ns == 1) {
            Component const& c = comp[index[0]];
            size_t const cw = ((size_t)width * (size_t)c.h + (size_t)hmax - 1) / (size_t)hmax;
            size_t const ch = ((size_t)height * (size_t)c.v + (size_t)vmax - 1) / (size_t)vmax;
            mcux = (cw + 7) / 8;
            mcuy = (ch + 7) / 8;
        } else {
            mcux = plane_w / (8 * (size_t)hmax);
            mcuy = plane_h / (8 * (size_t)vmax);
        }
        int until_restart = restart, next_rst = 0, eobrun = 0;
        auto block = [&](Component& c, bool luma, size_t bx, size_t by) -> bool {
            int32_t dummy[64] = {};
            int32_t* blk = luma ? &coefs[(by * bw + bx) * 64] : dummy;
            if (b.ss == 0) {
                if (b.ah == 0) {   // DC, first pass
                    int const s = decode_symbol(br, dc[c.td]);
                    if (s < 0 || s > 11) return false;
                    if (s) c.pred += extend(br.bits(s), s);
                    if (c.pred < -32768 || c.pred > 32767) return false;
                    blk[0] = c.pred * (1 << b.al);
                } else if (br.bit()) {   // DC, one more bit
                    blk[0] |= 1 << b.al;
                }
                return true;
            }
            return b.ah == 0 ? ac_first(br, ac[c.ta], blk, b, eobrun) : ac_refine(br, ac[c.ta], blk, b, eobrun);
        };
        for (size_t my = 0; my < mcuy; ++my)
            for (size_t mx = 0; mx < mcux; ++mx) {
                if (restart && until_restart == 0) {
                    uint8_t const* q = br.p;
                    while (q + 2 < br.end && q[0] == 0xFF && q[1] == 0xFF) ++q;
                    if (q + 1 >= br.end || q[0] != 0xFF || q[1] != 0xD0 + next_rst) return 0;
                    br.p = q + 2;
                    br.reset();
                    next_rst = (next_rst + 1) & 7;
                    until_restart = restart;
                    eobrun = 0;
                    for (int i = 0; i < ns; ++i) comp[index[i]].pred = 0;
                }
                for (int i = 0; i < ns; ++i) {
                    Component& c = comp[index[i]];
                    bool const luma = index[i] == 0;
                    if (ns == 1) {
                        if (!block(c, luma, mx, my)) return 0;
                    } else {
                        for (int v = 0; v < c.v; ++v)
                            for (int h = 0; h < c.h; ++h)
                                if (!block(c, luma, mx * (size_t)c.h + (size_t)h, my * (size_t)c.v + (size_t)v)) return 0;
                    }
                }
                if (br.overrun) return 0;
                if (restart) --until_restart;
            }
        uint8_t const* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF)) ++q;
        return (size_t)(q - f.data());
    }

    // all scans are in: dequantise and transform every luma block
    void finish_progressive() {
        size_t const bw = plane_w / 8, bh = plane_h / 8;
        for (size_t by = 0; by < bh; ++by)
            for (size_t bx = 0; bx < bw; ++bx) {
                int32_t const* c = &coefs[(by * bw + bx) * 64];
                int32_t dq[64];
                for (int i = 0; i < 64; ++i) {
                    int64_t const v = (int64_t)c[i] * luma_quant[i];
                    dq[i] = (int32_t)(v < -(1 << 30) ? -(1 << 30) : v > (1 << 30) ? (1 << 30) : v);
                }
                idct_block(dq, &plane[by * 8 * plane_w + bx * 8], plane_w);
            }
    }

    cv::Mat run() {
        if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) return {};
        bool luma_done = false;
        while (pos + 4 <= f.size()) {
            if (f[pos] != 0xFF) return {};
            int const m = f[pos + 1];
            if (m == 0xFF) { ++pos; continue; }   // fill byte
            if (m == 0xD9) break;                 // end of image
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { pos += 2; continue; }
            size_t const len = (size_t)((f[pos + 2] << 8) | f[pos + 3]);
            if (len < 2 || pos + 2 + len > f.size()) return {};
            size_t const p = pos + 4, e = pos + 2 + len;
            if (m == 0xDB) {
                if (!read_dqt(p, e)) return {};
            } else if (m == 0xC4) {
                if (!read_dht(p, e)) return {};
            } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                progressive = m == 0xC2;
                if (!read_sof(p, e)) return {};
            } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
                return {};   // lossless, differential or arithmetic coding
            } else if (m == 0xDD) {
                if (len != 4) return {};
                restart = (f[p] << 8) | f[p + 1];
            } else if (m == 0xEE) {   // Adobe: transform 0 with three components means RGB
                if (len >= 14 && std::memcmp(&f[p], "Adobe", 5) == 0 && f[p + 11] == 0) adobe_rgb = true;
                if (have_frame && ncomp == 3 && adobe_rgb) return {};
            } else if (m == 0xDA) {
                if (!have_frame || e - p < 1) return {};
                int const ns = f[p];
                if (ns < 1 || ns > ncomp || e - p < (size_t)1 + 2 * (size_t)ns + 3) return {};
                int index[4];
                size_t const s = p + 1 + 2 * (size_t)ns;
                Band const band{f[s], f[s + 1], f[s + 2] >> 4, f[s + 2] & 15};
                if (progressive) {
                    // a DC scan covers coefficient 0 alone and may interleave components; an AC band
                    // belongs to one component
                    if (band.ss > band.se || band.se > 63 || band.ah > 13 || band.al > 13 || (band.ss == 0 && band.se != 0) ||
                        (band.ss > 0 && ns != 1))
                        return {};
                } else if (band.ss != 0 || band.se != 63 || band.ah != 0 || band.al != 0) {
                    return {};   // sequential: full spectrum, no approximation
                }
                bool has_luma = false;
                for (int i = 0; i < ns; ++i) {
                    int const id = f[p + 1 + 2 * i], t = f[p + 2 + 2 * i];
                    int k = 0;
                    while (k < ncomp && comp[k].id != id) ++k;
                    if (k == ncomp) return {};
                    for (int j = 0; j < i; ++j)
                        if (index[j] == k) return {};   // a component listed twice
                    comp[k].td = t >> 4;
                    comp[k].ta = t & 15;
                    // tables a scan of this kind reads: sequential both; progressive DC first pass the DC
                    // table, AC passes the AC table, a DC refinement none
                    bool const need_dc = !progressive || (band.ss == 0 && band.ah == 0);
                    bool const need_ac = !progressive || band.ss > 0;
                    if (comp[k].td > 3 || comp[k].ta > 3 || (need_dc && !dc[comp[k].td].present) ||
                        (need_ac && !ac[comp[k].ta].present) || !have_quant[comp[k].tq])
                        return {};
                    index[i] = k;
                    if (k == 0) has_luma = true;
                }
                if (has_luma) luma_done = true;
                size_t next;
                if (!progressive) {
                    next = decode_scan(e, ns, index);
                } else if (!has_luma) {
                    next = skip_scan(e);   // chroma only: nothing of it is kept
                } else {
                    if (!luma_quant_latched) {
                        std::memcpy(luma_quant, quant[comp[0].tq], sizeof(luma_quant));
                        luma_quant_latched = true;
                    }
                    next = decode_scan_progressive(e, ns, index, band);
                }
                if (next == 0) return {};
                pos = next;
                continue;
            }
            pos = e;
        }
        if (!have_frame || !luma_done) return {};
        if (progressive) finish_progressive();
        cv::Mat img(height, width, CV_8UC1);
        for (int y = 0; y < height; ++y) std::memcpy(img.ptr<uint8_t>(y), &plane[(size_t)y * plane_w], (size_t)width);
        return img;
    }
};

}  // namespace

auto decode_jpeg_gray(std::vector<uint8_t> const& file) -> cv::Mat {
    try {
        Decoder d(file);
        return d.run();
    } catch (std::bad_alloc const&) {
        return {};
    }
}

}  // namespace visual_odometry::detail
