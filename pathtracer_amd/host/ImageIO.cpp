// PNG in and out for the headless host, with no external library: the reference loads its textures (head/*.png, named by head.mtl)
// through FW's importer and writes "pt-%03dppp.png" per pass with FW's exportLodePngImage (reference src/PathTraceRenderer.cpp:900-902);
// neither is in the reference's tree, so this is the build's own reader/writer of the file format (RFC 2083 / RFC 1950 / RFC 1951).
// Reader: 8-bit grey, grey+alpha, RGB, RGBA and palette images, non-interlaced; every chunk's CRC and the zlib Adler-32 are checked.
// Writer: 8-bit RGB, filter 0, stored (uncompressed) deflate blocks -- valid PNG that any viewer opens; size is not a concern here.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

#include "3d/Mesh.hpp"

namespace FW {

namespace {

bool fail(std::string* err, const std::string& msg) { if (err) *err = msg; return false; }

uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

uint32_t adler32(const uint8_t* p, size_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;  // largest run that cannot overflow 32 bits before the modulo
        n -= k;
        while (k--) { a += *p++; b += a; }
        a %= 65521u; b %= 65521u;
    }
    return (b << 16) | a;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

// ---- inflate (RFC 1951) ----
struct Bits {
    const uint8_t* p;
    size_t n, pos = 0;
    uint64_t buf = 0;
    int cnt = 0;
    bool bad = false;
    Bits(const uint8_t* data, size_t len) : p(data), n(len) {}
    void fill() { while (cnt <= 56 && pos < n) { buf |= (uint64_t)p[pos++] << cnt; cnt += 8; } }
    uint32_t peek(int k) { if (cnt < k) fill(); return (uint32_t)(buf & ((1ull << k) - 1)); }
    void drop(int k) { if (k > cnt) { bad = true; cnt = 0; buf = 0; return; } buf >>= k; cnt -= k; }
    uint32_t get(int k) { if (k == 0) return 0; uint32_t v = peek(k); drop(k); return v; }
    void align() { drop(cnt & 7); }
};

// canonical Huffman code decoded through a (first code, first index, count) table per length
struct Huffman {
    uint16_t count[16] = {0};
    uint16_t symbol[288] = {0};
    bool build(const uint8_t* lengths, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; ++len) { left <<= 1; left -= count[len]; if (left < 0) return false; }  // over-subscribed
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = offs[len] + count[len];
        for (int i = 0; i < n; ++i) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(Bits& b) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            code |= (int)b.get(1);
            if (b.bad) return -1;
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect, std::string* err) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Bits b(src, n);
    out.clear();
    out.reserve(expect);
    for (;;) {
        const uint32_t last = b.get(1), type = b.get(2);
        if (b.bad) return fail(err, "deflate stream ends early");
        if (type == 0) {
            b.align();
            const uint32_t len = b.get(16), nlen = b.get(16);
            if (b.bad || (len ^ 0xffffu) != nlen) return fail(err, "bad stored block");
            for (uint32_t i = 0; i < len; ++i) { out.push_back((uint8_t)b.get(8)); }
            if (b.bad) return fail(err, "deflate stream ends early");
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lengths[i] = 8;
                for (; i < 256; ++i) lengths[i] = 9;
                for (; i < 280; ++i) lengths[i] = 7;
                for (; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
                if (nlen > 286 || ndist > 30) return fail(err, "bad dynamic block header");
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)b.get(3);
                Huffman lencode;
                if (!lencode.build(cl, 19)) return fail(err, "bad code-length code");
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = lencode.decode(b);
                    if (sym < 0) return fail(err, "bad code-length symbol");
                    if (sym < 16) lengths[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (i == 0) return fail(err, "repeat without a previous length"); val = lengths[i - 1]; rep = 3 + (int)b.get(2); }
                        else if (sym == 17) rep = 3 + (int)b.get(3);
                        else rep = 11 + (int)b.get(7);
                        if (i + rep > nlen + ndist) return fail(err, "code lengths overrun");
                        while (rep--) lengths[i++] = (uint8_t)val;
                    }
                }
                if (lengths[256] == 0) return fail(err, "no end-of-block code");
                if (!lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) return fail(err, "over-subscribed Huffman code");
            }
            for (;;) {
                int sym = lit.decode(b);
                if (sym < 0 || b.bad) return fail(err, "bad literal/length symbol");
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return fail(err, "bad length symbol");
                    const int len = lbase[sym] + (int)b.get(lext[sym]);
                    const int ds = dist.decode(b);
                    if (ds < 0 || ds >= 30) return fail(err, "bad distance symbol");
                    const size_t d = dbase[ds] + b.get(dext[ds]);
                    if (b.bad || d > out.size()) return fail(err, "distance beyond the start of the output");
                    const size_t from = out.size() - d;
                    for (int k = 0; k < len; ++k) out.push_back(out[from + k]);
                }
            }
        } else
            return fail(err, "reserved block type");
        if (last) break;
    }
    return true;
}

}  // namespace

std::shared_ptr<Image> importPNG(const std::string& path, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fail(err, "cannot open " + path); return nullptr; }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) { fail(err, path + ": not a PNG file"); return nullptr; }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool end = false;
    for (size_t at = 8; !end;) {
        if (at + 12 > file.size()) { fail(err, path + ": truncated chunk"); return nullptr; }
        const uint32_t len = be32(&file[at]);
        if ((size_t)len > file.size() - at - 12) { fail(err, path + ": chunk length beyond the file"); return nullptr; }
        const uint8_t* type = &file[at + 4];
        const uint8_t* data = &file[at + 8];
        if (crc32(type, (size_t)len + 4) != be32(data + len)) { fail(err, path + ": chunk CRC mismatch"); return nullptr; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) { fail(err, path + ": bad IHDR"); return nullptr; }
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) { fail(err, path + ": unknown compression/filter method"); return nullptr; }
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        at += (size_t)len + 12;
    }
    int channels = 0;
    switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; }
    if (w == 0 || h == 0 || w > 32768 || h > 32768 || channels == 0) { fail(err, path + ": bad image header"); return nullptr; }
    if (depth != 8 || interlace != 0) { fail(err, path + ": only 8-bit non-interlaced PNG images are read"); return nullptr; }
    if (ctype == 3 && (plte.empty() || plte.size() % 3)) { fail(err, path + ": palette image without a palette"); return nullptr; }
    if (idat.size() < 6 || (idat[0] & 0x0f) != 8 || ((idat[0] << 8) | idat[1]) % 31 != 0 || (idat[1] & 0x20)) { fail(err, path + ": bad zlib header"); return nullptr; }
    const size_t stride = (size_t)w * channels, expect = (stride + 1) * h;
    std::vector<uint8_t> raw;
    std::string why;
    if (!inflate(idat.data() + 2, idat.size() - 6, raw, expect, &why)) { fail(err, path + ": " + why); return nullptr; }
    if (raw.size() != expect) { fail(err, path + ": image data has the wrong size"); return nullptr; }
    if (adler32(raw.data(), raw.size()) != be32(&idat[idat.size() - 4])) { fail(err, path + ": Adler-32 mismatch"); return nullptr; }
    // undo the per-row filters in place (RFC 2083 section 6)
    const int bpp = channels;
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t* row = &raw[(stride + 1) * y + 1];
        const uint8_t* up = y ? &raw[(stride + 1) * (y - 1) + 1] : zero.data();
        const int ft = row[-1];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)bpp ? row[x - bpp] : 0, b = up[x], c = x >= (size_t)bpp ? up[x - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: fail(err, path + ": unknown row filter"); return nullptr;
            }
            row[x] = (uint8_t)(row[x] + pred);
        }
    }
    auto img = std::make_shared<Image>(Vec2i((int)w, (int)h), ImageFormat::R8_G8_B8_A8);
    uint8_t* dst = (uint8_t*)img->getMutablePtr();
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = &raw[(stride + 1) * y + 1];
        for (uint32_t x = 0; x < w; ++x, dst += 4) {
            const uint8_t* p = row + (size_t)x * channels;
            switch (ctype) {
                case 0: dst[0] = dst[1] = dst[2] = p[0]; dst[3] = 255; break;
                case 2: dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2]; dst[3] = 255; break;
                case 3: {
                    const size_t k = p[0];
                    if (k * 3 + 2 >= plte.size()) { fail(err, path + ": palette index out of range"); return nullptr; }
                    dst[0] = plte[k * 3]; dst[1] = plte[k * 3 + 1]; dst[2] = plte[k * 3 + 2]; dst[3] = k < trns.size() ? trns[k] : 255;
                    break;
                }
                case 4: dst[0] = dst[1] = dst[2] = p[0]; dst[3] = p[1]; break;
                default: dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2]; dst[3] = p[3]; break;
            }
        }
    }
    return img;
}

bool exportPNG(const std::string& path, const Image& img, std::string* err) {
    const Vec2i s = img.getSize();
    if (s.x <= 0 || s.y <= 0) return fail(err, "exportPNG: empty image");
    const size_t stride = (size_t)s.x * 3;
    std::vector<uint8_t> raw((stride + 1) * s.y);
    for (int y = 0; y < s.y; ++y) {
        uint8_t* row = &raw[(stride + 1) * y];
        *row++ = 0;  // filter type None
        for (int x = 0; x < s.x; ++x) {
            const Vec4f c = clamp(img.getVec4f(Vec2i(x, y)), 0.0f, 1.0f);
            *row++ = (uint8_t)(c.x * 255.0f + 0.5f); *row++ = (uint8_t)(c.y * 255.0f + 0.5f); *row++ = (uint8_t)(c.z * 255.0f + 0.5f);
        }
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    for (size_t at = 0; at < raw.size();) {
        const size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
        z.push_back(at + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored), byte aligned
        z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        at += n;
    }
    put_be32(z, adler32(raw.data(), raw.size()));
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        put_be32(out, (uint32_t)data.size());
        const size_t at = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        put_be32(out, crc32(&out[at], data.size() + 4));
    };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)s.x); put_be32(ihdr, (uint32_t)s.y);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) return fail(err, "cannot create " + path);
    f.write((const char*)out.data(), (std::streamsize)out.size());
    return (bool)f || fail(err, "short write to " + path);
}

std::shared_ptr<Image> importImage(const std::string& path, std::string* err) {
    const size_t dot = path.find_last_of('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    if (ext == ".png" || ext == ".PNG") return importPNG(path, err);
    if (ext == ".ppm" || ext == ".PPM") return importPPM(path, err);
    fail(err, "unsupported image format '" + ext + "'");
    return nullptr;
}

}  // namespace FW
