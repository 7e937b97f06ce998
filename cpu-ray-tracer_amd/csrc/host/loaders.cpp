// loaders.cpp — OBJ / scene-XML / texture readers of the host front (see loaders.h).
#include "loaders.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace crt {

namespace {

[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error(msg); }

std::string readFile(const std::string& path, bool binary)
{
    std::ifstream f(path.c_str(), binary ? std::ios::binary : std::ios::in);
    if (!f) fail("File not found: " + path);
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}

// ------------------------------------------------------------------------------------------------
// OBJ
// ------------------------------------------------------------------------------------------------
inline bool isDigit(char c) { return c >= '0' && c <= '9'; }

// decimal -> double with the digit-accumulation scheme of tinyobjloader (lib/tiny_obj_loader.h:891-1021): the value
// is then narrowed to float, so meshes get the same vertex bits as the reference's loader
bool parseNumber(const char* s, const char* end, double* out)
{
    if (s >= end) return false;
    double mant = 0.0; int exponent = 0; char sign = '+', esign = '+';
    const char* c = s; int read = 0; bool leadingDot = false;
    if (*c == '+' || *c == '-') { sign = *c; c++; if (c != end && *c == '.') leadingDot = true; }
    else if (isDigit(*c)) {}
    else if (*c == '.') leadingDot = true;
    else return false;
    bool more = (c != end);
    if (!leadingDot) {
        while (more && isDigit(*c)) { mant *= 10; mant += (int)(*c - '0'); c++; read++; more = (c != end); }
        if (read == 0) return false;
    }
    if (more) {
        bool expPart = false;
        if (*c == '.') {
            c++; read = 1; more = (c != end);
            static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            while (more && isDigit(*c)) {
                mant += (int)(*c - '0') * (read < 8 ? lut[read] : std::pow(10.0, -read));
                read++; c++; more = (c != end);
            }
            expPart = more && (*c == 'e' || *c == 'E');
        } else if (*c == 'e' || *c == 'E') expPart = true;
        if (expPart) {
            c++; more = (c != end);
            if (more && (*c == '+' || *c == '-')) { esign = *c; c++; }
            else if (more && isDigit(*c)) {}
            else return false;
            read = 0; more = (c != end);
            while (more && isDigit(*c)) {
                if (exponent > 2147483647 / 10) return false;
                exponent = exponent * 10 + (int)(*c - '0'); c++; read++; more = (c != end);
            }
            exponent *= (esign == '+' ? 1 : -1);
            if (read == 0) return false;
        }
    }
    *out = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mant * std::pow(5.0, exponent), exponent) : mant);
    return true;
}

float nextReal(const char*& p, double dflt = 0.0)
{
    p += strspn(p, " \t");
    const char* end = p + strcspn(p, " \t\r");
    double v = dflt;
    parseNumber(p, end, &v);
    p = end;
    return (float)v;
}

struct Corner { int v = -1, vt = -1, vn = -1; };

bool fixIndex(int idx, int n, int* out)   // lib/tiny_obj_loader.h fixIndex: 1-based, negative = relative, 0 invalid
{
    if (idx > 0) { *out = idx - 1; return true; }
    if (idx == 0) return false;
    *out = n + idx; return *out >= 0;
}

bool parseCorner(const char*& p, int nv, int nvn, int nvt, Corner* c)
{
    Corner r;
    if (!fixIndex(atoi(p), nv, &r.v)) return false;
    p += strcspn(p, "/ \t\r");
    if (*p != '/') { *c = r; return true; }
    p++;
    if (*p == '/') {                       // i//k
        p++;
        if (!fixIndex(atoi(p), nvn, &r.vn)) return false;
        p += strcspn(p, "/ \t\r");
        *c = r; return true;
    }
    if (!fixIndex(atoi(p), nvt, &r.vt)) return false;      // i/j or i/j/k
    p += strcspn(p, "/ \t\r");
    if (*p != '/') { *c = r; return true; }
    p++;
    if (!fixIndex(atoi(p), nvn, &r.vn)) return false;
    p += strcspn(p, "/ \t\r");
    *c = r; return true;
}

// even-odd point-in-triangle test on the projected coordinates (lib/tiny_obj_loader.h:1413-1423)
bool insideTri(const float* vx, const float* vy, float tx, float ty)
{
    bool c = false;
    for (int i = 0, j = 2; i < 3; j = i++)
        if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
    return c;
}

} // namespace

MeshCorners LoadObj(const std::string& path)
{
    const std::string text = readFile(path, false);
    std::vector<float> V, VN, VT;
    std::vector<Corner> out;           // three per triangle
    size_t pos = 0; int lineNo = 0;
    auto emit = [&](const Corner& a, const Corner& b, const Corner& c) { out.push_back(a); out.push_back(b); out.push_back(c); };
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos);
        pos = eol + 1; lineNo++;
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        const char* p = line.c_str();
        p += strspn(p, " \t");
        if (*p == '\0' || *p == '#') continue;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) { p += 2; float x = nextReal(p), y = nextReal(p), z = nextReal(p); V.push_back(x); V.push_back(y); V.push_back(z); continue; }
        if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) { p += 3; float x = nextReal(p), y = nextReal(p), z = nextReal(p); VN.push_back(x); VN.push_back(y); VN.push_back(z); continue; }
        if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) { p += 3; float x = nextReal(p), y = nextReal(p); VT.push_back(x); VT.push_back(y); continue; }
        if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2; p += strspn(p, " \t");
            std::vector<Corner> face;
            while (*p != '\0' && *p != '\r' && *p != '\n') {
                Corner c;
                if (!parseCorner(p, (int)(V.size() / 3), (int)(VN.size() / 3), (int)(VT.size() / 2), &c))
                    fail(path + ":" + std::to_string(lineNo) + ": failed to parse `f' line (invalid vertex index)");
                face.push_back(c);
                p += strspn(p, " \t\r");
            }
            const size_t n = face.size();
            if (n < 3) continue;                                    // degenerate face, dropped as tinyobj does
            for (const Corner& c : face)
                if (c.v < 0 || (size_t)c.v * 3 + 2 >= V.size() || (c.vn >= 0 && (size_t)c.vn * 3 + 2 >= VN.size()) || (c.vt >= 0 && (size_t)c.vt * 2 + 1 >= VT.size()))
                    fail(path + ":" + std::to_string(lineNo) + ": face references a vertex / normal / texcoord that does not exist");
            if (n == 3) { emit(face[0], face[1], face[2]); continue; }
            if (n == 4) {                                           // shorter diagonal, strict <  (tiny_obj_loader.h:1488-1588)
                const float* p0 = &V[3 * face[0].v]; const float* p1 = &V[3 * face[1].v]; const float* p2 = &V[3 * face[2].v]; const float* p3 = &V[3 * face[3].v];
                const float ax = p2[0] - p0[0], ay = p2[1] - p0[1], az = p2[2] - p0[2];
                const float bx = p3[0] - p1[0], by = p3[1] - p1[1], bz = p3[2] - p1[2];
                const float d02 = ax * ax + ay * ay + az * az, d13 = bx * bx + by * by + bz * bz;
                if (d02 < d13) { emit(face[0], face[1], face[2]); emit(face[0], face[2], face[3]); }
                else { emit(face[0], face[1], face[3]); emit(face[1], face[2], face[3]); }
                continue;
            }
            // n > 4: ear clipping in the plane of the two dominant axes (tiny_obj_loader.h:1714-1935)
            size_t axes[2] = {1, 2};
            for (size_t k = 0; k < n; ++k) {
                const float* a = &V[3 * face[(k + 0) % n].v]; const float* b = &V[3 * face[(k + 1) % n].v]; const float* c = &V[3 * face[(k + 2) % n].v];
                const float e0x = b[0] - a[0], e0y = b[1] - a[1], e0z = b[2] - a[2];
                const float e1x = c[0] - b[0], e1y = c[1] - b[1], e1z = c[2] - b[2];
                const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
                const float eps = std::numeric_limits<float>::epsilon();
                if (cx > eps || cy > eps || cz > eps) {
                    if (!(cx > cy && cx > cz)) { axes[0] = 0; if (cz > cx && cz > cy) axes[1] = 1; }
                    break;
                }
            }
            std::vector<Corner> rest = face;
            size_t guess = 0, budget = face.size(), prevSize = rest.size();
            while (rest.size() > 3 && budget > 0) {
                const size_t m = rest.size();
                if (guess >= m) guess -= m;
                if (prevSize != m) { prevSize = m; budget = m; } else budget--;
                Corner ind[3]; float vx[3], vy[3];
                for (size_t k = 0; k < 3; k++) { ind[k] = rest[(guess + k) % m]; vx[k] = V[3 * ind[k].v + axes[0]]; vy[k] = V[3 * ind[k].v + axes[1]]; }
                const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
                const float crs = e0x * e1y - e0y * e1x;
                const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
                if (crs * area < 0.0f) { guess += 1; continue; }
                bool overlap = false;
                for (size_t other = 3; other < m; ++other) {
                    const size_t oi = (guess + other) % m;
                    const float tx = V[3 * rest[oi].v + axes[0]], ty = V[3 * rest[oi].v + axes[1]];
                    if (insideTri(vx, vy, tx, ty)) { overlap = true; break; }
                }
                if (overlap) { guess += 1; continue; }
                emit(ind[0], ind[1], ind[2]);
                rest.erase(rest.begin() + (long)((guess + 1) % m));
            }
            if (rest.size() == 3) emit(rest[0], rest[1], rest[2]);
            continue;
        }
        // o, g, s, usemtl, mtllib, l, p, ...: no effect on the triangle list the path tracer consumes
    }
    MeshCorners m;
    m.pos.resize(out.size() * 3, 0.0f); m.nrm.resize(out.size() * 3, 0.0f); m.uv.resize(out.size() * 2, 0.0f);
    for (size_t i = 0; i < out.size(); i++) {
        const Corner& c = out[i];
        memcpy(&m.pos[3 * i], &V[3 * c.v], 12);
        if (c.vn >= 0) memcpy(&m.nrm[3 * i], &VN[3 * c.vn], 12);
        if (c.vt >= 0) memcpy(&m.uv[2 * i], &VT[2 * c.vt], 8);
    }
    if (out.empty()) fail(path + ": no faces");
    return m;
}

// ------------------------------------------------------------------------------------------------
// images
// ------------------------------------------------------------------------------------------------
namespace {

struct Raw { int w = 0, h = 0, n = 0; std::vector<uint8_t> px; };   // n interleaved 8-bit channels, top row first

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

Raw decodePng(const std::string& d, const std::string& path)
{
    const uint8_t* p = (const uint8_t*)d.data(); const size_t size = d.size();
    size_t o = 8; uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0; bool haveHdr = false;
    std::vector<uint8_t> idat, plte, trns;
    while (o + 8 <= size) {
        const uint32_t len = be32(p + o); const std::string typ((const char*)p + o + 4, 4);
        if (o + 12 + (size_t)len > size) fail(path + ": truncated PNG chunk");
        const uint8_t* body = p + o + 8;
        if (typ == "IHDR") { if (len != 13) fail(path + ": PNG IHDR chunk must hold 13 bytes"); w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; haveHdr = true; }
        else if (typ == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (typ == "PLTE") plte.assign(body, body + len);
        else if (typ == "tRNS") trns.assign(body, body + len);
        else if (typ == "IEND") break;
        o += 12 + (size_t)len;
    }
    if (!haveHdr || w == 0 || h == 0) fail(path + ": bad PNG header");
    if (w > (1u << 15) || h > (1u << 15)) fail(path + ": PNG larger than 32768 x 32768 is refused");     // stb_image's STBI_MAX_DIMENSIONS is 1 << 24; the texel pool addresses 2^32
    if (interlace) fail(path + ": interlaced PNG is not supported by this loader");
    int ch; switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: fail(path + ": bad PNG colour type"); }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) fail(path + ": unsupported PNG bit depth");
    const size_t bpp = (size_t)ch * depth;                         // bits per pixel
    const size_t stride = ((size_t)w * bpp + 7) / 8, fb = (bpp + 7) / 8;  // filter byte distance
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf dl = (uLongf)raw.size();
    if (uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size()) != Z_OK || dl != raw.size()) fail(path + ": PNG inflate failed");
    std::vector<uint8_t> img(stride * (size_t)h);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t ft = raw[(stride + 1) * y]; const uint8_t* in = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &img[stride * y]; const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= fb ? cur[x - fb] : 0, b = up ? up[x] : 0, c = (up && x >= fb) ? up[x - fb] : 0;
            int pr = 0;
            switch (ft) {
                case 0: pr = 0; break; case 1: pr = a; break; case 2: pr = b; break; case 3: pr = (a + b) >> 1; break;
                case 4: { const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: fail(path + ": bad PNG filter");
            }
            cur[x] = (uint8_t)(in[x] + pr);
        }
    }
    // to 8-bit samples (16-bit keeps the high byte; sub-byte grey is scaled, palette indices stay) — as stb_image does
    std::vector<uint8_t> s8((size_t)w * h * ch);
    for (uint32_t y = 0; y < h; y++) for (size_t i = 0; i < (size_t)w * ch; i++) {
        uint8_t v;
        const uint8_t* row = &img[stride * y];
        if (depth == 8) v = row[i];
        else if (depth == 16) v = row[2 * i];
        else { const int per = 8 / depth; const int sh = (per - 1 - (int)(i % per)) * depth; v = (uint8_t)((row[i / per] >> sh) & ((1 << depth) - 1)); if (ctype == 0) v = (uint8_t)(v * (255 / ((1 << depth) - 1))); }
        s8[(size_t)y * w * ch + i] = v;
    }
    Raw r; r.w = (int)w; r.h = (int)h;
    if (ctype == 3) {
        r.n = trns.empty() ? 3 : 4; r.px.resize((size_t)w * h * r.n);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            const size_t k = s8[i];
            if (k * 3 + 2 >= plte.size()) fail(path + ": palette index out of range");
            r.px[i * r.n] = plte[k * 3]; r.px[i * r.n + 1] = plte[k * 3 + 1]; r.px[i * r.n + 2] = plte[k * 3 + 2];
            if (r.n == 4) r.px[i * 4 + 3] = k < trns.size() ? trns[k] : 255;
        }
    } else if (!trns.empty() && (ctype == 0 || ctype == 2)) {   // colour-key transparency adds an alpha channel
        r.n = ch + 1; r.px.resize((size_t)w * h * r.n);
        for (size_t i = 0; i < (size_t)w * h; i++) { for (int k = 0; k < ch; k++) r.px[i * r.n + k] = s8[i * ch + k]; r.px[i * r.n + ch] = 255; }
    } else { r.n = ch; r.px.swap(s8); }
    return r;
}

Raw decodeTga(const std::string& d, const std::string& path)
{
    const uint8_t* p = (const uint8_t*)d.data();
    if (d.size() < 18) fail(path + ": truncated TGA");
    const int idlen = p[0], cmap = p[1], typ = p[2], w = p[12] | (p[13] << 8), h = p[14] | (p[15] << 8), bpp = p[16], desc = p[17];
    const bool rle = typ >= 8; const int base = typ & 7;
    if (cmap != 0 || !(base == 2 || base == 3) || !((base == 2 && (bpp == 24 || bpp == 32)) || (base == 3 && bpp == 8)))
        fail(path + ": unsupported TGA variant (only 8-bit grey and 24/32-bit true colour, raw or RLE)");
    const int n = bpp / 8; size_t o = 18 + (size_t)idlen;
    std::vector<uint8_t> px((size_t)w * h * n);
    if (!rle) { if (o + px.size() > d.size()) fail(path + ": truncated TGA"); memcpy(px.data(), p + o, px.size()); }
    else {
        size_t i = 0;
        while (i < px.size()) {
            if (o >= d.size()) fail(path + ": truncated TGA");
            const int c = p[o++]; const size_t cnt = (size_t)(c & 127) + 1;
            if (c & 128) { if (o + n > d.size()) fail(path + ": truncated TGA"); for (size_t k = 0; k < cnt && i < px.size(); k++, i += n) memcpy(&px[i], p + o, n); o += n; }
            else { const size_t bytes = cnt * n; if (o + bytes > d.size() || i + bytes > px.size()) fail(path + ": truncated TGA"); memcpy(&px[i], p + o, bytes); o += bytes; i += bytes; }
        }
    }
    Raw r; r.w = w; r.h = h; r.n = n; r.px.resize(px.size());
    const bool topDown = (desc & 0x20) != 0;
    for (int y = 0; y < h; y++) {
        const uint8_t* src = &px[(size_t)(topDown ? y : h - 1 - y) * w * n]; uint8_t* dst = &r.px[(size_t)y * w * n];
        for (int x = 0; x < w; x++) {
            if (n >= 3) { dst[x * n] = src[x * n + 2]; dst[x * n + 1] = src[x * n + 1]; dst[x * n + 2] = src[x * n]; if (n == 4) dst[x * n + 3] = src[x * n + 3]; }
            else dst[x] = src[x];
        }
    }
    return r;
}

// ---- JPEG ---------------------------------------------------------------------------------------------------------------
// Huffman JPEG, sequential (SOF0 / SOF1) and progressive (SOF2: spectral selection + successive approximation, ITU T.81 annex G),
// 8 bit, 1 or 3 components, any sampling factors, restart intervals.  A texture's texels must
// be the ones the reference gets from stbi_load (template/texture.h:18), and JPEG leaves the arithmetic of the decoder open, so this
// follows the arithmetic lib/stb_image.h uses, stage by stage:
//   coefficients  (short)(value * quantiser); progressive: 16-bit coefficients accumulated over the scans, then multiplied by the
//                 quantiser in 16-bit arithmetic                                         stbi__jpeg_decode_block, :2182-2233; :2236-2380, :3038-3062
//   inverse DCT   the 12-bit fixed-point "islow" butterfly, column pass rounded to 2 extra bits (+512 >> 10), row pass
//                 +65536 + (128 << 17) >> 17, clamped                                   stbi__idct_block, :2393-2495
//   upsampling    1x1 copy; 2x1 / 1x2 / 2x2 triangle filters (3:1 weights), anything else nearest; the row pairing of
//                 load_jpeg_image's line0 / line1 / ystep walk                          :3411-3604, :3845-3893
//   colour        YCbCr -> RGB in 20-bit fixed point with the 0xffff0000 mask on the Cb term of G; components tagged 'R','G','B', or an
//                 Adobe APP14 transform 0 without a JFIF header, are taken as RGB         stbi__YCbCr_to_RGB_row :3606-3631, :3825
// Arithmetic-coded, lossless, 12-bit and 4-component files are rejected with a message.
struct JpegComp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, w2 = 0, h2 = 0, x = 0, y = 0, pred = 0; std::vector<uint8_t> plane;
                  std::vector<short> coef; };   // coef: progressive files only, 64 per 8x8 block, (w2 / 8) blocks per row
struct JpegHuff { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool present = false; };

struct JpegBits {
    const uint8_t* p; size_t size, pos; uint32_t acc = 0; int cnt = 0; int marker = 0;      // marker: the marker that ended the entropy-coded segment (0 = none yet)
    void fill() {
        while (cnt <= 24) {
            uint32_t b = 0;
            if (!marker && pos < size) {
                b = p[pos++];
                if (b == 0xFF) {
                    int c = pos < size ? p[pos++] : 0xD9;
                    while (c == 0xFF) c = pos < size ? p[pos++] : 0xD9;                    // fill bytes
                    if (c != 0) { marker = c; b = 0; }                                      // a real marker: the segment ends, zeros follow
                }
            }
            acc |= b << (24 - cnt); cnt += 8;
        }
    }
    int bit() { if (cnt < 1) fill(); const int b = (int)(acc >> 31); acc <<= 1; cnt--; return b; }
    int receive(int n) { if (n == 0) return 0; if (cnt < n) fill(); const int v = (int)(acc >> (32 - n)); acc <<= n; cnt -= n; return v; }
    void reset() { acc = 0; cnt = 0; marker = 0; }
};

void jpegBuildHuff(JpegHuff& h, const std::string& path)
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {                                    // canonical codes, ITU T.81 annex C / F.2.2.3
        h.valptr[l] = k; h.mincode[l] = code;
        code += h.bits[l]; k += h.bits[l];
        h.maxcode[l] = h.bits[l] ? code - 1 : -1;
        if (code > (1 << l)) fail(path + ": bad JPEG Huffman code lengths");
        code <<= 1;
    }
    h.present = true;
}

int jpegDecodeSym(JpegBits& b, const JpegHuff& h, const std::string& path)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | b.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    fail(path + ": corrupt JPEG (bad Huffman code)");
    return 0;
}

inline int jpegExtend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }   // T.81 F.2.2.1 EXTEND

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp255(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// one 8-point pass of the fixed-point inverse DCT: constants are round(c * 4096); returns the even part in x[0..3] and the odd part in t[0..3]
inline void idct8(const int s[8], int x[4], int t[4])
{
    // (int)(c * 4096 + 0.5), truncating toward zero as the C conversion does for the negative ones
    const int C0_5411 = 2217, Cm1_8477 = -7567, C0_7653 = 3135, C1_1758 = 4816, C0_2986 = 1223, C2_0531 = 8410, C3_0727 = 12586,
              C1_5013 = 6149, Cm0_8999 = -3685, Cm2_5629 = -10497, Cm1_9615 = -8034, Cm0_3901 = -1597;
    int p1 = (s[2] + s[6]) * C0_5411;
    const int e2 = p1 + s[6] * Cm1_8477, e3 = p1 + s[2] * C0_7653;
    const int e0 = (s[0] + s[4]) * 4096, e1 = (s[0] - s[4]) * 4096;
    x[0] = e0 + e3; x[3] = e0 - e3; x[1] = e1 + e2; x[2] = e1 - e2;
    int t0 = s[7], t1 = s[5], t2 = s[3], t3 = s[1];
    int p3 = t0 + t2, p4 = t1 + t3; p1 = t0 + t3; int p2 = t1 + t2;
    const int p5 = (p3 + p4) * C1_1758;
    t0 *= C0_2986; t1 *= C2_0531; t2 *= C3_0727; t3 *= C1_5013;
    p1 = p5 + p1 * Cm0_8999; p2 = p5 + p2 * Cm2_5629; p3 *= Cm1_9615; p4 *= Cm0_3901;
    t[3] = t3 + p1 + p4; t[2] = t2 + p2 + p3; t[1] = t1 + p2 + p4; t[0] = t0 + p1 + p3;
}

void jpegIdct(uint8_t* out, int stride, const short d[64])
{
    int v[64];
    for (int c = 0; c < 8; c++) {                                       // columns; a column whose AC terms are all zero is just its DC term * 4
        if (!(d[c + 8] | d[c + 16] | d[c + 24] | d[c + 32] | d[c + 40] | d[c + 48] | d[c + 56])) {
            const int dc = d[c] * 4;
            for (int r = 0; r < 8; r++) v[c + 8 * r] = dc;
            continue;
        }
        const int s[8] = {d[c], d[c + 8], d[c + 16], d[c + 24], d[c + 32], d[c + 40], d[c + 48], d[c + 56]};
        int x[4], t[4]; idct8(s, x, t);
        for (int k = 0; k < 4; k++) x[k] += 512;
        v[c] = (x[0] + t[3]) >> 10; v[c + 56] = (x[0] - t[3]) >> 10;
        v[c + 8] = (x[1] + t[2]) >> 10; v[c + 48] = (x[1] - t[2]) >> 10;
        v[c + 16] = (x[2] + t[1]) >> 10; v[c + 40] = (x[2] - t[1]) >> 10;
        v[c + 24] = (x[3] + t[0]) >> 10; v[c + 32] = (x[3] - t[0]) >> 10;
    }
    for (int r = 0; r < 8; r++, out += stride) {                        // rows: remove 2^17 with rounding, level-shift by +128
        int x[4], t[4]; idct8(v + 8 * r, x, t);
        for (int k = 0; k < 4; k++) x[k] += 65536 + (128 << 17);
        out[0] = clamp255((x[0] + t[3]) >> 17); out[7] = clamp255((x[0] - t[3]) >> 17);
        out[1] = clamp255((x[1] + t[2]) >> 17); out[6] = clamp255((x[1] - t[2]) >> 17);
        out[2] = clamp255((x[2] + t[1]) >> 17); out[5] = clamp255((x[2] - t[1]) >> 17);
        out[3] = clamp255((x[3] + t[0]) >> 17); out[4] = clamp255((x[3] - t[0]) >> 17);
    }
}

// one output row of a component from its low-resolution rows `nr` (nearer) and `fr` (farther); w = low-resolution width
const uint8_t* jpegUpsampleRow(uint8_t* out, const uint8_t* nr, const uint8_t* fr, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) return nr;
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * nr[i] + fr[i] + 2) >> 2); return out; }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = nr[0]; return out; }
        out[0] = nr[0]; out[1] = (uint8_t)((nr[0] * 3 + nr[1] + 2) >> 2);
        int i = 1;
        for (; i < w - 1; i++) { const int n = 3 * nr[i] + 2; out[2 * i] = (uint8_t)((n + nr[i - 1]) >> 2); out[2 * i + 1] = (uint8_t)((n + nr[i + 1]) >> 2); }
        out[2 * i] = (uint8_t)((nr[w - 2] * 3 + nr[w - 1] + 2) >> 2); out[2 * i + 1] = nr[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nr[0] + fr[0] + 2) >> 2); return out; }
        int t1 = 3 * nr[0] + fr[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1; t1 = 3 * nr[i] + fr[i];
            out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++) for (int j = 0; j < hs; j++) out[i * hs + j] = nr[i];       // nearest
    return out;
}

Raw decodeJpeg(const std::string& d, const std::string& path)
{
    const uint8_t* p = (const uint8_t*)d.data(); const size_t size = d.size();
    uint16_t quant[4][64]; bool haveQ[4] = {false, false, false, false};
    JpegHuff hdc[4], hac[4];
    std::vector<JpegComp> comp; int W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, adobe = -1; bool jfif = false, haveFrame = false, decoded = false, progressive = false;
    size_t o = 2;
    auto need = [&](size_t n) { if (o + n > size) fail(path + ": truncated JPEG"); };
    for (;;) {
        need(2);
        if (p[o] != 0xFF) fail(path + ": corrupt JPEG (marker expected)");
        while (o < size && p[o] == 0xFF) o++;                           // fill bytes before a marker
        need(1);
        const int m = p[o++];
        if (m == 0xD9) break;                                           // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;            // standalone markers
        need(2);
        const size_t L = ((size_t)p[o] << 8) | p[o + 1];
        if (L < 2) fail(path + ": corrupt JPEG (segment length)");
        need(L);
        const uint8_t* q = p + o + 2; const size_t n = L - 2;
        if (m == 0xDB) {                                                // DQT
            size_t i = 0;
            while (i < n) {
                const int pq = q[i] >> 4, t = q[i] & 15; i++;
                if (t > 3 || pq > 1 || i + (size_t)64 * (pq + 1) > n) fail(path + ": corrupt JPEG (DQT)");
                for (int k = 0; k < 64; k++) { quant[t][kZigzag[k]] = pq ? (uint16_t)((q[i] << 8) | q[i + 1]) : q[i]; i += pq + 1; }
                haveQ[t] = true;
            }
        } else if (m == 0xC4) {                                         // DHT
            size_t i = 0;
            while (i < n) {
                if (i + 17 > n) fail(path + ": corrupt JPEG (DHT)");
                const int tc = q[i] >> 4, th = q[i] & 15; i++;
                if (tc > 1 || th > 3) fail(path + ": corrupt JPEG (DHT)");
                JpegHuff& h = tc ? hac[th] : hdc[th];
                int total = 0; for (int l = 1; l <= 16; l++) { h.bits[l] = q[i++]; total += h.bits[l]; }
                if (total > 256 || i + (size_t)total > n) fail(path + ": corrupt JPEG (DHT)");
                for (int k = 0; k < total; k++) h.vals[k] = q[i++];
                jpegBuildHuff(h, path);
            }
        } else if (m == 0xDD) { if (n < 2) fail(path + ": corrupt JPEG (DRI)"); restart = (q[0] << 8) | q[1]; }
        else if (m == 0xE0) { if (n >= 5 && memcmp(q, "JFIF\0", 5) == 0) jfif = true; }
        else if (m == 0xEE) { if (n >= 12 && memcmp(q, "Adobe\0", 6) == 0) adobe = q[11]; }
        else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {                 // SOF0 / SOF1 / SOF2 (progressive)
            if (haveFrame) fail(path + ": corrupt JPEG (two frame headers)");
            if (n < 6 || q[0] != 8) fail(path + ": only 8-bit JPEG is supported");
            H = (q[1] << 8) | q[2]; W = (q[3] << 8) | q[4];
            const int nc = q[5];
            if (W == 0 || H == 0) fail(path + ": JPEG with zero size / DNL height is not supported");
            if (nc != 1 && nc != 3) fail(path + ": only 1- and 3-component JPEG is supported (no CMYK / YCCK)");
            if (n < (size_t)6 + 3 * nc) fail(path + ": corrupt JPEG (SOF)");
            comp.resize(nc);
            for (int k = 0; k < nc; k++) {
                comp[k].id = q[6 + 3 * k]; comp[k].h = q[7 + 3 * k] >> 4; comp[k].v = q[7 + 3 * k] & 15; comp[k].tq = q[8 + 3 * k];
                if (comp[k].h < 1 || comp[k].h > 4 || comp[k].v < 1 || comp[k].v > 4 || comp[k].tq > 3) fail(path + ": corrupt JPEG (sampling factors)");
                if (comp[k].h > hmax) hmax = comp[k].h;
                if (comp[k].v > vmax) vmax = comp[k].v;
            }
            for (auto& c : comp) if (hmax % c.h || vmax % c.v) fail(path + ": corrupt JPEG (sampling factors)");
            const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (auto& c : comp) {
                c.x = (W * c.h + hmax - 1) / hmax; c.y = (H * c.v + vmax - 1) / vmax;            // samples that carry image data
                c.w2 = mcux * c.h * 8; c.h2 = mcuy * c.v * 8;                                    // padded to whole MCUs
                c.plane.assign((size_t)c.w2 * c.h2 + 15, 0);
                if (m == 0xC2) c.coef.assign((size_t)(c.w2 / 8) * (c.h2 / 8) * 64, 0);
            }
            progressive = (m == 0xC2);
            haveFrame = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            fail(path + ": lossless / hierarchical / arithmetic-coded JPEG is not supported (Huffman sequential and progressive only)");
        } else if (m == 0xDA) {                                         // SOS + entropy-coded data
            if (!haveFrame) fail(path + ": corrupt JPEG (scan before frame)");
            if (n < 1) fail(path + ": corrupt JPEG (SOS)");
            const int ns = q[0];
            if (ns < 1 || ns > (int)comp.size() || n < (size_t)1 + 2 * ns + 3) fail(path + ": corrupt JPEG (SOS)");
            int order[4];
            for (int k = 0; k < ns; k++) {
                int ci = -1; for (size_t c = 0; c < comp.size(); c++) if (comp[c].id == q[1 + 2 * k]) ci = (int)c;
                if (ci < 0) fail(path + ": corrupt JPEG (scan component)");
                comp[ci].td = q[2 + 2 * k] >> 4; comp[ci].ta = q[2 + 2 * k] & 15;
                if (comp[ci].td > 3 || comp[ci].ta > 3 || !haveQ[comp[ci].tq]) fail(path + ": corrupt JPEG (missing table)");
                order[k] = ci;
            }
            const int Ss = q[1 + 2 * ns], Se = q[2 + 2 * ns], Ah = q[3 + 2 * ns] >> 4, Al = q[3 + 2 * ns] & 15;
            if (progressive) {
                if (Ss > 63 || Se > 63 || Ss > Se || Ah > 13 || Al > 13 || (Ss == 0 && Se != 0) || (Ss != 0 && ns != 1)) fail(path + ": corrupt JPEG (progressive scan parameters)");
            } else if (Ss != 0 || Ah != 0 || Al != 0) fail(path + ": corrupt JPEG (scan parameters)");
            for (int k = 0; k < ns; k++) {
                const JpegComp& c = comp[order[k]];
                if ((!progressive || Ss == 0) && !hdc[c.td].present) fail(path + ": corrupt JPEG (missing DC table)");
                if ((!progressive || Ss != 0) && !hac[c.ta].present) fail(path + ": corrupt JPEG (missing AC table)");
            }
            JpegBits b{p, size, o + L};
            for (auto& c : comp) c.pred = 0;
            int todo = restart ? restart : 0x7fffffff;
            short blk[64];
            auto block = [&](JpegComp& c, int bx, int by) {
                memset(blk, 0, sizeof(blk));
                const int t = jpegDecodeSym(b, hdc[c.td], path);
                if (t > 15) fail(path + ": corrupt JPEG (DC size)");
                c.pred += jpegExtend(b.receive(t), t);
                blk[0] = (short)(c.pred * quant[c.tq][0]);
                for (int k = 1; k < 64;) {
                    const int rs = jpegDecodeSym(b, hac[c.ta], path), r = rs >> 4, sz = rs & 15;
                    if (sz == 0) { if (rs != 0xF0) break; k += 16; continue; }
                    k += r;
                    if (k > 63) fail(path + ": corrupt JPEG (AC run)");
                    const int z = kZigzag[k++];
                    blk[z] = (short)(jpegExtend(b.receive(sz), sz) * quant[c.tq][z]);
                }
                jpegIdct(&c.plane[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, blk);
            };
            int eobrun = 0;
            // one block of a progressive scan (T.81 G.1.2): DC first / refinement, AC first / refinement with end-of-band runs
            auto blockProg = [&](JpegComp& c, int bx, int by) {
                short* d = &c.coef[((size_t)by * (c.w2 / 8) + bx) * 64];
                if (Ss == 0) {
                    if (Ah == 0) {
                        const int t = jpegDecodeSym(b, hdc[c.td], path);
                        if (t > 15) fail(path + ": corrupt JPEG (DC size)");
                        c.pred += jpegExtend(b.receive(t), t);
                        d[0] = (short)(c.pred * (1 << Al));
                    } else if (b.bit()) d[0] += (short)(1 << Al);
                    return;
                }
                if (Ah == 0) {
                    if (eobrun) { eobrun--; return; }
                    int k = Ss;
                    do {
                        const int rs = jpegDecodeSym(b, hac[c.ta], path), r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r < 15) { eobrun = (1 << r); if (r) eobrun += b.receive(r); eobrun--; break; }
                            k += 16;
                        } else {
                            k += r;
                            if (k > 63) fail(path + ": corrupt JPEG (AC run)");
                            d[kZigzag[k++]] = (short)(jpegExtend(b.receive(sz), sz) * (1 << Al));
                        }
                    } while (k <= Se);
                    return;
                }
                const short bit = (short)(1 << Al);
                auto refine = [&](short* p) { if (b.bit() && (*p & bit) == 0) { if (*p > 0) *p += bit; else *p -= bit; } };
                if (eobrun) {
                    eobrun--;
                    for (int k = Ss; k <= Se; k++) { short* p = &d[kZigzag[k]]; if (*p != 0) refine(p); }
                    return;
                }
                int k = Ss;
                do {
                    const int rs = jpegDecodeSym(b, hac[c.ta], path); int r = rs >> 4, sv = rs & 15;
                    if (sv == 0) {
                        if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += b.receive(r); r = 64; }    // end of band: only refinements follow in this block
                    } else {
                        if (sv != 1) fail(path + ": corrupt JPEG (refinement scan)");
                        sv = b.bit() ? bit : -bit;
                    }
                    while (k <= Se) {                                   // skip r zero-history coefficients, refining the non-zero ones passed
                        short* p = &d[kZigzag[k++]];
                        if (*p != 0) refine(p);
                        else { if (r == 0) { *p = (short)sv; break; } r--; }
                    }
                } while (k <= Se);
            };
            auto mcuDone = [&]() -> bool {                              // false: the segment ended without the expected restart marker
                if (--todo > 0) return true;
                if (b.cnt < 24) b.fill();
                if (!(b.marker >= 0xD0 && b.marker <= 0xD7)) return false;
                b.reset(); for (auto& c : comp) c.pred = 0; todo = restart ? restart : 0x7fffffff; eobrun = 0;
                return true;
            };
            auto anyBlock = [&](JpegComp& c, int bx, int by) { if (progressive) blockProg(c, bx, by); else block(c, bx, by); };
            if (ns == 1) {                                              // non-interleaved: the component's own blocks in raster order
                JpegComp& c = comp[order[0]];
                const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3; bool go = true;
                for (int j = 0; j < bh && go; j++) for (int i = 0; i < bw && go; i++) { anyBlock(c, i, j); go = mcuDone(); }
            } else {
                const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax); bool go = true;
                for (int j = 0; j < mcuy && go; j++) for (int i = 0; i < mcux && go; i++) {
                    for (int k = 0; k < ns; k++) { JpegComp& c = comp[order[k]]; for (int y = 0; y < c.v; y++) for (int x = 0; x < c.h; x++) anyBlock(c, i * c.h + x, j * c.v + y); }
                    go = mcuDone();
                }
            }
            decoded = true;
            // continue behind the entropy-coded segment: at the marker that ended it, or search for the next one
            size_t e = b.pos;
            if (b.marker) { o = e - 2; while (o > 0 && !(p[o] == 0xFF && p[o + 1] == (uint8_t)b.marker)) o--; continue; }
            while (e + 1 < size && !(p[e] == 0xFF && p[e + 1] != 0 && p[e + 1] != 0xFF && !(p[e + 1] >= 0xD0 && p[e + 1] <= 0xD7))) e++;
            if (e + 1 >= size) break;
            o = e; continue;
        }
        o += L;
    }
    if (!haveFrame || !decoded) fail(path + ": JPEG without image data");
    if (progressive) {                                                  // all scans are in: dequantise (16-bit products) and transform the blocks that carry image data
        for (auto& c : comp) {
            if (!haveQ[c.tq]) fail(path + ": corrupt JPEG (missing quantisation table)");
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; j++) for (int i = 0; i < bw; i++) {
                short* d = &c.coef[((size_t)j * (c.w2 / 8) + i) * 64];
                for (int k = 0; k < 64; k++) d[k] = (short)(d[k] * quant[c.tq][k]);
                jpegIdct(&c.plane[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, d);
            }
        }
    }
    const int nc = (int)comp.size();
    int rgbIds = 0; if (nc == 3) { const char* tag = "RGB"; for (int k = 0; k < 3; k++) if (comp[k].id == tag[k]) rgbIds++; }
    const bool isRgb = nc == 3 && (rgbIds == 3 || (adobe == 0 && !jfif));
    Raw r; r.w = W; r.h = H; r.n = nc == 3 ? 3 : 1; r.px.resize((size_t)W * H * r.n);
    struct Walk { int hs, vs, ystep, wl, ypos; const uint8_t *l0, *l1; std::vector<uint8_t> buf; };
    std::vector<Walk> wk(nc);
    for (int k = 0; k < nc; k++) {
        Walk& w = wk[k]; w.hs = hmax / comp[k].h; w.vs = vmax / comp[k].v; w.ystep = w.vs >> 1; w.wl = (W + w.hs - 1) / w.hs; w.ypos = 0;
        w.l0 = w.l1 = comp[k].plane.data(); w.buf.resize((size_t)W + 8);
    }
    const uint8_t* row[3] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < H; j++) {
        for (int k = 0; k < nc; k++) {
            Walk& w = wk[k];
            const bool bot = w.ystep >= (w.vs >> 1);
            row[k] = jpegUpsampleRow(w.buf.data(), bot ? w.l1 : w.l0, bot ? w.l0 : w.l1, w.wl, w.hs, w.vs);
            if (++w.ystep >= w.vs) { w.ystep = 0; w.l0 = w.l1; if (++w.ypos < comp[k].y) w.l1 += comp[k].w2; }
        }
        uint8_t* out = &r.px[(size_t)j * W * r.n];
        if (nc == 1) memcpy(out, row[0], (size_t)W);
        else if (isRgb) for (int i = 0; i < W; i++) { out[3 * i] = row[0][i]; out[3 * i + 1] = row[1][i]; out[3 * i + 2] = row[2][i]; }
        else for (int i = 0; i < W; i++) {
            const int yf = (row[0][i] << 20) + (1 << 19), cb = row[1][i] - 128, cr = row[2][i] - 128;
            // fixed-point factors: round(c * 4096) << 8
            int rr = yf + cr * (5743 << 8);
            int gg = yf + cr * -(2925 << 8) + (int)((uint32_t)(cb * -(1410 << 8)) & 0xffff0000u);
            int bb = yf + cb * (7258 << 8);
            out[3 * i] = clamp255(rr >> 20); out[3 * i + 1] = clamp255(gg >> 20); out[3 * i + 2] = clamp255(bb >> 20);
        }
    }
    return r;
}

Raw decodePnm(const std::string& d, const std::string& path)
{
    size_t o = 2; int vals[3], got = 0;
    while (got < 3 && o < d.size()) {
        while (o < d.size() && (d[o] == ' ' || d[o] == '\n' || d[o] == '\r' || d[o] == '\t')) o++;
        if (o < d.size() && d[o] == '#') { while (o < d.size() && d[o] != '\n') o++; continue; }
        int v = 0; bool any = false; while (o < d.size() && isDigit(d[o])) { v = v * 10 + (d[o] - '0'); o++; any = true; }
        if (!any) fail(path + ": bad PNM header");
        vals[got++] = v;
    }
    o++;    // single whitespace after maxval
    Raw r; r.w = vals[0]; r.h = vals[1]; r.n = d[1] == '6' ? 3 : 1;
    if (vals[2] != 255 || o + (size_t)r.w * r.h * r.n > d.size()) fail(path + ": unsupported / truncated PNM");
    r.px.assign((const uint8_t*)d.data() + o, (const uint8_t*)d.data() + o + (size_t)r.w * r.h * r.n);
    return r;
}

} // namespace

Image LoadImage(const std::string& path)
{
    const std::string d = readFile(path, true);
    Raw r;
    if (d.size() >= 8 && memcmp(d.data(), "\x89PNG\r\n\x1a\n", 8) == 0) r = decodePng(d, path);
    else if (d.size() >= 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) r = decodePnm(d, path);
    else if (d.size() >= 3 && (uint8_t)d[0] == 0xFF && (uint8_t)d[1] == 0xD8) r = decodeJpeg(d, path);
    else r = decodeTga(d, path);
    Image img; img.width = r.w; img.height = r.h; img.pixels.resize((size_t)r.w * r.h);
    const size_t s = (size_t)r.w * r.h;
    if (r.n == 1) for (size_t i = 0; i < s; i++) { const uint32_t p = r.px[i]; img.pixels[i] = p + (p << 8) + (p << 16); }      // texture.h:25-31
    else for (size_t i = 0; i < s; i++) {                                                                                          // texture.h:33-36
        const size_t b = i * r.n;
        const uint32_t c2 = (b + 2 < r.px.size()) ? r.px[b + 2] : 0u;      // n == 2 reads one byte past the pixel in the reference
        img.pixels[i] = ((uint32_t)r.px[b] << 16) + ((uint32_t)r.px[b + 1] << 8) + c2;
    }
    return img;
}

// ------------------------------------------------------------------------------------------------
// scene XML
// ------------------------------------------------------------------------------------------------
namespace {

struct XNode { std::string name, text; std::vector<XNode> kids; const XNode* child(const char* n) const { for (auto& k : kids) if (k.name == n) return &k; return nullptr; } };

struct XParser {
    const std::string& s; size_t i = 0; const std::string& path;
    [[noreturn]] void err(const char* what) { fail(path + ": XML parse error: " + what); }
    void skipMisc()
    {
        for (;;) {
            while (i < s.size() && isspace((unsigned char)s[i])) i++;
            if (s.compare(i, 4, "<!--") == 0) { size_t e = s.find("-->", i + 4); if (e == std::string::npos) err("unterminated comment"); i = e + 3; }
            else if (s.compare(i, 2, "<?") == 0) { size_t e = s.find("?>", i + 2); if (e == std::string::npos) err("unterminated declaration"); i = e + 2; }
            else if (s.compare(i, 2, "<!") == 0) { size_t e = s.find('>', i); if (e == std::string::npos) err("unterminated doctype"); i = e + 1; }
            else return;
        }
    }
    static void decode(std::string& t)
    {
        static const struct { const char* e; char c; } ent[] = {{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
        for (auto& en : ent) { size_t p = 0; const size_t l = strlen(en.e); while ((p = t.find(en.e, p)) != std::string::npos) { t.replace(p, l, 1, en.c); p++; } }
    }
    XNode element()
    {
        if (i >= s.size() || s[i] != '<') err("expected element");
        i++;
        XNode n; size_t b = i;
        while (i < s.size() && !isspace((unsigned char)s[i]) && s[i] != '>' && s[i] != '/') i++;
        n.name = s.substr(b, i - b);
        if (n.name.empty()) err("empty element name");
        // attributes are not used by the scene schema: skip to the end of the tag
        while (i < s.size() && s[i] != '>') { if (s[i] == '"' || s[i] == '\'') { const char q = s[i++]; while (i < s.size() && s[i] != q) i++; } if (s[i] == '/' && i + 1 < s.size() && s[i + 1] == '>') { i += 2; return n; } i++; }
        if (i >= s.size()) err("unterminated tag");
        i++;
        bool firstData = true;
        for (;;) {
            const size_t contentStart = i;
            // whitespace between '>' and '<' is not a data node; text keeps its leading whitespace (rapidxml parse<0>)
            size_t j = i; while (j < s.size() && isspace((unsigned char)s[j])) j++;
            if (j >= s.size()) err("unexpected end of file");
            if (s[j] == '<') {
                i = j;
                if (s.compare(i, 2, "</") == 0) { size_t e = s.find('>', i); if (e == std::string::npos) err("unterminated end tag"); i = e + 1; return n; }
                if (s.compare(i, 4, "<!--") == 0) { size_t e = s.find("-->", i + 4); if (e == std::string::npos) err("unterminated comment"); i = e + 3; continue; }
                if (s.compare(i, 9, "<![CDATA[") == 0) { size_t e = s.find("]]>", i + 9); if (e == std::string::npos) err("unterminated CDATA"); if (firstData) { n.text = s.substr(i + 9, e - i - 9); firstData = false; } i = e + 3; continue; }
                n.kids.push_back(element());
            } else {
                size_t e = s.find('<', j); if (e == std::string::npos) err("unexpected end of file");
                if (firstData) { n.text = s.substr(contentStart, e - contentStart); decode(n.text); firstData = false; }
                i = e;
            }
        }
    }
};

float toFloat(const XNode* n, const std::string& path, const char* what)
{
    if (!n) fail(path + ": missing <" + what + ">");
    try { return std::stof(n->text); } catch (...) { fail(path + ": <" + what + "> is not a number"); }
}
float3 toXYZ(const XNode* n, const std::string& path, const char* what)
{
    if (!n) fail(path + ": missing <" + what + ">");
    float3 v(0, 0, 0);
    for (const XNode& k : n->kids) {
        const int idx = k.name[0] - 'x';                      // 'x','y','z' -> 0,1,2 (file_scene.cpp:82)
        if (idx < 0 || idx > 2) fail(path + ": <" + what + "> has a child that is not x / y / z");
        try { v[idx] = std::stof(k.text); } catch (...) { fail(path + ": <" + what + "> component is not a number"); }
    }
    return v;
}
const std::string& toText(const XNode* n, const std::string& path, const char* what)
{
    if (!n) fail(path + ": missing <" + what + ">");
    return n->text;
}

} // namespace

SceneData LoadSceneFile(const std::string& path)
{
    const std::string text = readFile(path, false);
    XParser xp{text, 0, path};
    xp.skipMisc();
    XNode root = xp.element();
    if (root.name != "scene") fail(path + ": root element must be <scene>");
    SceneData sd;
    sd.name = toText(root.child("scene_name"), path, "scene_name");
    sd.lightPos = toXYZ(root.child("light_position"), path, "light_position");
    sd.planeTextureLocation = toText(root.child("plane_texture_location"), path, "plane_texture_location");
    sd.skydomeLocation = toText(root.child("skydome_location"), path, "skydome_location");
    const XNode* objs = root.child("objects");
    if (!objs) fail(path + ": missing <objects>");
    bool started = false;
    for (const XNode& o : objs->kids) {
        if (!started) { if (o.name != "object") continue; started = true; }    // first_node("object"), then every next_sibling()
        ObjectData od;
        od.modelLocation = toText(o.child("model_location"), path, "model_location");
        try { od.materialIdx = std::stoi(toText(o.child("material_idx"), path, "material_idx")); } catch (const std::runtime_error&) { throw; } catch (...) { fail(path + ": <material_idx> is not an integer"); }
        od.position = toXYZ(o.child("position"), path, "position");
        od.rotation = toXYZ(o.child("rotation"), path, "rotation");
        od.scale = toXYZ(o.child("scale"), path, "scale");
        sd.objects.push_back(od);
    }
    const XNode* mats = root.child("materials");
    if (!mats) fail(path + ": missing <materials>");
    started = false;
    for (const XNode& m : mats->kids) {
        if (!started) { if (m.name != "material") continue; started = true; }
        MaterialData md;
        md.reflectivity = toFloat(m.child("reflectivity"), path, "reflectivity");
        md.refractivity = toFloat(m.child("refractivity"), path, "refractivity");
        md.absorption = toXYZ(m.child("absorption"), path, "absorption");
        md.textureLocation = toText(m.child("texture_location"), path, "texture_location");
        sd.materials.push_back(md);
    }
    return sd;
}

} // namespace crt
