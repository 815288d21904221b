// pngdec.cpp -- decoder for the mask images of the ingest path (8-bit greyscale, non-interlaced PNG:
// what `io.write_image(f, im, 'png')` makes of the uint8 masks of tasks/proc2d.py:376-381 and what
// plantdb.io.read_image hands back to cl.py:298 as an (H, W) uint8 array).
//
// Why it exists: decoding is the floor of the files -> volume time once the carve takes 0.2 ms, and the
// Python decoders do not scale over threads (PIL: 32 ms for 72 masks on 8, 16 or 32 threads).  This one
// is called through the C ABI, i.e. with the interpreter lock released, from the decode-ahead threads
// of Backprojection.process_label.  Anything but plain grey8 is refused (SC_ERR_INVALID) and the caller
// falls back to the Python reader.  Host code only: zlib's inflate + the five PNG row filters
// (PNG specification, section 9: None, Sub, Up, Average, Paeth).

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "spacecarve.h"

namespace {

thread_local char g_png_err[160] = "";

int png_fail(const char *msg) {
    strncpy(g_png_err, msg, sizeof g_png_err - 1);
    return SC_ERR_INVALID;
}

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

// IHDR of a PNG we decode: returns SC_OK and the size, or SC_ERR_INVALID with the reason
int parse_header(const uint8_t *d, int64_t len, int *W, int *H) {
    if (len < 33 || memcmp(d, kSig, 8) != 0) return png_fail("not a PNG");
    if (be32(d + 8) != 13 || memcmp(d + 12, "IHDR", 4) != 0) return png_fail("no IHDR");
    const uint32_t w = be32(d + 16), h = be32(d + 20);
    const int depth = d[24], colour = d[25], comp = d[26], filt = d[27], interlace = d[28];
    if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) return png_fail("bad size");
    // filter byte + row, all rows: below 2^31 (zlib counts its output in 32 bits; a mask is a few megabytes) and
    // not absurd for the file at hand -- deflate cannot expand more than 1032 : 1, so a header that declares
    // more pixels than that is not describing this file
    const uint64_t need = ((uint64_t)w + 1) * (uint64_t)h;
    if (need >= (1ull << 31) || need > (uint64_t)len * 1032ull + 65536ull) return png_fail("declared size does not fit the file");
    if (depth != 8 || colour != 0) return png_fail("not 8-bit greyscale");
    if (comp != 0 || filt != 0 || interlace != 0) return png_fail("interlaced or unknown method");
    *W = (int)w;
    *H = (int)h;
    return SC_OK;
}

inline uint8_t paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

}  // namespace

extern "C" {

const char *sc_png_last_error(void) { return g_png_err; }

int sc_png_info(const void *data, int64_t len, int *W, int *H) {
    if (!data || !W || !H) return png_fail("null argument");
    return parse_header(static_cast<const uint8_t *>(data), len, W, H);
}

int sc_png_decode_gray8(const void *data, int64_t len, uint8_t *out, int W, int H) {
    if (!data || !out) return png_fail("null argument");
    const uint8_t *d = static_cast<const uint8_t *>(data);
    int w = 0, h = 0;
    int rc = parse_header(d, len, &w, &h);
    if (rc) return rc;
    if (w != W || h != H) return png_fail("size differs from the header");
    // inflate the concatenated IDAT chunks into filter-byte + row records
    const size_t stride = (size_t)W + 1;
    std::vector<uint8_t> raw;
    try {
        raw.resize(stride * (size_t)H);
    } catch (const std::bad_alloc &) {  // never across the C boundary
        snprintf(g_png_err, sizeof g_png_err, "out of memory for a %d x %d image", W, H);
        return SC_ERR_NOMEM;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return png_fail("inflateInit failed");
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    int64_t pos = 8;
    bool done = false, seen_idat = false;
    while (pos + 12 <= len && !done) {
        const uint32_t clen = be32(d + pos);
        const uint8_t *type = d + pos + 4;
        if (pos + 12 + (int64_t)clen > len) { inflateEnd(&zs); return png_fail("truncated chunk"); }
        if (memcmp(type, "IDAT", 4) == 0) {
            // the chunk's CRC covers its type and data (PNG specification, section 5.3)
            if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + clen) != be32(d + pos + 8 + clen)) {
                inflateEnd(&zs);
                return png_fail("IDAT checksum mismatch");
            }
            seen_idat = true;
            zs.next_in = const_cast<uint8_t *>(d + pos + 8);
            zs.avail_in = clen;
            const int zr = inflate(&zs, Z_NO_FLUSH);
            if (zr == Z_STREAM_END) done = true;
            else if (zr != Z_OK && zr != Z_BUF_ERROR) { inflateEnd(&zs); return png_fail("inflate failed"); }
        } else if (memcmp(type, "IEND", 4) == 0) {
            break;
        } else if (memcmp(type, "PLTE", 4) == 0 || memcmp(type, "tRNS", 4) == 0) {
            inflateEnd(&zs);
            return png_fail("palette / transparency chunk");  // leave such files to the Python reader
        }
        pos += 12 + (int64_t)clen;
    }
    const bool complete = zs.avail_out == 0;
    inflateEnd(&zs);
    if (!seen_idat || !complete) return png_fail("image data incomplete");
    // undo the row filters (bytes per pixel = 1)
    const uint8_t *prev = nullptr;
    for (int y = 0; y < H; ++y) {
        const uint8_t *src = raw.data() + (size_t)y * stride;
        uint8_t *dst = out + (size_t)y * (size_t)W;
        const int ft = src[0];
        ++src;
        switch (ft) {
            case 0: memcpy(dst, src, (size_t)W); break;
            case 1:
                dst[0] = src[0];
                for (int x = 1; x < W; ++x) dst[x] = (uint8_t)(src[x] + dst[x - 1]);
                break;
            case 2:
                if (!prev) memcpy(dst, src, (size_t)W);
                else for (int x = 0; x < W; ++x) dst[x] = (uint8_t)(src[x] + prev[x]);
                break;
            case 3:
                for (int x = 0; x < W; ++x) {
                    const int a = x ? dst[x - 1] : 0, b = prev ? prev[x] : 0;
                    dst[x] = (uint8_t)(src[x] + ((a + b) >> 1));
                }
                break;
            case 4:
                for (int x = 0; x < W; ++x) {
                    const int a = x ? dst[x - 1] : 0, b = prev ? prev[x] : 0, c = (x && prev) ? prev[x - 1] : 0;
                    dst[x] = (uint8_t)(src[x] + paeth(a, b, c));
                }
                break;
            default: return png_fail("unknown row filter");
        }
        prev = dst;
    }
    return SC_OK;
}

}  // extern "C"
