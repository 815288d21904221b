// CPU-only fuzz harness for csrc/pngdec.cpp under ASan + UBSan: mutated and truncated PNG files must be
// decoded or refused, never read or written out of bounds.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "spacecarve.h"
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int rounds = atoi(argv[1]);
    unsigned long long accepted = 0, refused = 0;
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int f = 2; f < argc; ++f) {
        FILE *fp = fopen(argv[f], "rb");
        if (!fp) continue;
        std::vector<uint8_t> orig;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) orig.insert(orig.end(), buf, buf + n);
        fclose(fp);
        for (int r = 0; r < rounds; ++r) {
            std::vector<uint8_t> d = orig;
            const int kind = (int)(rnd() % 6);
            if (kind == 0 && !d.empty()) d.resize(rnd() % d.size());                       // truncation
            else if (kind == 1) for (int q = 0; q < 1 + (int)(rnd() % 8); ++q) d[rnd() % d.size()] = (uint8_t)rnd();  // byte noise
            else if (kind == 2 && d.size() > 40) for (int q = 16; q < 29; ++q) if (rnd() % 3 == 0) d[q] = (uint8_t)rnd();  // header fields
            else if (kind == 3 && d.size() > 60) { size_t a = 33 + rnd() % (d.size() - 40); d[a] = (uint8_t)rnd(); d[a + 1] = (uint8_t)rnd(); }  // chunk lengths / data
            else if (kind == 4) d.insert(d.begin() + (rnd() % (d.size() + 1)), (size_t)(rnd() % 64), (uint8_t)rnd());  // inserted bytes
            // kind 5: unchanged
            int W = 0, H = 0;
            // exact-size heap copy: any read past the end trips ASan
            uint8_t *in = (uint8_t *)malloc(d.size() ? d.size() : 1);
            memcpy(in, d.data(), d.size());
            if (sc_png_info(in, (int64_t)d.size(), &W, &H) == SC_OK && (int64_t)W * H <= (1 << 26)) {
                uint8_t *out = (uint8_t *)malloc((size_t)W * H);
                if (sc_png_decode_gray8(in, (int64_t)d.size(), out, W, H) == SC_OK) ++accepted; else ++refused;
                // wrong sizes must be refused, not written
                if (sc_png_decode_gray8(in, (int64_t)d.size(), out, W > 1 ? W - 1 : W + 1, H) == SC_OK) { printf("accepted a wrong width\n"); return 1; }
                free(out);
            } else ++refused;
            free(in);
        }
    }
    printf("accepted %llu, refused %llu\n", accepted, refused);
    return 0;
}
