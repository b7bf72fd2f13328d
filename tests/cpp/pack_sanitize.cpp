// Host-side sanitizer run of the weight packing (csrc/pack.cpp): built with -fsanitize=address,undefined together
// with pack.cpp and runtime.cpp themselves (no GPU, no HIP call), it packs every (dtype, cout, cin, tile, shuffle)
// combination the networks and the tests use plus a few ragged ones into exactly-sized heap buffers, so that an
// index outside the packed image or the padded bias is caught by ASan's red zones, and checks that every byte of
// the image was written.  Prints "ok <cases>".
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "srmi355.h"

int main() {
    struct Case { int cout, cin, tile, ps_r; };
    const Case cases[] = {
        {32, 64, 32, 1}, {32, 96, 32, 1}, {32, 128, 32, 1}, {32, 160, 32, 1}, {64, 192, 32, 1}, {64, 192, 64, 1},
        {64, 64, 64, 1}, {64, 64, 32, 1}, {256, 64, 64, 2}, {256, 64, 32, 2}, {576, 64, 64, 3}, {288, 32, 32, 3},
        {1024, 256, 64, 2}, {256, 256, 64, 1}, {3, 64, 32, 1}, {3, 256, 32, 1}, {96, 96, 32, 1}, {32, 32, 32, 1},
    };
    const int dtypes[] = {SRMI_F32, SRMI_F16, SRMI_BF16, SRMI_I8};
    int done = 0;
    for (int dt : dtypes) {
        for (const Case& c : cases) {
            if (dt == SRMI_I8 && (c.cin % 32 || c.ps_r != 1)) continue;   // int8 planes hold 32 channels; dense blocks only
            int nbias = 0;
            const size_t bytes = srmi_pack_conv3x3_bytes(dt, c.cout, c.cin, c.tile, &nbias);
            if (!bytes || nbias < c.cout) { fprintf(stderr, "bad sizes for cout %d cin %d\n", c.cout, c.cin); return 1; }
            std::vector<float> w((size_t)c.cout * c.cin * 9), b((size_t)c.cout);
            uint32_t s = 1234567u + (uint32_t)done;
            for (auto& v : w) { s = s * 1664525u + 1013904223u; v = dt == SRMI_I8 ? (float)((int)(s >> 24) - 128) : ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
            for (auto& v : b) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f; }
            // exactly-sized buffers: one byte too many written is a heap-buffer-overflow report
            std::vector<unsigned char> img(bytes, 0xCD);
            std::vector<float> bias((size_t)nbias, -777.0f);
            if (srmi_pack_conv3x3(dt, c.cout, c.cin, c.tile, c.ps_r, w.data(), b.data(), img.data(), bias.data())) {
                fprintf(stderr, "pack failed: %s\n", srmi_last_error());
                return 1;
            }
            for (int i = 0; i < nbias; ++i)
                if (bias[i] == -777.0f) { fprintf(stderr, "bias %d of %d not written (cout %d)\n", i, nbias, c.cout); return 1; }
            // a second pack into a buffer with another fill must give the same bytes: nothing is left unwritten
            std::vector<unsigned char> img2(bytes, 0x37);
            std::vector<float> bias2((size_t)nbias, 0.f);
            if (srmi_pack_conv3x3(dt, c.cout, c.cin, c.tile, c.ps_r, w.data(), nullptr, img2.data(), bias2.data())) return 1;
            if (memcmp(img.data(), img2.data(), bytes)) {
                fprintf(stderr, "dtype %d cout %d cin %d tile %d ps %d: packed image has unwritten bytes\n", dt, c.cout, c.cin, c.tile, c.ps_r);
                return 1;
            }
            for (int i = 0; i < nbias; ++i)
                if (bias2[i] != 0.f) { fprintf(stderr, "NULL bias must pack zeros\n"); return 1; }
            ++done;
        }
    }
    // the two Winograd forms of the dense-block convolutions (f16 images [tile][stage][position][half][32][8 halves])
    for (const Case& c : cases) {
        if (c.cin % 16 || c.ps_r != 1 || c.tile != 32) continue;
        for (int form = 0; form < 2; ++form) {
            int nbias = 0;
            const size_t bytes = form ? srmi_pack_wino1d3x3_bytes(c.cout, c.cin, &nbias) : srmi_pack_wino3x3_bytes(c.cout, c.cin, &nbias);
            if (!bytes || nbias < c.cout) { fprintf(stderr, "bad Winograd sizes for cout %d cin %d\n", c.cout, c.cin); return 1; }
            std::vector<float> w((size_t)c.cout * c.cin * 9), b((size_t)c.cout, 0.25f);
            uint32_t s = 7654321u + (uint32_t)done;
            for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
            std::vector<unsigned char> img(bytes, 0xCD), img2(bytes, 0x37);
            std::vector<float> bias((size_t)nbias, -777.0f), bias2((size_t)nbias, -777.0f);
            auto pack = form ? srmi_pack_wino1d3x3 : srmi_pack_wino3x3;
            if (pack(c.cout, c.cin, w.data(), b.data(), img.data(), bias.data()) || pack(c.cout, c.cin, w.data(), nullptr, img2.data(), bias2.data())) {
                fprintf(stderr, "Winograd pack failed: %s\n", srmi_last_error());
                return 1;
            }
            if (memcmp(img.data(), img2.data(), bytes)) { fprintf(stderr, "Winograd image (form %d, cout %d, cin %d) has unwritten bytes\n", form, c.cout, c.cin); return 1; }
            for (int i = 0; i < nbias; ++i)
                if (bias[i] == -777.0f || bias2[i] != 0.f) { fprintf(stderr, "Winograd bias %d not written\n", i); return 1; }
            ++done;
        }
    }
    // the sub-pixel form of an upsampling convolution (f16 / bf16 images [tile][stage][tap 2x2][row][32 bytes], bias per phase)
    for (const Case& c : cases) {
        if (c.cin % 16 || c.ps_r != 1 || c.cout % c.tile) continue;
        for (int dt : {SRMI_F16, SRMI_BF16}) {
            int nbias = 0;
            const size_t bytes = srmi_pack_conv3x3_up2_bytes(dt, c.cout, c.cin, c.tile, &nbias);
            if (!bytes || nbias != 4 * c.cout) { fprintf(stderr, "bad sub-pixel sizes for cout %d cin %d\n", c.cout, c.cin); return 1; }
            std::vector<float> w((size_t)c.cout * c.cin * 9), b((size_t)c.cout, 0.25f);
            uint32_t s = 2468013u + (uint32_t)done;
            for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
            std::vector<unsigned char> img(bytes, 0xCD), img2(bytes, 0x37);
            std::vector<float> bias((size_t)nbias, -777.0f), bias2((size_t)nbias, -777.0f);
            if (srmi_pack_conv3x3_up2(dt, c.cout, c.cin, c.tile, w.data(), b.data(), img.data(), bias.data()) ||
                srmi_pack_conv3x3_up2(dt, c.cout, c.cin, c.tile, w.data(), nullptr, img2.data(), bias2.data())) {
                fprintf(stderr, "sub-pixel pack failed: %s\n", srmi_last_error());
                return 1;
            }
            if (memcmp(img.data(), img2.data(), bytes)) { fprintf(stderr, "sub-pixel image (cout %d, cin %d) has unwritten bytes\n", c.cout, c.cin); return 1; }
            for (int i = 0; i < nbias; ++i)
                if (bias[i] != 0.25f || bias2[i] != 0.f) { fprintf(stderr, "sub-pixel bias %d not written\n", i); return 1; }
            ++done;
        }
    }
    if (srmi_pack_conv3x3_up2_bytes(SRMI_F32, 64, 64, 32, nullptr) || srmi_pack_conv3x3_up2_bytes(SRMI_F16, 48, 64, 32, nullptr)) {
        fprintf(stderr, "sub-pixel form accepted f32 or a ragged output count\n");
        return 1;
    }
    // rejected shapes return -1 with a message and write nothing
    unsigned char dummy[64];
    float fb[64];
    if (srmi_pack_conv3x3(SRMI_F16, 48, 64, 32, 1, nullptr, nullptr, dummy, fb) == 0) { fprintf(stderr, "null weights accepted\n"); return 1; }
    printf("ok %d\n", done);
    return 0;
}
