// scaled_ppm_diff -- |a-b| image stretched so that [min diff, max diff] fills 0..255.
// Same command line, messages and output as the reference tool
// (src/ppm_diff/scaled_ppm_diff.cpp:143-231):  scaled_ppm_diff <in1.ppm> <in2.ppm> <out.ppm>
#include <algorithm>
#include <climits>
#include "ppm_common.h"

int main(int argc, char** argv) {
    if (argc != 4) { std::fprintf(stderr, "Usage: %s <input1.ppm> <input2.ppm> <output.ppm>\n", argv[0]); return 1; }
    const Ppm a = read_ppm(argv[1]);
    if (!a.ok()) return 1;
    const Ppm b = read_ppm(argv[2]);
    if (!b.ok()) return 1;
    if (a.width != b.width || a.height != b.height) {
        std::fprintf(stderr, "Error: Image dimensions do not match.\n%s: %dx%d\n%s: %dx%d\n", argv[1], a.width, a.height, argv[2], b.width, b.height);
        return 1;
    }
    if (a.maxval != b.maxval)
        std::fprintf(stderr, "Warning: Max color values differ (%d vs %d). Differences will be calculated based on their original values.\n", a.maxval, b.maxval);
    const size_t n = a.rgb.size();
    std::vector<int> raw(n);
    int lo = INT_MAX, hi = INT_MIN;
    for (size_t k = 0; k < n; ++k) { raw[k] = std::abs((int)a.rgb[k] - (int)b.rgb[k]); lo = std::min(lo, raw[k]); hi = std::max(hi, raw[k]); }
    Ppm d;
    d.width = a.width; d.height = a.height; d.maxval = 255;
    d.rgb.assign(n, 0);
    if (hi == lo) {
        std::printf("Images are identical or have uniform difference. Outputting a black image.\n");
    } else {
        const double scale = 255.0 / (hi - lo);                    // scaled_ppm_diff.cpp:211
        for (size_t k = 0; k < n; ++k) d.rgb[k] = (unsigned char)std::max(0, std::min(255, (int)((raw[k] - lo) * scale)));
        std::printf("Minimum difference found: %d\nMaximum difference found: %d\nDifferences scaled to 0-255 range.\n", lo, hi);
    }
    if (!write_ppm_p3(argv[3], d)) return 1;
    std::printf("Successfully wrote scaled difference image to %s\n", argv[3]);
    return 0;
}
