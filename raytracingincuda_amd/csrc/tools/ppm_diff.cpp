// ppm_diff -- per-component |a-b| image of two PPMs, the parity gate of this repository.
// Same command line and output file as the reference tool (src/ppm_diff/ppm_diff.cpp:143-208):
//     ppm_diff <input1.ppm> <input2.ppm> <output.ppm>
// The reference tool has NO threshold (its README asks the user to look at a "rather dark
// image"); this one additionally prints the statistics on stdout and, when tolerances are
// given, turns them into the exit status:
//     --max-mean M   fail (exit 2) if the mean absolute difference exceeds M levels
//     --max-p99 P    fail if the 99th percentile exceeds P
//     --max-abs A    fail if any component differs by more than A
#include <algorithm>
#include <cstring>
#include "ppm_common.h"

int main(int argc, char** argv) {
    std::vector<std::string> pos;
    double max_mean = -1, max_p99 = -1, max_abs = -1;
    for (int k = 1; k < argc; ++k) {
        if (!std::strcmp(argv[k], "--max-mean") && k + 1 < argc) max_mean = std::atof(argv[++k]);
        else if (!std::strcmp(argv[k], "--max-p99") && k + 1 < argc) max_p99 = std::atof(argv[++k]);
        else if (!std::strcmp(argv[k], "--max-abs") && k + 1 < argc) max_abs = std::atof(argv[++k]);
        else pos.push_back(argv[k]);
    }
    if (pos.size() != 3) { std::fprintf(stderr, "Usage: %s <input1.ppm> <input2.ppm> <output.ppm>\n", argv[0]); return 1; }
    const Ppm a = read_ppm(pos[0]);
    if (!a.ok()) return 1;
    const Ppm b = read_ppm(pos[1]);
    if (!b.ok()) return 1;
    if (a.width != b.width || a.height != b.height) {
        std::fprintf(stderr, "Error: Image dimensions do not match.\n%s: %dx%d\n%s: %dx%d\n", pos[0].c_str(), a.width, a.height, pos[1].c_str(), b.width, b.height);
        return 1;
    }
    if (a.maxval != b.maxval)
        std::fprintf(stderr, "Warning: Max color values differ (%d vs %d). Using %d for output.\n", a.maxval, b.maxval, a.maxval);
    Ppm d;
    d.width = a.width; d.height = a.height; d.maxval = std::min(a.maxval, b.maxval);
    d.rgb.resize(a.rgb.size());
    std::vector<size_t> hist(256, 0);
    unsigned long long sum = 0;
    int worst = 0;
    for (size_t k = 0; k < a.rgb.size(); ++k) {
        const int diff = std::abs((int)a.rgb[k] - (int)b.rgb[k]);
        d.rgb[k] = (unsigned char)diff;
        ++hist[diff]; sum += (unsigned)diff; worst = std::max(worst, diff);
    }
    if (!write_ppm_p3(pos[2], d)) return 1;
    std::printf("Successfully wrote difference image to %s\n", pos[2].c_str());
    const size_t n = a.rgb.size();
    auto pct = [&](double q) { size_t need = (size_t)(q * (double)n), acc = 0; for (int v = 0; v < 256; ++v) { acc += hist[v]; if (acc > need || acc == n) return v; } return 255; };
    const double mean = (double)sum / (double)n;
    const int p50 = pct(0.50), p90 = pct(0.90), p99 = pct(0.99);
    std::printf("components %zu  mean %.6f  p50 %d  p90 %d  p99 %d  max %d  identical %s\n", n, mean, p50, p90, p99, worst, worst == 0 ? "yes" : "no");
    bool fail = false;
    if (max_mean >= 0 && mean > max_mean) { std::printf("FAIL: mean %.6f > %.6f\n", mean, max_mean); fail = true; }
    if (max_p99 >= 0 && p99 > max_p99) { std::printf("FAIL: p99 %d > %.0f\n", p99, max_p99); fail = true; }
    if (max_abs >= 0 && worst > max_abs) { std::printf("FAIL: max %d > %.0f\n", worst, max_abs); fail = true; }
    return fail ? 2 : 0;
}
