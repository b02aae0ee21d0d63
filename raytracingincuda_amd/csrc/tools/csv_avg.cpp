// csv_avg -- averages the timing columns of a benchmark CSV over its runs; the C++ counterpart
// of the reference's timing-benchmarks/process.py:16-33 (pandas groupby(...).mean()):
//     csv_avg <input.csv> <output.csv>
// input : scene_id,width,height,samples,bounces,threads,run,render_only_time_ms,end_to_end_time_ms
// output: scene_id,width,height,samples,bounces,threads,avg_render_only_time_ms,avg_end_to_end_time_ms
// Groups are written in ascending key order; empty timing cells (a failed launch leaves the row
// ending after `run,`) are skipped like pandas skips NaN, a group without any value gives empty
// cells.  Means use compensated (Kahan) summation and are printed as the shortest text that
// round-trips the double -- what pandas does, so the reference's own in->out CSV pairs are
// reproduced byte for byte (tests/test_tools.py).
#include <algorithm>
#include <array>
#include <charconv>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

struct Acc { double sum = 0, comp = 0; long n = 0;
    void add(double v) { const double y = v - comp, t = sum + y; comp = (t - sum) - y; sum = t; ++n; } };

static std::string shortest(double v) {           // Python's repr(float)
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    std::string s(buf, r.ptr);
    if (s.find_first_of(".en") == std::string::npos) s += ".0";   // "100" -> "100.0" (not for inf/nan)
    return s;
}

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "Usage: %s <input.csv> <output.csv>\n", argv[0]); return 1; }
    std::ifstream in(argv[1]);
    if (!in) { std::fprintf(stderr, "Error: Input file '%s' not found.\n", argv[1]); return 1; }
    std::string line;
    if (!std::getline(in, line)) { std::fprintf(stderr, "Error: empty input\n"); return 1; }
    std::map<std::array<long, 6>, std::array<Acc, 2>> groups;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> cell;
        std::stringstream ss(line);
        std::string c;
        while (std::getline(ss, c, ',')) cell.push_back(c);
        if (cell.size() < 7) continue;
        std::array<long, 6> key;
        for (int k = 0; k < 6; ++k) key[k] = std::strtol(cell[k].c_str(), nullptr, 10);
        auto& g = groups[key];
        for (int k = 0; k < 2; ++k) {
            if ((int)cell.size() <= 7 + k) continue;
            const char* b = cell[7 + k].c_str();
            char* e = nullptr;
            const double v = std::strtod(b, &e);
            if (e == b) continue;                                   // empty / non-numeric -> NaN -> skipped
            g[k].add(v);
        }
    }
    std::ofstream out(argv[2]);
    if (!out) { std::fprintf(stderr, "Error: Could not open file for writing: %s\n", argv[2]); return 1; }
    out << "scene_id,width,height,samples,bounces,threads,avg_render_only_time_ms,avg_end_to_end_time_ms\n";
    for (const auto& kv : groups) {
        for (int k = 0; k < 6; ++k) out << kv.first[k] << ",";
        for (int k = 0; k < 2; ++k) {
            if (kv.second[k].n) out << shortest(kv.second[k].sum / (double)kv.second[k].n);
            out << (k == 0 ? "," : "\n");
        }
    }
    std::printf("Averaged data saved to '%s'\n", argv[2]);
    return 0;
}
