// ppm_common.h -- PPM (P3 text / P6 binary) reader and the P3 writer shared by ppm_diff and
// scaled_ppm_diff.  Behavioural counterpart of the reference's src/ppm_diff/ppm_diff.cpp:37-141
// (read: '#' comments between header tokens, values stored as bytes; write: P3, twelve values
// per text line).  Written from scratch around a byte-level tokenizer.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Ppm {
    int width = 0, height = 0, maxval = 0;
    std::vector<unsigned char> rgb;
    bool ok() const { return width > 0 && height > 0 && maxval > 0 && rgb.size() == (size_t)width * height * 3; }
};

namespace ppm_detail {
inline bool slurp(const std::string& path, std::string& out) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    std::fclose(f);
    return true;
}
// next whitespace-delimited token of the header, skipping "# ..." comment lines
inline bool header_token(const std::string& s, size_t& pos, std::string& tok) {
    for (;;) {
        while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) ++pos;
        if (pos < s.size() && s[pos] == '#') { while (pos < s.size() && s[pos] != '\n') ++pos; continue; }
        break;
    }
    const size_t b = pos;
    while (pos < s.size() && !(s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) ++pos;
    tok = s.substr(b, pos - b);
    return !tok.empty();
}
}  // namespace ppm_detail

inline Ppm read_ppm(const std::string& path) {
    Ppm img;
    std::string data;
    if (!ppm_detail::slurp(path, data)) { std::fprintf(stderr, "Error: Could not open file for reading: %s\n", path.c_str()); return Ppm(); }
    size_t pos = 0;
    std::string magic, w, h, m;
    if (!ppm_detail::header_token(data, pos, magic) || !ppm_detail::header_token(data, pos, w) ||
        !ppm_detail::header_token(data, pos, h) || !ppm_detail::header_token(data, pos, m)) {
        std::fprintf(stderr, "Error: Invalid dimensions or max color value in %s\n", path.c_str());
        return Ppm();
    }
    if (magic != "P3" && magic != "P6") {
        std::fprintf(stderr, "Error: Unsupported PPM format. Expected P3 or P6, got %s in %s\n", magic.c_str(), path.c_str());
        return Ppm();
    }
    img.width = std::atoi(w.c_str()); img.height = std::atoi(h.c_str()); img.maxval = std::atoi(m.c_str());
    if (img.width <= 0 || img.height <= 0 || img.maxval <= 0) {
        std::fprintf(stderr, "Error: Invalid dimensions or max color value in %s\n", path.c_str());
        return Ppm();
    }
    const size_t need = (size_t)img.width * img.height * 3;
    img.rgb.resize(need);
    if (magic == "P6") {
        ++pos;                                   // the single whitespace byte after maxval
        if (data.size() < pos + need) { std::fprintf(stderr, "Error: Failed to read binary pixel data from %s\n", path.c_str()); return Ppm(); }
        for (size_t k = 0; k < need; ++k) img.rgb[k] = (unsigned char)data[pos + k];
    } else {
        for (size_t k = 0; k < need; ++k) {
            while (pos < data.size() && (data[pos] == ' ' || data[pos] == '\t' || data[pos] == '\n' || data[pos] == '\r')) ++pos;
            if (pos >= data.size() || !((data[pos] >= '0' && data[pos] <= '9') || data[pos] == '-' || data[pos] == '+')) {
                std::fprintf(stderr, "Error: Failed to read ASCII pixel data from %s\n", path.c_str());
                return Ppm();
            }
            char* end = nullptr;
            const long v = std::strtol(data.c_str() + pos, &end, 10);
            pos = (size_t)(end - data.c_str());
            img.rgb[k] = (unsigned char)v;      // same narrowing as the reference (ppm_diff.cpp:91)
        }
    }
    return img;
}

inline bool write_ppm_p3(const std::string& path, const Ppm& img) {
    if (!img.ok()) { std::fprintf(stderr, "Error: Invalid image data provided for writing.\n"); return false; }
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "Error: Could not open file for writing: %s\n", path.c_str()); return false; }
    std::string out = "P3\n" + std::to_string(img.width) + " " + std::to_string(img.height) + "\n" + std::to_string(img.maxval) + "\n";
    const size_t n = img.rgb.size();
    for (size_t k = 0; k < n; ++k) {            // twelve values (four pixels) per line, ppm_diff.cpp:119-131
        out += std::to_string((int)img.rgb[k]);
        const bool end_of_line = ((k + 1) % 12 == 0) || (k + 1 == n);
        out += end_of_line ? '\n' : ' ';
    }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    if (std::fclose(f) != 0 || !ok) { std::fprintf(stderr, "Error: Writing to file failed: %s\n", path.c_str()); return false; }
    return true;
}
