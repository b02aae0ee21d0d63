// xorwow_jump.h -- the 2^67-stride subsequence jump matrices of XORWOW (curand_init skip-ahead), built on the host
// Host side of librtiow_hip.so; part of the single translation unit rtiow_hip.hip (internal linkage).
#pragma once
#include "../device/xorwow.h"

namespace {

// =====================================================================================
// XORWOW jump matrices
// =====================================================================================
struct Mat160 { uint32_t col[XW_BITS][XW_WORDS]; };

void mat_vec(const Mat160& m, const uint32_t* in, uint32_t* out) {
    uint32_t acc[XW_WORDS] = {0, 0, 0, 0, 0};
    for (int w = 0; w < XW_WORDS; ++w)
        for (uint32_t bits = in[w]; bits; bits &= bits - 1) {          // the set bits only
            const uint32_t* c = m.col[w * 32 + __builtin_ctz(bits)];
            for (int k = 0; k < XW_WORDS; ++k) acc[k] ^= c[k];
        }
    std::memcpy(out, acc, sizeof acc);
}

// Jump matrices A^(2^(67+b)), b = 0..31, of the xorshift part of XORWOW (A = the one-step matrix,
// built by pushing the 160 basis vectors through the generator).  A^(2^67) is a committed constant
// (xorwow_jump67.inc, written by gen/gen_xorwow_jump67.cpp), so a process pays 31 squarings instead
// of 98; `from_scratch` derives everything from A and is what the tests compare the constant with.
// `count` = how many of the 32 to build: rng_init_kernel reads matrix b only when bit b of a pixel index is set.
const uint32_t kJump67[XW_BITS * XW_WORDS] = {
#include "xorwow_jump67.inc"
};

std::vector<uint32_t> build_sequence_jump_matrices(bool from_scratch = false, int count = XW_JUMPS) {
    Mat160 cur, nxt;
    int done = 0;
    if (from_scratch) {
        for (int b = 0; b < XW_BITS; ++b) {
            uint32_t v[XW_WORDS] = {0, 0, 0, 0, 0};
            v[b >> 5] = 1u << (b & 31);
            const uint32_t t = v[0] ^ (v[0] >> 2);
            const uint32_t n4 = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
            cur.col[b][0] = v[1]; cur.col[b][1] = v[2]; cur.col[b][2] = v[3]; cur.col[b][3] = v[4]; cur.col[b][4] = n4;
        }
    } else {
        std::memcpy(&cur.col[0][0], kJump67, sizeof kJump67);
        done = 67;
    }
    std::vector<uint32_t> out;
    out.reserve((size_t)count * XW_MAT_WORDS);
    for (int e = done; e < 67 + count; ++e) {
        if (e >= 67) out.insert(out.end(), &cur.col[0][0], &cur.col[0][0] + XW_MAT_WORDS);
        if (e + 1 == 67 + count) break;
        for (int b = 0; b < XW_BITS; ++b) mat_vec(cur, cur.col[b], nxt.col[b]);
        cur = nxt;
    }
    return out;
}

}  // namespace
