// snake_place.hpp — _place_food (snake_env.py:121-129) over a window of 8 pre-drawn 4-bit digits, without a per-digit loop.
//
// CPython: food = (randint(0,G-1), randint(0,G-1)) redrawn while it lies on the snake; randint(0,G-1) = _randbelow(G): r =
// getrandbits(4) (8 <= G <= 15), redrawn while r >= G.  So a placement is a scan over the stream of 4-bit digits: digits >= G are
// skipped, the valid ones pair up (row, column), the first pair whose cell is free wins.  The rollout's digit queue hands the
// scan 8 digits at a time; `phase` / `row` carry a half-finished pair from one window into the next.
//
// scan8 examines at most the first TWO pairs a window completes (the first one is accepted 97+ % of the time) and reports how many
// digits it consumed, so a caller loops until `done`.  place_loop8 is the digit-by-digit statement of the same rule; the two are
// checked against each other exhaustively-at-random by tools/probes/test_place_scan.hip (host build, no GPU needed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cge {
namespace snake {

struct PlaceScan {
    uint32_t food, used, phase, row;
    bool done;
};

__host__ __device__ __forceinline__ bool occ128(uint64_t lo, uint64_t hi, uint32_t cell) {
    return (((cell < 64u ? lo : hi) >> (cell & 63u)) & 1ull) != 0ull;
}

// reference statement: one digit after the other
template <int G>
__host__ __device__ inline PlaceScan place_loop8(uint32_t bits, uint32_t phase, uint32_t row, uint64_t occ_lo, uint64_t occ_hi) {
    PlaceScan r{0u, 0u, phase, row, false};
    for (int j = 0; j < 8 && !r.done; ++j) {
        const uint32_t d = (bits >> (4 * j)) & 15u;
        r.used = (uint32_t)j + 1u;
        if (d < (uint32_t)G) {
            if (r.phase == 0u) { r.row = d; r.phase = 1u; }
            else {
                r.phase = 0u;
                const uint32_t cell = r.row * (uint32_t)G + d;
                if (!occ128(occ_lo, occ_hi, cell)) { r.food = cell; r.done = true; }
            }
        }
    }
    return r;
}

// bit 4j of the result is set iff digit j of `bits` is < G
template <int G>
__host__ __device__ __forceinline__ uint32_t valid_digits(uint32_t bits) {
    static_assert(G >= 8 && G <= 15, "4-bit digits");
    if constexpr (G == 8) return ~(bits >> 3) & 0x11111111u;
    else if constexpr (G == 10) return ~((bits >> 3) & ((bits >> 2) | (bits >> 1))) & 0x11111111u;        // d >= 10 <=> d3 & (d2 | d1)
    else if constexpr (G == 12) return ~((bits >> 3) & (bits >> 2)) & 0x11111111u;                          // d >= 12 <=> d3 & d2
    else {
        uint32_t v = 0;
        for (int j = 0; j < 8; ++j) v |= (((bits >> (4 * j)) & 15u) < (uint32_t)G ? 1u : 0u) << (4 * j);
        return v;
    }
}

template <int G>
__host__ __device__ __forceinline__ PlaceScan place_scan8(uint32_t bits, uint32_t phase, uint32_t row, uint64_t occ_lo, uint64_t occ_hi) {
    const uint32_t v0 = valid_digits<G>(bits);
    const uint32_t count = (uint32_t)__builtin_popcount(v0);
    // bit positions (4j) of the first four valid digits; 32 = none
    const uint32_t v1 = v0 & (v0 - 1u), v2 = v1 & (v1 - 1u), v3 = v2 & (v2 - 1u);
    const uint32_t q1 = v0 ? (uint32_t)__builtin_ctz(v0) : 32u, q2 = v1 ? (uint32_t)__builtin_ctz(v1) : 32u;
    const uint32_t q3 = v2 ? (uint32_t)__builtin_ctz(v2) : 32u, q4 = v3 ? (uint32_t)__builtin_ctz(v3) : 32u;
    const uint32_t d1 = (bits >> (q1 & 31u)) & 15u, d2 = (bits >> (q2 & 31u)) & 15u, d3 = (bits >> (q3 & 31u)) & 15u, d4 = (bits >> (q4 & 31u)) & 15u;
    // pair 1 = (carried row, d1) or (d1, d2); pair 2 = the next two valid digits
    const bool carry = phase != 0u;
    const uint32_t need1 = carry ? 1u : 2u;
    const uint32_t r1 = carry ? row : d1, c1 = carry ? d1 : d2, e1 = carry ? q1 : q2;             // e: bit position of the completing digit
    const uint32_t r2 = carry ? d2 : d3, c2 = carry ? d3 : d4, e2 = carry ? q3 : q4;
    const bool have1 = count >= need1, have2 = count >= need1 + 2u;
    const uint32_t cell1 = r1 * (uint32_t)G + c1, cell2 = r2 * (uint32_t)G + c2;
    const bool ok1 = have1 && !occ128(occ_lo, occ_hi, cell1);
    const bool ok2 = !ok1 && have2 && !occ128(occ_lo, occ_hi, cell2);
    PlaceScan r;
    r.done = ok1 || ok2;
    r.food = ok1 ? cell1 : cell2;
    // digits consumed: through the accepted pair; through pair 2 if both pairs were examined and refused (the rest of the window
    // is looked at by the next call); otherwise the whole window
    r.used = ok1 ? (e1 >> 2) + 1u : (have2 ? (e2 >> 2) + 1u : 8u);
    // state carried out: a pair is half finished iff an odd number of valid digits follows the last completed pair
    const uint32_t left = have2 ? 0u : (have1 ? count - need1 : count);       // valid digits that did not complete a pair (whole window consumed)
    const bool odd = !r.done && !have2 && (have1 ? (left & 1u) : ((left + phase) & 1u));
    // ... and then `row` is the last valid digit of the window (with carry and no valid digit at all the old row stays)
    const uint32_t last = count >= 4u ? d4 : count == 3u ? d3 : count == 2u ? d2 : count == 1u ? d1 : row;
    r.phase = odd ? 1u : 0u;
    r.row = odd ? last : row;
    return r;
}

}  // namespace snake
}  // namespace cge
