// host-only check of snake_place.hpp: place_scan8 (the bit-parallel scan the rollout kernel runs) against place_loop8 (the
// digit-by-digit statement of snake_env.py:121-129) on random windows, carries and occupancies, iterated to completion.
//   hipcc -O2 -o /tmp/test_place_scan tools/probes/test_place_scan.hip && /tmp/test_place_scan      (runs on the CPU)
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../custom_gymnasium_environments_amd/csrc/snake_place.hpp"

using namespace cge::snake;

template <int G>
long check(std::mt19937_64 &rng, long trials) {
    long bad = 0;
    for (long t = 0; t < trials; ++t) {
        // occupancy: sparse, dense or almost full boards
        uint64_t lo = 0, hi = 0;
        const int mode = (int)(rng() % 4);
        const int fill = mode == 0 ? 3 : mode == 1 ? 30 : mode == 2 ? G * G - 2 : G * G / 2;
        for (int k = 0; k < fill; ++k) { const uint32_t c = (uint32_t)(rng() % (G * G)); if (c < 64) lo |= 1ull << c; else hi |= 1ull << (c - 64); }
        // a stream of windows; both implementations walk it to completion and must agree on the cell and on every digit consumed
        uint32_t win[16];
        for (auto &w : win) w = (uint32_t)rng();
        if (rng() % 8 == 0) for (auto &w : win) w |= 0xCCCCCCCCu & (uint32_t)rng();        // many invalid digits
        uint32_t pa = 0, ra = 0, pb = 0, rb = 0;
        long ca = 0, cb = 0;                                  // digit cursors into the 128-digit stream
        bool da = false, db = false;
        uint32_t fa = 0, fb = 0;
        auto window = [&](long cur) {                         // 8 digits starting at digit `cur`
            const long w = cur / 8, o = (cur % 8) * 4;
            const uint64_t two = ((uint64_t)win[(w + 1) % 16] << 32) | win[w % 16];
            return (uint32_t)(two >> o);
        };
        for (int it = 0; it < 14 && !da && ca + 8 <= 120; ++it) { PlaceScan r = place_loop8<G>(window(ca), pa, ra, lo, hi); ca += r.used; pa = r.phase; ra = r.row; da = r.done; fa = r.food; }
        for (int it = 0; it < 28 && !db && cb + 8 <= 120; ++it) { PlaceScan r = place_scan8<G>(window(cb), pb, rb, lo, hi); cb += r.used; pb = r.phase; rb = r.row; db = r.done; fb = r.food; }
        const bool same = da == db && (!da || (fa == fb && ca == cb));
        if (!same && (da || db) && !(da != db && (ca + 8 > 120 || cb + 8 > 120))) {
            if (++bad < 5) printf("G=%d mismatch: loop done=%d food=%u used=%ld | scan done=%d food=%u used=%ld\n", G, da, fa, ca, db, fb, cb);
        }
    }
    return bad;
}

int main() {
    std::mt19937_64 rng(12345);
    long bad = 0;
    bad += check<10>(rng, 2000000);
    bad += check<8>(rng, 300000);
    bad += check<12>(rng, 300000);
    bad += check<9>(rng, 300000);
    bad += check<15>(rng, 300000);
    printf(bad ? "FAILED: %ld mismatches\n" : "place_scan8 == place_loop8 on all trials (%ld mismatches)\n", bad);
    return bad ? 1 : 0;
}
