// StlOrderMap (csrc/host/stl_order_map.hpp) against std::unordered_map<uint64_t, ...>: same emplace() results and the
// same iteration order after every insert sequence (test infrastructure, built by tests/test_host_cpu.py).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>
#include <vector>

#include "stl_order_map.hpp"

static uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

int main()
{
    int bad = 0;
    std::mt19937_64 rng(12345);
    const size_t sizes[] = {0, 1, 2, 10, 11, 12, 13, 14, 100, 1000, 5000, 100000, 1500000};
    for (size_t n : sizes) {
        for (int style = 0; style < 4; ++style) {
            std::unordered_map<uint64_t, uint32_t> ref;
            vgh::StlOrderMap mine(4);
            for (size_t i = 0; i < n; ++i) {
                uint64_t k;
                switch (style) {
                    case 0: k = mix(i) >> 8 << 8 | 27; break;                       // graph keys: hash << 8 | k
                    case 1: k = rng() % (n / 2 + 1); break;                         // many duplicates
                    case 2: k = i * 13; break;                                      // arithmetic progression (bucket clashes)
                    default: k = (rng() & 0xFFFFF) | ((uint64_t)(i % 7) << 40);     // clustered
                }
                const auto a = ref.emplace(k, (uint32_t)i);
                const auto b = mine.emplace(k);
                if (a.second != b.second) { ++bad; break; }
                if (b.second) {
                    const uint32_t v = (uint32_t)i;
                    memcpy(mine.payload(b.first), &v, 4);
                } else {
                    uint32_t v;
                    memcpy(&v, mine.payload(b.first), 4);
                    if (v != a.first->second) { ++bad; break; }
                }
                if (i % 1000 == 0 && (mine.find(k + 1) != vgh::StlOrderMap::kNil) != (ref.find(k + 1) != ref.end())) { ++bad; break; }
            }
            if (ref.size() != mine.size()) ++bad;
            const std::vector<uint32_t> order = mine.order();
            if (order.size() != ref.size()) ++bad;
            size_t steps = 0;
            for (const auto& kv : ref) {
                if (bad) break;
                const uint32_t id = order[steps];
                if (mine.key(id) != kv.first) { ++bad; break; }
                uint32_t v;
                memcpy(&v, mine.payload(id), 4);
                if (v != kv.second) { ++bad; break; }
                ++steps;
            }
            if (bad) { printf("MISMATCH n=%zu style=%d\n", n, style); return 1; }
        }
    }
    printf("order identical\n");
    return 0;
}
