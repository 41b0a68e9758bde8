// ubench_x87.cpp -- what bounds the HMM's forward / backward recursion on the host: the loop of genotyper.cpp (A)
// against the same sums from a table of products (B: no multiply) and with the 80-bit operands as exact pairs of doubles (C).
// g++ -O3 -std=c++17 tools/ubench_x87.cpp -o tools/bin/x87micro.  EPYC 9575F: A 0.80, B 0.80, C 0.60 ns per term: the 80-bit
// load is the bound.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
    const size_t NP = 120, NG = 120, REP = 20000;
    std::vector<long double> step(NP * 3), obs(NG), ptab(NP * 3);
    std::vector<double> hi(NP * 3), lo(NP * 3);
    std::vector<uint8_t> keep(NG * NP);
    srand(1);
    for (auto& v : step) v = 1e-3L * (1 + rand() % 1000) / 7.0L;
    for (auto& v : obs) v = 1e-5L * (1 + rand() % 1000) / 3.0L;
    for (auto& v : keep) v = rand() % 3;
    for (size_t q = 0; q < NP * 3; ++q) { ptab[q] = step[q] * obs[0]; hi[q] = (double)step[q]; lo[q] = (double)(step[q] - (long double)hi[q]); }
    long double sink = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (size_t rep = 0; rep < REP; ++rep) {           // A: current
        for (size_t g = 0; g + 3 <= NG; g += 3) {
            const uint8_t *k0 = &keep[g * NP], *k1 = k0 + NP, *k2 = k1 + NP;
            const long double o0 = obs[g], o1 = obs[g + 1], o2 = obs[g + 2];
            long double r0 = 0, r1 = 0, r2 = 0;
            const long double* sp = step.data();
            for (size_t pi = 0; pi < NP; ++pi, sp += 3) { r0 += sp[k0[pi]] * o0; r1 += sp[k1[pi]] * o1; r2 += sp[k2[pi]] * o2; }
            sink += r0 + r1 + r2;
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    for (size_t rep = 0; rep < REP; ++rep) {           // B: products from an L1-resident table, four chains
        for (size_t g = 0; g + 4 <= NG; g += 4) {
            const uint8_t *k0 = &keep[g * NP], *k1 = k0 + NP, *k2 = k1 + NP, *k3 = k2 + NP;
            long double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            const long double* sp = ptab.data();
            for (size_t pi = 0; pi < NP; ++pi, sp += 3) { r0 += sp[k0[pi]]; r1 += sp[k1[pi]]; r2 += sp[k2[pi]]; r3 += sp[k3[pi]]; }
            sink += r0 + r1 + r2 + r3;
        }
    }
    auto t2 = std::chrono::steady_clock::now();
    for (size_t rep = 0; rep < REP; ++rep) {           // C: step as double hi + double lo (exact), then multiply and add
        for (size_t g = 0; g + 3 <= NG; g += 3) {
            const uint8_t *k0 = &keep[g * NP], *k1 = k0 + NP, *k2 = k1 + NP;
            const long double o0 = obs[g], o1 = obs[g + 1], o2 = obs[g + 2];
            long double r0 = 0, r1 = 0, r2 = 0;
            const double *h = hi.data(), *l = lo.data();
            for (size_t pi = 0; pi < NP; ++pi, h += 3, l += 3) {
                r0 += ((long double)h[k0[pi]] + l[k0[pi]]) * o0;
                r1 += ((long double)h[k1[pi]] + l[k1[pi]]) * o1;
                r2 += ((long double)h[k2[pi]] + l[k2[pi]]) * o2;
            }
            sink += r0 + r1 + r2;
        }
    }
    auto t3 = std::chrono::steady_clock::now();
    const double terms = (double)REP * NG * NP;
    auto ns = [&](auto a, auto b) { return std::chrono::duration<double, std::nano>(b - a).count() / terms; };
    printf("A fld80+fmul+fadd: %.3f ns/term   B fld80+fadd (table): %.3f   C fld64+fadd64+fmul+fadd: %.3f   (%Lg)\n", ns(t0, t1), ns(t1, t2), ns(t2, t3), sink);
}
