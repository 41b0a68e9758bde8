// vgh::CpuBudget (csrc/host/genotyper.hpp): never more holders than tokens, nobody starves, set(0) lifts the limit.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "genotyper.hpp"

int main(int argc, char** argv)
{
    const unsigned limit = argc > 1 ? (unsigned)atoi(argv[1]) : 3, n_threads = argc > 2 ? (unsigned)atoi(argv[2]) : 16;
    vgh::CpuBudget::set(limit);
    std::atomic<int> inside{0}, worst{0}, done{0};
    auto work = [&] {
        for (int i = 0; i < 200; ++i) {
            vgh::CpuBudget::Hold h;
            const int now = ++inside;
            for (int w = worst.load(); now > w && !worst.compare_exchange_weak(w, now);) {}
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            --inside;
        }
        ++done;
    };
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < n_threads; ++t) ts.emplace_back(work);
    for (auto& t : ts) t.join();
    const int limited = worst.load();
    vgh::CpuBudget::set(0);
    worst = 0;
    ts.clear();
    done = 0;
    for (unsigned t = 0; t < n_threads; ++t) ts.emplace_back(work);
    for (auto& t : ts) t.join();
    std::printf("%d %d %d\n", limited, worst.load(), done.load());
    return 0;
}
