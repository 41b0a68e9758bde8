// mem_advice.hpp -- large host arrays that are read at random (per-key tables indexed by key number) ask for 2 MiB pages:
// with 4 KiB pages every access is a TLB miss and every first touch a page fault (graph.bin key index: 0.27 -> 0.07 s).
#pragma once
#include <sys/mman.h>

#include <cstddef>
#include <cstdint>

namespace vgh {

inline void advise_huge_pages(const void* p, size_t bytes)
{
#ifdef MADV_HUGEPAGE
    constexpr uintptr_t kHuge = (uintptr_t)2 << 20;
    if (bytes < 2 * kHuge) return;
    const uintptr_t a = ((uintptr_t)p + kHuge - 1) & ~(kHuge - 1), b = ((uintptr_t)p + bytes) & ~(kHuge - 1);
    if (b > a) (void)madvise(reinterpret_cast<void*>(a), b - a, MADV_HUGEPAGE);   // advice only: failure changes nothing
#else
    (void)p;
    (void)bytes;
#endif
}

}  // namespace vgh
