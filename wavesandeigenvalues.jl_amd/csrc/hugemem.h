// Large host vectors of the set-up on transparent huge pages.  The set-up's host phases are bound by first-touch page faults
// (sixteen threads filling fresh 100..300-MB vectors through 4-KB pages: a triple product runs 1.6 s cold against 0.33 s on memory
// the process has touched before); where the kernel offers transparent huge pages on request (`madvise` mode, the setting of the
// GPU boxes) the interior of a reserved range is advised before it is first written, which divides the number of faults by 512.
// No effect (and no error) where THP is off.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include <sys/mman.h>

inline void advise_huge(const void *p, size_t bytes) {
    static const bool on = !(getenv("WAE_HUGE_PAGES") && atoi(getenv("WAE_HUGE_PAGES")) == 0);       // (A/B switch)
    if (!on) return;
    constexpr uintptr_t H = (uintptr_t)2 << 20;
    const uintptr_t lo = ((uintptr_t)p + H - 1) & ~(H - 1), hi = ((uintptr_t)p + bytes) & ~(H - 1);
    if (hi > lo) (void)madvise((void *)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
}
// reserve + advise; a following resize / assign / push_back up to n elements does not reallocate
template <class T> inline void huge_reserve(std::vector<T> &v, size_t n) {
    if (n > v.capacity()) v.reserve(n);
    if (n * sizeof(T) >= ((size_t)4 << 20)) advise_huge(v.data(), n * sizeof(T));
}
