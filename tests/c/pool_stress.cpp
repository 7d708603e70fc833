// Six callers post parallel_for jobs of 1 .. 40 tasks to one zl_pool at the same time (what the completion threads of a proof's MSMs do with their window
// Horners): every task of every job must run exactly once.  Built and run by tests/test_abi.py (also under -fsanitize=thread when the compiler has it).
#include "zl_pool.h"
#include <cstdio>
#include <atomic>
int main() {
    zl_pool pool(7);
    std::atomic<long> bad{0};
    std::vector<std::thread> callers;
    for (int c = 0; c < 6; c++) callers.emplace_back([&, c]() {
        for (int it = 0; it < 20000; it++) {
            const size_t n = 1 + (size_t)((it * 7 + c * 13) % 40);
            std::vector<int> hit(n, 0);
            pool.parallel_for(n, [&](size_t i) { hit[i]++; });
            for (size_t i = 0; i < n; i++) if (hit[i] != 1) bad++;
        }
    });
    for (auto& t : callers) t.join();
    printf("bad = %ld\n", bad.load());
    return bad.load() != 0;
}
