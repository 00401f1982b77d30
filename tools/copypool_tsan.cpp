// ThreadSanitizer / ASan check of the host thread pool and the pack / widen work items of the host-buffer pipeline
// (b2f_ctx.h; host code only, no GPU needed):
//   g++ -std=c++17 -O1 -g -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iback2future_amd/csrc \
//       tools/copypool_tsan.cpp -o /tmp/copypool_tsan -lpthread && /tmp/copypool_tsan
#include "b2f_ctx.h"

#include <cstdio>
#include <cstdlib>

using namespace b2f;

int main()
{
    int bad = 0;
    for (int workers : {0, 1, 3, 7}) {
        CopyPool pa(workers), pb(workers);
        auto body = [&](CopyPool &pool, unsigned seed) {
            for (int it = 0; it < 40; ++it) {
                seed = seed * 1664525u + 1013904223u;
                const size_t n = 1000 + (seed >> 8) % (3u << 20);            // floats
                std::vector<float> src(n);
                std::vector<unsigned char> q(n);
                for (size_t i = 0; i < n; ++i) { q[i] = (unsigned char)((i * 7 + seed) & 255); src[i] = (float)q[i] / 255.0f; }
                if (it % 5 == 4) src[n / 2] = 0.3f;                             // not k / 255
                std::vector<float> cp(n);
                std::vector<double> wide(n);
                std::vector<unsigned char> packed(n);
                std::atomic<int> inexact{0};
                pool.run({{cp.data(), src.data(), n * 4},
                          {wide.data(), src.data(), n * 4, JOB_F32_TO_F64, 1.5, nullptr},
                          {packed.data(), src.data(), n * 4, JOB_PACK_U8, 1.0, &inexact}});
                for (size_t i = 0; i < n; ++i) {
                    if (cp[i] != src[i] || wide[i] != (double)src[i] * 1.5) ++bad;
                    if (it % 5 != 4 && packed[i] != q[i]) ++bad;
                }
                if ((inexact.load() != 0) != (it % 5 == 4)) ++bad;
            }
        };
        std::thread t1([&] { body(pa, 1u); }), t2([&] { body(pb, 2u); });
        t1.join();
        t2.join();
    }
    printf("copy pool: %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
