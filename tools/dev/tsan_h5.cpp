// build + run: g++ -O1 -g -std=c++17 -fsanitize=thread -pthread -Iinclude -o /tmp/tsan_h5 tools/dev/tsan_h5.cpp poreplex_amd/csrc/pxg_h5.cpp poreplex_amd/csrc/pxg_zcodec.cpp poreplex_amd/csrc/pxg_text.cpp -lz -ldl && /tmp/tsan_h5 some_multi_read.fast5
// TSan driver: worker threads decoding stretches of one multi-read FAST5 file at once (the pool busy -> small jobs on
// their callers), plus threads opening single-read files by the batch (pxg_h5_open_many)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "pxg.h"
int main(int argc, char** argv)
{
    pxg_h5* f = nullptr;
    if (pxg_h5_open_mt(argv[1], 4, &f)) { printf("open failed: %s\n", pxg_h5_last_error()); return 1; }
    const int64_t n = pxg_h5_n_reads(f);
    std::vector<pxg_h5_read_info> info((size_t)n);
    if (pxg_h5_info_mt(f, 0, n, info.data(), 4)) return 2;
    std::vector<std::thread> ts;
    for (int t = 0; t < 8; t++)
        ts.emplace_back([&, t] {
            for (int rep = 0; rep < 30; rep++) {
                const int64_t k = 16, lo = ((t * 7 + rep * 3) * k) % (n - k);
                std::vector<const pxg_h5*> files((size_t)k, f);
                std::vector<int64_t> idx((size_t)k), dst((size_t)k), ns((size_t)k), s0((size_t)k), sl((size_t)k), m0((size_t)k), nm((size_t)k);
                int64_t total = 0, st = 0, mt = 0;
                for (int64_t i = 0; i < k; i++) {
                    idx[i] = lo + i; dst[i] = total; ns[i] = info[lo + i].n_samples; total += ns[i];
                    s0[i] = st; sl[i] = info[lo + i].bc_present ? info[lo + i].bc_seq_len : 0; st += sl[i];
                    m0[i] = mt; nm[i] = info[lo + i].bc_present && info[lo + i].bc_n_moves > 0 ? info[lo + i].bc_n_moves : 0; mt += nm[i];
                }
                std::vector<int16_t> arena((size_t)total);
                std::vector<uint8_t> sa((size_t)st), qa((size_t)st), ma((size_t)mt);
                std::vector<int32_t> s1((size_t)k), s2((size_t)k);
                pxg_h5_load_signals(k, files.data(), idx.data(), dst.data(), ns.data(), arena.data(), (rep & 1) ? 4 : 1, s1.data());
                pxg_h5_basecall_many(k, files.data(), idx.data(), s0.data(), sl.data(), sa.data(), qa.data(), m0.data(), nm.data(), ma.data(), 4, s2.data());
                for (int64_t i = 0; i < k; i++) if (s1[i] || s2[i]) { printf("status %d %d\n", s1[i], s2[i]); exit(3); }
            }
        });
    for (auto& t : ts) t.join();
    pxg_h5_close(f);
    printf("ok: %lld reads, 8 threads x 30 calls\n", (long long)n);
    return 0;
}
