// How the host tail of detect() (aprilgrid-rs_amd/csrc/host_tail.cpp: board search + decode, reference
// src/detector.rs:510-539) scales over host threads, without a GPU and without the library: T threads take
// frames from an atomic counter and run detect_tail on them.  Beside it ("spin") the same threads run a
// fixed amount of register arithmetic: the box's own ceiling for T threads (cgroup quota, SMT, clocks).
//   g++ -O3 -std=c++17 -pthread -I. tools/tail_scaling/tail_scaling.cpp aprilgrid-rs_amd/csrc/host_tail.cpp -o /tmp/tail_scaling
//   python tools/tail_scaling/dump_cases.py /tmp/cases.bin 64
//   /tmp/tail_scaling /tmp/cases.bin 1,2,4,8 [total frames per measurement = 512]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "aprilgrid-rs_amd/csrc/host_tail.hpp"

using namespace agx;
using clk = std::chrono::steady_clock;

struct Case {
    std::vector<agx_saddle> s;
    std::vector<uint8_t> g;
};

static double spin_rate(int T)
{
    std::atomic<long> sink{0};
    const long iters = 200000000L;
    std::vector<std::thread> th;
    const auto t0 = clk::now();
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            unsigned long a = 88172645463325252ull + (unsigned long)t;
            for (long i = 0; i < iters; ++i) {
                a ^= a << 13;
                a ^= a >> 7;
                a ^= a << 17;
            }
            sink += (long)a;
        });
    for (auto &x : th) x.join();
    const double dt = std::chrono::duration<double>(clk::now() - t0).count();
    return (double)T * (double)iters / dt * 1e-9;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int w, h, n;
    if (fread(&w, 4, 1, f) != 1 || fread(&h, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) return 2;
    std::vector<Case> cases((size_t)n);
    for (Case &c : cases) {
        int ns;
        if (fread(&ns, 4, 1, f) != 1) return 2;
        c.s.resize((size_t)ns);
        c.g.resize((size_t)w * h);
        if (fread(c.s.data(), 20, (size_t)ns, f) != (size_t)ns || fread(c.g.data(), 1, c.g.size(), f) != c.g.size()) return 2;
    }
    fclose(f);
    const long total = argc > 3 ? atol(argv[3]) : 512;
    FamilyInfo fam;
    family_info(AGX_T36H11, fam);
    std::vector<int> counts;
    for (char *p = strtok(argv[2], ","); p; p = strtok(nullptr, ",")) counts.push_back(atoi(p));
    double base = 0, spin_base = 0;
    printf("%d frames %dx%d; %ld tails per measurement; hardware_concurrency %u\n", n, w, h, total, std::thread::hardware_concurrency());
    printf("%8s %12s %10s %12s %10s\n", "threads", "frames/s", "x 1 thread", "spin G/s", "x 1 thread");
    for (int T : counts) {
        std::atomic<long> next{0}, tag_sum{0};
        std::vector<std::thread> th;
        const auto t0 = clk::now();
        for (int t = 0; t < T; ++t)
            th.emplace_back([&] {
                std::vector<agx_tag> tags;
                long mine = 0;
                for (;;) {
                    const long i = next.fetch_add(1);
                    if (i >= total) break;
                    const Case &c = cases[(size_t)(i % n)];
                    detect_tail(fam, 2, c.s, c.g.data(), w, h, (size_t)w, tags);
                    mine += (long)tags.size();
                }
                tag_sum += mine;
            });
        for (auto &x : th) x.join();
        const double dt = std::chrono::duration<double>(clk::now() - t0).count();
        const double fps = (double)total / dt;
        const double sp = getenv("NO_SPIN") ? 0.0 : spin_rate(T);
        if (base == 0) base = fps / T, spin_base = sp / T;
        printf("%8d %12.1f %10.2f %12.2f %10.2f   (%.1f tags per frame)\n", T, fps, fps / base, sp, spin_base > 0 ? sp / spin_base : 0.0,
               (double)tag_sum.load() / (double)total);
        fflush(stdout);
    }
    return 0;
}
