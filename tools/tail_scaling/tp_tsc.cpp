#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <x86intrin.h>
#include "aprilgrid-rs_amd/csrc/host_tail.hpp"
namespace agx { extern unsigned long long g_tsc[16], g_cnt2[16]; }
using namespace agx;
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int w, h, n;
    if (fread(&w, 4, 1, f) != 1 || fread(&h, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) return 2;
    std::vector<agx_saddle> s(n);
    std::vector<uint8_t> g((size_t)w * h);
    if (fread(s.data(), 20, n, f) != (size_t)n || fread(g.data(), 1, g.size(), f) != g.size()) return 2;
    fclose(f);
    FamilyInfo fam;
    family_info(AGX_T36H11, fam);
    std::vector<agx_tag> tags;
    const int reps = 200;
    detect_tail(fam, 2, s, g.data(), w, h, (size_t)w, tags);
    for (int i = 0; i < 16; ++i) g_tsc[i] = g_cnt2[i] = 0;
    const auto t0 = std::chrono::steady_clock::now();
    const unsigned long long c0 = __rdtsc();
    for (int rep = 0; rep < reps; ++rep) detect_tail(fam, 2, s, g.data(), w, h, (size_t)w, tags);
    const unsigned long long c1 = __rdtsc();
    const double ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / reps;
    const double cyc_per_ms = (double)(c1 - c0) / reps / ms;
    printf("%zu tags, %.3f ms per call\n", tags.size(), ms);
    const char *names[] = {"pair miss", "valid_quad miss", "Board ctor (incl. expand, nested)", "init_quads", "  50-NN", "decode_quad", "pair_candidates total", "valid_quad total", "index reset"};
    for (int i = 0; i < 9; ++i) printf("  %-36s %8.3f ms per call  %8.1f calls  %7.0f ns each\n", names[i], g_tsc[i] / cyc_per_ms / reps, (double)g_cnt2[i] / reps, g_cnt2[i] ? g_tsc[i] / cyc_per_ms / g_cnt2[i] * 1e6 : 0.0);
}
