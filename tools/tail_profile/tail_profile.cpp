// Where the host tail of detect() (board search + decode, aprilgrid-rs_amd/csrc/host_tail.cpp) spends
// its time, per round of TagDetector::detect's loop (src/detector.rs:510-539).  Builds host_tail.cpp
// with -DAGX_TAIL_PROFILE (timers compiled in only then) -- no GPU, no library:
//   g++ -O3 -std=c++17 -DAGX_TAIL_PROFILE -I. tools/tail_profile/tail_profile.cpp aprilgrid-rs_amd/csrc/host_tail.cpp -o /tmp/tail_profile
//   python tools/tail_profile/dump_case.py /tmp/case.bin [frame index | image name]     (oracle saddles + u8 luma)
//   /tmp/tail_profile /tmp/case.bin
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "aprilgrid-rs_amd/csrc/host_tail.hpp"

namespace agx {
extern double g_tail_prof[12];
extern long g_tail_cnt[12];
}
using namespace agx;

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int w, h, n;
    if (fread(&w, 4, 1, f) != 1 || fread(&h, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) return 2;
    std::vector<agx_saddle> s(n);
    std::vector<uint8_t> g((size_t)w * h);
    if (fread(s.data(), 20, n, f) != (size_t)n || fread(g.data(), 1, g.size(), f) != g.size()) return 2;
    fclose(f);
    FamilyInfo fam;
    family_info(AGX_T36H11, fam);
    std::vector<agx_tag> tags;
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;
    double best = 1e9;
    for (int rep = 0; rep < reps; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        detect_tail(fam, getenv("BOARDS") ? atoi(getenv("BOARDS")) : 2, s, g.data(), w, h, (size_t)w, tags);
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
    }
    printf("%d saddles -> %zu tags; best %.3f ms per call\n", n, tags.size(), best);
    const char *names[12] = {"index build", "init_quads", "boards (expand)", "fix_missing + collect", "decode", "boards built", "50-NN of init_quads", "pair look-ups", "pair misses (2 x 3-NN each)", "valid_quad misses", "angle_degree (atan2f) calls", "points scanned by the 1- / 3-NN queries"};
    for (int i = 0; i < 12; ++i)
        printf("  %-22s %8.3f ms per call   (%ld per call)\n", names[i], g_tail_prof[i] * 1e3 / reps, g_tail_cnt[i] / reps);
    return 0;
}
