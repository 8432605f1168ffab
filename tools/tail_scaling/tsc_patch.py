"""Inserts rdtsc section counters (pair-candidate misses, valid_quad misses, Board construction, init_quads and its 50-NN, decode_quad,
the memo look-ups, the index build) into a copy of host_tail.cpp: python tsc_patch.py <host_tail.cpp> <out.cpp>; tp_tsc.cpp prints them."""
import re,sys
src=open(sys.argv[1]).read()
# add tsc accumulators
src=src.replace('namespace agx {\n','namespace agx {\n#include <x86intrin.h>\nunsigned long long g_tsc[16]={0}; unsigned long long g_cnt2[16]={0};\nstruct Tsc{int s; unsigned long long t0; Tsc(int s_):s(s_),t0(__rdtsc()){} ~Tsc(){g_tsc[s]+=__rdtsc()-t0; g_cnt2[s]++;}};\n',1)
# pair miss
src=src.replace('            AGX_TAIL_COUNT(8, 1);\n            PairCands e;','            AGX_TAIL_COUNT(8, 1);\n            Tsc tsc_pm(0);\n            PairCands e;')
# valid quad miss
src=src.replace('            AGX_TAIL_COUNT(9, 1);\n            const bool v = is_valid_quad','            AGX_TAIL_COUNT(9, 1);\n            Tsc tsc_vq(1);\n            const bool v = is_valid_quad')
# board ctor
src=src.replace('        st_.begin(refined.size());\n        for (int i = 1; i < 4; ++i) st_.use(seed[i]);  // board.rs:35-37','        Tsc tsc_b(2);\n        st_.begin(refined.size());\n        for (int i = 1; i < 4; ++i) st_.use(seed[i]);  // board.rs:35-37')
# init_quads whole + 50NN
src=src.replace('    out.clear();\n    const agx_saddle &s0 = refined[s0_idx];\n    SaddleIndex::Hit near[50];','    Tsc tsc_iq(3);\n    out.clear();\n    const agx_saddle &s0 = refined[s0_idx];\n    SaddleIndex::Hit near[50];')
src=src.replace('        m = index.nearest(s0.x, s0.y, 50, near);','        Tsc tsc_nn(4);\n        m = index.nearest(s0.x, s0.y, 50, near);')
# decode
for fn in ('std::round', 'round_half_away'):
    src=src.replace('    for (int i = 0; i < 4; ++i) {\n        const uint32_t x = f32_as_u32(%s(quad_xy[2 * i]))' % fn,'    Tsc tsc_dec(5);\n    for (int i = 0; i < 4; ++i) {\n        const uint32_t x = f32_as_u32(%s(quad_xy[2 * i]))' % fn)
# pair_candidates total
src=src.replace('        const std::vector<agx_saddle> &pts_ = *pts_p_;\n        AGX_TAIL_COUNT(7, 1);','        const std::vector<agx_saddle> &pts_ = *pts_p_;\n        Tsc tsc_pc(6);\n        AGX_TAIL_COUNT(7, 1);')
# valid_quad total
src=src.replace('        const std::vector<agx_saddle> &pts_ = *pts_p_;\n        if (pts_.size() >= 65535u)','        const std::vector<agx_saddle> &pts_ = *pts_p_;\n        Tsc tsc_vt(7);\n        if (pts_.size() >= 65535u)')
# index reset
src=src.replace('        pts_p_ = &pts_in;\n','        Tsc tsc_ir(8);\n        pts_p_ = &pts_in;\n')
open(sys.argv[2],'w').write(src)
