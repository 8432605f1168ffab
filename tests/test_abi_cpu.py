"""The C-ABI library without a GPU: it loads, exports every symbol include/aprilgrid_amd.h
declares, mirrors the reference's constructor-level behaviour, refuses to work without a
device (no CPU fallback), and its host tail agrees with the oracle's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests.util import ROOT, bits_equal, load_image, synth_module


@pytest.fixture(scope="module")
def lib():
    from aprilgrid_rs_amd import _ffi
    return _ffi.lib()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "aprilgrid_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(agx_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from aprilgrid_rs_amd import _ffi
    names = declared_symbols()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
    assert sorted(_ffi.SYMBOLS) == names, "ctypes table and header disagree"
    assert lib.agx_abi_version() == 1


def test_family_from_str_like_the_reference(lib):
    """src/tag_families.rs:661-685: lower and upper case accepted, 'invalid' rejected."""
    import aprilgrid_rs_amd as A
    assert A.TagFamily.from_str("t36h11") == A.TagFamily.T36H11 == A.TagFamily.from_str("T36H11")
    assert A.TagFamily.from_str("t16h5") == A.TagFamily.T16H5
    assert A.TagFamily.from_str("T36H11B1") == A.TagFamily.T36H11B1
    with pytest.raises(ValueError):
        A.TagFamily.from_str("invalid")


def test_default_params(lib):
    import aprilgrid_rs_amd as A
    p = A.DetectorParams.default_params()
    assert (p.min_saddle_angle, p.max_saddle_angle, p.max_num_of_boards) == (30.0, 60.0, 2)
    assert abs(p.tag_spacing_ratio - 0.3) < 1e-6


def test_no_device_means_no_results(lib):
    """There is no CPU fallback: without a gfx950 device the constructor fails loudly."""
    import torch
    import aprilgrid_rs_amd as A
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(A.AgxError) as e:
        A.TagDetector(A.TagFamily.T36H11)
    assert e.value.status == -5


def test_constants_equal_the_oracles(lib):
    from oracle import oracle as O
    w = np.zeros(7, np.float32)
    cone = np.zeros(25, np.float32)
    pm = np.zeros((25, 6), np.float32)
    assert lib.agx_detector_constants(None, w.ctypes.data, cone.ctypes.data, pm.ctypes.data) == 0
    op, ok = O.refine_constants(2)
    assert bits_equal(w, O.blur_weights(1.5)) and bits_equal(cone, ok) and bits_equal(pm, op)


def test_luma8_matches_oracle(lib):
    import aprilgrid_rs_amd as A
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    for img in (rng.integers(0, 256, (9, 13), dtype=np.uint8), rng.integers(0, 65536, (9, 13), dtype=np.uint16),
                rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)):
        assert np.array_equal(A.TagDetector.luma8(img), O.luma_u8(img))


@pytest.mark.parametrize("name", ["EuRoC.png", "TUM_VI.png", "r45.png", "two_boards.png", "iphone.png"])
def test_host_tail_matches_oracle_tail_on_fixtures(name):
    """Board search + decode (host C++, aprilgrid-rs_amd/csrc/host_tail.cpp) against the oracle's
    tail, both fed the oracle's saddles: identical ids, bit-identical corners, and the reference's
    tag count (tests/test_detector.rs:26-32)."""
    import aprilgrid_rs_amd as A
    from oracle import oracle as O
    from tests.util import REFERENCE_TAG_COUNTS
    img = load_image(name)
    saddles = O.refined_saddle_points(img)
    grey = O.luma_u8(img)
    got = A.TagDetector.detect_tail("t36h11", saddles, grey)
    ref = O.detect_tail(grey, saddles)
    assert sorted(got) == sorted(ref)
    assert len(got) == dict(REFERENCE_TAG_COUNTS)[name]
    for tid in ref:
        assert bits_equal(got[tid], ref[tid])


def test_host_tail_matches_oracle_tail_on_synthetic_and_other_families():
    import aprilgrid_rs_amd as A
    from oracle import oracle as O
    synth = synth_module()
    for i in range(12):
        frame, gt = synth.render_frame(20 + i, 640, 400)
        img = frame.numpy()
        saddles = O.refined_saddle_points(img)
        got = A.TagDetector.detect_tail(A.TagFamily.T36H11, saddles, img)
        ref = O.detect_tail(img, saddles)
        assert sorted(got) == sorted(ref) and len(got) >= 30
        for tid in ref:
            assert bits_equal(got[tid], ref[tid])
    # a family that does not match the drawn tags decodes (almost) nothing, identically
    for fam in ("T16H5", "T25H9", "T36H11B1"):
        got = A.TagDetector.detect_tail(fam, saddles, img)
        ref = O.detect_tail(img, saddles, family=fam)
        assert sorted(got) == sorted(ref)
    # max_num_of_boards = 1 vs 2 on the two-board fixture
    two = load_image("two_boards.png")
    s2, g2 = O.refined_saddle_points(two), O.luma_u8(two)
    p1 = A.DetectorParams(0.3, 30.0, 60.0, 1)
    op = O.default_params(); op.max_num_of_boards = 1
    a1 = A.TagDetector.detect_tail("t36h11", s2, g2, p1)
    assert sorted(a1) == sorted(O.detect_tail(g2, s2, params=op)) and 30 <= len(a1) <= 36
    assert A.TagDetector.detect_tail("t36h11", s2[:0], g2) == {}
    for nb in (0, 3, 5):  # detector.rs:510: the loop runs max_num_of_boards times, found or not
        op.max_num_of_boards = nb
        got = A.TagDetector.detect_tail("t36h11", s2, g2, A.DetectorParams(0.3, 30.0, 60.0, nb))
        assert sorted(got) == sorted(O.detect_tail(g2, s2, params=op)) and len(got) == (0 if nb == 0 else 72)


def test_host_tail_matches_oracle_tail_on_many_and_perturbed_saddle_sets():
    """The product's board search evaluates is_valid_quad from per-pair / per-diagonal tables and
    memoised neighbour queries; the oracle restates the reference's loops literally.  Both must give
    the same tags on many frames and on saddle sets the chain would never produce: random subsets,
    jittered positions, shuffled orientations (many rejected quads, ties, partial boards)."""
    import aprilgrid_rs_amd as A
    from oracle import oracle as O
    synth = synth_module()
    rng = np.random.default_rng(12)
    n_tags = 0
    for i in range(48):
        w, h = [(640, 400), (800, 600), (1280, 800)][i % 3]
        img = synth.render_frame(1000 + i, w, h)[0].numpy()
        base = O.refined_saddle_points(img)
        variants = [base]
        keep = rng.random(len(base)) < 0.8
        variants.append(base[keep])                                    # missing corners
        j = base.copy()
        j["x"] += rng.normal(0, 0.7, len(j)).astype(np.float32)        # geometry off by a fraction of a pixel
        j["y"] += rng.normal(0, 0.7, len(j)).astype(np.float32)
        variants.append(j)
        t = base.copy()
        flip = rng.random(len(t)) < 0.15
        t["theta"][flip] = rng.uniform(-90, 90, int(flip.sum())).astype(np.float32)  # wrong orientations
        variants.append(t)
        for v, sad in enumerate(variants):
            got = A.TagDetector.detect_tail("t36h11", sad, img)
            ref = O.detect_tail(img, sad)
            assert sorted(got) == sorted(ref), (i, v, len(got), len(ref))
            # the same frame's board search on several threads (option "tail_threads"): same tags, same order
            par = A.TagDetector.detect_tail("t36h11", sad, img, n_threads=2 + (i + v) % 5)
            assert list(par) == list(got) and all(bits_equal(par[t], got[t]) for t in got), (i, v, "threads")
            for tid in ref:
                assert bits_equal(got[tid], ref[tid]), (i, v, tid)
            n_tags += len(ref)
    assert n_tags > 3000


def test_host_tail_library_entry_points_via_cpu_handle(lib):
    """The host tail itself is CPU code; exercise it here through the exported helpers that do
    not need a handle."""
    assert lib.agx_luma8(None, 4, 4, 4, 0, None) == -1
    fam = C.c_int(-1)
    assert lib.agx_family_from_str(b"t25h9", C.byref(fam)) == 0 and fam.value == 2
    assert lib.agx_family_from_str(b"nope", C.byref(fam)) == -6
    assert lib.agx_status_string(-3) == b"capacity exceeded"


def _build_c_client(tmp_path):
    """examples/c_client.c: a plain-C99 program against include/aprilgrid_amd.h, linked with the
    library -- no Python, no torch in that process."""
    import subprocess
    exe = str(tmp_path / "c_client")
    pkg = os.path.join(ROOT, "aprilgrid-rs_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_client.c"), "-L", pkg, "-laprilgrid_amd",
                    "-Wl,-rpath," + pkg, "-o", exe], check=True)
    return exe


def _build_c_group_client(tmp_path):
    """examples/c_group_client.c: the detector groups of the ABI from plain C (uses the HIP runtime
    only to place the frames on the devices, as a Rust host would)."""
    import subprocess
    exe = str(tmp_path / "c_group_client")
    pkg = os.path.join(ROOT, "aprilgrid-rs_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__",
                    "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "examples", "c_group_client.c"), "-L", pkg, "-laprilgrid_amd", "-L", "/opt/rocm/lib",
                    "-lamdhip64", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_c_group_client_builds(tmp_path):
    assert os.path.exists(_build_c_group_client(tmp_path))


def test_c_client_builds_and_fails_loudly_without_a_device(tmp_path):
    import subprocess
    import numpy as np
    exe = _build_c_client(tmp_path)
    raw = tmp_path / "flat.raw"
    np.full((64, 64), 128, np.uint8).tofile(raw)
    r = subprocess.run([exe, str(raw), "64", "64"], capture_output=True, text=True)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        assert r.returncode == 0 and "0 saddles, 0 tags" in r.stdout, (r.stdout, r.stderr)
    else:  # the product path has no CPU fallback
        assert r.returncode == 1 and "no usable gfx950 device" in r.stderr, (r.stdout, r.stderr)



def test_compiled_blur_kernel_keeps_its_pending_poll_register_alone():
    """tools/check_isa.py on the gfx950 image inside the built library: the register that receives the
    asynchronous scalar poll of K1 is not touched before the wait that follows it, in any variant."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    n, problems = check_isa.check(os.path.join(ROOT, "aprilgrid-rs_amd", "libaprilgrid_amd.so"))
    assert n == 24 and not problems, problems  # 4 formats x {aligned, generic, unaligned (UF)} x {product, stored response}


def test_bounded_angle_approximation_of_the_board_search(lib):
    """The host tail decides `|a0 - a2| > 10` and `60 <= |a| <= 120` from a polynomial approximation
    of atan2 wherever the value is farther than its guard band (0.005 degrees per angle) from the
    threshold, and from the reference's expression inside the band.  That is exact as long as the
    approximation stays within the band of angle_degree: checked here on random, axis-aligned, nearly
    parallel / antiparallel, tiny and huge vectors."""
    rng = np.random.default_rng(4)
    n = 1 << 20
    parts = [rng.normal(0, 100, (n, 4)), rng.uniform(-2000, 2000, (n, 4)),
             rng.normal(0, 1, (n, 4)) * 10.0 ** rng.uniform(-18, 18, (n, 1))]
    a = rng.uniform(-np.pi, np.pi, n)
    eps = rng.normal(0, 1e-4, n) * rng.integers(0, 2, n)
    flip = rng.integers(0, 2, n) * np.pi
    r0, r1 = rng.uniform(0.1, 500, n), rng.uniform(0.1, 500, n)
    parts.append(np.stack([r0 * np.cos(a), r0 * np.sin(a), r1 * np.cos(a + eps + flip), r1 * np.sin(a + eps + flip)], 1))
    q = rng.integers(-3, 4, (n, 4)).astype(np.float64)  # lattice vectors: exact 0 / 45 / 90 / 180 degrees, zero vectors
    parts.append(q)
    v = np.ascontiguousarray(np.concatenate(parts), np.float32)
    m = len(v)
    exact, approx, has = np.zeros(m, np.float32), np.zeros(m, np.float32), np.zeros(m, np.uint8)
    assert lib.agx_debug_angle_pairs(v.ctypes.data, m, exact.ctypes.data, approx.ctypes.data, has.ctypes.data) == 0
    ref = np.degrees(np.arctan2((v[:, 3] * v[:, 0] - v[:, 2] * v[:, 1]).astype(np.float32).astype(np.float64),
                                (v[:, 0] * v[:, 2] + v[:, 1] * v[:, 3]).astype(np.float32).astype(np.float64)))
    used = has != 0
    assert used.mean() > 0.9
    assert np.abs(exact[used].astype(np.float64) - ref[used]).max() < 2e-4      # angle_degree vs binary64
    err = np.abs(approx[used].astype(np.float64) - exact[used].astype(np.float64))
    assert err.max() < 5e-4, err.max()                                           # a tenth of the guard band
    # operands the approximation refuses: zero cross product (sign of zero decides 0 / +-180) and zero vectors
    y = (v[:, 3] * v[:, 0] - v[:, 2] * v[:, 1]).astype(np.float32)
    assert not used[y == 0].any()
    # round 5: the coarser level in front of it (three-term float polynomial, guard band 0.1 degrees per angle)
    coarse, hasc = np.zeros(m, np.float32), np.zeros(m, np.uint8)
    assert lib.agx_debug_angle_pairs_coarse(v.ctypes.data, m, coarse.ctypes.data, hasc.ctypes.data) == 0
    usedc = hasc != 0
    assert usedc.mean() > 0.9 and not usedc[y == 0].any()
    errc = np.abs(coarse[usedc].astype(np.float64) - exact[usedc].astype(np.float64))
    assert errc.max() < 0.04, errc.max()                                         # 0.4 of the coarse band


@pytest.mark.parametrize("family", ["T16H5", "T25H7", "T25H9", "T36H11B1"])
def test_host_tail_decodes_every_family_on_rendered_boards(family):
    """Boards drawn with the other families' code tables and cell layouts (4x4 / 5x5 code bits, border of
    1 or 2 cells; src/detector.rs:369-405): the product's tail and the oracle's decode the same tags with
    bit-identical corners, and those are the 25 drawn ids at the drawn positions."""
    import aprilgrid_rs_amd as A
    from oracle import oracle as O
    synth = synth_module()
    for seed in (11, 12):
        img, gt = synth.render_frame(seed, 800, 600, spec=synth.BoardSpec(rows=5, cols=5), family=family)
        img = img.numpy()
        saddles = O.refined_saddle_points(img)
        got = A.TagDetector.detect_tail(family, saddles, img)
        ref = O.detect_tail(img, saddles, family=family)
        assert sorted(got) == sorted(ref) == sorted(gt), (family, seed, sorted(got), sorted(ref))
        for tid in ref:
            assert bits_equal(got[tid], ref[tid])
            err = np.abs(np.sort(got[tid], axis=0) - np.sort(gt[tid].astype(np.float32), axis=0)).max()
            assert err < 0.5, (family, tid, err)


def test_rccl_prototypes_group_cpp_binds_by_name():
    """group.cpp opens librccl with dlopen and calls seven entry points through hand-written function-pointer types
    (rccl.h is not included: the library is optional at run time).  The installed header must agree with them: the
    parameter lists, ncclUint8 == 1 (the only data type used) and ncclSuccess == 0 -- and the test suite's stand-in
    (tests/stub_rccl) must export the same names."""
    import re
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h in this image")
    h = re.sub(r"\s+", " ", open(hdr).read())
    want = {
        "ncclCommInitAll": "ncclComm_t* comm, int ndev, const int* devlist",
        "ncclCommDestroy": "ncclComm_t comm",
        "ncclGroupStart": "",
        "ncclGroupEnd": "",
        "ncclSend": "const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream",
        "ncclRecv": "void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream",
        "ncclGetErrorString": "ncclResult_t result",
    }
    for name, params in want.items():
        m = re.search(r"[\w\* ]+ %s\(([^)]*)\);" % name, h)
        assert m, name
        got = m.group(1).strip()
        assert got.replace("void", "") .strip() == params or got == params, (name, got)
    assert re.search(r"ncclUint8 = 1,", h) and re.search(r"ncclSuccess = 0,", h)
    src = open(os.path.join(ROOT, "aprilgrid-rs_amd", "csrc", "group.cpp")).read()
    stub = open(os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp")).read()
    for name in want:
        assert 'sym("%s")' % name in src, name
        assert re.search(r"\b%s\(" % name, stub), name
    assert "kNcclUint8 = 1" in src


def test_group_of_several_ranks_reports_an_unloadable_rccl_instead_of_crashing(lib):
    """load_rccl built its message from a second dlerror() call (NULL: the first consumed it) -- std::string +
    nullptr, a crash no guard catches, on any box where librccl cannot be loaded.  The library is bound before any device
    is touched, so the path runs without a GPU: a two-rank RCCL group with an unloadable library is an error + message."""
    from aprilgrid_rs_amd import _ffi
    os.environ["AGX_RCCL_LIBRARY"] = "/nonexistent/librccl_not_here.so"
    try:
        g = C.c_void_p()
        devs = (C.c_int * 2)(0, 1)
        st = lib.agx_group_create(3, None, devs, 2, _ffi.AGX_GATHER_RCCL, C.byref(g))
        assert st == _ffi.AGX_ERR_HIP and not g.value
        msg = lib.agx_group_last_error(None).decode()
        assert "dlopen librccl" in msg and "librccl_not_here" in msg, msg
        # a group of ONE never needs the library (two device-to-device copies move its slabs): with the same unloadable
        # library its creation gets as far as the device -- which this machine lacks --, not to dlopen
        devs1 = (C.c_int * 1)(0)
        st = lib.agx_group_create(3, None, devs1, 1, _ffi.AGX_GATHER_RCCL, C.byref(g))
        if st != _ffi.AGX_OK:
            assert "dlopen" not in lib.agx_group_last_error(None).decode()
        else:  # (a GPU box running the CPU suite)
            lib.agx_group_destroy(g)
    finally:
        del os.environ["AGX_RCCL_LIBRARY"]


def test_stand_in_rccl_builds_and_exports_the_bound_symbols(tmp_path):
    """tests/stub_rccl compiles here (host code only) and exports every entry point group.cpp binds."""
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this box")
    so = str(tmp_path / "librccl_stub.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    nm = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString",
                 "stub_rccl_stats"):
        assert re.search(r"\bT %s\b" % name, nm), name


_NO_UNWIND_CHILD = r"""
import ctypes as C, os, resource, sys
import numpy as np
sys.path.insert(0, %r)
from aprilgrid_rs_amd import _ffi
lib = _ffi.lib()
mode = sys.argv[1]
rng = np.random.default_rng(5)
n = 30000 if mode == "threads" else 3000000
sad = np.zeros((n, 5), np.float32)
sad[:, 0] = rng.uniform(0, 1280, n); sad[:, 1] = rng.uniform(0, 800, n); sad[:, 2] = 1.0
sad[:, 3] = rng.uniform(-90, 90, n); sad[:, 4] = 45.0
luma = np.zeros((800, 1280), np.uint8)
out = (C.c_uint8 * (36 * 64))()
n_out = C.c_uint32(0)
def vm_bytes():
    for line in open("/proc/self/status"):
        if line.startswith("VmSize:"):
            return int(line.split()[1]) * 1024
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
# threads: 64 worker threads need 64 x 8 MiB of stack address space; alloc: the saddle copy alone is 60 MB
resource.setrlimit(resource.RLIMIT_AS, (vm_bytes() + (48 << 20 if mode == "threads" else 16 << 20), hard))
rc = lib.agx_detect_tail_threads(3, None, sad.ctypes.data, n, luma.ctypes.data, 1280, 800, 1280, out, 64, C.byref(n_out),
                                 64 if mode == "threads" else 1)
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
print("RC", rc, lib.agx_status_string(rc).decode())
# the library is still usable afterwards
small = sad[:100].copy()
rc2 = lib.agx_detect_tail_threads(3, None, small.ctypes.data, 100, luma.ctypes.data, 1280, 800, 1280, out, 64, C.byref(n_out), 2)
print("RC2", rc2)
"""


@pytest.mark.parametrize("mode", ["threads", "alloc"])
def test_nothing_unwinds_across_the_boundary_when_the_host_runs_out_of_resources(mode, lib):
    """include/aprilgrid_amd.h, Conventions: a failed worker-thread creation / host allocation inside an entry point is a
    negative status (AGX_ERR_NOMEM), not an exception in the caller's frame -- the reference's failure model is a panic
    the caller can see (src/detector.rs:500), a C++ exception through extern "C" into Rust is an abort.  In a subprocess,
    with the address space capped right before the call."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _NO_UNWIND_CHILD % ROOT, mode], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "RC -8 out of host memory or threads" in r.stdout, r.stdout
    assert "RC2 0" in r.stdout, r.stdout


def test_status_codes_of_the_header_and_the_ctypes_table_agree():
    from aprilgrid_rs_amd import _ffi
    hdr = open(os.path.join(ROOT, "include", "aprilgrid_amd.h")).read()
    for name, val in re.findall(r"\b(AGX_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", hdr):
        assert getattr(_ffi, name) == int(val), name
    assert _ffi.AGX_ERR_NOMEM == -8


def test_host_parallelism_reads_the_cgroup_cpu_quota(lib, tmp_path):
    """agx_host_parallelism() = affinity mask AND cgroup CPU quota (the default thread count of agx_detect_batch: a 1-GPU box of
    this pool shows 256 CPUs and grants 16).  The quota half on made-up cgroup trees: v2 cpu.max of the process's cgroup and its
    ancestors (the smallest wins, "max" = none, fractions round up), v1 cfs files, nothing found = 0."""
    def quota(files, proc):
        root = tmp_path / ("cg%d" % quota.n)
        quota.n += 1
        for rel, text in files.items():
            f = root / rel
            f.parent.mkdir(parents=True, exist_ok=True)
            f.write_text(text)
        pc = root / "proc_self_cgroup"
        pc.parent.mkdir(parents=True, exist_ok=True)
        pc.write_text(proc)
        return lib.agx_debug_cgroup_cpu_quota(str(root).encode(), str(pc).encode())
    quota.n = 0
    assert quota({"cpu.max": "1600000 100000\n"}, "0::/\n") == 16                      # the GPU boxes of this pool
    assert quota({"cpu.max": "max 100000\n"}, "0::/\n") == 0
    assert quota({"cpu.max": "150000 100000\n"}, "0::/\n") == 2                        # 1.5 CPUs -> 2 threads
    assert quota({"cpu.max": "max 100000\n", "a/cpu.max": "800000 100000\n", "a/b/cpu.max": "max 100000\n"}, "0::/a/b\n") == 8
    assert quota({"cpu.max": "400000 100000\n", "a/b/cpu.max": "3200000 100000\n"}, "0::/a/b\n") == 4   # an ancestor's is tighter
    assert quota({"cpu/cpu.cfs_quota_us": "-1\n", "cpu/cpu.cfs_period_us": "100000\n"}, "3:cpuset:/jobs\n1:cpu:/\n0::/\n") == 0
    assert quota({"cpu/cpu.cfs_quota_us": "600000\n", "cpu/cpu.cfs_period_us": "100000\n"}, "2:cpuacct:/\n1:cpu:/\n") == 6
    assert quota({"cpu,cpuacct/k/cpu.cfs_quota_us": "250000\n", "cpu,cpuacct/k/cpu.cfs_period_us": "50000\n"}, "4:cpu,cpuacct:/k\n") == 5
    assert quota({}, "") == 0
    n = lib.agx_host_parallelism()
    assert 1 <= n <= (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())


def test_host_tail_survives_non_finite_saddles():
    """agx_detect_tail takes a caller's saddle list (agx_detect_from_saddles: any floats).  NaN / infinite coordinates and
    angles must not crash or hang the search (the grid index clamps its cells, the selection orders keys, the angle
    approximations refuse such operands); what it returns for them is unspecified."""
    import aprilgrid_rs_amd as A
    from aprilgrid_rs_amd.detector import SADDLE_DTYPE
    rng = np.random.default_rng(0)
    g = np.full((200, 300), 128, np.uint8)
    for trial in range(120):
        n = int(rng.integers(1, 400)) if trial % 10 else 1500  # (1500: beyond the brute-force 50-NN's range)
        s = np.zeros(n, SADDLE_DTYPE)
        s["x"], s["y"] = rng.uniform(0, 300, n), rng.uniform(0, 200, n)
        s["k"], s["phi"] = 1.0, 45.0
        s["theta"] = rng.choice([10.0, -80.0], n) + rng.normal(0, 1, n)
        bad = rng.integers(0, n, max(1, n // 10))
        kind = trial % 4
        if kind == 0:
            s["x"][bad] = np.nan
        elif kind == 1:
            s["y"][bad] = np.inf
        elif kind == 2:
            s["theta"][bad] = np.nan
        else:
            s["x"][bad] = -np.inf
            s["theta"][bad] = 1e30
        A.TagDetector.detect_tail("t36h11", s, g)


def test_this_libm_atan2f_is_the_routine_the_device_tail_restates():
    """csrc/libm_f32.h restates glibc's single-precision atan2f (the fdlibm routine, all binary32 operations) so that the device
    tail can evaluate the reference's angle_degree (math_util.rs:31-33) with the bits the host's libm gives.  The library refuses
    the device tail where the two differ; in this image they must not: any two floats, cross / dot products of image-sized
    vectors, ratios at the routine's interval ends, the special cases -- 2^24 pairs per seed."""
    import ctypes as C
    from aprilgrid_rs_amd import _ffi
    lib = _ffi.lib()
    for seed in (1, 2, 3):
        bad = C.c_uint64(12345)
        assert lib.agx_debug_libm_atan2f_check(1 << 24, seed, C.byref(bad)) == 0
        assert bad.value == 0, (seed, bad.value)
    assert lib.agx_debug_libm_atan2f_check(16, 1, None) != 0  # (null output: an argument error, not a crash)


def test_the_device_tails_white_block_band_covers_the_references_roundings():
    """The device tail decides the white-block test (saddle.rs:26-38) from a binary64 evaluation when the angle is farther than
    1e-4 degrees (kBandAbs, csrc/tail_kernels.hip) from 60 and 120.  That is sound if the reference's own binary32 value -- cosf,
    sinf, six roundings, atan2f, the conversion to degrees -- is never that far from the binary64 one: on ten million random
    (theta, vector) triples, and on vectors whose angle sits at the thresholds, the two differ by less than 5e-5 degrees."""
    import numpy as np
    from aprilgrid_rs_amd import _ffi
    lib = _ffi.lib()
    rng = np.random.default_rng(11)
    worst = 0.0
    for part in range(10):
        n = 1_000_000
        t = np.empty((n, 3), np.float32)
        t[:, 0] = rng.uniform(-90.0, 90.0, n)
        r = np.exp(rng.uniform(np.log(2.0), np.log(2000.0), n))
        if part % 2:  # directions at the thresholds: theta + 60 / + 120 degrees (both signs), jittered by up to a millidegree
            a = np.deg2rad(t[:, 0].astype(np.float64) + rng.choice([60.0, 120.0, -60.0, -120.0], n) + rng.uniform(-1e-3, 1e-3, n))
        else:
            a = rng.uniform(-np.pi, np.pi, n)
        t[:, 1] = (r * np.cos(a)).astype(np.float32)
        t[:, 2] = (r * np.sin(a)).astype(np.float32)
        ref = np.empty(n, np.float32)
        f64 = np.empty(n, np.float64)
        assert lib.agx_debug_white_block_angles(t.ctypes.data, n, ref.ctypes.data, f64.ctypes.data) == 0
        worst = max(worst, float(np.max(np.abs(ref.astype(np.float64) - f64))))
    assert worst < 5e-5, worst
