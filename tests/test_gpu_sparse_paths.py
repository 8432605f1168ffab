"""The sparse stages as ONE workgroup per frame (k_sparse_frame): path 2 = verify + seeds + floods + refine + emission in one
launch, path 3 = k_verify_seeds, then floods + refine + emission in one launch (what batches that fill the chip take by
themselves).  The parity tests of tests/test_gpu_parity.py that exercise the sparse stages run again with the path forced
-- single frames, ragged sizes, noise (seed lists beyond the LDS list, more than 1024 refined records), components beyond
the flood windows (generic path inside the workgroup), capacity overflow -- and the three paths must agree bit for bit."""
import numpy as np
import pytest

from tests.util import check_frame, check_saddles, load_image, synth_module
from tests.test_gpu_parity import (oracle, test_fixture_images_chain, test_ragged_sizes, test_flat_and_empty,  # noqa: F401
                                   test_batch_matches_oracle_and_is_deterministic, test_batch_other_formats,
                                   test_noise_frames_many_clusters, test_oversized_components_take_second_tier_and_generic_path,
                                   test_randomised_sizes_formats_batches, test_unaligned_width_device_batches,
                                   test_more_than_16384_saddles_per_frame, test_4k_pure_noise_frame,
                                   test_detect_batch_equals_per_frame_detect)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[2, 3], ids=["one_launch", "verify_then_one_launch"])
def det(request):
    import aprilgrid_rs_amd as A
    d = A.TagDetector(A.TagFamily.T36H11, None, device=0)
    d.set_option("sparse_path", request.param)
    yield d
    assert d.get_option("last_sparse_path") == request.param  # the forced path was the one that ran
    d.close()


def test_three_paths_agree_bit_for_bit():
    """Records, counts, status flags and cluster counts of a mixed batch (boards, noise, a frame whose components leave the
    flood windows) are identical on the three paths."""
    import torch
    import aprilgrid_rs_amd as A
    synth = synth_module()
    h, w = 240, 320
    boards, _ = synth.render_batch(11, 5, w, h, device="cpu")
    noise, _ = synth.render_batch(3, 2, w, h, device="cpu", pure_noise=True)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    big = (128 + 100 * np.sin(2 * np.pi / 150.0 * xx) * np.sin(2 * np.pi / 150.0 * yy)).astype(np.uint8)
    frames = torch.cat([boards, noise, torch.from_numpy(big)[None]]).cuda()
    out = {}
    for path in (1, 2, 3):
        d = A.TagDetector("t36h11", None, device=0)
        d.set_option("sparse_path", path)
        d.saddles_batch_enqueue(frames)
        res, status = d.saddles_batch_fetch()
        out[path] = (res, status, [d.debug_fetch(i, "counters") for i in range(len(res))])
        d.close()
    for path in (2, 3):
        assert np.array_equal(out[path][1], out[1][1])
        for i, (a, b) in enumerate(zip(out[path][0], out[1][0])):
            assert a.tobytes() == b.tobytes(), (path, i)
        for i, (ca, cb) in enumerate(zip(out[path][2], out[1][2])):
            for k in ("flags", "clusters", "refined", "saddles", "big_seeds"):
                assert ca[k] == cb[k], (path, i, k, ca, cb)


@pytest.mark.parametrize("path", [2, 3])
def test_capacity_overflow_is_reported_on_the_fused_paths(oracle, path):
    import aprilgrid_rs_amd as A
    d = A.TagDetector("T36H11", None, device=0)
    d.set_option("sparse_path", path)
    d.set_limits(max_candidates=4096, max_clusters=256, max_saddles=64)
    img = load_image("EuRoC.png")
    with pytest.raises(A.AgxError) as e:
        d.refined_saddle_points(img)
    assert e.value.status == -3
    d.set_limits(max_candidates=0, max_clusters=0, max_saddles=64)  # only the output list is too short
    with pytest.raises(A.AgxError) as e:
        d.refined_saddle_points(img)
    assert e.value.status == -3
    d.set_limits(0, 0, 0)
    check_saddles(d.refined_saddle_points(img, as_array=True), oracle.refined_saddle_points(img), "after reset")
    d.close()


def test_batch_size_selects_the_path():
    """256 frames take k_verify_seeds + k_sparse_frame by themselves, 64 frames the three launches."""
    import aprilgrid_rs_amd as A
    synth = synth_module()
    d = A.TagDetector("t36h11", None, device=0)
    for n, expect in ((64, 1), (256, 3), (300, 1)):
        frames, _ = synth.render_batch(0, 4, 320, 200, device="cuda")
        frames = frames.repeat((n // 4 + 1, 1, 1))[:n].contiguous()
        d.saddles_batch_enqueue(frames)
        d.sync()
        assert d.get_option("last_sparse_path") == expect, n
    d.close()


@pytest.mark.parametrize("fmt", ["L8", "L16", "RGB8"])
def test_k1_poll_forms_agree_bit_for_bit(monkeypatch, fmt):
    """K1 polls the frame's running minimum by an awaited scalar load (batches that fill the chip) or by an asynchronous vector
    load (few waves: plan_k1).  Both forms forced on a batch either would take: the blur plane, the frame minimum and the lists are
    identical, and a batch of each size takes the form the planner documents."""
    import torch
    import aprilgrid_rs_amd as A
    synth = synth_module()
    frames, _ = synth.render_batch(21, 6, 640, 416, device="cuda", fmt=fmt)
    got = {}
    for form in ("0", "1"):
        monkeypatch.setenv("AGX_K1_ASYNC_POLL", form)
        d = A.TagDetector("t36h11", None, device=0)
        d.saddles_batch_enqueue(frames)
        res, status = d.saddles_batch_fetch()
        assert (status == 0).all()
        got[form] = ([r.tobytes() for r in res], [float(d.debug_fetch(i, "min")) for i in range(len(res))],
                     d.debug_fetch(2, "blur", (416, 640)).tobytes())
        d.close()
    assert got["0"] == got["1"]
    monkeypatch.delenv("AGX_K1_ASYNC_POLL")
    d = A.TagDetector("t36h11", None, device=0)  # (creating a detector reads the environment again: the override is gone)
    d.close()
    torch.cuda.synchronize()
