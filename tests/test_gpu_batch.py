"""GPU parity of the batched device-pointer path (extract batch + consecutive-frame matching),
the path bench.py times, against the oracle run frame by frame."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import _lib, synth
from openvslam_amd.feature import orb_extractor, orb_params

pytestmark = pytest.mark.gpu


# the rectangle masks of the reference's tutorial configuration (docs/simple_tutorial.rst:297-301)
TUTORIAL_RECTS = [(0.0, 1.0, 0.0, 0.1), (0.0, 1.0, 0.84, 1.0), (0.0, 0.2, 0.7, 1.0), (0.8, 1.0, 0.7, 1.0)]


@pytest.mark.parametrize("w,h,n_kp,n_frames,rects", [(640, 480, 1000, 5, ()), (1241, 376, 2000, 4, ()),
                                                     # the equirectangular configuration at full size: more than 8000
                                                     # keypoint slots per frame, the resolve kernel's largest LDS plan
                                                     (1920, 960, 4000, 3, TUTORIAL_RECTS)])
def test_batch_extract_and_match(w, h, n_kp, n_frames, rects):
    dev = torch.device("cuda", 0)
    dx, dy = 3, 1
    frames = synth.make_sequence(w, h, n_frames, seed=77, dx=dx, dy=dy)
    d_frames = torch.from_numpy(frames).to(dev)
    ext = orb_extractor(orb_params(max_num_keypts=n_kp, mask_rects=rects), device=0)
    cap = ext.keypoint_capacity(w, h)
    kps = torch.zeros((n_frames, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_frames, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros((n_frames,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ext.extract_batch_dev(d_frames.data_ptr(), n_frames, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)
    reproj = torch.full((n_frames, cap, 2), -1e38, dtype=torch.float32, device=dev)
    reproj[1:] = kps[:-1, :, 0:2] - torch.tensor([float(dx), float(dy)], device=dev)
    matched = torch.zeros((n_frames, cap), dtype=torch.int32, device=dev)
    num = torch.zeros((n_frames,), dtype=torch.int32, device=dev)
    grid = A.make_grid(0.0, 0.0, float(w), float(h))
    sf = ext.get_scale_factors()
    _lib.check(_lib.lib().ovs_match_batch_current_and_last_dev(
        kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), n_frames, cap, reproj.data_ptr(), C.byref(grid),
        A.ptr(sf, A.f32p), 8, 20.0, 1, matched.data_ptr(), num.data_ptr(), 0, stream))
    torch.cuda.synchronize(dev)

    h_counts = counts.cpu().numpy()
    h_kps = kps.cpu().numpy().view(np.uint8).reshape(n_frames, cap, 28).copy().view(A.KEYPOINT_DTYPE).reshape(n_frames, cap)
    h_desc = desc.cpu().numpy()
    h_matched = matched.cpu().numpy()
    h_num = num.cpu().numpy()

    ref = oracle.OrbExtractor(max_num_keypts=n_kp, mask_rects=rects)
    r_kps = np.zeros((n_frames, cap), dtype=A.KEYPOINT_DTYPE)
    r_desc = np.zeros((n_frames, cap, 32), dtype=np.uint8)
    r_counts = np.zeros(n_frames, dtype=np.int32)
    for f in range(n_frames):
        k, d = ref.extract(frames[f])
        r_counts[f] = len(k)
        r_kps[f, :len(k)] = k
        r_desc[f, :len(k)] = d
        assert h_counts[f] == len(k)
        assert np.array_equal(h_kps[f, :len(k)].view(np.uint8), k.view(np.uint8)), "frame %d keypoints differ" % f
        assert np.array_equal(h_desc[f, :len(k)], d), "frame %d descriptors differ" % f
    r_reproj = np.full((n_frames, cap, 2), -1e38, dtype=np.float32)
    r_reproj[1:, :, 0] = r_kps["x"][:-1] - np.float32(dx)
    r_reproj[1:, :, 1] = r_kps["y"][:-1] - np.float32(dy)
    r_matched = np.zeros((n_frames, cap), dtype=np.int32)
    r_num = np.zeros(n_frames, dtype=np.uint32)
    oracle.check(oracle.lib().ovo_match_batch_current_and_last(
        A.ptr(r_kps, A.kpp), A.ptr(r_desc, A.u8p), A.ptr(r_counts, A.i32p), n_frames, cap, A.ptr(r_reproj, A.f32p),
        C.byref(grid), A.ptr(sf, A.f32p), 8, 20.0, 1, A.ptr(r_matched, A.i32p), A.ptr(r_num, A.u32p)))
    assert np.array_equal(h_num.astype(np.uint32), r_num)
    for f in range(n_frames):
        n = r_counts[f]
        assert np.array_equal(h_matched[f, :n], r_matched[f, :n]), "frame %d matches differ" % f
    assert r_num[1:].min() > n_kp // 10
    if rects:
        # nothing inside the masked bands (test/openvslam/feature/orb_extractor.cc:212-217)
        k = h_kps[0, :h_counts[0]]
        assert k["y"].min() >= 0.1 * h - 1 and k["y"].max() <= 0.84 * h + 1


def test_more_frames_than_one_grid_dimension_holds():
    """The FAST kernel takes the frame from grid y, which is limited to 65535: a larger batch is cut into
    several launches.  65 540 small frames (16 distinct images, repeated); the frames around the cut and
    the last one are compared with the oracle, every repetition of an image with its first occurrence."""
    dev = torch.device("cuda", 0)
    w = h = 72
    n_frames, distinct = 65540, 16
    base = synth.make_sequence(w, h, distinct, seed=41, dx=5, dy=3)
    frames = torch.from_numpy(base).to(dev).repeat((n_frames + distinct - 1) // distinct, 1, 1)[:n_frames].contiguous()
    params = dict(max_num_keypts=60, num_levels=2, ini_fast_thr=12, min_fast_thr=5)
    ext = orb_extractor(orb_params(**params), device=0)
    cap = ext.keypoint_capacity(w, h)
    kps = torch.zeros((n_frames, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_frames, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros((n_frames,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ext.extract_batch_dev(frames.data_ptr(), n_frames, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, stream)
    torch.cuda.synchronize(dev)
    ref = oracle.OrbExtractor(**params)
    for f in (0, 1, 65534, 65535, 65536, n_frames - 1):
        k, d = ref.extract(base[f % distinct])
        assert len(k) > 5
        assert int(counts[f]) == len(k), f
        got = kps[f, :len(k)].cpu().numpy().view(np.uint8).reshape(len(k), 28)
        assert np.array_equal(got, k.view(np.uint8).reshape(len(k), 28)), "frame %d keypoints differ" % f
        assert np.array_equal(desc[f, :len(k)].cpu().numpy(), d), "frame %d descriptors differ" % f
    first = counts[:distinct]
    assert torch.equal(counts, first.repeat((n_frames + distinct - 1) // distinct)[:n_frames])
    rep = kps.view(torch.int32)[:distinct].repeat((n_frames + distinct - 1) // distinct, 1, 1)[:n_frames]
    assert torch.equal(kps.view(torch.int32), rep)


@pytest.mark.parametrize("w,h,base_off,pad", [(641, 479, 1, 0), (1241, 376, 3, 5), (322, 247, 2, 1), (640, 480, 0, 0)])
def test_unaligned_caller_images(w, h, base_off, pad):
    """Level-0 images that start at any byte, with any row stride, and end exactly at the end of
    their allocation: the kernels' aligned-dword and buffer-bounded loads must see every pixel of
    the last rows and columns (compared plane by plane and keypoint by keypoint)."""
    dev = torch.device("cuda", 0)
    n_frames = 2
    frames = synth.make_sequence(w, h, n_frames, seed=5 + w, dx=2, dy=1)
    stride = w + pad
    frame_stride = stride * h + 7
    # one exactly sized allocation: the last image ends at its last byte
    total = base_off + frame_stride * (n_frames - 1) + stride * (h - 1) + w
    buf = torch.zeros(total, dtype=torch.uint8, device=dev)
    host = np.zeros(total, dtype=np.uint8)
    for f in range(n_frames):
        for y in range(h):
            o = base_off + f * frame_stride + y * stride
            host[o:o + w] = frames[f, y]
    buf.copy_(torch.from_numpy(host))
    ext = orb_extractor(orb_params(max_num_keypts=800), device=0)
    cap = ext.keypoint_capacity(w, h)
    kps = torch.zeros((n_frames, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_frames, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros((n_frames,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ext.extract_batch_dev(buf.data_ptr() + base_off, n_frames, w, h, stride, frame_stride, kps.data_ptr(), desc.data_ptr(),
                          counts.data_ptr(), cap, stream)
    torch.cuda.synchronize(dev)
    h_counts = counts.cpu().numpy()
    h_kps = kps.cpu().numpy().view(np.uint8).reshape(n_frames, cap, 28).copy().view(A.KEYPOINT_DTYPE).reshape(n_frames, cap)
    h_desc = desc.cpu().numpy()
    ref = oracle.OrbExtractor(max_num_keypts=800)
    for f in range(n_frames):
        k, d = ref.extract(frames[f])
        for lvl in range(1, 8):
            assert np.array_equal(ext.image_pyramid(lvl, frame=f), ref.pyramid_level(lvl)), "frame %d pyramid level %d" % (f, lvl)
        for lvl in range(8):
            want = ref.blurred_level(lvl)
            if want is not None:
                assert np.array_equal(ext.blurred_level(lvl, frame=f), want), "frame %d blurred level %d" % (f, lvl)
        assert h_counts[f] == len(k)
        assert np.array_equal(h_kps[f, :len(k)].view(np.uint8), k.view(np.uint8))
        assert np.array_equal(h_desc[f, :len(k)], d)


def test_two_streams_in_flight_share_no_scratch():
    """Two sub-batches queued on two HIP streams from one host thread (two extractor handles, the
    matcher's per-stream scratch) give the results of the same work done on one stream."""
    dev = torch.device("cuda", 0)
    w, h, n_frames = 640, 480, 9
    frames = torch.from_numpy(synth.make_sequence(w, h, n_frames, seed=91, dx=3, dy=1)).to(dev)
    grid = A.make_grid(0.0, 0.0, float(w), float(h))

    def run(first, count, stream, rounds):
        # everything (allocation, zero fill, kernels) is ordered on `stream`
        with torch.cuda.stream(stream):
            ext = orb_extractor(orb_params(max_num_keypts=1000), device=0)
            cap = ext.keypoint_capacity(w, h)
            kps = torch.zeros((count, cap, 7), dtype=torch.float32, device=dev)
            desc = torch.zeros((count, cap, 32), dtype=torch.uint8, device=dev)
            counts = torch.zeros((count,), dtype=torch.int32, device=dev)
            reproj = torch.full((count, cap, 2), -1e38, dtype=torch.float32, device=dev)
            matched = torch.zeros((count, cap), dtype=torch.int32, device=dev)
            num = torch.zeros((count,), dtype=torch.int32, device=dev)
            sf = ext.get_scale_factors()
            shift = torch.tensor([3.0, 1.0], device=dev)
            for _ in range(rounds):
                st = stream.cuda_stream
                ext.extract_batch_dev(frames[first:first + count].data_ptr(), count, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(),
                                      counts.data_ptr(), cap, st)
                torch.sub(kps[:-1, :, 0:2], shift, out=reproj[1:])
                _lib.check(_lib.lib().ovs_match_batch_current_and_last_dev(
                    kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), count, cap, reproj.data_ptr(), C.byref(grid),
                    A.ptr(sf, A.f32p), 8, 20.0, 1, matched.data_ptr(), num.data_ptr(), 0, st))
        return ext, sf, (kps, desc, counts, matched, num), (reproj, shift)

    torch.cuda.synchronize(dev)
    s0, s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    # interleaved submission: both streams have several rounds queued at once; nothing is freed
    # before the final synchronisation
    alive = []
    for _ in range(4):
        ra = run(0, 5, s1, 2)
        rb = run(4, 5, s2, 2)
        alive += [ra, rb]
    torch.cuda.synchronize(dev)
    ref_a = run(0, 5, s0, 1)
    torch.cuda.synchronize(dev)
    ref_b = run(4, 5, s0, 1)
    torch.cuda.synchronize(dev)
    names = ("keypoints", "descriptors", "counts", "matched", "num_matches")
    for which, got, want in (("a", ra, ref_a), ("b", rb, ref_b)):
        for name, g, r in zip(names, got[2], want[2]):
            if g.dtype == torch.float32:  # class_id = -1 reads as NaN: compare bit patterns
                g, r = g.view(torch.int32), r.view(torch.int32)
            assert torch.equal(g, r), "sub-batch %s: %s differ (%d elements)" % (which, name, int((g != r).sum()))
    assert int(ra[2][4][1:].min()) > 100


def test_pack_batch_results_for_the_trip_to_the_host():
    """ovs_pack_batch_results_dev: the first counts[f] records of every frame of capacity-strided arrays end
    up at offsets[f] of dense arrays (the exactly sized containers the reference returns,
    orb_extractor.cc:116-120); counts above the capacity are clamped, empty frames take no room, more frames
    than the scan's workgroup has threads."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    n_frames, cap = 1500, 300
    counts = rng.integers(0, cap + 1, n_frames).astype(np.int32)
    counts[5], counts[6], counts[1499] = 0, cap + 50, cap
    kps = rng.integers(0, 2 ** 31, (n_frames, cap, 7), dtype=np.int32)
    desc = rng.integers(0, 256, (n_frames, cap, 32), dtype=np.uint8)
    matched = rng.integers(-2, cap, (n_frames, cap), dtype=np.int32)
    d = {k: torch.from_numpy(v).to(dev) for k, v in (("counts", counts), ("kps", kps), ("desc", desc), ("matched", matched))}
    off = torch.zeros(n_frames + 1, dtype=torch.int32, device=dev)
    o_kps = torch.zeros((n_frames * cap, 7), dtype=torch.int32, device=dev)
    o_desc = torch.zeros((n_frames * cap, 32), dtype=torch.uint8, device=dev)
    o_matched = torch.zeros(n_frames * cap, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().ovs_pack_batch_results_dev(d["counts"].data_ptr(), n_frames, cap, d["kps"].data_ptr(), d["desc"].data_ptr(), d["matched"].data_ptr(),
                                                     off.data_ptr(), o_kps.data_ptr(), o_desc.data_ptr(), o_matched.data_ptr(), 0,
                                                     torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    c = np.minimum(counts, cap)
    want_off = np.concatenate([[0], np.cumsum(c)]).astype(np.int32)
    assert np.array_equal(off.cpu().numpy(), want_off)
    total = int(want_off[-1])
    keep = np.arange(cap)[None, :] < c[:, None]
    assert np.array_equal(o_kps.cpu().numpy()[:total], kps[keep])
    assert np.array_equal(o_desc.cpu().numpy()[:total], desc[keep])
    assert np.array_equal(o_matched.cpu().numpy()[:total], matched[keep])
    # descriptors only
    off2 = torch.zeros_like(off)
    o_desc2 = torch.zeros_like(o_desc)
    _lib.check(_lib.lib().ovs_pack_batch_results_dev(d["counts"].data_ptr(), n_frames, cap, None, d["desc"].data_ptr(), None, off2.data_ptr(), None,
                                                     o_desc2.data_ptr(), None, 0, torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    assert torch.equal(off2, off) and torch.equal(o_desc2[:total], o_desc[:total])
