"""GPU: the stage hand-overs of the pipeline (lightness plane from the denoiser to the bilateral stage, image
metrics fed through a MetricsAccumulator) give the SAME results as the separate wrapper calls they replace
(reference pipeline/image_processor.py:257-300 runs the separate calls)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('size', [(192, 256), (250, 334), (1024, 1536)])
def test_luminance_handover_and_metrics_accumulator_equal_the_chain(td, dev, dtype, size):
    from torch_darktable.synthetic import synthetic_rgb

    h, w = size
    rgb = synthetic_rgb(h, w, seed=61, device=dev).to(dtype)
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bil = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)

    # the chain of separate calls
    den = wiener.process_log_luminance(rgb, 0.075)
    loc = bil.process_rgb(den, 0.4)
    metrics = td.compute_image_metrics([loc], stride=8)

    # with hand-overs
    lum = torch.empty((h, w), dtype=torch.float32, device=dev)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)
    den2 = wiener.process_log_luminance(rgb, 0.075, luminance_out=lum)
    assert torch.equal(den2, den)
    assert torch.equal(lum, td.extension.extension._extract_luminance(den, False, 1e-6, torch.float32))   # == compute_luminance(den) as fp32
    loc2 = bil.process_rgb(den2, 0.4, luminance=lum, metrics=acc)
    assert torch.equal(loc2, loc)
    m2 = acc.finish()
    assert torch.allclose(m2, metrics, rtol=2e-5, atol=1e-7)      # float atomics: order of the sums differs
    assert torch.count_nonzero(acc.acc).item() == 0               # the accumulator is clean for the next frame
    # a second frame through the same accumulator
    loc3 = bil.process_rgb(den2, 0.4, luminance=lum, metrics=acc)
    assert torch.equal(loc3, loc) and torch.allclose(acc.finish(), metrics, rtol=2e-5, atol=1e-7)


def test_metrics_accumulator_equals_compute_image_metrics(td, dev):
    from torch_darktable.synthetic import synthetic_rgb

    imgs = [synthetic_rgb(96, 131, seed=70 + i, device=dev) * 1.2 for i in range(3)]
    acc = td.tonemap.MetricsAccumulator(dev, stride=4, min_gray=1e-3)
    for im in imgs:
        acc.add(im)
    assert torch.allclose(acc.finish(), td.compute_image_metrics(imgs, stride=4, min_gray=1e-3), rtol=2e-5, atol=1e-7)


def test_metrics_accumulator_on_the_grid_path(td, dev):
    """sigma_s = 8 uses the four-kernel grid path; the accumulator is fed the same way."""
    from torch_darktable.synthetic import synthetic_rgb

    rgb = synthetic_rgb(160, 224, seed=5, device=dev)
    bil = td.Bilateral(dev, (224, 160), sigma_s=8.0, sigma_r=0.1)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)
    out = bil.process_rgb(rgb, 0.3, metrics=acc)
    assert torch.equal(out, bil.process_rgb(rgb, 0.3))
    assert torch.allclose(acc.finish(), td.compute_image_metrics([out], stride=8), rtol=2e-5, atol=1e-7)


def test_workspace_scratch_is_per_stream(td, dev):
    """One workspace object used from two streams: each stream gets its own scratch (slabs / planes), results agree."""
    from torch_darktable.synthetic import synthetic_rgb

    rgb = synthetic_rgb(256, 320, seed=9, device=dev)
    wiener = td.Wiener(dev, (320, 256))
    ref = wiener.process_log_luminance(rgb, 0.05)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs = []
    for s in (s1, s2, s1, s2):
        with torch.cuda.stream(s):
            outs.append(wiener.process_log_luminance(rgb, 0.05))
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    assert len(wiener._wiener._scratch) == 3


def test_batch_on_two_streams_equals_back_to_back(td, dev):
    """sharding.FrameStreams (used by bench.py) issues the frames of a batch round-robin on two HIP streams, each with
    its own chain object (op workspaces, accumulators), so that kernels of neighbouring frames overlap: every frame must
    come out bit-identical to the same chain run back to back on one stream."""
    import bench
    from torch_darktable.sharding import FrameStreams
    from torch_darktable.synthetic import synthetic_bayer

    w, h, frames = 1536, 1024, 6
    inputs = [synthetic_bayer(h, w, seed=900 + i, device=dev).to(torch.float16) for i in range(frames)]
    make = lambda: bench.build_pipeline(td, dev, w, h, 'f16', 'isp')[1]
    serial = make()
    ref = [serial(b).clone() for b in inputs]
    for nstreams in (2, 3):  # 3 = the bench's setting: the rotation over the streams carries over from batch to batch (6 frames, then 5)
        runner = FrameStreams(dev, make, streams=nstreams)
        for rep in range(3):   # repeated: a race would not hit every time
            batch = inputs if rep != 1 else inputs[:5]
            outs = runner.run(batch)          # joined: usable on the current stream
            for i in range(len(batch)):
                # (the serial reference runs rs::rcd_stream, the streamed frames the register-blocked strips: TDK_RCD_CONCURRENT)
                assert torch.equal(outs[i], ref[i]), f'frame {i} differs between the {nstreams}-stream and the one-stream run'
    one = FrameStreams(dev, make, streams=1).run(inputs)
    assert all(torch.equal(a, b) for a, b in zip(one, ref))
    # which RCD strips variant ran: the register-blocked one only when the frames have company (several frames in a batch, or an
    # earlier batch not yet joined); a lone frame takes the stand-alone kernel, which is the faster one with the GPU to itself
    from torch_darktable import _native

    runner = FrameStreams(dev, make, streams=3)
    for batch, want in ((inputs[:3], 'tdk_rcd(concurrent)'), (inputs[:1], 'tdk_rcd')):
        torch.cuda.synchronize()
        _native.profile_enable(True)
        runner.run(batch)
        torch.cuda.synchronize()
        names = set(_native.profile_report())
        _native.profile_enable(False)
        assert want in names and ({'tdk_rcd', 'tdk_rcd(concurrent)'} - {want}).isdisjoint(names), (len(batch), names)
    _native.profile_enable(True)
    runner.issue(inputs[:1])  # not joined ...
    runner.issue(inputs[1:2])  # ... so this single frame has company
    torch.cuda.synchronize()
    names = set(_native.profile_report())
    _native.profile_enable(False)
    runner.join()
    assert {'tdk_rcd', 'tdk_rcd(concurrent)'} <= names, names


def test_concurrent_frames_context(td, dev):
    """concurrent_frames() sets TDK_RCD_CONCURRENT for the calling thread only and restores what was there; RCD.process under it
    gives the bits of the default call (the register-blocked strips against rs::rcd_stream) -- and the library refuses flag bits
    it does not know."""
    import ctypes as C
    import threading

    from torch_darktable import torch_darktable_extension as ext
    from torch_darktable._native import lib
    from torch_darktable.synthetic import synthetic_bayer

    seen = {}
    with ext.concurrent_frames():
        t = threading.Thread(target=lambda: seen.setdefault('other', getattr(ext._verify, 'concurrent', 0)))
        t.start()
        t.join()
        seen['mine'] = ext._verify.concurrent
        with ext.concurrent_frames(False):
            seen['inner'] = ext._verify.concurrent
        seen['back'] = ext._verify.concurrent
    assert seen == {'other': 0, 'mine': ext.TDK_RCD_CONCURRENT, 'inner': 0, 'back': ext.TDK_RCD_CONCURRENT}
    assert getattr(ext._verify, 'concurrent', 0) == 0
    w, h = 1200, 700
    for dt in (torch.float32, torch.float16):
        bayer = synthetic_bayer(h, w, seed=5, device=dev).to(dt)
        ws = td.RCD(dev, (w, h), td.BayerPattern.GRBG)
        plain = ws.process(bayer).clone()
        with ext.concurrent_frames():
            quad = ws.process(bayer).clone()
        assert torch.equal(plain, quad)  # (float16: both strip kernels in the approximate flavour)
        with ext.verification_paths(rcd_exact=True):  # the tile kernel is always the exact flavour
            exact = ws.process(bayer).clone()
            with ext.concurrent_frames(), ext.verification_paths(rcd_tiles=True, rcd_exact=True):  # both flags at once: the tile kernel wins (it serves any frame)
                tiles = ws.process(bayer).clone()
        assert torch.equal(exact, tiles)
    out = torch.empty(h, w, 3, device=dev)
    b32 = synthetic_bayer(h, w, seed=5, device=dev)
    rc = lib.tdk_rcd_ex(C.c_void_p(b32.data_ptr()), C.c_void_p(out.data_ptr()), None, w, h, C.c_uint32(0x94949494), 0, 8, None)
    assert rc != 0 and b'unknown flags' in lib.tdk_last_error()


def test_metrics_one_launch_equals_two_launches(td, dev):
    """tdk_image_metrics (sums + finish by the workgroup that draws the last ticket, ONE launch) gives the bits of
    tdk_image_metrics_accumulate_rows + tdk_image_metrics_finish_reset (two stream-ordered launches), leaves the accumulator
    (ticket included) zero, and keeps doing so frame after frame on the same accumulator; a list of images = accumulate the
    first ones, fuse the last."""
    import ctypes as C

    from torch_darktable._native import lib
    from torch_darktable.synthetic import synthetic_rgb

    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bounds = torch.tensor([0.0, 1.0], device=dev)
    acc1 = torch.zeros(8192, device=dev)
    acc2 = torch.zeros(8192, device=dev)
    imgs = [synthetic_rgb(h, w, seed=s, device=dev).to(dt) for (h, w, s, dt) in
            [(3072, 4096, 5, torch.float16), (250, 334, 6, torch.float32), (64, 64, 7, torch.float32), (1500, 2100, 8, torch.float16)]]
    p = lambda t: C.c_void_p(t.data_ptr())
    tag = lambda t: 1 if t.dtype == torch.float16 else 0
    for rep in range(3):
        for x in imgs:
            m1, m2 = torch.empty(5, device=dev), torch.empty(5, device=dev)
            assert lib.tdk_image_metrics(p(x), x.size(1), x.size(0), 8, 1e-4, p(bounds), p(acc1), p(m1), tag(x), stream) == 0
            assert lib.tdk_image_metrics_accumulate_rows(p(x), x.size(1), x.size(0), 8, 1e-4, p(bounds), p(acc2), tag(x), stream) == 0
            assert lib.tdk_image_metrics_finish_reset(p(acc2), p(m2), stream) == 0
            assert torch.equal(m1, m2), (rep, tuple(x.shape), m1, m2)
            assert not acc1.any() and not acc2.any()  # rows and ticket back to zero
    # a list: the first images accumulate, the last launch finishes
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)
    for x in imgs[1:]:
        acc.add(x)
    got = acc.finish()
    assert torch.equal(got, td.compute_image_metrics(imgs[1:], stride=8))
    mlist = torch.empty(5, device=dev)
    for x in imgs[1:]:
        assert lib.tdk_image_metrics_accumulate_rows(p(x), x.size(1), x.size(0), 8, 1e-4, p(bounds), p(acc2), tag(x), stream) == 0
    assert lib.tdk_image_metrics_finish_reset(p(acc2), p(mlist), stream) == 0
    assert torch.equal(got, mlist)
    # an accumulator nobody added to: the metrics of an empty list
    assert torch.equal(acc.finish(), torch.zeros(5, device=dev))


def test_metrics_one_launch_on_many_grids(td, dev):
    """The run-time side of tests/test_isa_contract.py: the fence-less last-ticket hand-off of tdk_image_metrics against the
    two-launch form on grids from ONE workgroup to the 4 096-workgroup cap (more workgroups than accumulator rows: rows shared),
    repeated back to back on one accumulator -- a stale row, a lost add or a ticket left behind shows as a mismatch or a
    non-zero accumulator."""
    import ctypes as C

    from torch_darktable._native import lib

    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bounds = torch.tensor([0.0, 1.0], device=dev)
    acc1, acc2 = torch.zeros(8192, device=dev), torch.zeros(8192, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    g = torch.Generator(device=dev).manual_seed(11)
    # grid = ceil(samples / 512), samples = ceil(w / stride) * ceil(h / stride)
    shapes = [(16, 16, 1), (23, 22, 1), (64, 64, 2), (100, 300, 1), (512, 384, 1), (700, 999, 1), (1024, 1024, 1), (1536, 2048, 2),
              (1024, 2048, 1), (2048, 2048, 1), (3072, 4096, 8), (3072, 4096, 3)]
    grids = set()
    for rep in range(2):
        for (h, w, stride) in shapes:
            x = torch.rand(h, w, 3, generator=g, device=dev) * 1.05
            grid = min(4096, -(-(-(-w // stride) * -(-h // stride)) // 512))
            grids.add(grid)
            for _ in range(3):
                m1, m2 = torch.empty(5, device=dev), torch.empty(5, device=dev)
                assert lib.tdk_image_metrics(p(x), w, h, stride, 1e-4, p(bounds), p(acc1), p(m1), 0, stream) == 0
                assert lib.tdk_image_metrics_accumulate_rows(p(x), w, h, stride, 1e-4, p(bounds), p(acc2), 0, stream) == 0
                assert lib.tdk_image_metrics_finish_reset(p(acc2), p(m2), stream) == 0
                # up to two workgroups per accumulator row add in either order to the same bits; three or more may not
                same = torch.equal(m1, m2) if grid <= 2048 else torch.allclose(m1, m2, rtol=2e-6, atol=0)
                assert same, ((h, w, stride), m1, m2)
            assert not acc1.any() and not acc2.any()
    assert min(grids) == 1 and max(grids) == 4096 and len(grids) >= 9, sorted(grids)


def test_bilateral_prepared_workspace_and_flags(td, dev):
    """The C ABI of the bilateral tile kernel: tdk_bilateral_prepare once + TDK_BILATERAL_PREPARED on every call == the plain
    entry points (which build the axis tables themselves) == the four-kernel path (TDK_BILATERAL_GENERAL_PATH), for the plane
    and the RGB layout sharing ONE prepared workspace, across different `detail` values; unknown flags are refused."""
    import ctypes as C

    from torch_darktable._native import lib

    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    g = torch.Generator(device=dev).manual_seed(3)
    for (w, h, ss, sr) in [(334, 250, 2.0, 0.2), (1000, 96, 3.3, 0.1), (128, 64, 1.0, 0.25)]:
        lum = torch.rand(h, w, generator=g, device=dev)
        rgb = torch.rand(h, w, 3, generator=g, device=dev)
        nbytes = max(lib.tdk_bilateral_workspace_bytes(w, h, ss, sr), lib.tdk_bilateral_rgb_workspace_bytes(w, h, ss, sr))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws.fill_(0xFF)  # garbage: nothing may depend on what the workspace held
        assert lib.tdk_bilateral_prepare(p(ws), w, h, ss, sr, stream) == 0
        for detail in (0.4, -0.3):
            outs = []
            for flags, wsp in ((1, ws), (0, ws2), (2, ws2)):
                o = torch.empty_like(lum)
                assert lib.tdk_bilateral_ex(p(lum), p(o), p(wsp), w, h, ss, sr, detail, 0, flags, stream) == 0
                outs.append(o)
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (w, h, ss, sr, detail)
            o_plain = torch.empty_like(lum)
            assert lib.tdk_bilateral(p(lum), p(o_plain), p(ws2), w, h, ss, sr, detail, 0, stream) == 0
            assert torch.equal(o_plain, outs[0])
            routs = []
            for flags, wsp in ((1, ws), (0, ws2), (2, ws2)):
                o = torch.empty_like(rgb)
                assert lib.tdk_bilateral_rgb_ex(p(rgb), p(None), p(o), p(wsp), w, h, ss, sr, detail, 0, 1e-6, 0, flags, stream) == 0
                routs.append(o)
            assert torch.equal(routs[0], routs[1]) and torch.equal(routs[0], routs[2]), (w, h, ss, sr, detail)
        o = torch.empty_like(lum)
        assert lib.tdk_bilateral_ex(p(lum), p(o), p(ws), w, h, ss, sr, 0.4, 0, 8, stream) != 0
        assert b'unknown flags' in lib.tdk_last_error()


def test_verification_paths_are_thread_local(td, dev):
    """verification_paths() selects the second kernel path of RCD / Bilateral for the calling thread only (the library takes
    the path per call: tdk_rcd_ex / tdk_bilateral_ex flags); another thread keeps the default meanwhile."""
    import threading

    from torch_darktable import torch_darktable_extension as ext

    seen = {}

    def other():
        seen['other'] = (getattr(ext._verify, 'rcd', 0), getattr(ext._verify, 'bil', 0))

    with ext.verification_paths(rcd_tiles=True, bilateral_general=True):
        t = threading.Thread(target=other)
        t.start()
        t.join()
        seen['mine'] = (ext._verify.rcd, ext._verify.bil)
    assert seen['other'] == (0, 0) and seen['mine'] == (ext.TDK_RCD_TILE_KERNEL, ext.TDK_BILATERAL_GENERAL_PATH)
    assert (ext._verify.rcd, ext._verify.bil) == (0, 0)


def _isp_chain(td, dev, w, h, dtype):
    rcd = td.RCD(dev, (w, h), td.BayerPattern.RGGB)
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bil = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    params = td.TonemapParameters(gamma=0.75, intensity=2.0, light_adapt=1.0, vibrance=0.0)
    lum = torch.empty((h, w), dtype=torch.float32, device=dev)
    ab = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)

    def frame(bayer):
        rgb = rcd.process(bayer)
        wiener.process_log_luminance_lab(rgb, 0.075, luminance_out=lum, chroma_out=ab)
        rgb = bil.process_lab(lum, ab, 0.4, out_dtype=dtype, metrics=acc)
        return td.reinhard_tonemap(rgb, acc.finish(), params)

    return frame


def test_frame_chain_replays_as_a_hip_graph(td, dev):
    """DESIGN.md 2 says nothing allocates or synchronises inside an op, so a frame's whole chain (RCD -> Wiener -> bilateral + metrics
    -> tone map; 7 launches) can be captured into a HIP graph: capture it, refill the static input with OTHER frames, replay, and
    get the bits of the eager chain -- the metrics accumulator's ticket and rows included (they must come back to zero inside the
    graph for the next replay)."""
    from torch_darktable.synthetic import synthetic_bayer

    h, w = 512, 768
    frames = [synthetic_bayer(h, w, seed=80 + i, device=dev).half() for i in range(3)]
    chain = _isp_chain(td, dev, w, h, torch.float16)
    eager = [chain(f).clone() for f in frames]
    static_in = frames[0].clone()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):  # warm-up on the capture's side stream: workspaces are cached per (object, stream)
        for _ in range(2):
            chain(static_in)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        static_out = chain(static_in)
    for rep in range(2):
        for f, want in zip(frames, eager):
            static_in.copy_(f)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(static_out, want), (rep, (static_out != want).sum().item())


def test_frame_streams_batch_replays_as_one_graph(td, dev):
    """FrameStreams.capture: a batch of frames on three streams, fork and join included, as ONE graph; replays on refilled inputs
    give the bits of the eager batch."""
    from torch_darktable.sharding import FrameStreams
    from torch_darktable.synthetic import synthetic_bayer

    h, w = 384, 512
    runner = FrameStreams(dev, lambda: _isp_chain(td, dev, w, h, torch.float16), streams=3)
    batch_a = [synthetic_bayer(h, w, seed=90 + i, device=dev).half() for i in range(5)]
    batch_b = [synthetic_bayer(h, w, seed=190 + i, device=dev).half() for i in range(5)]
    want_a = [o.clone() for o in runner.run(batch_a)]
    want_b = [o.clone() for o in runner.run(batch_b)]
    static = [f.clone() for f in batch_a]
    cap = runner.capture(static)
    for rep in range(2):
        for batch, want in ((batch_b, want_b), (batch_a, want_a)):
            outs = cap.replay(batch)
            torch.cuda.synchronize()
            for i, (o, e) in enumerate(zip(outs, want)):
                assert torch.equal(o, e), (rep, i, (o != e).sum().item())


def test_bounds_one_launch_equals_init_plus_accumulate(td, dev):
    """compute_image_bounds through tdk_image_bounds (no init launch, a persistent state that is idle between calls, the last ticket of
    the list writes the result) == the reference-shaped tdk_image_bounds_init + _accumulate calls: lists of 1 - 3 images, grids of 1 to
    256 workgroups, both storage types, negative and large values, back-to-back calls on the same cached state."""
    from torch_darktable._native import lib
    from torch_darktable.torch_darktable_extension import _dtype_tag, _ptr, _stream, check

    def two_call(images, stride):
        b = torch.empty(2, dtype=torch.float32, device=dev)
        check(lib.tdk_image_bounds_init(_ptr(b), _stream()))
        for x in images:
            check(lib.tdk_image_bounds_accumulate(_ptr(x), x.size(1), x.size(0), stride, _ptr(b), _dtype_tag(x), _stream()))
        return b

    g = torch.Generator(device='cpu').manual_seed(5)
    for rep in range(3):   # the cached state must come back idle every time
        for shapes, stride in ((((8, 8),), 8), (((96, 128),), 2), (((512, 768), (100, 60)), 1), (((3072, 4096),), 8), (((40, 40), (64, 64), (700, 900)), 4)):
            for dtype in (torch.float32, torch.float16):
                images = [((torch.rand(h, w, 3, generator=g) - 0.3) * (1.0 + 7.0 * k)).to(dtype).to(dev) for k, (h, w) in enumerate(shapes)]
                got = td.compute_image_bounds(images, stride)
                want = two_call(images, stride)
                assert torch.equal(got, want), (shapes, stride, dtype, got, want)
    # a rejected list (second image on the wrong device / of the wrong type) leaves the state idle: the next call is right
    good = (torch.rand(64, 64, 3, generator=g) * 2.0 - 0.5).to(dev)
    with pytest.raises(RuntimeError):
        td.compute_image_bounds([good, good.to(torch.int32)], 4)
    assert torch.equal(td.compute_image_bounds([good], 4), two_call([good], 4))
