"""rtow_render_noalloc: the render entry point that allocates nothing (its radiance sums and counters
live in a fixed per-GPU array inside the library) against rtow_render_async, through the C ABI.
Pixels and ray-segment counts must be equal bit for bit, on two calls in a row - the second smaller,
so stale sums of the first call would show - and with a device-resident row list."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_render_equals_allocating_render():
    import torch
    from raytracinginoneweekend_amd import _lib, raytracer as R

    with R.Renderer(0) as r:
        r.setWorld(R.randomScene())
        for W, H, spp, depth, rows in ((240, 160, 6, 50, None), (96, 64, 3, 50, None), (200, 120, 4, 8, [119, 0, 57, 58, 3])):
            r.setCamera(R.bookCamera(W / H))
            n_rows = H if rows is None else len(rows)
            want = torch.zeros((n_rows, W, 3), dtype=torch.uint8, device="cuda:0")
            got = torch.full((n_rows, W, 3), 7, dtype=torch.uint8, device="cuda:0")
            rows_dev = None if rows is None else torch.tensor(rows, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()
            r.renderAsync(W, H, spp, depth, rows=rows, rgb_dev=want.data_ptr())
            r.sync()
            seg_want = r.stats()["ray_segments"]
            r.renderNoAlloc(W, H, spp, depth, got.data_ptr(), rows_dev=None if rows is None else rows_dev.data_ptr(),
                            n_rows=n_rows)
            r.sync()
            st = r.stats()
            assert st["ray_segments"] == seg_want and st["paths"] == n_rows * W * spp
            assert torch.equal(got, want), f"{W}x{H}: {(got != want).sum().item()} bytes differ"
            assert want.any()
        # the same context still renders through the allocating entry points afterwards
        r.setCamera(R.bookCamera(96 / 64))
        rgb = r.renderRows(96, 64, 3, 50)[0]
        # nothing of a no-alloc image stays in the context
        dev = torch.zeros((64, 96, 3), dtype=torch.uint8, device="cuda:0")
        r.renderNoAlloc(96, 64, 3, 50, dev.data_ptr())
        with pytest.raises(_lib.RtowError) as e:
            r.readImage()
        assert e.value.status == -3
        with pytest.raises(_lib.RtowError) as e:
            r.renderPass(96, 64, 3, 1, 50)
        assert e.value.status == -3
        r.sync()
        assert (dev.cpu().numpy() == rgb).all()
        # limits and arguments
        big = torch.zeros(8, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(_lib.RtowError) as e:      # 1025 x 1024 pixels: one row over the fixed work memory
            r.renderNoAlloc(1025, 1024, 1, 1, big.data_ptr())
        assert e.value.status == -1 and "1048576" in str(e.value)
        with pytest.raises(_lib.RtowError) as e:
            r.renderNoAlloc(96, 64, 1, 1, None)
        assert e.value.status == -1
        with pytest.raises(_lib.RtowError):
            r.renderNoAlloc(96, 64, 1, 1, big.data_ptr(), n_rows=10)   # rows_dev NULL needs n_rows == h
        with pytest.raises(_lib.RtowError) as e:      # the 64-bit sums hold 2^22 samples of a pixel, here as everywhere
            r.renderNoAlloc(96, 64, 2 ** 22 + 1, 1, dev.data_ptr())
        assert e.value.status == -1 and "4194304" in str(e.value)
    with R.Renderer(0) as fresh:
        with pytest.raises(_lib.RtowError) as e:
            fresh.renderNoAlloc(96, 64, 1, 1, big.data_ptr())
        assert e.value.status == -3


def test_noalloc_largest_frame_and_two_contexts_in_turn():
    """The largest frame the fixed work memory holds (1024 x 1024), then a second context on the same
    GPU - it shares the array - after the first has finished."""
    import torch
    from raytracinginoneweekend_amd import raytracer as R

    W = H = 1024
    with R.Renderer(0) as a, R.Renderer(0) as b:
        for r in (a, b):
            r.setWorld(R.randomScene())
            r.setCamera(R.bookCamera(1.0))
        want = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
        got = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        a.renderAsync(W, H, 2, 50, rgb_dev=want.data_ptr())
        a.sync()
        a.renderNoAlloc(W, H, 2, 50, got.data_ptr())
        a.sync()
        assert torch.equal(got, want)
        small_want = torch.zeros((64, 64, 3), dtype=torch.uint8, device="cuda:0")
        small_got = torch.zeros((64, 64, 3), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        b.renderAsync(64, 64, 5, 50, rgb_dev=small_want.data_ptr())
        b.sync()
        b.renderNoAlloc(64, 64, 5, 50, small_got.data_ptr())
        b.sync()
        assert torch.equal(small_got, small_want)


def test_noalloc_calls_allocate_no_device_memory():
    """The contract of the entry point: after set-up, a call takes no device memory.  Free device memory (as the
    HIP runtime reports it) must not move across a series of calls of different sizes, the largest frame included,
    while the allocating entry point, asked for a frame larger than any before, does take some."""
    import torch
    from raytracinginoneweekend_amd import raytracer as R

    with R.Renderer(0) as r:
        r.setWorld(R.randomScene())
        r.setCamera(R.bookCamera(1.0))
        out = torch.zeros((1024, 1024, 3), dtype=torch.uint8, device="cuda:0")
        rows_dev = torch.arange(0, 300, 3, dtype=torch.int32, device="cuda:0")
        r.renderNoAlloc(1024, 1024, 1, 4, out.data_ptr())   # warm-up at full grid: code objects, the runtime's first-use buffers
        r.sync()
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        for W, H, spp, kw in ((1024, 1024, 1, {}), (64, 64, 3, {}), (512, 300, 2, {"rows_dev": rows_dev.data_ptr(), "n_rows": 100}),
                              (1024, 1024, 2, {})):
            r.renderNoAlloc(W, H, spp, 10, out.data_ptr(), **kw)
            r.sync()
            assert torch.cuda.mem_get_info(0)[0] == free_before, (W, H)
        r.renderAsync(2048, 2048, 1, 4)                      # the allocating path: 100 MB of sums for a frame of this size
        r.sync()
        assert torch.cuda.mem_get_info(0)[0] < free_before


def test_noalloc_overlap_on_one_gpu_is_refused():
    """All contexts of a GPU share the work memory: while one context's frame is in flight (it has not been
    synchronised), a call on another context of the same GPU returns RTOW_ERR_STATE instead of racing on the sums;
    after the first context's rtow_sync the second may go, and both images are right."""
    import torch
    from raytracinginoneweekend_amd import _lib, raytracer as R

    W, H, spp = 320, 200, 8
    with R.Renderer(0) as a, R.Renderer(0) as b:
        for r in (a, b):
            r.setWorld(R.randomScene())
            r.setCamera(R.bookCamera(W / H))
        want = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
        got_a = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
        got_b = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        a.renderAsync(W, H, spp, 50, rgb_dev=want.data_ptr())
        a.sync()
        a.renderNoAlloc(W, H, spp, 50, got_a.data_ptr())
        with pytest.raises(_lib.RtowError) as e:
            b.renderNoAlloc(W, H, spp, 50, got_b.data_ptr())
        assert e.value.status == -3 and "another context" in str(e.value)
        a.renderNoAlloc(W, H, spp, 50, got_a.data_ptr())     # the owner itself may go on: its calls follow each other on its stream
        a.sync()
        b.renderNoAlloc(W, H, spp, 50, got_b.data_ptr())
        with pytest.raises(_lib.RtowError):
            a.renderNoAlloc(W, H, spp, 50, got_a.data_ptr())
        b.sync()
        assert torch.equal(got_a, want) and torch.equal(got_b, want)
    with R.Renderer(0) as c:                                 # a destroyed owner gives the memory back
        c.setWorld(R.randomScene())
        c.setCamera(R.bookCamera(W / H))
        c.renderNoAlloc(W, H, 1, 50, got_a.data_ptr())
    with R.Renderer(0) as d:
        d.setWorld(R.randomScene())
        d.setCamera(R.bookCamera(W / H))
        d.renderNoAlloc(W, H, spp, 50, got_b.data_ptr())
        d.sync()
        assert torch.equal(got_b, want)
