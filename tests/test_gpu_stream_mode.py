"""Stream mode (rife_hip_frame_upload / rife_hip_process_frames / rife_hip_frame_release): frames uploaded once
and shared by consecutive pairs give exactly the pixels of the host-buffer call, for every model family and mode."""
import importlib
import threading

import numpy as np
import pytest

from tools import gen_frames

pytestmark = pytest.mark.gpu
amd = importlib.import_module("rife-ncnn-vulkan_amd")

CASES = [("rife-v4.6", {}), ("rife-v4.6", dict(tta_mode=True, tta_temporal_mode=True)), ("rife-v4", {}), ("rife-v2.3", {}),
         ("rife-v2.3", dict(uhd_mode=True)), ("rife-v3.1", {}), ("rife", {}), ("rife-HD", {})]


def _engine(modeldirs, fam, kw):
    fl = dict(kw, rife_v2=fam.startswith(("rife-v2", "rife-v3")), rife_v4=fam.startswith("rife-v4"))
    g = amd.RIFE(0, **fl)
    g.load(modeldirs[fam])
    return g


@pytest.mark.parametrize("fam,kw", CASES)
def test_sequence_through_resident_frames_equals_host_calls(modeldirs, fam, kw):
    g = _engine(modeldirs, fam, kw)
    frames = [gen_frames.smooth_pair(192, 128, 300 + i)[0] for i in range(4)]
    res = [g.upload(f) for f in frames]
    for i in range(3):
        for t in (0.5, 0.25):
            want = g.process(frames[i], frames[i + 1], t)
            got = g.process_frames(res[i], res[i + 1], t)
            assert np.array_equal(got, want), (fam, i, t)
    # the reference's early-outs (rife.cpp:2470-2480)
    assert np.array_equal(g.process_frames(res[0], res[1], 0.0), frames[0])
    assert np.array_equal(g.process_frames(res[0], res[1], 1.0), frames[1])
    for r in res:
        r.release()


def test_resident_frames_shared_by_concurrent_calls(modeldirs):
    """One frame is the second frame of one pair and the first frame of the next while both are in flight."""
    g = _engine(modeldirs, "rife-v4.6", {})
    frames = [gen_frames.smooth_pair(256, 160, 320 + i)[0] for i in range(5)]
    want = [g.process(frames[i], frames[i + 1], 0.5) for i in range(4)]
    res = [g.upload(f) for f in frames]
    got = [None] * 4

    def work(i):
        got[i] = g.process_frames(res[i], res[i + 1], 0.5)
    for _ in range(3):
        th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        [t.start() for t in th]; [t.join() for t in th]
        for i in range(4):
            assert np.array_equal(got[i], want[i]), i


def test_frame_errors_and_lifetime(modeldirs):
    g = _engine(modeldirs, "rife-v4.6", {})
    a = g.upload(gen_frames.smooth_pair(64, 64, 1)[0])
    b = g.upload(gen_frames.smooth_pair(96, 64, 1)[0])
    with pytest.raises(amd.RifeError, match="differ in size"):
        g.process_frames(a, b, 0.5)
    b.release()
    with pytest.raises(ValueError):
        g.process_frames(a, b, 0.5)
    # buffers of released frames are recycled; a frame may outlive its engine
    for i in range(40):
        f = g.upload(gen_frames.smooth_pair(64, 64, i)[0])
        f.release()
    c = g.upload(gen_frames.smooth_pair(64, 64, 2)[0])
    ref = g.process(gen_frames.smooth_pair(64, 64, 1)[0], gen_frames.smooth_pair(64, 64, 2)[0], 0.5)
    assert np.array_equal(g.process_frames(a, c, 0.5), ref)
    del g
    a.release(); c.release()


def test_page_locked_frames_give_the_same_pixels(modeldirs):
    """rife_hip_host_alloc / rife_hip_host_register: page-locked host frames (asynchronous DMA) in, page-locked frame out."""
    import ctypes
    g = amd.RIFE(0, rife_v4=True)
    g.load(modeldirs["rife-v4.6"])
    a, b = gen_frames.smooth_pair(320, 200, 31)
    want = g.process(a, b, 0.5)
    pa, pb, po = amd.pinned_empty(a.shape), amd.pinned_empty(a.shape), amd.pinned_empty(a.shape)
    pa[...] = a; pb[...] = b
    assert np.array_equal(g.process(pa, pb, 0.5, outimage=po), want)
    # a buffer the caller owns, page-locked in place for as long as it is used
    ra, ro = a.copy(), np.empty_like(a)
    L = amd.lib()
    assert L.rife_hip_host_register(ra.ctypes.data_as(ctypes.c_void_p), ra.nbytes) == 0
    assert L.rife_hip_host_register(ro.ctypes.data_as(ctypes.c_void_p), ro.nbytes) == 0
    try:
        assert np.array_equal(g.process(ra, pb, 0.5, outimage=ro), want)
    finally:
        assert L.rife_hip_host_unregister(ra.ctypes.data_as(ctypes.c_void_p)) == 0
        assert L.rife_hip_host_unregister(ro.ctypes.data_as(ctypes.c_void_p)) == 0
    del pa, pb, po                                      # the last view gone: rife_hip_host_free


@pytest.mark.parametrize("w,h,n", [(1920, 1080, 5), (640, 360, 4), (100, 60, 3)])
def test_device_batch_equals_single_device_calls(modeldirs, w, h, n):
    """rife_hip_process_device_batch (lockstep groups of two resident pairs, coarse trunks batched per layer; include/rife_hip.h) against n
    rife_hip_process_device calls: same bytes, ordered on the caller's stream, timestep 0 / 1 entries are copies."""
    import torch
    g = amd.RIFE(0, rife_v4=True); g.load(modeldirs["rife-v4.6"])
    prs = [gen_frames.smooth_pair(w, h, 40 + i) for i in range(n)]
    ts = [0.5, 0.25, 1.0, 0.7, 0.125][:n]
    d0 = [torch.from_numpy(p[0]).cuda() for p in prs]; d1 = [torch.from_numpy(p[1]).cuda() for p in prs]
    st = torch.cuda.Stream()
    want = []
    for i in range(n):
        o = torch.empty_like(d0[i])
        g.process_device(d0[i].data_ptr(), d1[i].data_ptr(), w, h, ts[i], o.data_ptr(), st.cuda_stream)
        want.append(o)
    st.synchronize()
    outs = [torch.zeros_like(x) for x in d0]
    for rep in range(2):                                    # the second call reuses the leased workspaces
        g.process_device_batch([x.data_ptr() for x in d0], [x.data_ptr() for x in d1], w, h, ts, [o.data_ptr() for o in outs], st.cuda_stream)
        with torch.cuda.stream(st):
            got = [o.clone() for o in outs]                # enqueued on the caller's stream AFTER the batch: must see every result
        st.synchronize()
        for i in range(n):
            assert torch.equal(got[i], want[i]), "pair %d (rep %d)" % (i, rep)
    g.process_device_batch([x.data_ptr() for x in d0], [x.data_ptr() for x in d1], w, h, ts, [o.data_ptr() for o in outs], None)      # NULL stream: synchronous
    for i in range(n):
        assert torch.equal(outs[i], want[i])


def test_partition_streams_give_the_same_bytes(modeldirs):
    """rife_hip_stream_create: streams that own a quarter of the compute units each (hipExtStreamCreateWithCUMask), four pairs in flight from four
    threads: the same bytes as the host-frame call, whatever part of the chip a pair ran on; errors for bad partitions and foreign streams."""
    import ctypes
    import torch
    g = amd.RIFE(0, rife_v4=True); g.load(modeldirs["rife-v4.6"])
    w, h = 1920, 1080
    prs = [gen_frames.smooth_pair(w, h, 60 + i) for i in range(4)]
    want = [g.process(p[0], p[1], 0.5) for p in prs]
    strs = [g.stream_create(i, 4) for i in range(4)]
    d0 = [torch.from_numpy(p[0]).cuda() for p in prs]; d1 = [torch.from_numpy(p[1]).cuda() for p in prs]
    outs = [torch.zeros_like(x) for x in d0]
    def worker(i):
        for _ in range(3):
            g.process_device(d0[i].data_ptr(), d1[i].data_ptr(), w, h, 0.5, outs[i].data_ptr(), strs[i])
    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    for i in range(4):
        assert np.array_equal(outs[i].cpu().numpy(), want[i]), "pair %d on part %d of 4" % (i, i)
    for s in strs:
        g.stream_destroy(s)
    with pytest.raises(amd.RifeError):
        g.stream_create(4, 4)
    with pytest.raises(amd.RifeError):
        g.stream_destroy(torch.cuda.Stream().cuda_stream)
    one = g.stream_create(0, 1)                             # nparts = 1: an ordinary stream; left to rife_hip_destroy
    g.process_device(d0[0].data_ptr(), d1[0].data_ptr(), w, h, 0.5, outs[0].data_ptr(), one)
    torch.cuda.synchronize()
    assert np.array_equal(outs[0].cpu().numpy(), want[0])


def test_workspace_pool_is_trimmed_to_the_callers_in_flight(modeldirs):
    """The pool of the host-buffer entry points (csrc/engine_abi.h lease_ctx / release_ctx): a burst of four concurrent callers leaves at most
    four workspaces pooled, and once 32 leases have gone by with one caller in flight the pool is back to ONE workspace - the burst's memory is returned.  The
    frames are the same bytes before, during and after."""
    t = amd.test_build()
    g = t.RIFE(0, rife_v4=True); g.load(modeldirs["rife-v4.6"])
    w, h = 640, 360
    prs = [gen_frames.smooth_pair(w, h, 80 + i) for i in range(4)]
    want = [g.process(p[0], p[1], 0.5) for p in prs]
    assert g.pool_state() == (1, 0, 1)
    got = [[None] * 6 for _ in range(4)]
    start = threading.Barrier(4)
    def worker(i):
        start.wait()
        for k in range(6):
            got[i][k] = g.process(prs[i][0], prs[i][1], 0.5)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [x.start() for x in th]; [x.join() for x in th]
    for i in range(4):
        for k in range(6):
            assert np.array_equal(got[i][k], want[i])
    pooled, leased, hw = g.pool_state()
    assert leased == 0 and 2 <= hw <= 4 and 1 <= pooled <= hw, (pooled, leased, hw)
    for k in range(33):                                             # 32 leases with one caller in flight flush the history
        assert np.array_equal(g.process(prs[k % 4][0], prs[k % 4][1], 0.5), want[k % 4])
    assert g.pool_state() == (1, 0, 1)
