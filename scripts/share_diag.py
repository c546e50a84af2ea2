"""streams sharing one GPU: what a lone pipeline with share n does, and n of them side by side (device-resident scans, one host thread each)"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kiss-icp_amd', 'python'))
import numpy as np, torch
from kiss_icp_amd import _cabi
from kiss_icp_amd.config import load_config
from kiss_icp_amd.datasets import generate_scans, kitti_like_vegetated
from kiss_icp_amd.kiss_icp import KissICP
W, K = 10, 150
for a in sys.argv[1:]:
    n, v = a.split('=')
    _cabi.set_option(n, int(v))
scans = generate_scans(kitti_like_vegetated, dict(seed=0, n_frames=W + K), range(W + K))
dev = torch.device('cuda:0')
keep = [torch.from_numpy(p).to(dev) for p, _ in scans]
items = [(d.data_ptr(), d.shape[0], None, 0) for d in keep]
def lone(share):
    _cabi.set_option('icp_device_streams', share)
    k = KissICP(load_config(deskew=False))
    _cabi.set_option('icp_device_streams', 1)
    for f in items[:W]: k.register_frame_device(*f)
    k.sync(); k.icp_timing(reset=True); k.host_stats(reset=True)
    t0 = time.perf_counter()
    for f in items[W:]: k.register_frame_device(*f)
    k.sync(); dt = time.perf_counter() - t0
    icp, hs = k.icp_timing(), k.host_stats()
    print('lone, share %d: %7.1f scans/s | k_icp %.4f ms/launch (%d workgroups) %.2f us/iter | device gap %.4f ms/frame' % (
        share, K / dt, icp['total_ms'] / icp['launches'], k.icp_profile()['workgroups'], 1e3 * icp['total_ms'] / icp['iterations'], hs['device_gap_ms'] / K))
def together(n):
    _cabi.set_option('icp_device_streams', n)
    ks = [KissICP(load_config(deskew=False)) for _ in range(n)]
    _cabi.set_option('icp_device_streams', 1)
    for k in ks:
        for f in items[:W]: k.register_frame_device(*f)
    for k in ks: k.sync(); k.icp_timing(reset=True); k.host_stats(reset=True)
    def drive(k):
        for f in items[W:]: k.register_frame_device(*f)
        k.sync()
    th = [threading.Thread(target=drive, args=(k,)) for k in ks]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print('%d together: aggregate %7.1f scans/s (%.1f per stream)' % (n, n * K / dt, K / dt))
    for k in ks:
        icp, hs = k.icp_timing(), k.host_stats()
        print('    k_icp %.4f ms/launch (%d workgroups) %.2f us/iter | device gap %.4f ms/frame (max %.3f)' % (
            icp['total_ms'] / icp['launches'], k.icp_profile()['workgroups'], 1e3 * icp['total_ms'] / icp['iterations'], hs['device_gap_ms'] / K, hs['max_device_gap_ms']))
for s in (1, 2, 4): lone(s)
for n in (2, 4): together(n)
