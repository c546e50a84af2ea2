"""Why is the device-resident entry slower than the host-input one?  Same frames, fresh pipelines, host-side statistics and
the device time between registrations for: host float64 (staged), device float64, device float32."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kiss-icp_amd', 'python'))
import numpy as np, torch
from kiss_icp_amd import _cabi
from kiss_icp_amd.config import load_config
from kiss_icp_amd.datasets import generate_scans, kitti_like_vegetated
from kiss_icp_amd.kiss_icp import KissICP
opts = dict(a.split('=') for a in sys.argv[1:])
keep_alive = opts.pop('keep_alive', '0') == '1'  # keep every pipeline of the run alive (bench.py keeps its main pipeline while the extras run)
alive = []
for k, v in opts.items():
    _cabi.set_option(k, int(v))
W, K = 10, 200
scans = generate_scans(kitti_like_vegetated, dict(seed=0, n_frames=W + K), range(W + K))
dev = torch.device('cuda:0')
def run(kind):
    k = KissICP(load_config(deskew=False))
    for p, t in scans[:W]:
        k.register_frame_async(p, t)
    k.sync()
    if kind == 'host64':
        items = [(p, t) for p, t in scans[W:]]
        fn = lambda f: k.register_frame_async(*f)
    elif kind == 'host32':
        items = [(p.astype(np.float32), t) for p, t in scans[W:]]
        fn = lambda f: k.register_frame_async(*f)
    else:
        keep = [torch.from_numpy(p).to(dev) for p, _ in scans[W:]]
        items = [(d.data_ptr(), d.shape[0], None, 0) for d in keep]
        fn = lambda f: k.register_frame_device(*f)
    k.host_stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in items:
        fn(f)
    k.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hs = k.host_stats()
    icp = k.icp_timing()
    if keep_alive:
        alive.append(k)
    print('%-8s %7.1f scans/s  %.4f ms/frame | k_icp %.4f ms/launch, %.1f iters/frame | device gap %.4f ms/frame (max %.3f) | waits: backpressure %d (%.2f ms) capacity %d staging %d | enqueue %.3f ms/frame stage %.3f ms/frame' % (
        kind, K / dt, 1e3 * dt / K, icp['total_ms'] / max(1, icp['launches']), icp['iterations'] / max(1, icp['launches']), hs['device_gap_ms'] / K, hs['max_device_gap_ms'],
        hs['backpressure_waits'], hs['backpressure_ms'], hs['capacity_waits'], hs['staging_waits'], hs['enqueue_ms'] / K, hs['stage_ms'] / K))
for rep in range(2):
    for kind in ('host64', 'dev64', 'host32'):
        run(kind)
