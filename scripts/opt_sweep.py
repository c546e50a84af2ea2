"""GPU sweep of one option over the bench workloads: scans generated once, one pipeline per setting,
the device time of k_icp per launch and per iteration from the pipeline's own events.
usage: opt_sweep.py NAME v1,v2,...  [livox=1] [kitti=1] [frames=N] [other_option=value ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kiss-icp_amd', 'python'))
import numpy as np  # noqa: E402
from kiss_icp_amd import _cabi  # noqa: E402
from kiss_icp_amd.config import load_config  # noqa: E402
from kiss_icp_amd.datasets import generate_scans, kitti_like_vegetated, livox_like  # noqa: E402
from kiss_icp_amd.kiss_icp import KissICP  # noqa: E402

name, values = sys.argv[1], [int(v) for v in sys.argv[2].split(',')]
kw = dict(a.split('=') for a in sys.argv[3:])
do_livox, do_kitti = kw.pop('livox', '0') == '1', kw.pop('kitti', '1') == '1'
frames = int(kw.pop('frames', '0'))
livox_frames = int(kw.pop('livox_frames', '16'))
outer = kw.pop('outer', None)  # outer=name:v1,v2 -- a second option swept around the first
outer_name, outer_vals = (outer.split(':')[0], [int(v) for v in outer.split(':')[1].split(',')]) if outer else (None, [None])
for k, v in kw.items():
    _cabi.set_option(k, int(v))
work = []
if do_kitti:
    nf = frames or 120
    work.append(('kitti', generate_scans(kitti_like_vegetated, dict(seed=0, n_frames=nf), range(nf)), load_config(deskew=False), nf // 2))
if do_livox:
    nf = livox_frames
    work.append(('livox', generate_scans(livox_like, dict(seed=2, n_frames=nf), range(nf)), load_config(deskew=False, voxel_size=0.1), max(6, nf // 2)))
for wname, scans, cfg, warm in work:
    ref = None
    for ov, v in [(ov, v) for ov in outer_vals for v in values]:
        if outer_name:
            _cabi.set_option(outer_name, ov)
        _cabi.set_option(name, v)
        k = KissICP(cfg)
        for i in range(warm):
            k.register_frame_async(scans[i][0])
        k.sync()
        k.icp_timing(reset=True)
        t0 = time.perf_counter()
        for i in range(warm, len(scans)):
            k.register_frame_async(scans[i][0])
        k.sync()
        wall = time.perf_counter() - t0
        t = k.icp_timing()
        ms, launches, iters = t['total_ms'], t['launches'], t['iterations']
        pose = k.last_pose.copy()
        same = True if ref is None else bool(np.array_equal(ref, pose))
        ref = pose if ref is None else ref
        print('%s%s %s=%-8d frame %.3f ms  k_icp %.3f ms/launch  %.2f us/iter  iters/frame %.1f  n_src %d  same_pose %s' % (
            wname, (' %s=%d' % (outer_name, ov)) if outer_name else '', name, v, 1e3 * wall / (len(scans) - warm), ms / launches, 1e3 * ms / iters, iters / launches, k.last_stats()['n_source'], same), flush=True)
        del k
