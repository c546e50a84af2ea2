import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/kiss-icp_amd/python")
import numpy as np
from kiss_icp_amd.datasets import generate_scans, kitti_like_vegetated
N = 110
scans = generate_scans(kitti_like_vegetated, dict(seed=0, n_frames=N), range(N))
from kiss_icp_amd import _cabi
from kiss_icp_amd.config import load_config
from kiss_icp_amd.kiss_icp import KissICP
def run(sync_every=None, mode="async"):
    k = KissICP(load_config(deskew=False))
    poses = []
    devs = []
    for i, (p, t) in enumerate(scans):
        if mode == "dev":
            d = _cabi.DeviceArray(p); devs.append(d)
            k.register_frame_device(d.ptr, d.shape[0])
        elif mode == "f32":
            k.register_frame_async(p.astype(np.float32), t)
        else:
            k.register_frame_async(p, t)
        if sync_every and (i + 1) % sync_every == 0:
            k.sync(); poses.extend(k.synced_poses())
    k.sync(); poses.extend(k.synced_poses())
    return np.array(poses), k.last_stats()
a, sa = run()
def first_diff(x, y):
    for i in range(min(len(x), len(y))):
        if not np.array_equal(x[i], y[i]): return i, float(np.abs(x[i]-y[i]).max())
    return None
for name, kw in (("again", {}), ("sync10", dict(sync_every=10)), ("sync1", dict(sync_every=1)), ("dev", dict(mode="dev")), ("f32", dict(mode="f32"))):
    b, sb = run(**kw)
    print(name, len(b), first_diff(a, b), sb["map_voxels"], sa["map_voxels"])
