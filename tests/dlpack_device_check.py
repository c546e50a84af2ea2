"""Helper of tests/test_cpp_api.py (GPU): the pybind pipeline fed with torch ROCm tensors through DLPack."""
import os
import sys

import numpy as np
import torch

torch.zeros(1, device="cuda:0")  # torch's HIP runtime first
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kiss-icp_amd", "cpp"))
sys.path.insert(0, os.path.join(ROOT, "kiss-icp_amd", "python"))
import kiss_icp_pybind as m  # noqa: E402
from kiss_icp_amd.datasets import kitti_like, mulran_like  # noqa: E402

for deskew, factory in ((False, kitti_like), (True, mulran_like)):
    ds = factory(seed=4, n_frames=6, beams=32, azimuth_steps=512)
    cfg = m._KISSConfig()
    cfg.deskew = deskew
    host, dev = m._KissICP(cfg), m._KissICP(cfg)
    for i in range(6):
        pts, ts = ds[i]
        a = host._register_frame(pts, ts)
        d_pts = torch.from_numpy(pts).to("cuda:0")
        d_ts = torch.from_numpy(np.asarray(ts, dtype=np.float64)).to("cuda:0") if len(ts) else None
        b = dev._register_frame(d_pts, d_ts)
        assert np.array_equal(np.asarray(a[0]), np.asarray(b[0])) and np.array_equal(np.asarray(a[1]), np.asarray(b[1])), (deskew, i)
        assert np.array_equal(host._pose(), dev._pose()), (deskew, i)
    assert np.linalg.norm(host._pose()[:3, 3]) > 3.0
pts = ds[0][0]
for bad, exc in ((torch.from_numpy(pts).to("cuda:0").float(), TypeError), (torch.from_numpy(pts).to("cuda:0")[:, :2], TypeError),
                 (torch.from_numpy(pts).to("cuda:0").t(), TypeError)):
    try:
        dev._register_frame(bad, None)
    except exc:
        pass
    else:
        raise SystemExit("accepted a malformed device tensor")
try:
    m._voxel_down_sample(torch.from_numpy(pts).to("cuda:0"), 0.5)  # device tensors: the pipeline entry only
except TypeError:
    pass
else:
    raise SystemExit("a device tensor reached a host-points entry")
print("device tensors: ok")
