"""GPU probe: per-iteration phase profile of the ICP kernel on the bench workload."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kiss-icp_amd', 'python'))
import numpy as np
from kiss_icp_amd import _cabi
from kiss_icp_amd.config import load_config
from kiss_icp_amd.datasets import kitti_like
from kiss_icp_amd.kiss_icp import KissICP
opts = dict(a.split('=') for a in sys.argv[1:])
opts.setdefault('icp_profile', '1')
for k, v in opts.items():
    _cabi.set_option(k, int(v))
nf = 14
ds = kitti_like(seed=0, n_frames=nf)
k = KissICP(load_config(deskew=False))
for i in range(nf):
    k.register_frame(ds[i][0])
prof = k.icp_iteration_profile()
print(opts, k.icp_profile(), 'n_src', k.last_stats()['n_source'])
print(' it  assoc publish gather solve | max_assoc passes   (us)')
for i, r in enumerate(prof):
    print('%3d %6.2f %6.2f %6.2f %6.2f | %6.2f %4d' % (i, r[0] / 100, r[1] / 100, r[2] / 100, r[3] / 100, r[4] / 100, r[5]))
