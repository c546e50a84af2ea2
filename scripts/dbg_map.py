import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'kiss-icp_amd', 'python'))
import numpy as np
from kiss_icp_amd.mapping import VoxelHashMap
case = sys.argv[1]
rng = np.random.default_rng(12)
g = VoxelHashMap(1.0, 100.0, 20)
if case == 'a': pts = rng.uniform(0, 3, (30000, 3))
elif case == 'b': pts = rng.uniform(0, 3, (1000, 3))
elif case == 'c': pts = rng.uniform(-60, 60, (30000, 3))
elif case == 'd': pts = rng.uniform(0, 3, (600, 3))
elif case == 'e': pts = rng.uniform(0, 3, (16000, 3))
g.add_points(pts)
print(case, 'ok voxels', g.num_voxels(), 'pts', len(g.point_cloud()))
