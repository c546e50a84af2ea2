"""The data formats either side of the registration path: the pybind _Vector3dVector's list
semantics and repr (python/kiss_icp/pybind/stl_vector_eigen.h:44-117), point arguments given as arrays / DLPack
tensors, the KITTI and TUM pose writers and the config round trip (python/kiss_icp/pipeline.py:116-134,
config/parser.py:50-90).  No GPU needed: nothing here reaches a device entry with real work."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "kiss-icp_amd", "cpp")


@pytest.fixture(scope="module")
def m():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "kiss-icp_amd", "csrc")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)
    if CPP not in sys.path:
        sys.path.insert(0, CPP)
    import kiss_icp_pybind

    return kiss_icp_pybind


def test_vector3dvector_is_a_list_of_points(m):
    a = np.arange(12, dtype=np.float64).reshape(4, 3)
    v = m._Vector3dVector(a)
    assert repr(v) == "std::vector<Eigen::Vector3d> with 4 elements.\nUse numpy.asarray() to access data."  # stl_vector_eigen.h:103-106
    assert np.array_equal(v[1], a[1]) and np.array_equal(v[-1], a[3])
    with pytest.raises(IndexError):
        v[4]
    assert np.array_equal(np.asarray(v[1:3]), a[1:3]) and np.array_equal(np.asarray(v[::2]), a[::2])
    assert [p.tolist() for p in v] == a.tolist()
    v.append([1, 2, 3])
    v.extend([[4, 5, 6], (7, 8, 9)])
    v.extend(m._Vector3dVector(np.ones((1, 3))))
    v.insert(0, (9, 9, 9))
    assert len(v) == 9 and np.array_equal(v.pop(), [1, 1, 1]) and np.array_equal(v.pop(0), [9, 9, 9])
    v[0] = [7, 7, 7]
    v[1:3] = m._Vector3dVector(np.zeros((2, 3)))
    assert np.array_equal(np.asarray(v)[:3], [[7, 7, 7], [0, 0, 0], [0, 0, 0]])
    del v[0]
    del v[0:2]
    assert np.array_equal(np.asarray(v), [[9, 10, 11], [1, 2, 3], [4, 5, 6], [7, 8, 9]])
    with pytest.raises(RuntimeError):
        v[0:2] = m._Vector3dVector(np.zeros((3, 3)))
    w = m._Vector3dVector(v)  # copy constructor
    assert w == v and not (w != v) and [1, 2, 3] in w and w.count([1, 2, 3]) == 1 and [0, 0, 1] not in w
    w.remove([1, 2, 3])
    with pytest.raises(ValueError):
        w.remove([1, 2, 3])
    assert w != v and len(copy.copy(w)) == 3 and len(copy.deepcopy(w)) == 3
    assert np.array_equal(np.asarray(m._Vector3dVector([[1, 2, 3], (4, 5, 6)])), [[1, 2, 3], [4, 5, 6]])
    v.clear()
    assert len(v) == 0 and not v
    with pytest.raises(IndexError):
        v.pop()
    with pytest.raises(RuntimeError):
        m._Vector3dVector([[1, 2]])
    # the buffer is a live view of the vector's storage
    v = m._Vector3dVector(a)
    view = np.asarray(v)
    view[0, 0] = 42.0
    assert v[0][0] == 42.0


def test_point_arguments_accept_arrays_and_dlpack(m):
    """every points argument takes a _Vector3dVector, an (N,3) array, a host DLPack tensor; the argument is resolved
    before the device is touched, so on a box without a GPU a well-formed argument reaches the 'no device' error
    and a malformed one is rejected first"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the GPU tests")
    a = np.random.default_rng(0).normal(size=(10, 3))
    for good in (a, a.astype(np.float32), a.tolist(), m._Vector3dVector(a), torch.from_numpy(a), np.asfortranarray(a)):
        with pytest.raises(RuntimeError, match="no gfx950 device|NO_DEVICE|device"):
            m._voxel_down_sample(good, 1.0)
    for bad in (np.zeros((3, 2)), np.zeros(3), "points"):
        with pytest.raises((RuntimeError, TypeError), match="points|array"):
            m._voxel_down_sample(bad, 1.0)


def test_pose_writers(tmp_path):
    from kiss_icp_amd.pipeline import rotation_to_quaternion_wxyz, save_poses_kitti_format, save_poses_tum_format

    rng = np.random.default_rng(3)
    rots = Rotation.from_rotvec(rng.normal(0, 1.5, (40, 3)))
    rots = Rotation.concatenate([rots, Rotation.from_rotvec([[np.pi, 0, 0], [0, np.pi - 1e-9, 0], [0, 0, 0], [0, 0, 3.1]])])
    poses = np.tile(np.eye(4), (len(rots), 1, 1))
    poses[:, :3, :3] = rots.as_matrix()
    poses[:, :3, 3] = rng.normal(0, 50, (len(rots), 3))
    for R in poses[:, :3, :3]:
        q = rotation_to_quaternion_wxyz(R)  # (w, x, y, z), w >= 0, and the same rotation
        assert q[0] >= 0.0 and abs(np.linalg.norm(q) - 1.0) < 1e-12
        np.testing.assert_allclose(Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(), R, atol=1e-9)
    base = str(tmp_path / "seq_poses")
    stamps = 0.1 * np.arange(len(poses))
    save_poses_kitti_format(base, poses)
    save_poses_tum_format(base, poses, stamps)
    kitti = np.loadtxt(base + "_kitti.txt")
    assert kitti.shape == (len(poses), 12)
    assert np.array_equal(kitti.reshape(-1, 3, 4), poses[:, :3, :])  # np.savetxt's default %.18e round-trips float64
    tum = np.loadtxt(base + "_tum.txt")
    assert tum.shape == (len(poses), 8)
    np.testing.assert_allclose(tum[:, 0], stamps, atol=5e-5)
    np.testing.assert_allclose(tum[:, 1:4], poses[:, :3, 3], atol=5e-5)
    back = Rotation.from_quat(tum[:, 4:8]).as_matrix()  # columns qx qy qz qw
    np.testing.assert_allclose(back, poses[:, :3, :3], atol=5e-4)
    with open(base + "_tum.txt") as f:
        assert all(len(tok.split(".")[1]) == 4 for tok in f.readline().split())  # fmt="%.4f"


def test_config_yaml_round_trip(tmp_path):
    from kiss_icp_amd.config import KISSConfig, load_config, write_config

    cfg = load_config(max_range=80.0, deskew=False, max_points_per_voxel=13, out_dir=str(tmp_path / "out"))
    assert cfg.mapping.voxel_size == 0.8
    path = tmp_path / "kiss.yaml"
    write_config(cfg, str(path))
    again = load_config(path)
    assert again == cfg
    (tmp_path / "partial.yaml").write_text("data:\n  max_range: 50.0\n  min_range: 60.0\nmapping:\n  max_points_per_voxel: 7\n")
    c = load_config(tmp_path / "partial.yaml", convergence_criterion=1e-5)
    assert c.data.min_range == 0.0 and c.mapping.voxel_size == 0.5 and c.mapping.max_points_per_voxel == 7  # parser.py:73-79
    assert c.registration.convergence_criterion == 1e-5 and c.out_dir == KISSConfig().out_dir
    (tmp_path / "bad.yaml").write_text("mapping:\n  voxel: 1.0\n")
    with pytest.raises(KeyError):
        load_config(tmp_path / "bad.yaml")


def test_synthetic_dataset_offers_what_the_pipeline_asks_for():
    from kiss_icp_amd.datasets import kitti_like

    ds = kitti_like(seed=1, n_frames=5, beams=8, azimuth_steps=64)
    gt = ds.gt_poses
    assert gt.shape == (5, 4, 4) and np.allclose(gt[0], np.eye(4))
    assert np.allclose(np.linalg.norm(gt[1, :3, 3]), 1.0, atol=1e-3)  # one metre per frame
    assert len(ds.get_frames_timestamps()) == 5 and isinstance(ds.sequence_id, str)
