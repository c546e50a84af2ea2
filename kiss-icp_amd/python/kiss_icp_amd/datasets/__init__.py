from .synthetic import SyntheticLidar, kitti_like, livox_like, mulran_like, plane_pair  # noqa: F401
