from .synthetic import (SyntheticLidar, generate_scans, kitti_like, kitti_like_vegetated, livox_like,  # noqa: F401
                        mulran_like, plane_pair)
