"""OdometryPipeline -- the dataset -> KissICP -> poses -> files loop of python/kiss_icp/pipeline.py:41-231, without
its visualizer, progress bar and CLI (out of scope: SURVEY 8, "control plane / UI").  What is kept is what sits either
side of the registration path: the per-scan loop, the result files (numpy, KITTI and TUM pose formats:
pipeline.py:116-134), the metrics when the dataset has ground truth, and the timing summary.

Difference by design: scans are QUEUED on the device pipeline (kicp_pipeline_register_frame_async: staged, uploaded
under the previous frame's registration) and synchronised every `queue_depth` scans instead of once per scan -- the
poses are the same bit for bit (tests/test_gpu_paths.py), the throughput is what bench.py reports.  queue_depth=1 is
the reference's scan-at-a-time behaviour."""
import datetime
import os
import time
from pathlib import Path
from typing import Optional

import numpy as np

from .config import load_config, write_config
from .kiss_icp import KissICP


def rotation_to_quaternion_wxyz(R: np.ndarray) -> np.ndarray:
    """unit quaternion (w, x, y, z), w >= 0, of a rotation matrix (what pyquaternion's Quaternion(matrix=...)
    .elements gives the reference, up to the sign convention w >= 0)"""
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0.0:
        s = 2.0 * np.sqrt(1.0 + t)
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(max(1.0 + R[i, i] - R[j, j] - R[k, k], 0.0))
        q = np.empty(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q /= np.linalg.norm(q)
    return -q if q[0] < 0.0 else q


class PipelineResults:
    """tools/pipeline_results.py: an ordered list of (description, units, value) with a text rendering"""

    def __init__(self):
        self._results = []

    def empty(self):
        return len(self._results) == 0

    def append(self, desc, units, value, trunc=False):
        self._results.append({"desc": desc, "units": units, "value": int(value) if trunc else float(value)})

    def __iter__(self):
        return iter(self._results)

    def as_dict(self):
        return {r["desc"]: r["value"] for r in self._results}

    def text(self, title=""):
        w = max([len(r["desc"]) for r in self._results] + [6])
        lines = [title] if title else []
        lines += ["%-*s | %12s | %s" % (w, "Metric", "Value", "Units")]
        lines += ["%-*s | %12s | %s" % (w, r["desc"], ("%d" % r["value"]) if isinstance(r["value"], int) else ("%.3f" % r["value"]), r["units"])
                  for r in self._results]
        return "\n".join(lines)

    def log_to_file(self, filename, title):
        with open(filename, "w") as f:
            f.write(self.text(title) + "\n")


class OdometryPipeline:
    def __init__(self, dataset, config: Optional[Path] = None, n_scans: int = -1, jump: int = 0, device_id: int = 0,
                 queue_depth: int = 32, write_results: bool = True):
        self._dataset = dataset
        self._n_scans = len(dataset) - jump if n_scans == -1 else min(len(dataset) - jump, n_scans)
        self._jump = jump
        self._first = jump
        self._last = jump + self._n_scans
        self._queue_depth = max(1, int(queue_depth))
        self._write_results = write_results
        self.config = config if hasattr(config, "mapping") else load_config(config)
        self.results_dir = None
        self.odometry = KissICP(config=self.config, device_id=device_id)
        self.results = PipelineResults()
        self.times = np.zeros(self._n_scans)
        self.poses = np.zeros((self._n_scans, 4, 4))
        self.has_gt = hasattr(dataset, "gt_poses")
        self.gt_poses = np.asarray(dataset.gt_poses)[self._first:self._last] if self.has_gt else None
        self.dataset_name = dataset.__class__.__name__
        self.dataset_sequence = (dataset.sequence_id if hasattr(dataset, "sequence_id")
                                 else os.path.basename(getattr(dataset, "data_dir", "sequence")))

    # ---- public --------------------------------------------------------------------------------------
    def run(self):
        self._run_pipeline()
        self._run_evaluation()
        if self._write_results:
            self._create_output_dir()
            self._write_result_poses()
            self._write_gt_poses()
            write_config(self.config, os.path.join(self.results_dir, "config.yml"))
            if not self.results.empty():
                self.results.log_to_file(os.path.join(self.results_dir, "result_metrics.log"),
                                         f"Results for {self.dataset_name} Sequence {self.dataset_sequence}")
        return self.results

    # ---- the per-scan loop (pipeline.py:96-113) ---------------------------------------------------------
    def _run_pipeline(self):
        k = self.odometry
        pending = []  # indices queued since the last synchronisation
        t_batch = time.perf_counter_ns()

        def flush():
            nonlocal t_batch
            if not pending:
                return
            k.sync()
            got = k.synced_poses()
            assert len(got) == len(pending)
            per = (time.perf_counter_ns() - t_batch) / len(pending)
            for j, idx in enumerate(pending):
                self.poses[idx - self._first] = got[j]
                self.times[idx - self._first] = per
            pending.clear()
            t_batch = time.perf_counter_ns()

        for idx in range(self._first, self._last):
            raw_frame, timestamps = self._dataset[idx]
            k.register_frame_async(raw_frame, timestamps)
            pending.append(idx)
            if len(pending) >= self._queue_depth:
                flush()
        flush()

    # ---- files (pipeline.py:116-175) -----------------------------------------------------------------------
    @staticmethod
    def save_poses_kitti_format(filename: str, poses: np.ndarray):
        """one line per pose: the 12 numbers of the upper 3x4 block, row-major"""
        np.savetxt(fname=f"{filename}_kitti.txt", X=np.asarray(poses)[:, :3].reshape(-1, 12))

    @staticmethod
    def save_poses_tum_format(filename, poses, timestamps):
        """one line per pose: timestamp tx ty tz qx qy qz qw, 4 decimals"""
        poses = np.asarray(poses)
        tum = np.zeros((len(poses), 8))
        for i in range(len(poses)):
            qw, qx, qy, qz = rotation_to_quaternion_wxyz(poses[i, :3, :3])
            tum[i] = np.r_[float(timestamps[i]), poses[i, :3, 3], qx, qy, qz, qw]
        np.savetxt(fname=f"{filename}_tum.txt", X=tum, fmt="%.4f")

    def _calibrate_poses(self, poses):
        return self._dataset.apply_calibration(poses) if hasattr(self._dataset, "apply_calibration") else poses

    def _get_frames_timestamps(self):
        return (self._dataset.get_frames_timestamps() if hasattr(self._dataset, "get_frames_timestamps")
                else np.arange(0, self._n_scans, 1.0))

    def _save_poses(self, filename: str, poses, timestamps):
        np.save(filename, poses)
        self.save_poses_kitti_format(filename, poses)
        self.save_poses_tum_format(filename, poses, timestamps)

    def _write_result_poses(self):
        self._save_poses(f"{self.results_dir}/{self.dataset_sequence}_poses", self._calibrate_poses(self.poses),
                         self._get_frames_timestamps())

    def _write_gt_poses(self):
        if self.has_gt:
            self._save_poses(f"{self.results_dir}/{self.dataset_sequence}_gt", self._calibrate_poses(self.gt_poses),
                             self._get_frames_timestamps())

    # ---- evaluation (pipeline.py:177-196) --------------------------------------------------------------------
    def _get_fps(self):
        t = self.times[self.times != 0]
        total_s = np.sum(t) * 1e-9
        return float(t.shape[0] / total_s) if total_s > 0 else 0.0

    def _run_evaluation(self):
        if self.has_gt:
            from .metrics import absolute_trajectory_error, sequence_error

            avg_tra, avg_rot = sequence_error(self.gt_poses, self.poses)
            ate_rot, ate_trans = absolute_trajectory_error(self.gt_poses, self.poses)
            self.results.append(desc="Average Translation Error", units="%", value=avg_tra)
            self.results.append(desc="Average Rotational Error", units="deg/m", value=avg_rot)
            self.results.append(desc="Absolute Trajectory Error (ATE)", units="m", value=ate_trans)
            self.results.append(desc="Absolute Rotational Error (ARE)", units="rad", value=ate_rot)
        fps = self._get_fps()
        if int(np.floor(fps)) > 0:
            self.results.append(desc="Average Frequency", units="Hz", value=int(np.floor(fps)), trunc=True)
            self.results.append(desc="Average Runtime", units="ms", value=int(np.ceil(1e3 / fps)), trunc=True)

    @staticmethod
    def _get_results_dir(out_dir: str):
        stamp = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
        results_dir = os.path.join(os.path.realpath(out_dir), stamp)
        latest = os.path.join(os.path.realpath(out_dir), "latest")
        os.makedirs(results_dir, exist_ok=True)
        if os.path.exists(latest) or os.path.islink(latest):
            os.unlink(latest)
        os.symlink(results_dir, latest)
        return results_dir

    def _create_output_dir(self):
        self.results_dir = self._get_results_dir(self.config.out_dir)
