"""KISSConfig -- field names and defaults of the reference's python/kiss_icp/config/config.py:28-48
and pipeline/KissICP.hpp:36-54, as plain dataclasses (pydantic_settings is not available here; the
YAML/env loading of config/parser.py is application glue outside the hot path)."""
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class DataConfig:
    max_range: float = 100.0
    min_range: float = 0.0
    deskew: bool = True


@dataclass
class MappingConfig:
    voxel_size: Optional[float] = None  # default: max_range / 100 (config/parser.py:78-79)
    max_points_per_voxel: int = 20


@dataclass
class RegistrationConfig:
    max_num_iterations: int = 500
    convergence_criterion: float = 0.0001
    max_num_threads: int = 0


@dataclass
class AdaptiveThresholdConfig:
    fixed_threshold: Optional[float] = None
    initial_threshold: float = 2.0
    min_motion_th: float = 0.1


@dataclass
class KISSConfig:
    out_dir: str = "results"
    data: DataConfig = field(default_factory=DataConfig)
    registration: RegistrationConfig = field(default_factory=RegistrationConfig)
    mapping: MappingConfig = field(default_factory=MappingConfig)
    adaptive_threshold: AdaptiveThresholdConfig = field(default_factory=AdaptiveThresholdConfig)

    def __post_init__(self):
        if self.data.max_range < self.data.min_range:  # config/parser.py:73-75
            self.data.min_range = 0.0
        if self.mapping.voxel_size is None:  # config/parser.py:78-79
            self.mapping.voxel_size = float(self.data.max_range / 100.0)


def load_config(**overrides) -> KISSConfig:
    """KISSConfig from flat keyword overrides, e.g. load_config(max_range=80, deskew=False);
    voxel_size defaults to max_range / 100 like config/parser.py:78-79."""
    sections = {"data": DataConfig(), "mapping": MappingConfig(), "registration": RegistrationConfig(),
                "adaptive_threshold": AdaptiveThresholdConfig()}
    for k, v in overrides.items():
        for section in sections.values():
            if hasattr(section, k):
                setattr(section, k, v)
                break
        else:
            raise KeyError(k)
    return KISSConfig(**sections)
