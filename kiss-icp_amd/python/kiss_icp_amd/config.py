"""KISSConfig -- field names and defaults of the reference's python/kiss_icp/config/config.py:28-48
and pipeline/KissICP.hpp:36-54, as plain dataclasses (pydantic_settings is not available here), with the
YAML loading / writing of config/parser.py:50-90 (environment-variable overrides are not restated)."""
import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional, Union


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


def load_config(config_file: Optional[Union[str, Path]] = None, **overrides) -> KISSConfig:
    """KISSConfig from an optional YAML file with the reference's layout (config/parser.py:50-81: sections data /
    registration / mapping / adaptive_threshold, and out_dir) and / or flat keyword overrides, e.g.
    load_config(max_range=80, deskew=False); voxel_size defaults to max_range / 100 like config/parser.py:78-79."""
    sections = {"data": DataConfig(), "mapping": MappingConfig(), "registration": RegistrationConfig(),
                "adaptive_threshold": AdaptiveThresholdConfig()}
    out_dir = {}
    if config_file is not None:
        import yaml

        with open(config_file) as f:
            data = yaml.safe_load(f) or {}
        for name, values in data.items():
            if name == "out_dir":
                out_dir["out_dir"] = str(values)
            elif name in sections and isinstance(values, dict):
                for k, v in values.items():
                    if not hasattr(sections[name], k):
                        raise KeyError(f"{name}.{k}")
                    setattr(sections[name], k, v)
            else:
                raise KeyError(name)
    if "out_dir" in overrides:
        out_dir["out_dir"] = overrides.pop("out_dir")
    for k, v in overrides.items():
        for section in sections.values():
            if hasattr(section, k):
                setattr(section, k, v)
                break
        else:
            raise KeyError(k)
    return KISSConfig(**out_dir, **sections)


def write_config(config: KISSConfig = None, filename: str = "kiss_icp.yaml"):
    """config/parser.py:84-90: the configuration as YAML"""
    import yaml

    with open(filename, "w") as outfile:
        yaml.dump(dataclasses.asdict(config if config is not None else KISSConfig()), outfile, default_flow_style=False)
