"""custom_gymnasium_environments_amd — MI355X-native batched stepper for the Gymnasium environments of
hasnainfarid/Custom_Gymnasium_Environments.  Hand-written HIP kernels (csrc/*.hip) behind a C ABI
(include/cge_amd.h) behind gymnasium.vector.VectorEnv-shaped Python classes on torch-ROCm tensors."""
from ._native import NativeLibraryError, lib as native_lib  # noqa: F401
from .vector_env import AutoresetMode, DeviceVectorEnv  # noqa: F401
from .snake import SnakeVectorEnv  # noqa: F401
from .crypto import CryptoVectorEnv  # noqa: F401
from .traffic import TrafficVectorEnv  # noqa: F401
from .parking import ParkingVectorEnv  # noqa: F401
from .climate import ClimateVectorEnv  # noqa: F401
from .fleet import FleetVectorEnv  # noqa: F401
from .manufacturing import ManufacturingVectorEnv  # noqa: F401
from .hospital import HospitalVectorEnv  # noqa: F401
from . import sharding  # noqa: F401
from .sharding import gather_obs, make_sharded, shard_range  # noqa: F401
from .registry import NumpyVectorEnv, make_vec, registered_ids  # noqa: F401

__all__ = ["SnakeVectorEnv", "CryptoVectorEnv", "TrafficVectorEnv", "ParkingVectorEnv", "ClimateVectorEnv", "FleetVectorEnv", "ManufacturingVectorEnv", "HospitalVectorEnv", "sharding", "make_sharded", "gather_obs", "shard_range", "make_vec", "registered_ids", "NumpyVectorEnv", "AutoresetMode", "DeviceVectorEnv", "NativeLibraryError", "native_lib"]
__version__ = "0.1.0"
