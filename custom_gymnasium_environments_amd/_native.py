"""ctypes binding of libcge_amd.so (the C ABI declared in include/cge_amd.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, importing an
env class still works (so the package can be inspected on a CPU box) but creating one raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CGE_AMD_LIBRARY: A/B measurements against another build of the SAME library (tools/ab/); not a backend switch
LIB_PATH = os.environ.get("CGE_AMD_LIBRARY") or os.path.join(_HERE, "libcge_amd.so")

CGE_OK = 0
STATUS_NAMES = {0: "CGE_OK", -1: "CGE_ERR_INVALID_ARG", -2: "CGE_ERR_HIP", -3: "CGE_ERR_UNSUPPORTED",
                -4: "CGE_ERR_NO_DEVICE"}
AUTORESET_NEXT_STEP, AUTORESET_SAME_STEP, AUTORESET_DISABLED = 0, 1, 2


class NativeLibraryError(RuntimeError):
    pass


class SnakeConfig(C.Structure):
    _fields_ = [("grid_size", C.c_int32), ("max_steps", C.c_int32), ("autoreset_mode", C.c_int32),
                ("reserved", C.c_int32)]


class CryptoConfig(C.Structure):
    _fields_ = [("initial_balance", C.c_double), ("trading_fee_rate", C.c_double), ("slippage_rate", C.c_double),
                ("min_price", C.c_double), ("max_price", C.c_double), ("volatility_base", C.c_double),
                ("market_psychology_factor", C.c_double), ("max_steps", C.c_int32), ("action_type", C.c_int32),
                ("autoreset_mode", C.c_int32), ("reserved", C.c_int32)]


class TrafficConfig(C.Structure):
    _fields_ = [("grid_rows", C.c_int32), ("grid_cols", C.c_int32), ("num_intersections", C.c_int32),
                ("max_vehicles", C.c_int32), ("spawn_rate", C.c_double), ("max_steps", C.c_int32),
                ("autoreset_mode", C.c_int32)]


class ParkingConfig(C.Structure):
    _fields_ = [("max_steps", C.c_int32), ("autoreset_mode", C.c_int32)]


class ClimateConfig(C.Structure):
    _fields_ = [("max_occupancy", C.c_int32), ("episode_minutes", C.c_int32), ("autoreset_mode", C.c_int32), ("reserved", C.c_int32)]


class FleetConfig(C.Structure):
    _fields_ = [("max_timesteps", C.c_int32), ("autoreset_mode", C.c_int32)]


class HospitalConfig(C.Structure):
    _fields_ = [("max_episode_length", C.c_int32), ("autoreset_mode", C.c_int32)]


class ManufacturingConfig(C.Structure):
    _fields_ = [("max_steps", C.c_int32), ("autoreset_mode", C.c_int32)]


# name -> (restype, argtypes); also the list tests check against include/cge_amd.h
_vp, _i32, _i64, _u32, _u64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_size_t
SIGNATURES = {
    "cge_version": (C.c_char_p, []),
    "cge_hash_action": (_u32, [_u64, _u64, _u64, _u32, _u32]),
    "cge_snake_create": (C.c_int, [C.POINTER(SnakeConfig), _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "cge_snake_destroy": (C.c_int, [_vp]),
    "cge_snake_seed": (C.c_int, [_vp, _vp, _u64, _vp]),
    "cge_snake_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cge_snake_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cge_snake_rollout": (C.c_int, [_vp, _i32, _vp, _u64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "cge_snake_rollout_final_obs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "cge_snake_final_obs_segment": (_i64, [_vp]),
    "cge_snake_info": (C.c_int, [_vp, _i32, _vp, _vp]),
    "cge_snake_render_rgb": (C.c_int, [_vp, _vp, _vp]),
    "cge_snake_state_bytes": (_sz, [_vp]),
    "cge_snake_get_state": (C.c_int, [_vp, _vp, _vp]),
    "cge_snake_set_state": (C.c_int, [_vp, _vp, _vp]),
    "cge_snake_error_count": (_i64, [_vp, _vp]),
    "cge_snake_device_bytes": (_sz, [_vp]),
    "cge_snake_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "cge_snake_last_error": (C.c_char_p, [_vp]),
    "cge_snake_last_kernel": (C.c_char_p, [_vp]),
    "cge_crypto_default_config": (None, [C.POINTER(CryptoConfig)]),
    "cge_crypto_create": (C.c_int, [C.POINTER(CryptoConfig), _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "cge_crypto_destroy": (C.c_int, [_vp]),
    "cge_crypto_seed": (C.c_int, [_vp, _vp, _u64, _vp]),
    "cge_crypto_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cge_crypto_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cge_crypto_rollout": (C.c_int, [_vp, _i32, _vp, _u64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "cge_crypto_rollout_final_obs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "cge_crypto_final_obs_segment": (_i64, [_vp]),
    "cge_crypto_info": (C.c_int, [_vp, _i32, _vp, _vp]),
    "cge_crypto_state_bytes": (_sz, [_vp]),
    "cge_crypto_get_state": (C.c_int, [_vp, _vp, _vp]),
    "cge_crypto_set_state": (C.c_int, [_vp, _vp, _vp]),
    "cge_crypto_device_bytes": (_sz, [_vp]),
    "cge_crypto_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "cge_crypto_last_error": (C.c_char_p, [_vp]),
    "cge_crypto_last_kernel": (C.c_char_p, [_vp]),
    "cge_traffic_default_config": (None, [C.POINTER(TrafficConfig)]),
    "cge_traffic_create": (C.c_int, [C.POINTER(TrafficConfig), _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "cge_traffic_destroy": (C.c_int, [_vp]),
    "cge_traffic_seed": (C.c_int, [_vp, _vp, _u64, _vp]),
    "cge_traffic_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cge_traffic_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cge_traffic_rollout": (C.c_int, [_vp, _i32, _vp, _u64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "cge_traffic_rollout_final_obs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "cge_traffic_final_obs_segment": (_i64, [_vp]),
    "cge_traffic_info": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "cge_traffic_total_reward": (C.c_int, [_vp, _vp, _vp]),
    "cge_traffic_state_bytes": (_sz, [_vp]),
    "cge_traffic_get_state": (C.c_int, [_vp, _vp, _vp]),
    "cge_traffic_set_state": (C.c_int, [_vp, _vp, _vp]),
    "cge_traffic_device_bytes": (_sz, [_vp]),
    "cge_traffic_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "cge_traffic_last_error": (C.c_char_p, [_vp]),
    "cge_traffic_last_kernel": (C.c_char_p, [_vp]),
    "cge_parking_create": (C.c_int, [C.POINTER(ParkingConfig), _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "cge_parking_destroy": (C.c_int, [_vp]),
    "cge_parking_seed": (C.c_int, [_vp, _vp, _u64, _vp]),
    "cge_parking_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cge_parking_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cge_parking_rollout": (C.c_int, [_vp, _i32, _vp, _u64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "cge_parking_rollout_final_obs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "cge_parking_final_obs_segment": (_i64, [_vp]),
    "cge_parking_info": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "cge_parking_info64": (C.c_int, [_vp, _i32, _vp, _vp]),
    "cge_parking_snapshot_bytes": (_sz, [_vp]),
    "cge_parking_snapshot_get": (C.c_int, [_vp, _vp, _vp]),
    "cge_parking_snapshot_set": (C.c_int, [_vp, _vp, _vp]),
    "cge_parking_device_bytes": (_sz, [_vp]),
    "cge_parking_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "cge_parking_last_error": (C.c_char_p, [_vp]),
    "cge_parking_last_kernel": (C.c_char_p, [_vp]),
    "cge_climate_create": (C.c_int, [C.POINTER(ClimateConfig), _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "cge_climate_destroy": (C.c_int, [_vp]),
    "cge_climate_seed": (C.c_int, [_vp, _vp, _u64, _vp]),
    "cge_climate_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cge_climate_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cge_climate_rollout": (C.c_int, [_vp, _i32, _vp, _vp, _u64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "cge_climate_rollout_final_obs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "cge_climate_final_obs_segment": (_i64, [_vp]),
    "cge_climate_info": (C.c_int, [_vp, _i32, _vp, _vp]),
    "cge_climate_snapshot_bytes": (_sz, [_vp]),
    "cge_climate_snapshot_get": (C.c_int, [_vp, _vp, _vp]),
    "cge_climate_snapshot_set": (C.c_int, [_vp, _vp, _vp]),
    "cge_climate_device_bytes": (_sz, [_vp]),
    "cge_climate_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "cge_climate_last_error": (C.c_char_p, [_vp]),
    "cge_climate_last_kernel": (C.c_char_p, [_vp]),
    "cge_fleet_create": (C.c_int, [C.POINTER(FleetConfig), _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "cge_fleet_destroy": (C.c_int, [_vp]),
    "cge_fleet_seed": (C.c_int, [_vp, _vp, _u64, _vp]),
    "cge_fleet_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cge_fleet_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cge_fleet_rollout": (C.c_int, [_vp, _i32, _vp, _u64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "cge_fleet_rollout_final_obs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "cge_fleet_final_obs_segment": (_i64, [_vp]),
    "cge_fleet_info": (C.c_int, [_vp, _i32, _vp, _vp]),
    "cge_fleet_snapshot_bytes": (_sz, [_vp]),
    "cge_fleet_snapshot_get": (C.c_int, [_vp, _vp, _vp]),
    "cge_fleet_snapshot_set": (C.c_int, [_vp, _vp, _vp]),
    "cge_fleet_device_bytes": (_sz, [_vp]),
    "cge_fleet_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "cge_fleet_done_mask": (C.c_int, [_vp, _vp]),
    "cge_fleet_last_error": (C.c_char_p, [_vp]),
    "cge_fleet_last_kernel": (C.c_char_p, [_vp]),
    "cge_manufacturing_create": (C.c_int, [C.POINTER(ManufacturingConfig), _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "cge_manufacturing_destroy": (C.c_int, [_vp]),
    "cge_manufacturing_seed": (C.c_int, [_vp, _vp, _u64, _vp]),
    "cge_manufacturing_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cge_manufacturing_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cge_manufacturing_rollout": (C.c_int, [_vp, _i32, _vp, _u64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "cge_manufacturing_rollout_final_obs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "cge_manufacturing_final_obs_segment": (_i64, [_vp]),
    "cge_manufacturing_info": (C.c_int, [_vp, _i32, _vp, _vp]),
    "cge_manufacturing_snapshot_bytes": (_sz, [_vp]),
    "cge_manufacturing_snapshot_get": (C.c_int, [_vp, _vp, _vp]),
    "cge_manufacturing_snapshot_set": (C.c_int, [_vp, _vp, _vp]),
    "cge_manufacturing_device_bytes": (_sz, [_vp]),
    "cge_manufacturing_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "cge_manufacturing_done_mask": (C.c_int, [_vp, _vp]),
    "cge_manufacturing_last_error": (C.c_char_p, [_vp]),
    "cge_manufacturing_last_kernel": (C.c_char_p, [_vp]),
    "cge_hospital_create": (C.c_int, [C.POINTER(HospitalConfig), _i64, C.c_int, _i64, C.POINTER(_vp)]),
    "cge_hospital_destroy": (C.c_int, [_vp]),
    "cge_hospital_seed": (C.c_int, [_vp, _vp, _u64, _vp]),
    "cge_hospital_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cge_hospital_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cge_hospital_rollout": (C.c_int, [_vp, _i32, _vp, _u64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "cge_hospital_rollout_final_obs": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "cge_hospital_final_obs_segment": (_i64, [_vp]),
    "cge_hospital_info": (C.c_int, [_vp, _i32, _vp, _vp]),
    "cge_hospital_snapshot_bytes": (_sz, [_vp]),
    "cge_hospital_snapshot_get": (C.c_int, [_vp, _vp, _vp]),
    "cge_hospital_snapshot_set": (C.c_int, [_vp, _vp, _vp]),
    "cge_hospital_device_bytes": (_sz, [_vp]),
    "cge_hospital_episode_stats": (C.c_int, [_vp, _vp, _vp]),
    "cge_hospital_done_mask": (C.c_int, [_vp, _vp]),
    "cge_hospital_last_error": (C.c_char_p, [_vp]),
    "cge_hospital_last_kernel": (C.c_char_p, [_vp]),
}

_lib = None


def lib():
    """Load libcge_amd.so once; raise NativeLibraryError (never fall back) if that is impossible."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `python -m custom_gymnasium_environments_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                if os.environ.get("CGE_AMD_LIBRARY"):          # A/B against an older build: entry points added since are simply absent
                    continue
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, handle=None, last_error=None, what=""):
    if status == CGE_OK:
        return
    msg = ""
    if handle is not None and last_error is not None:
        raw = last_error(handle)
        msg = raw.decode() if raw else ""
    raise NativeLibraryError(f"{what} failed: {STATUS_NAMES.get(status, status)} {msg}".strip())
