"""ctypes binding of the C ABI in include/bsuite_amd.h (libbsuite_amd.so: hand-written HIP, gfx950).

There is no CPU fallback: if the shared library cannot be loaded the import of this module raises,
and every environment in bsuite_amd.environments depends on it.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: its libamdhip64 is the HIP runtime we bind to)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, '_lib', 'libbsuite_amd.so')

FIRST, MID, LAST = 0, 1, 2
BSX_EINVAL, BSX_ENULL, BSX_EALIGN, BSX_ERANGE, BSX_EMODE, BSX_ENOMEM = -1, -2, -3, -4, -5, -6
WRAP_NONE, WRAP_SCALE, WRAP_NOISE, WRAP_SCALE_NOISE, WRAP_NOISE_SCALE = 0, 1, 2, 3, 4
COUNTER_SHARDS, COUNTER_STRIDE = 256, 16
DEEP_SEA_MAX_SIZE = 64
BANDIT_MAX_ACTIONS = 32
FUSED_CATCH_MAX_CELLS = 128   # BSX_FUSED_CATCH_MAX_CELLS (include/bsuite_amd.h)


class NativeLibraryError(RuntimeError):
  pass


def _load():
  override = os.environ.get('BSX_NATIVE_LIB')     # A/B of kernel variants: load this build instead
  if override:
    try:
      return ctypes.CDLL(override)
    except OSError as e:
      raise NativeLibraryError(f'cannot load BSX_NATIVE_LIB={override}: {e}') from e
  # Build in-tree on first use, and rebuild when a kernel source is newer than the library (a
  # stale .so silently running old kernels is worse than a slow import).  No toolchain and no
  # library -> fail loudly.
  try:
    from bsuite_amd import build as _build  # pylint: disable=import-outside-toplevel
    _build.build()
  except Exception as e:  # pylint: disable=broad-except
    if not os.path.exists(SO_PATH):
      raise NativeLibraryError(
          f'{SO_PATH} is missing and could not be built with hipcc ({e}). bsuite_amd has no CPU '
          'fallback: run `python -m bsuite_amd.build`.') from e
    # a rebuild was needed (a source is newer than the library) and failed: running the old kernels
    # silently would be worse than stopping.  BSX_ALLOW_STALE_LIB=1 overrides (e.g. read-only trees).
    if os.environ.get('BSX_ALLOW_STALE_LIB') != '1':
      raise NativeLibraryError(
          f'{SO_PATH} is older than its sources and rebuilding it failed ({e}); fix the build or set '
          'BSX_ALLOW_STALE_LIB=1 to load the stale library anyway.') from e
    import warnings  # pylint: disable=import-outside-toplevel
    warnings.warn(f'loading a STALE {SO_PATH}: rebuild failed ({e})')
  try:
    return ctypes.CDLL(SO_PATH)
  except OSError as e:
    raise NativeLibraryError(f'cannot load {SO_PATH}: {e}') from e


class Stream(ctypes.Structure):
  _fields_ = [('seed', ctypes.c_uint64), ('lane_offset', ctypes.c_uint64),
              ('step_index', ctypes.c_uint64), ('step_base', ctypes.c_void_p),
              ('mt_state', ctypes.c_void_p), ('mt_pos', ctypes.c_void_p),
              ('mt_gauss', ctypes.c_void_p), ('mt_has_gauss', ctypes.c_void_p)]


class RewardWrap(ctypes.Structure):
  _fields_ = [('kind', ctypes.c_int32), ('_pad', ctypes.c_int32), ('param', ctypes.c_double),
              ('seed', ctypes.c_uint64), ('mt_state', ctypes.c_void_p), ('mt_pos', ctypes.c_void_p),
              ('mt_gauss', ctypes.c_void_p), ('mt_has_gauss', ctypes.c_void_p), ('param2', ctypes.c_double)]


class TimeStepPtrs(ctypes.Structure):
  _fields_ = [('reward', ctypes.c_void_p), ('discount', ctypes.c_void_p),
              ('step_type', ctypes.c_void_p), ('observation', ctypes.c_void_p)]


class Logging(ctypes.Structure):
  _fields_ = [('steps', ctypes.c_void_p), ('episode', ctypes.c_void_p),
              ('total_return', ctypes.c_void_p), ('episode_len', ctypes.c_void_p),
              ('episode_return', ctypes.c_void_p), ('rows', ctypes.c_void_p),
              ('n_rows', ctypes.c_void_p), ('info', ctypes.c_void_p),
              ('log_points', ctypes.c_void_p), ('n_log_points', ctypes.c_int32),
              ('max_rows', ctypes.c_int32), ('n_info', ctypes.c_int32),
              ('log_by_step', ctypes.c_int32), ('log_every', ctypes.c_int32),
              ('_pad', ctypes.c_int32)]


class Call(ctypes.Structure):
  _fields_ = [('n_lanes', ctypes.c_int64), ('force_reset', ctypes.c_int32), ('n_steps', ctypes.c_int32),
              ('stream', Stream), ('wrap', RewardWrap), ('counters', ctypes.c_void_p),
              ('hip_stream', ctypes.c_void_p), ('logging', ctypes.POINTER(Logging)),
              ('obs_paint', ctypes.c_void_p), ('reward_f64', ctypes.c_void_p),
              ('state_alt', ctypes.c_void_p), ('action_ring', ctypes.c_int32), ('flags', ctypes.c_int32),
              ('row_scratch', ctypes.c_void_p)]


class DeepSeaCfg(ctypes.Structure):
  _fields_ = [('size', ctypes.c_int32), ('deterministic', ctypes.c_int32),
              ('move_cost', ctypes.c_double), ('inv_size', ctypes.c_double),
              ('mapping_bits', ctypes.c_uint32 * (DEEP_SEA_MAX_SIZE * DEEP_SEA_MAX_SIZE // 32))]


class CatchCfg(ctypes.Structure):
  _fields_ = [('rows', ctypes.c_int32), ('columns', ctypes.c_int32)]


class BanditCfg(ctypes.Structure):
  _fields_ = [('num_actions', ctypes.c_int32), ('_pad', ctypes.c_int32),
              ('rewards', ctypes.c_double * BANDIT_MAX_ACTIONS)]


class MemoryChainCfg(ctypes.Structure):
  _fields_ = [('memory_length', ctypes.c_int32), ('num_bits', ctypes.c_int32)]


class UmbrellaChainCfg(ctypes.Structure):
  _fields_ = [('chain_length', ctypes.c_int32), ('n_distractor', ctypes.c_int32)]


class DiscountingChainCfg(ctypes.Structure):
  _fields_ = [('bonus_chain', ctypes.c_int32), ('_pad', ctypes.c_int32)]


class CartpoleCfg(ctypes.Structure):
  _fields_ = [('swingup', ctypes.c_int32), ('last_step', ctypes.c_int32),
              ('height_threshold', ctypes.c_float), ('x_threshold', ctypes.c_float),
              ('theta_dot_threshold', ctypes.c_float), ('x_reward_threshold', ctypes.c_float),
              ('timescale', ctypes.c_float),
              ('mass_cart', ctypes.c_float), ('mass_pole', ctypes.c_float),
              ('length', ctypes.c_float), ('force_mag', ctypes.c_float),
              ('gravity', ctypes.c_float),
              ('move_cost', ctypes.c_double), ('init_range', ctypes.c_double),
              ('theta_offset', ctypes.c_double), ('time_frac', ctypes.c_void_p)]


class MnistCfg(ctypes.Structure):
  _fields_ = [('num_data', ctypes.c_int32), ('num_pixels', ctypes.c_int32),
              ('images', ctypes.c_void_p), ('labels', ctypes.c_void_p),
              ('pixel_lut', ctypes.c_float * 256)]


class MountainCarCfg(ctypes.Structure):
  _fields_ = [('max_steps', ctypes.c_int32), ('_pad', ctypes.c_int32)]


IMAGE_SMALL, IMAGE_BILINEAR = 0, 1
IMAGE_MAX_RADIUS = 64


class ImageCfg(ctypes.Structure):
  _fields_ = [('mode', ctypes.c_int32), ('in_rows', ctypes.c_int32), ('in_cols', ctypes.c_int32),
              ('out_rows', ctypes.c_int32), ('out_cols', ctypes.c_int32), ('tail', ctypes.c_int32),
              ('radius_y', ctypes.c_int32), ('radius_x', ctypes.c_int32),
              ('gauss_y', ctypes.c_double * (IMAGE_MAX_RADIUS + 1)),
              ('gauss_x', ctypes.c_double * (IMAGE_MAX_RADIUS + 1))]


lib = _load()

_P = ctypes.c_void_p
_SIGS = {
    'bsx_abi_version': ([], ctypes.c_int),
    'bsx_row_scratch_bytes': ([ctypes.c_int32, ctypes.c_int32, ctypes.c_int64], ctypes.c_int64),
    'bsx_bsuite_info': ([ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, _P, _P, ctypes.c_int32, ctypes.c_int32, _P, _P],
                        ctypes.c_int),
    'bsx_strerror': ([ctypes.c_int], ctypes.c_char_p),
    'bsx_calib_fill': ([_P, ctypes.c_int64, ctypes.c_int32, _P], ctypes.c_int),
    'bsx_calib_copy': ([_P, _P, ctypes.c_int64, ctypes.c_int32, _P], ctypes.c_int),
    'bsx_counter_add': ([_P, ctypes.c_uint64, _P], ctypes.c_int),
    'bsx_image_observation': ([ctypes.POINTER(ImageCfg), ctypes.c_int64, _P, _P, _P], ctypes.c_int),
    'bsx_stream_dump': ([ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_uint64,
                         ctypes.c_int32, ctypes.c_int32, _P, _P, _P], ctypes.c_int),
    'bsx_deep_sea_step': ([ctypes.POINTER(DeepSeaCfg), ctypes.POINTER(Call), _P, _P, TimeStepPtrs, _P],
                          ctypes.c_int),
    'bsx_catch_step': ([ctypes.POINTER(CatchCfg), ctypes.POINTER(Call), _P, _P, TimeStepPtrs, _P],
                       ctypes.c_int),
    'bsx_bandit_step': ([ctypes.POINTER(BanditCfg), ctypes.POINTER(Call), _P, _P, TimeStepPtrs, _P],
                        ctypes.c_int),
    'bsx_memory_chain_step': ([ctypes.POINTER(MemoryChainCfg), ctypes.POINTER(Call), _P, _P, _P,
                               TimeStepPtrs, _P], ctypes.c_int),
    'bsx_umbrella_chain_step': ([ctypes.POINTER(UmbrellaChainCfg), ctypes.POINTER(Call), _P, _P,
                                 TimeStepPtrs, _P], ctypes.c_int),
    'bsx_discounting_chain_step': ([ctypes.POINTER(DiscountingChainCfg), ctypes.POINTER(Call), _P, _P,
                                    TimeStepPtrs], ctypes.c_int),
    'bsx_cartpole_step': ([ctypes.POINTER(CartpoleCfg), ctypes.POINTER(Call), _P, _P, _P,
                           TimeStepPtrs, _P], ctypes.c_int),
    'bsx_mnist_step': ([ctypes.POINTER(MnistCfg), ctypes.POINTER(Call), _P, _P, TimeStepPtrs, _P],
                       ctypes.c_int),
    'bsx_mountain_car_step': ([ctypes.POINTER(MountainCarCfg), ctypes.POINTER(Call), _P, _P, _P,
                               TimeStepPtrs, _P], ctypes.c_int),
}
_G = ctypes.c_void_p   # bsx_group_t*
_SIGS.update({
    'bsx_group_create': ([ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_G)], ctypes.c_int),
    'bsx_group_commit': ([_G], ctypes.c_int),
    'bsx_group_step': ([_G, _P], ctypes.c_int),
    'bsx_group_phases': ([_G], ctypes.c_int),
    'bsx_group_small_class': ([ctypes.c_int32], ctypes.c_int),
    'bsx_group_step_phase': ([_G, ctypes.c_int32, _P], ctypes.c_int),
    'bsx_group_step_pipelined': ([_G, _G, _P], ctypes.c_int),
    'bsx_group_step_split': ([_G, _P], ctypes.c_int),
    'bsx_group_trace': ([_G, _P, ctypes.c_int64], ctypes.c_int),
    'bsx_group_destroy': ([_G], ctypes.c_int),
})
for _fam in ('deep_sea', 'catch', 'bandit', 'memory_chain', 'umbrella_chain', 'discounting_chain',
             'cartpole', 'mountain_car', 'mnist'):
  _step_args, _ = _SIGS[f'bsx_{_fam}_step']
  _SIGS[f'bsx_group_set_{_fam}'] = ([_G, ctypes.c_int32] + list(_step_args), ctypes.c_int)
FAMILY_IDS = dict(deep_sea=0, catch=1, bandit=2, memory_chain=3, umbrella_chain=4, discounting_chain=5,
                  cartpole=6, mountain_car=7, mnist=8, small_mixed=9, pair_mixed=10, sweep_mixed=11)
EXPORTED = tuple(sorted(_SIGS))
MISSING = []
for _name, (_args, _res) in _SIGS.items():
  try:
    _fn = getattr(lib, _name)
  except AttributeError:
    MISSING.append(_name)
    continue
  _fn.argtypes = _args
  _fn.restype = _res
if MISSING:
  raise NativeLibraryError(f'{SO_PATH} does not export {MISSING}; rebuild with `python -m bsuite_amd.build --force`')
ABI_VERSION = 12
CALL_STATE_TAGGED = 1   # BSX_CALL_STATE_TAGGED
if lib.bsx_abi_version() != ABI_VERSION:
  raise NativeLibraryError('ABI version mismatch between bsuite_amd/_native.py and libbsuite_amd.so')


def check(rc, what):
  if rc != 0:
    raise RuntimeError(f'{what} failed: {lib.bsx_strerror(rc).decode()} (code {rc})')
