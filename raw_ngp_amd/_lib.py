"""Loader of libngp_hip.so and the `_backend` shims.

The reference's op wrappers call `_backend.<fn>(tensors..., scalars...)` on three pybind11
modules (`_gridencoder`, `_shencoder`, `_raymarching_mob`, + `_freqencoder`).
The objects exported here (`gridencoder_backend`, `shencoder_backend`, `raymarching_backend`,
`freqencoder_backend`) offer the same attribute names and positional argument orders, convert
tensors to raw device pointers, append the caller's current HIP stream and raise RuntimeError
when the C ABI (include/ngp_hip.h) reports a failure.

There is NO fallback: if the HIP library is missing or a tensor is not a contiguous device
tensor of the right dtype the call fails loudly.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NGP_HIP_LIB") or os.path.join(_HERE, "csrc", "libngp_hip.so")   # override: kernel experiments

_f, _u, _i, _p, _d = ctypes.c_float, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_double

# name -> argument ctypes (the trailing stream pointer is appended automatically)
_SIGNATURES = {
    "ngp_grid_encode_forward": [_p, _p, _p, _p, _u, _u, _u, _u, _u, _f, _u, _p, _u, _i, _u],
    "ngp_grid_encode_backward": [_p, _p, _p, _p, _p, _u, _u, _u, _u, _u, _f, _u, _p, _p, _u, _i, _u],
    "ngp_grad_total_variation": [_p, _p, _p, _p, _f, _u, _u, _u, _u, _f, _u, _u, _i],
    "ngp_grad_weight_decay": [_p, _p, _p, _f, _u, _u, _u],
    "ngp_sh_encode_forward": [_p, _p, _u, _u, _u, _p],
    "ngp_sh_encode_backward": [_p, _p, _u, _u, _u, _p, _p],
    "ngp_freq_encode_forward": [_p, _u, _u, _u, _u, _p],
    "ngp_freq_encode_backward": [_p, _p, _u, _u, _u, _u, _p],
    "ngp_near_far_from_aabb": [_p, _p, _p, _u, _f, _p, _p],
    "ngp_sph_from_ray": [_p, _p, _f, _u, _p],
    "ngp_morton3D": [_p, _u, _p],
    "ngp_morton3D_invert": [_p, _u, _p],
    "ngp_packbits": [_p, _u, _f, _p],
    "ngp_flatten_rays": [_p, _u, _u, _p],
    "ngp_march_rays_train": [_p, _p, _p, _p, _f, _i, _f, _u, _u, _u, _u, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "ngp_composite_rays_train_forward": [_p, _p, _p, _p, _u, _u, _f, _p, _p, _p, _p],
    "ngp_composite_rays_train_backward": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _u, _u, _f, _p, _p],
    "ngp_march_rays": [_u, _u, _p, _p, _p, _p, _f, _i, _f, _u, _u, _u, _p, _p, _p, _p, _p, _p, _p],
    "ngp_composite_rays": [_u, _u, _f, _p, _p, _p, _p, _p, _p, _p, _p],
    "ngp_x_grid_encode_backward_binned": [_p, _p, _p, _p, _p, _u, _u, _u, _u, _f, _u, _u, _i, _u, _u, _u, _p,
                                          ctypes.c_size_t],
    "ngp_x_grid_backward_binned_prepare": [_p, _f, _p, _p, _u, _u, _u, _f, _u, _u, _i, _u, _u, _u, _i, _u, _i, _p,
                                           ctypes.c_size_t],
    "ngp_x_grid_backward_binned_apply": [_p, _p, _p, _p, _p, _u, _u, _u, _u, _f, _u, _u, _i, _u, _u, _u, _p,
                                         ctypes.c_size_t, _p, _p, _p, _p, _f, _f, _f, _i, _p],
    "ngp_x_grid_backward_binned_apply_list": [_p, _p, _p, _p, _p, _p, _u, _u, _u, _u, _f, _u, _u, _i, _u, _u, _u, _p,
                                              ctypes.c_size_t, _p, _p, _p, _p, _f, _f, _f, _i, _p],
    "ngp_x_grid_backward_binned_apply_mlp": [_p, _p, _p, _p, _p, _u, _u, _u, _u, _f, _u, _u, _i, _u, _u, _u, _p,
                                             ctypes.c_size_t, _p, _p, _p, _p, _f, _f, _f, _i,
                                             _u, _f, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p, _p, _p, _p, _u, _p, _f,
                                             _f, _f, _p, _p],
    "ngp_x_grid_backward_binned_apply_mlp_list": [_p, _p, _p, _p, _p, _p, _u, _u, _u, _u, _f, _u, _u, _i, _u, _u, _u, _p,
                                                  ctypes.c_size_t, _p, _p, _p, _p, _f, _f, _f, _i,
                                                  _u, _f, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p, _p, _p, _p, _u, _p,
                                                  _f, _f, _f, _p, _p],
    "ngp_x_grid_input_backward": [_p, _p, _p, _u, _u, _u, _u, _i],
    "ngp_x_grid_encode_forward_jac": [_p, _p, _p, _p, _u, _u, _u, _u, _u, _f, _u, _p, _u, _i, _u, _i],
    "ngp_x_mlp_prepare": [_p, _p, _p, _p, _p, _p, _p],
    "ngp_x_mlp_forward": [_p, _u, _p, _p, _u, _p, _p, _p],
    "ngp_x_mlp_forward_act": [_p, _u, _p, _p, _u, _p, _p, _p, _u, _u, _u, _f],
    "ngp_x_mlp_density_scatter": [_p, _u, _u, _p, _p, _p, _u, _u, _f],
    "ngp_x_mlp_backward_act": [_p, _u, _p, _p, _p, _p, _u, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t,
                               _p, _u, _u, _u, _f],
    "ngp_x_mlp_backward": [_p, _u, _p, _p, _p, _p, _u, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t],
    "ngp_x_mlp_backward_dirs": [_p, _u, _p, _p, _p, _p, _u, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t],
    "ngp_x_mlp_backward_list": [_p, _u, _p, _p, _p, _p, _u, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t,
                                _p],
    "ngp_x_mlp_reduce_dw": [_u, _f, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p, _p, _p, _p, _u, _p, _f, _f, _f, _p, _p],
    "ngp_x_mlp_rf_prepare": [_p, _p, _p, _p, _p, _p, _p],
    "ngp_x_mlp_rf_forward": [_p, _u, _p, _p, _p, _p, _u, _p, _p, _p],
    "ngp_x_mlp_rf_forward_act": [_p, _u, _p, _p, _p, _p, _u, _p, _p, _p, _u, _u, _f],
    "ngp_x_mlp_rf_backward_act": [_p, _u, _p, _p, _p, _p, _p, _p, _u, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                  ctypes.c_size_t, _p, _u, _u, _f],
    "ngp_x_mlp_rf_backward": [_p, _u, _p, _p, _p, _p, _p, _p, _u, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                              ctypes.c_size_t],
    "ngp_x_mlp_rf_backward_list": [_p, _u, _p, _p, _p, _p, _p, _p, _u, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                   ctypes.c_size_t, _p],
    "ngp_x_grid_encode_forward_slab": [_p, _f, _p, _p, _p, _p, _p, _u, _u, _u, _u, _f, _u, _u, _i, _u, _p, _u],
    "ngp_x_grid_encode_forward_slab_levels": [_p, _f, _p, _p, _p, _p, _p, _u, _u, _u, _u, _u, _f, _u, _u, _i, _u, _p],
    "ngp_x_grid_backward_binned_reduce_range": [_p, _p, _p, _u, _u, _f, _u, _u, _u, _p, ctypes.c_size_t, _i, _u, _u, _p],
    "ngp_x_grid_encode_forward_slab_jac": [_p, _f, _p, _p, _p, _p, _p, _u, _u, _u, _u, _f, _u, _u, _i, _u, _p, _u, _p],
    "ngp_x_grid_encode_forward_slab_placed": [_p, _f, _p, _p, _p, _p, _p, _u, _u, _u, _u, _f, _u, _u, _i, _u, _p, _u, _p, _p],
    "ngp_x_composite_hdr_train": [_p, _p, _f, _p, _p, _f, _p, _p, _p, _p, _u, _u, _f, _p, _p, _p, _p, _p, _p],
    "ngp_x_sample_rays_lit": [_p, _u, _u, _u, _u, _p, _f, _f, _f, _f, _u, ctypes.c_uint64, _p, _u, _p, _p, _p, _p, _p, _p,
                              _p, _p],
    "ngp_x_sample_rays_adaptive": [_p, _u, _u, _u, _u, _p, _f, _f, _f, _f, _u, ctypes.c_uint64, _p, _u, _p, _p, _p, _p, _p,
                                   _p, _p, _p, _p, _p, _p, _u, _p, _p],
    "ngp_x_composite_train_live": [_p, _p, _f, _p, _p, _f, _p, _f, _p, _p, _p, _p, _u, _u, _f, _p, _p, _p, _p, _p, _p],
    "ngp_x_composite_train_live_idx": [_p, _p, _f, _p, _p, _f, _p, _f, _p, _p, _p, _p, _u, _u, _f, _p, _p, _p, _p, _p, _p,
                                       _p, _p, _p, _p],
    "ngp_x_composite_train_terms": [_p, _p, _f, _p, _p, _f, _p, _f, _p, _f, _p, _p, _p, _p, _u, _u, _f, _p, _p, _p, _p, _p,
                                    _p, _p, _p, _p, _p, _p],
    "ngp_x_orientation_term": [_p, _p, _u, _u, _f, _p, _p, _p, _u, _p, _p],
    "ngp_x_orientation_term_act": [_p, _p, _u, _u, _f, _p, _p, _p, _u, _p, _p, _u, _f],
    "ngp_x_ray_gradients_terms": [_p, _p, _u, _u, _f, _p, _p, _p, _p, _p, _p, _p, _u, _u, _p, _p],
    "ngp_x_mlp_density_gradient": [_p, _u, _p, _u, _p, _p],
    "ngp_x_mlp_rf_density_gradient": [_p, _u, _p, _p, _u, _p, _p],
    "ngp_x_step_window": [_p, _u, _d, _f, _f, _u, _p, _p],
    "ngp_x_step_window_baa": [_p, _u, _d, _f, _f, _u, _p, _p],
    "ngp_x_slab_window": [_p, _u, _u, _p, _p, _u, _i],
    "ngp_x_ray_gradients": [_p, _p, _u, _u, _f, _p, _p, _p, _u, _u, _p, _p],
    "ngp_x_ray_gradients_list": [_p, _p, _u, _u, _f, _p, _p, _p, _p, _p, _u, _u, _p, _p],
    "ngp_x_pose_gradient": [_p, _p, _p, _u, _u, _u, _f, _f, _f, _f, _p],
    "ngp_x_pose_update": [_p, _p, _p, _u, _p, _p, _p, _f, _f, _f, _f, _f, _p, _p, _p],
    "ngp_x_composite_rays_train_forward": [_p, _p, _p, _p, _u, _u, _f, _p, _p, _p, _p],
    "ngp_x_composite_rays_train_backward": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _u, _u, _f, _p, _p],
    "ngp_x_composite_mse_backward": [_p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _u, _u, _f, _p, _p, _p],
    "ngp_x_composite_mse_train": [_p, _p, _f, _p, _p, _p, _p, _u, _u, _f, _p, _p, _p, _p, _p, _p],
    "ngp_x_composite_mse_train_idx": [_p, _p, _f, _p, _p, _p, _p, _u, _u, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "ngp_x_adam_step": [_p, _p, _p, _p, ctypes.c_uint64, _f, _d, _d, _f, _u, _i],
    "ngp_x_near_far_from_aabb_v2": [_p, _p, _p, _u, _f, _p, _p],
    "ngp_x_adam_step_dev": [_p, _p, _p, _p, ctypes.c_uint64, _p, _f, _f, _f, _i, _p],
    "ngp_x_schedule_step": [_p, _p, _d, _d, _d, _d],
    "ngp_x_step_begin": [_p, _p, _d, _d, _d, _d, _p, _p, _p, _p, _u, _u, _i, _p, _d, _d, _u],
    "ngp_x_mlp_forward_step_begin": [_p, _u, _p, _p, _u, _p, _p, _p, _p, _p, _d, _d, _d, _d, _p, _p, _p, _p, _u, _u, _i,
                                     _p, _d, _d, _u],
    "ngp_x_adam_step_dev2": [_p, _p, _p, _p, ctypes.c_uint64, _i, _p, _p, _p, _p, ctypes.c_uint64, _i, _p, _f, _f, _f, _i, _p],
    "ngp_x_counter_add": [_p, _u],
    "ngp_x_sample_rays": [_p, _u, _u, _u, _u, _p, _f, _f, _f, _f, _u, ctypes.c_uint64, _p, _u, _p, _p, _p, _p, _p, _p],
    "ngp_x_march_rays_train_backward": [_p, _p, _p, _p, _u, _u, _p, _p],
    "ngp_x_march_rays_train_arena": [_p, _p, _p, _p, _f, _i, _f, _u, _u, _u, _u, _p, _p, _p, _p, _u, _p, _p, _p,
                                     _p, _p, _p, _p, _p, _p, _p, _p, _u],
    "ngp_x_march_rays_train_arena_stage": [_p, _p, _p, _p, _f, _i, _f, _u, _u, _u, _u, _p, _p, _p, _p, _u, _p, _p, _p,
                                     _p, _p, _p, _p, _p, _p, _p, _p, _u, _i],
    "ngp_x_build_occupancy_index": [_p, _u, _u, _p],
    "ngp_x_density_grid_sample": [_p, _u, _f, _f, _u, _u, _i, ctypes.c_uint64, _p, _u, _p, ctypes.c_size_t, _p, _p],
    "ngp_x_density_grid_scatter": [_p, _p, _u, _p],
    "ngp_x_density_grid_update": [_p, _p, _u, _f, _p],
    "ngp_x_packbits_mean": [_p, _u, _p, _f, _p],
}

_lib = None


def load():
    """dlopen the in-tree HIP library (built by `python -c 'import __graft_entry__ as g; g.build()'`
    or `make -C raw_ngp_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"raw_ngp_amd: {LIB_PATH} is missing -- build it with `make -C raw_ngp_amd/csrc` "
                "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        lib.ngp_last_error.restype = ctypes.c_char_p
        lib.ngp_abi_version.restype = ctypes.c_int
        lib.ngp_x_grid_backward_workspace_bytes.argtypes = [_u, _u, _u]
        lib.ngp_x_grid_backward_workspace_bytes.restype = ctypes.c_size_t
        lib.ngp_x_grid_backward_binned_counts.argtypes = [_u, _u, _u, _u]
        lib.ngp_x_grid_backward_binned_counts.restype = ctypes.c_int
        lib.ngp_x_grid_backward_binned_geometry.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
        lib.ngp_x_grid_backward_binned_geometry.restype = ctypes.c_int
        lib.ngp_x_occupancy_index_bytes.argtypes = [_u, _u]
        lib.ngp_x_occupancy_index_bytes.restype = ctypes.c_size_t
        lib.ngp_x_density_grid_workspace_bytes.argtypes = [_u]
        lib.ngp_x_density_grid_workspace_bytes.restype = ctypes.c_size_t
        lib.ngp_x_mlp_image_bytes.argtypes = []
        lib.ngp_x_mlp_image_bytes.restype = ctypes.c_size_t
        lib.ngp_x_mlp_backward_workspace_bytes.argtypes = [_u]
        lib.ngp_x_mlp_backward_workspace_bytes.restype = ctypes.c_size_t
        lib.ngp_x_mlp_rf_image_bytes.argtypes = []
        lib.ngp_x_mlp_rf_image_bytes.restype = ctypes.c_size_t
        lib.ngp_x_mlp_rf_backward_workspace_bytes.argtypes = [_u]
        lib.ngp_x_mlp_rf_backward_workspace_bytes.restype = ctypes.c_size_t
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = list(args) + [_p]
            fn.restype = ctypes.c_int
        _lib = lib
    return _lib


def declared_symbols():
    return ["ngp_abi_version", "ngp_last_error", "ngp_x_grid_backward_workspace_bytes", "ngp_x_grid_backward_binned_counts", "ngp_x_grid_backward_binned_geometry",
            "ngp_x_mlp_image_bytes", "ngp_x_mlp_backward_workspace_bytes", "ngp_x_mlp_rf_image_bytes",
            "ngp_x_mlp_rf_backward_workspace_bytes", "ngp_x_occupancy_index_bytes",
            "ngp_x_density_grid_workspace_bytes"] + list(_SIGNATURES)


_DT = {"f": torch.float32, "i": torch.int32, "b": torch.uint8, "u": torch.int32,   # counters: int32 storage
       "h": torch.bfloat16}


def _ptr(t, kind, name, optional=False):
    if t is None:
        if optional:
            return None
        raise RuntimeError(f"{name} must not be None")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # wording of the reference's CHECK_CUDA
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != _DT[kind]:
        raise RuntimeError(f"{name} must be a {_DT[kind]} tensor, got {t.dtype}")
    return t.data_ptr()


class LossScaler:
    """torch.cuda.amp.GradScaler's state as eight device words the kernels read and settle themselves (include/ngp_hip.h,
    "Dynamic loss scale"; the reference: train_utils.py:404,897-904).  Defaults are GradScaler's: 2^16, x2 after 2000
    clean steps, x0.5 on overflow.  Nothing here reads the device; `state()` does, for logging and tests."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.words = torch.zeros(8, dtype=torch.float32, device=device)
        self.words[0], self.words[1] = float(init_scale), 1.0 / float(init_scale)
        self.counters = self.words.view(torch.int32)
        self.found = self.counters[2:3]             # the `skip` word of the optimiser kernels
        self.growth, self.backoff, self.interval = float(growth_factor), float(backoff_factor), int(growth_interval)

    def state(self):
        w, c = self.words.tolist(), self.counters.tolist()      # host read
        return dict(scale=w[0], found_inf=c[2], growth_tracker=c[3], steps_taken=c[4], steps_skipped=c[5])


def _scaler_ptr(scaler):
    """Device pointer of a LossScaler's words (or of a bare 8-element float32 tensor); None passes through."""
    if scaler is None:
        return None
    words = scaler.words if isinstance(scaler, LossScaler) else scaler
    if words.dtype != torch.float32 or words.numel() < 8:
        raise RuntimeError("loss scaler: eight float32 device words expected")
    return _ptr(words, "f", "loss_scaler")


def _scaler_args(scaler):
    """(words, growth, backoff, growth_interval) as the step_begin entry points take them."""
    if scaler is None:
        return None, 2.0, 0.5, 2000
    return _scaler_ptr(scaler), scaler.growth, scaler.backoff, scaler.interval


# Optional per-entry-point probe (bench.py): HIP events recorded on the launch stream around every call of ONE
# C symbol, plus the value of one integer argument (the number of samples the launch processes).  Events cannot be
# timed inside a captured graph, so the fused engine keeps the probed entry point out of its graphs (it asks
# `probed_symbol()` when it captures).
_probe = {"names": (), "arg": 0, "events": [], "every": 1, "phase": 0, "calls": {}}


def set_probe(names, units_arg=0, every=1, phase=0):
    """names: one C symbol or a tuple of them (e.g. the two halves of one operation); units_arg applies to the first.
    every = k times only every k-th call of each symbol (timing events drain the queue around the launch: sampling
    keeps the measurement from slowing down what it measures); phase: which of every k calls (call number % k == phase)."""
    if names is None:
        names = ()
    elif isinstance(names, str):
        names = (names,)
    every = max(int(every), 1)
    _probe.update(names=tuple(names), arg=units_arg, events=[], every=every, phase=int(phase) % every, calls={})


def probed_symbols():
    return _probe["names"]


class probe_paused:
    """Context manager: calls inside are neither timed nor counted (other uses of a probed entry point)."""

    def __enter__(self):
        self.saved = _probe["names"]
        _probe["names"] = ()

    def __exit__(self, *exc):
        _probe["names"] = self.saved
        return False


def probe_next_timed():
    """True when the next call of a probed symbol will be bracketed by timing events (see `every`)."""
    if not _probe["names"]:
        return False
    return _probe["calls"].get(_probe["names"][0], 0) % _probe["every"] == _probe["phase"]


def probe_untimed_run():
    """Number of upcoming calls of the probed symbols that will NOT be timed (a large number when nothing is probed)."""
    if not _probe["names"]:
        return 1 << 30
    k = _probe["calls"].get(_probe["names"][0], 0) % _probe["every"]
    return (_probe["phase"] - k) % _probe["every"]


def probe_skip(names=None):
    """Account for calls that happened inside a captured graph (where nothing can be timed)."""
    for n in (names or _probe["names"]):
        _probe["calls"][n] = _probe["calls"].get(n, 0) + 1


def probe_reset():
    """Forget the measurements taken so far, keep probing."""
    _probe["events"] = []


def probe_results(names=None):
    """(timed launches of the first of `names`, its total units, seconds summed over all of `names`) since
    set_probe() / probe_reset(); names defaults to every probed symbol."""
    torch.cuda.synchronize()
    names = tuple(names or _probe["names"])
    ev, first = [e for e in _probe["events"] if e[0] in names], (names or (None,))[0]
    return (sum(1 for n, *_ in ev if n == first), sum(u for n, _, _, u in ev if n == first),
            sum(a.elapsed_time(b) for _, a, b, _ in ev) * 1e-3)


def _call(name, anchor, *args, probe_as=None, probe_shift=0):
    lib = load()
    dev = anchor.device
    pname = probe_as or name      # (variants of one operation are probed under the operation's name)
    probing = pname in _probe["names"]
    timed = False
    if probing and not torch.cuda.is_current_stream_capturing():
        k = _probe["calls"].get(pname, 0)
        _probe["calls"][pname] = k + 1
        timed = k % _probe["every"] == _probe["phase"]
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        if timed:
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record(stream)
        rc = getattr(lib, name)(*args, stream.cuda_stream)
        if timed:
            stop.record(stream)
            units = int(args[_probe["arg"] + probe_shift]) if pname == _probe["names"][0] else 0
            _probe["events"].append((pname, start, stop, units))
    if rc != 0:
        raise RuntimeError(lib.ngp_last_error().decode())


class _GridBackend:
    """Stands in for `_gridencoder` (gridencoder/src/bindings.cpp:5-9)."""

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype,
                            align_corners, interp):
        _call("ngp_grid_encode_forward", inputs, _ptr(inputs, "f", "inputs"), _ptr(embeddings, "f", "embeddings"),
              _ptr(offsets, "i", "offsets"), _ptr(outputs, "f", "outputs"), B, D, C, L, max_level, float(S), H,
              _ptr(dy_dx, "f", "dy_dx", True), gridtype, int(bool(align_corners)), interp)

    # D = 3, C = 2 (the field's encoder) takes the atomic-free binned scatter; everything else, or
    # use_binned_backward = False, takes the reference-shaped float-atomic kernel.
    use_binned_backward = True
    _level_rows = {}   # id(offsets tensor) -> (weakref, rows of its largest level): one host read per tensor object

    @staticmethod
    def _max_level_rows(offsets):
        """Rows of the largest level (sizes the per-level LDS histograms).  Cached per tensor OBJECT -- a data pointer
        can be recycled by the allocator for a different table."""
        import weakref
        hit = _GridBackend._level_rows.get(id(offsets))
        if hit is not None and hit[0]() is offsets and hit[2] == offsets._version:
            return hit[1]
        o = offsets.detach().cpu()
        val = int((o[1:] - o[:-1]).max())
        if len(_GridBackend._level_rows) > 64:
            _GridBackend._level_rows.clear()
        _GridBackend._level_rows[id(offsets)] = (weakref.ref(offsets), val, offsets._version)
        return val

    @staticmethod
    def level_major_jacobian(B, D, C, L):
        """True when the op pair keeps dy_dx level-major (the route that does not go through the reference-shaped
        ngp_grid_encode_backward, which reads the reference layout)."""
        return bool(_GridBackend.use_binned_backward and D == 3 and C == 2 and B * L * 8 < 2 ** 32)

    @staticmethod
    def grid_encode_forward_jac(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype,
                                align_corners, interp, level_major):
        _call("ngp_x_grid_encode_forward_jac", inputs, _ptr(inputs, "f", "inputs"), _ptr(embeddings, "f", "embeddings"),
              _ptr(offsets, "i", "offsets"), _ptr(outputs, "f", "outputs"), B, D, C, L, max_level, float(S), H,
              _ptr(dy_dx, "f", "dy_dx"), gridtype, int(bool(align_corners)), interp, int(bool(level_major)))

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, max_level, S, H, dy_dx,
                             grad_inputs, gridtype, align_corners, interp, dy_dx_level_major=False):
        if _GridBackend.use_binned_backward and D == 3 and C == 2 and B * L * 8 < 2 ** 32:
            rows = embeddings.shape[0]
            nbytes = load().ngp_x_grid_backward_workspace_bytes(B, L, rows)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=grad.device)
            _call("ngp_x_grid_encode_backward_binned", grad, _ptr(grad, "f", "grad"), _ptr(inputs, "f", "inputs"),
                  _ptr(offsets, "i", "offsets"), _ptr(grad_embeddings, "f", "grad_embeddings"), None, B, B, L, max_level,
                  float(S), H, gridtype, int(bool(align_corners)), interp, rows,
                  _GridBackend._max_level_rows(offsets), ws.data_ptr(), nbytes)
            if dy_dx is not None and grad_inputs is not None:
                _call("ngp_x_grid_input_backward", grad, _ptr(grad, "f", "grad"), _ptr(dy_dx, "f", "dy_dx"),
                      _ptr(grad_inputs, "f", "grad_inputs"), B, D, C, L, int(bool(dy_dx_level_major)))
            return
        if dy_dx_level_major:
            raise RuntimeError("a level-major dy_dx needs the binned backward route")
        _call("ngp_grid_encode_backward", grad, _ptr(grad, "f", "grad"), _ptr(inputs, "f", "inputs"),
              _ptr(embeddings, "f", "embeddings"), _ptr(offsets, "i", "offsets"),
              _ptr(grad_embeddings, "f", "grad_embeddings"), B, D, C, L, max_level, float(S), H,
              _ptr(dy_dx, "f", "dy_dx", True), _ptr(grad_inputs, "f", "grad_inputs", True), gridtype,
              int(bool(align_corners)), interp)

    @staticmethod
    def grid_backward_binned(grad, inputs, offsets, grad_embeddings, B_dev, B_cap, grad_stride, L, max_level, S, H,
                             workspace, gridtype=0, align_corners=False, interp=0):
        """Extension entry for the fused step: caller-owned workspace, live count read from the device."""
        _call("ngp_x_grid_encode_backward_binned", grad, _ptr(grad, "f", "grad"), _ptr(inputs, "f", "inputs"),
              _ptr(offsets, "i", "offsets"), _ptr(grad_embeddings, "f", "grad_embeddings"),
              _ptr(B_dev, "i", "B_dev", True), B_cap, grad_stride, L, max_level, float(S), H, gridtype,
              int(bool(align_corners)), interp, grad_embeddings.shape[0], _GridBackend._max_level_rows(offsets),
              workspace.data_ptr(), workspace.numel())

    @staticmethod
    def grid_backward_binned_prepare(inputs, in_bound, offsets, n_rows, B_dev, B_cap, L, max_level, S, H, workspace,
                                     gridtype=0, align_corners=False, interp=0, single_segment=False, merge_max_res=0,
                                     stage=0):
        """Positions-only half (plan, count, scan).  in_bound > 0: `inputs` are world positions.
        single_segment: one reduce workgroup per chunk (what the fused-Adam apply needs)."""
        _call("ngp_x_grid_backward_binned_prepare", offsets, _ptr(inputs, "f", "inputs", stage != 0), float(in_bound),
              _ptr(offsets, "i", "offsets"), _ptr(B_dev, "i", "B_dev", True), B_cap, L, max_level, float(S), H, gridtype,
              int(bool(align_corners)), interp, n_rows, _GridBackend._max_level_rows(offsets),
              int(bool(single_segment)), int(merge_max_res), int(stage), workspace.data_ptr(), workspace.numel())

    @staticmethod
    def grid_backward_binned_apply(grad, inputs, offsets, grad_embeddings, B_dev, B_cap, grad_stride, L, max_level, S, H,
                                   workspace, gridtype=0, align_corners=False, interp=0, adam=None, overwrite=False,
                                   mlp_tail=None, sample_index=None, scaler=None, n_rows=None):
        """Fill + reduce on a workspace prepared for the same positions.  adam = (param, exp_avg, exp_avg_sq, hyper,
        beta1, beta2, eps): apply the optimiser inside the reduce kernel instead of writing grad_embeddings.
        overwrite: grad_embeddings = sums for every row (no +=); a bfloat16 grad_embeddings selects the 16-bit store.
        mlp_tail = (M, loss_scale, dws, workspace, adam, image), the arguments of mlp_backend.reduce_dw: that reduction
        rides along as extra workgroups of the fill kernel (ngp_x_grid_backward_binned_apply_mlp).
        sample_index (int32): the call runs over a LIST of samples -- `inputs` by sample, `grad` in list order,
        B_dev[0] entries (ngp_x_grid_backward_binned_apply_mlp_list / ..._apply_list).
        scaler: the eight device words of the dynamic loss scale (LossScaler.words); the reduce launch then settles the
        step -- overflow word, table untouched on overflow, the MLP weights' Adam step as ITS passengers."""
        sc = _scaler_ptr(scaler)
        # (grad_embeddings None without adam: FILL ONLY -- grid_backward_binned_reduce_range follows; n_rows = the table's rows)
        if adam is not None or grad_embeddings is not None:
            n_rows = (adam[0] if adam is not None else grad_embeddings).shape[0]
        elif n_rows is None:
            raise RuntimeError("grid_backward_binned_apply: a fill-only call needs n_rows")
        wire16 = grad_embeddings is not None and grad_embeddings.dtype == torch.bfloat16
        if wire16 and not overwrite:
            raise RuntimeError("a bfloat16 grad_embeddings needs overwrite=True")
        extra = [None, None, None, None, 0.0, 0.0, 0.0]
        if adam is not None:
            p_, m_, v_, hyper, b1, b2, eps = adam
            extra = [_ptr(p_, "f", "adam_param"), _ptr(m_, "f", "adam_exp_avg"), _ptr(v_, "f", "adam_exp_avg_sq"),
                     _ptr(hyper, "f", "adam_hyper"), float(b1), float(b2), float(eps)]
        args = [_ptr(grad, "f", "grad"), _ptr(inputs, "f", "inputs"), _ptr(offsets, "i", "offsets"),
                _ptr(grad_embeddings, "h" if wire16 else "f", "grad_embeddings", True),
                _ptr(B_dev, "i", "B_dev", True), B_cap, grad_stride, L, max_level, float(S), H, gridtype,
                int(bool(align_corners)), interp, n_rows, _GridBackend._max_level_rows(offsets),
                workspace.data_ptr(), workspace.numel(), *extra, 2 if wire16 else int(bool(overwrite))]
        if mlp_tail is None:
            if sample_index is not None:
                _call("ngp_x_grid_backward_binned_apply_list", grad, *args[:2], _ptr(sample_index, "i", "sample_index"), *args[2:],
                      sc, probe_as="ngp_x_grid_backward_binned_apply", probe_shift=1)
            else:
                _call("ngp_x_grid_backward_binned_apply", grad, *args, sc)
            return
        M, loss_scale, dws, mlp_ws, mlp_adam, image = mlp_tail
        mextra = [None, None, None, None, 0, None, 0.0, 0.0, 0.0]
        if mlp_adam is not None:
            p_, g_, m_, v_, hyper, b1, b2, eps = mlp_adam
            mextra = [_ptr(p_, "f", "mlp_adam_param"), _ptr(g_, "f", "mlp_adam_grad"), _ptr(m_, "f", "mlp_adam_exp_avg"),
                      _ptr(v_, "f", "mlp_adam_exp_avg_sq"), g_.numel(), _ptr(hyper, "f", "mlp_adam_hyper"), float(b1),
                      float(b2), float(eps)]
        tail = [M, float(loss_scale), *[_ptr(w, "f", f"dw{i + 1}") for i, w in enumerate(dws)], mlp_ws.data_ptr(), mlp_ws.numel(),
                *mextra, image.data_ptr() if image is not None else None, sc]
        if sample_index is not None:
            # (the probe and the engine know the operation by ONE name: the list variant reports as the plain one)
            _call("ngp_x_grid_backward_binned_apply_mlp_list", grad, *args[:2], _ptr(sample_index, "i", "sample_index"), *args[2:],
                  *tail, probe_as="ngp_x_grid_backward_binned_apply_mlp", probe_shift=1)
        else:
            _call("ngp_x_grid_backward_binned_apply_mlp", grad, *args, *tail)

    @staticmethod
    def grid_backward_binned_reduce_range(offsets, grad_embeddings, B_dev, B_cap, L, S, H, workspace, chunk_lo, chunk_hi,
                                          scaler=None):
        """The reduce half over the chunks [chunk_lo, chunk_hi) of a workspace that grid_backward_binned_apply(...,
        grad_embeddings=None, adam=None) -- fill only -- has filled: grad_embeddings[rows of those chunks] = sums
        (bfloat16 grad_embeddings: the 16-bit store).  level_chunks() turns a level range into a chunk range."""
        wire16 = grad_embeddings.dtype == torch.bfloat16
        _call("ngp_x_grid_backward_binned_reduce_range", grad_embeddings, _ptr(offsets, "i", "offsets"),
              _ptr(grad_embeddings, "h" if wire16 else "f", "grad_embeddings"), _ptr(B_dev, "i", "B_dev", True), B_cap, L, float(S),
              H, grad_embeddings.shape[0], _GridBackend._max_level_rows(offsets), workspace.data_ptr(), workspace.numel(),
              2 if wire16 else 1, int(chunk_lo), int(chunk_hi), _scaler_ptr(scaler))

    @staticmethod
    def level_chunks(offsets, level_lo, level_hi):
        """(first chunk, one past the last chunk) of the levels [level_lo, level_hi): chunks are numbered level-major,
        ceil(rows of the level / chunk rows) per level."""
        rows = _GridBackend.binned_geometry()[0]
        o = offsets.detach().cpu().tolist() if torch.is_tensor(offsets) else list(offsets)
        per = [-(-(o[l + 1] - o[l]) // rows) for l in range(len(o) - 1)]
        return sum(per[:level_lo]), sum(per[:level_hi])

    @staticmethod
    def backward_workspace_bytes(B, L, rows):
        return int(load().ngp_x_grid_backward_workspace_bytes(B, L, rows))

    @staticmethod
    def binned_geometry():
        """(table rows per chunk, samples per fill tile, record slots per tile-local region, tiles per reduce batch)."""
        out = (ctypes.c_uint32 * 4)()
        if load().ngp_x_grid_backward_binned_geometry(out) != 0:
            raise RuntimeError("ngp_x_grid_backward_binned_geometry failed")
        return tuple(int(v) for v in out)

    @staticmethod
    def backward_needs_counts(B, L, offsets):
        """Does grid_backward_binned_apply on a workspace of this shape need the per-chunk record counts (global-bins
        layout: a counting forward pass + scan), or only prepare's header reset (tile-local layout)?"""
        rows = int(offsets[-1]) if not torch.is_tensor(offsets) else None
        if rows is None:
            rows = int(offsets[-1].item())
        return bool(load().ngp_x_grid_backward_binned_counts(B, L, rows, _GridBackend._max_level_rows(offsets)))

    @staticmethod
    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
        _call("ngp_grad_total_variation", inputs, _ptr(inputs, "f", "inputs"), _ptr(embeddings, "f", "embeddings"),
              _ptr(grad, "f", "grad"), _ptr(offsets, "i", "offsets"), float(weight), B, D, C, L, float(S), H,
              gridtype, int(bool(align_corners)))

    @staticmethod
    def grad_weight_decay(embeddings, grad, offsets, weight, B, C, L):
        _call("ngp_grad_weight_decay", embeddings, _ptr(embeddings, "f", "embeddings"), _ptr(grad, "f", "grad"),
              _ptr(offsets, "i", "offsets"), float(weight), B, C, L)


class _SHBackend:
    """Stands in for `_shencoder` (shencoder/src/bindings.cpp:5-7)."""

    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
        _call("ngp_sh_encode_forward", inputs, _ptr(inputs, "f", "inputs"), _ptr(outputs, "f", "outputs"), B, D, C,
              _ptr(dy_dx, "f", "dy_dx", True))

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
        _call("ngp_sh_encode_backward", grad, _ptr(grad, "f", "grad"), _ptr(inputs, "f", "inputs"), B, D, C,
              _ptr(dy_dx, "f", "dy_dx"), _ptr(grad_inputs, "f", "grad_inputs"))


class _FreqBackend:
    """Stands in for `_freqencoder`."""

    @staticmethod
    def freq_encode_forward(inputs, B, D, deg, C, outputs):
        _call("ngp_freq_encode_forward", inputs, _ptr(inputs, "f", "inputs"), B, D, deg, C,
              _ptr(outputs, "f", "outputs"))

    @staticmethod
    def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
        _call("ngp_freq_encode_backward", grad, _ptr(grad, "f", "grad"), _ptr(outputs, "f", "outputs"), B, D, deg, C,
              _ptr(grad_inputs, "f", "grad_inputs"))


class _RayBackend:
    """Stands in for `_raymarching_mob` (raymarching/src/bindings.cpp:5-19)."""

    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        _call("ngp_near_far_from_aabb", rays_o, _ptr(rays_o, "f", "rays_o"), _ptr(rays_d, "f", "rays_d"),
              _ptr(aabb, "f", "aabb"), N, float(min_near), _ptr(nears, "f", "nears"), _ptr(fars, "f", "fars"))

    @staticmethod
    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        _call("ngp_sph_from_ray", rays_o, _ptr(rays_o, "f", "rays_o"), _ptr(rays_d, "f", "rays_d"), float(radius), N,
              _ptr(coords, "f", "coords"))

    @staticmethod
    def morton3D(coords, N, indices):
        _call("ngp_morton3D", coords, _ptr(coords, "i", "coords"), N, _ptr(indices, "i", "indices"))

    @staticmethod
    def morton3D_invert(indices, N, coords):
        _call("ngp_morton3D_invert", indices, _ptr(indices, "i", "indices"), N, _ptr(coords, "i", "coords"))

    @staticmethod
    def packbits(grid, N, density_thresh, bitfield):
        _call("ngp_packbits", grid, _ptr(grid, "f", "grid"), N, float(density_thresh),
              _ptr(bitfield, "b", "bitfield"))

    @staticmethod
    def flatten_rays(rays, N, M, res):
        _call("ngp_flatten_rays", rays, _ptr(rays, "i", "rays"), N, M, _ptr(res, "i", "res"))

    @staticmethod
    def march_rays_train(rays_o, rays_d, rays_ldir, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars,
                         xyzs, dirs, ts, ldirs, rays, counter, noises):
        _call("ngp_march_rays_train", rays_o, _ptr(rays_o, "f", "rays_o"), _ptr(rays_d, "f", "rays_d"),
              _ptr(rays_ldir, "f", "rays_ldir", True), _ptr(grid, "b", "grid"), float(bound), int(bool(contract)),
              float(dt_gamma), max_steps, N, C, H, _ptr(nears, "f", "nears"), _ptr(fars, "f", "fars"),
              _ptr(xyzs, "f", "xyzs", True), _ptr(dirs, "f", "dirs", True), _ptr(ts, "f", "ts", True),
              _ptr(ldirs, "f", "ldirs", True), _ptr(rays, "i", "rays"), _ptr(counter, "i", "counter"),
              _ptr(noises, "f", "noises"))

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, ts, rays, M, N, T_thresh, weights, weights_sum, depth, image):
        _call("ngp_composite_rays_train_forward", rays, _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"),
              _ptr(ts, "f", "ts"), _ptr(rays, "i", "rays"), M, N, float(T_thresh), _ptr(weights, "f", "weights"),
              _ptr(weights_sum, "f", "weights_sum"), _ptr(depth, "f", "depth"), _ptr(image, "f", "image"))

    @staticmethod
    def composite_rays_train_backward(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                      weights_sum, depth, image, M, N, T_thresh, grad_sigmas, grad_rgbs):
        _call("ngp_composite_rays_train_backward", rays, _ptr(grad_weights, "f", "grad_weights"),
              _ptr(grad_weights_sum, "f", "grad_weights_sum"), _ptr(grad_depth, "f", "grad_depth"),
              _ptr(grad_image, "f", "grad_image"), _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"),
              _ptr(ts, "f", "ts"), _ptr(rays, "i", "rays"), _ptr(weights_sum, "f", "weights_sum"),
              _ptr(depth, "f", "depth"), _ptr(image, "f", "image"), M, N, float(T_thresh),
              _ptr(grad_sigmas, "f", "grad_sigmas"), _ptr(grad_rgbs, "f", "grad_rgbs"))

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps, C, H,
                   grid, nears, fars, xyzs, dirs, ts, noises):
        _call("ngp_march_rays", rays_o, n_alive, n_step, _ptr(rays_alive, "i", "rays_alive"),
              _ptr(rays_t, "f", "rays_t"), _ptr(rays_o, "f", "rays_o"), _ptr(rays_d, "f", "rays_d"), float(bound),
              int(bool(contract)), float(dt_gamma), max_steps, C, H, _ptr(grid, "b", "grid"),
              _ptr(nears, "f", "nears"), _ptr(fars, "f", "fars"), _ptr(xyzs, "f", "xyzs"), _ptr(dirs, "f", "dirs"),
              _ptr(ts, "f", "ts"), _ptr(noises, "f", "noises"))

    @staticmethod
    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image):
        _call("ngp_composite_rays", rays_alive, n_alive, n_step, float(T_thresh),
              _ptr(rays_alive, "i", "rays_alive"), _ptr(rays_t, "f", "rays_t"), _ptr(sigmas, "f", "sigmas"),
              _ptr(rgbs, "f", "rgbs"), _ptr(ts, "f", "ts"), _ptr(weights_sum, "f", "weights_sum"),
              _ptr(depth, "f", "depth"), _ptr(image, "f", "image"))

    # ---- extensions (ngp_x_*) ------------------------------------------------------------
    @staticmethod
    def march_rays_train_backward(grad_xyzs, grad_dirs, ts, rays, N, M, grad_rays_o, grad_rays_d):
        _call("ngp_x_march_rays_train_backward", rays, _ptr(grad_xyzs, "f", "grad_xyzs"),
              _ptr(grad_dirs, "f", "grad_dirs", True), _ptr(ts, "f", "ts"), _ptr(rays, "i", "rays"), N, M,
              _ptr(grad_rays_o, "f", "grad_rays_o"), _ptr(grad_rays_d, "f", "grad_rays_d"))

    @staticmethod
    def march_rays_train_arena(rays_o, rays_d, rays_ldir, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears,
                               fars, noises, t_scratch, M_cap, xyzs, dirs, ts, ldirs, rays, counter, ray_idx,
                               occ_index=None, chain=None, stage=0):
        """chain: (chain f32 [chain_cap, N], code int16 [chain_cap, N], len int32 [N]) scratch of the chain-parallel
        first pass, or None for the serial one.  stage 1 / 2 (chain variant): the grid-independent first kernel alone /
        everything after it."""
        ch, code, clen = chain if chain is not None else (None, None, None)
        if chain is not None and (ch.shape[0] != code.shape[0] or ch.shape[1] < N or code.shape[1] != ch.shape[1]
                                  or clen.numel() < N or code.dtype != torch.int16 or counter.numel() < 3):
            raise RuntimeError("march_rays_train_arena: malformed chain buffers")
        if chain is not None and ch.shape[1] != N:
            raise RuntimeError("march_rays_train_arena: chain buffers must be laid out for exactly N rays")
        _call("ngp_x_march_rays_train_arena_stage" if stage else "ngp_x_march_rays_train_arena", rays_o,
              _ptr(rays_o, "f", "rays_o"), _ptr(rays_d, "f", "rays_d"),
              _ptr(rays_ldir, "f", "rays_ldir", True), _ptr(grid, "b", "grid"), float(bound), int(bool(contract)),
              float(dt_gamma), max_steps, N, C, H, _ptr(nears, "f", "nears"), _ptr(fars, "f", "fars"),
              _ptr(noises, "f", "noises"), _ptr(t_scratch, "f", "t_scratch"), M_cap, _ptr(xyzs, "f", "xyzs"),
              _ptr(dirs, "f", "dirs"), _ptr(ts, "f", "ts"), _ptr(ldirs, "f", "ldirs", True),
              _ptr(rays, "i", "rays"), _ptr(counter, "i", "counter"), _ptr(ray_idx, "i", "ray_idx", True),
              _ptr(occ_index, "i", "occ_index", True), _ptr(ch, "f", "chain", True),
              code.data_ptr() if code is not None else None, _ptr(clen, "i", "chain_len", True),
              ch.shape[0] if ch is not None else 0, *((int(stage),) if stage else ()))

    @staticmethod
    def occupancy_index_bytes(C, H):
        return int(load().ngp_x_occupancy_index_bytes(C, H))

    @staticmethod
    def build_occupancy_index(grid, C, H, index):
        """index: int32 tensor of occupancy_index_bytes(C, H) / 4 words (rebuilt after every packbits)."""
        if index.numel() * 4 < load().ngp_x_occupancy_index_bytes(C, H):
            raise RuntimeError("build_occupancy_index: index buffer too small")
        _call("ngp_x_build_occupancy_index", grid, _ptr(grid, "b", "grid"), C, H, _ptr(index, "i", "index"))


def field_activations(opt):
    """(color_act, density_act, beta, internal_act) of an options object for the fused MLP kernels, or None for a configuration
    they do not implement (network.py:31-34,111-135: colour clamped_exp / exp / sigmoid, density clamped_exp (= trunc_exp) /
    softplus, hidden layers relu / softplus)."""
    color = {"clamped_exp": 0, "exp": 1, "sigmoid": 2}.get(getattr(opt, "color_activation", "clamped_exp"))
    density = 0 if getattr(opt, "density_activation", "clamped_exp") == "clamped_exp" else 1
    internal = {"relu": 0, "softplus": 1}.get(getattr(opt, "internal_activation", "relu"))
    return None if color is None or internal is None else (color, density, float(getattr(opt, "beta", 1.0)), internal)


def _default_act(act):
    return act is None or (tuple(act[:2]) == (0, 0) and (len(act) < 4 or act[3] == 0))


class _MlpBackend:
    """Fused tiny-MLP field (extension; no counterpart among the reference's bindings)."""

    @staticmethod
    def image_bytes():
        return int(load().ngp_x_mlp_image_bytes())

    @staticmethod
    def prepare(weights, image):
        """weights: the six fp32 matrices (grid_mlp.net.0..2, view_mlp.net.0..2) in torch layout."""
        _call("ngp_x_mlp_prepare", image, *[_ptr(w, "f", f"w{i + 1}") for i, w in enumerate(weights)],
              image.data_ptr())

    @staticmethod
    def density_scatter(enc, stride, M, image, cells, tmp_cas, act=None):
        """Density of the M rows of `enc`, max-scattered into tmp_cas[cells[i]] (cells[i] < 0: dropped): the refresh's field
        evaluation and its scatter as one launch."""
        act = act or (0, 0, 1.0, 0)
        _call("ngp_x_mlp_density_scatter", enc, _ptr(enc, "f", "enc"), stride, M, image.data_ptr(), _ptr(cells, "i", "cells"),
              _ptr(tmp_cas, "f", "tmp_cas"), int(act[1]), int(act[3]) if len(act) > 3 else 0, float(act[2]))

    @staticmethod
    def forward(enc, stride, dirs, M_dev, M, image, sigma, rgb, step_begin=None, act=None):
        """step_begin = (step_counter, hyper, lr0, decay_steps, beta1, beta2, loss_out, samples_seen, sample_counter,
        binned_workspace, L, n_rows_total, single_segment), the arguments of engine_backend.step_begin: that bookkeeping
        rides along as one more workgroup of this launch (ngp_x_mlp_forward_step_begin).
        act = (color_act, density_act, beta[, internal_act]): the field's non-default activations (field_activations())."""
        args = [_ptr(enc, "f", "enc"), stride, _ptr(dirs, "f", "dirs", True), _ptr(M_dev, "i", "M_dev", True), M,
                image.data_ptr(), _ptr(sigma, "f", "sigma"), _ptr(rgb, "f", "rgb", True)]
        if not _default_act(act):
            if step_begin is not None:
                raise RuntimeError("mlp forward: the step_begin passenger rides on the default activations only")
            _call("ngp_x_mlp_forward_act", enc, *args, int(act[0]), int(act[1]), int(act[3]) if len(act) > 3 else 0,
                  float(act[2]), probe_as="ngp_x_mlp_forward")
            return
        if step_begin is None:
            _call("ngp_x_mlp_forward", enc, *args)
            return
        ctr, hyper, lr0, decay, b1, b2, loss_out, seen, counter, ws, L, n_rows, single, *scaling = step_begin
        if seen is not None and (seen.dtype != torch.int64 or not seen.is_cuda):
            raise RuntimeError("samples_seen must be an int64 CUDA tensor")
        _call("ngp_x_mlp_forward_step_begin", enc, *args, _ptr(ctr, "u", "step_counter"), _ptr(hyper, "f", "hyper"),
              float(lr0), float(decay), float(b1), float(b2), _ptr(loss_out, "f", "loss_out", True),
              seen.data_ptr() if seen is not None else None, _ptr(counter, "i", "sample_counter", True),
              ws.data_ptr() if ws is not None else None, int(L), int(n_rows), int(bool(single)),
              *_scaler_args(scaling[0] if scaling else None))


    @staticmethod
    def backward_workspace_bytes(M):
        return int(load().ngp_x_mlp_backward_workspace_bytes(M))

    @staticmethod
    def backward(enc, stride, dirs, dsigma, drgb, M_dev, M, image, loss_scale, denc, dws, workspace=None, ddirs=None,
                 sample_index=None, scaler=None, act=None):
        """dws: six pre-allocated fp32 tensors shaped like the weights (overwritten), or None to leave the partial
        sums in `workspace` for reduce_dw().  workspace: uint8 tensor of backward_workspace_bytes(M) (allocated per
        call when omitted).  ddirs [M,3] (optional): d loss / d (un-normalised view direction).  sample_index (int32,
        optional): run over a LIST of M_dev[0] samples; enc / dirs / dsigma / drgb by sample, denc in list order.
        scaler: the device words of the dynamic loss scale (LossScaler.words) instead of the static `loss_scale`."""
        nbytes = load().ngp_x_mlp_backward_workspace_bytes(M)
        ws = workspace if workspace is not None else torch.empty(nbytes, dtype=torch.uint8, device=enc.device)
        if ws.numel() < nbytes or not ws.is_cuda:
            raise RuntimeError("mlp backward: workspace too small")
        grads = [_ptr(w, "f", f"dw{i + 1}") for i, w in enumerate(dws)] if dws is not None else [None] * 6
        head = (_ptr(enc, "f", "enc"), stride, _ptr(dirs, "f", "dirs"), _ptr(dsigma, "f", "dsigma"), _ptr(drgb, "f", "drgb"),
                _ptr(M_dev, "i", "M_dev", True), M, image.data_ptr(), float(loss_scale), _ptr(denc, "f", "denc"))
        if not _default_act(act):                               # the field's non-default activations
            _call("ngp_x_mlp_backward_act", enc, *head[:7], _ptr(sample_index, "i", "sample_index", True), *head[7:],
                  _ptr(ddirs, "f", "ddirs", True), *grads, ws.data_ptr(), nbytes, _scaler_ptr(scaler), int(act[0]), int(act[1]),
                  int(act[3]) if len(act) > 3 else 0, float(act[2]), probe_as="ngp_x_mlp_backward")
        elif sample_index is not None or scaler is not None:
            _call("ngp_x_mlp_backward_list", enc, *head[:7], _ptr(sample_index, "i", "sample_index", True), *head[7:],
                  _ptr(ddirs, "f", "ddirs", True), *grads, ws.data_ptr(), nbytes, _scaler_ptr(scaler))
        elif ddirs is not None:
            _call("ngp_x_mlp_backward_dirs", enc, *head, _ptr(ddirs, "f", "ddirs"), *grads, ws.data_ptr(), nbytes)
        else:
            _call("ngp_x_mlp_backward", enc, *head, *grads, ws.data_ptr(), nbytes)

    @staticmethod
    def density_gradient(enc, stride, M_dev, M, image, denc):
        """denc [L, stride, 2] <- d h0 / d enc per sample, h0 = the density network's first output (sigma = trunc_exp(h0)):
        the MLP's part of autograd.grad(sigma, pos) in the orientation term (nerf/renderer.py:558-566)."""
        _call("ngp_x_mlp_density_gradient", enc, _ptr(enc, "f", "enc"), stride, _ptr(M_dev, "i", "M_dev", True), M,
              image.data_ptr(), _ptr(denc, "f", "denc"))

    @staticmethod
    def reduce_dw(M, loss_scale, dws, workspace, adam=None, image=None, scaler=None):
        """Second half of backward(..., dws=None): weight gradients from the partial sums left in `workspace`.
        adam = (param, grad, exp_avg, exp_avg_sq, hyper, beta1, beta2, eps) with dws views of the flat `grad`: also
        apply Adam to the flat weights, element by element; image (a prepared operand image): keep it in step with the
        updated weights."""
        extra = [None, None, None, None, 0, None, 0.0, 0.0, 0.0]
        if adam is not None:
            p_, g_, m_, v_, hyper, b1, b2, eps = adam
            extra = [_ptr(p_, "f", "adam_param"), _ptr(g_, "f", "adam_grad"), _ptr(m_, "f", "adam_exp_avg"),
                     _ptr(v_, "f", "adam_exp_avg_sq"), g_.numel(), _ptr(hyper, "f", "adam_hyper"), float(b1), float(b2),
                     float(eps)]
        _call("ngp_x_mlp_reduce_dw", workspace, M, float(loss_scale),
              *[_ptr(w, "f", f"dw{i + 1}") for i, w in enumerate(dws)], workspace.data_ptr(), workspace.numel(), *extra,
              image.data_ptr() if image is not None else None, _scaler_ptr(scaler))


class _MlpRfBackend:
    """Fused field of the light-conditioned configuration (rfield: 47 -> 80 -> 80 -> 3 view MLP), optional level window."""

    @staticmethod
    def image_bytes():
        return int(load().ngp_x_mlp_rf_image_bytes())

    @staticmethod
    def prepare(weights, image):
        _call("ngp_x_mlp_rf_prepare", image, *[_ptr(w, "f", f"w{i + 1}") for i, w in enumerate(weights)],
              image.data_ptr())

    @staticmethod
    def forward(enc, stride, dirs, ldirs, level_w, M_dev, M, image, sigma, rgb, act=None):
        """act = (color_act, density_act, beta[, internal_act = 0]): the field's non-default OUTPUT activations."""
        args = [_ptr(enc, "f", "enc"), stride, _ptr(dirs, "f", "dirs", True),
                _ptr(ldirs, "f", "ldirs", True), _ptr(level_w, "f", "level_w", True), _ptr(M_dev, "i", "M_dev", True), M,
                image.data_ptr(), _ptr(sigma, "f", "sigma"), _ptr(rgb, "f", "rgb", True)]
        if _default_act(act):
            _call("ngp_x_mlp_rf_forward", enc, *args)
        else:
            if len(act) > 3 and act[3]:
                raise RuntimeError("mlp_rf forward: the light-conditioned field's hidden layers are ReLU")
            _call("ngp_x_mlp_rf_forward_act", enc, *args, int(act[0]), int(act[1]), float(act[2]), probe_as="ngp_x_mlp_rf_forward")

    @staticmethod
    def density_gradient(enc, stride, M_dev, M, image, denc, level_w=None):
        """As mlp_backend.density_gradient, for this field's operand image (the density network is the same); level_w: the
        level window the field's kernels apply."""
        _call("ngp_x_mlp_rf_density_gradient", enc, _ptr(enc, "f", "enc"), stride, _ptr(level_w, "f", "level_w", True),
              _ptr(M_dev, "i", "M_dev", True), M, image.data_ptr(), _ptr(denc, "f", "denc"))

    @staticmethod
    def backward_workspace_bytes(M):
        return int(load().ngp_x_mlp_rf_backward_workspace_bytes(M))

    @staticmethod
    def backward(enc, stride, dirs, ldirs, level_w, dsigma, drgb, M_dev, M, image, loss_scale, denc, ddirs, dws,
                 workspace=None, sample_index=None, scaler=None, act=None):
        """sample_index (int32, optional): run over a LIST of M_dev[0] samples -- inputs and ddirs by sample, denc in list order.
        act: as forward."""
        nbytes = load().ngp_x_mlp_rf_backward_workspace_bytes(M)
        ws = workspace if workspace is not None else torch.empty(nbytes, dtype=torch.uint8, device=enc.device)
        if ws.numel() < nbytes or not ws.is_cuda:
            raise RuntimeError("mlp_rf backward: workspace too small")
        head = [_ptr(enc, "f", "enc"), stride, _ptr(dirs, "f", "dirs"),
                _ptr(ldirs, "f", "ldirs"), _ptr(level_w, "f", "level_w", True), _ptr(dsigma, "f", "dsigma"),
                _ptr(drgb, "f", "drgb"), _ptr(M_dev, "i", "M_dev", True), M]
        tail = [image.data_ptr(), float(loss_scale), _ptr(denc, "f", "denc"), _ptr(ddirs, "f", "ddirs", True),
                *[_ptr(w, "f", f"dw{i + 1}") for i, w in enumerate(dws)], ws.data_ptr(), nbytes]
        if not _default_act(act):
            if len(act) > 3 and act[3]:
                raise RuntimeError("mlp_rf backward: the light-conditioned field's hidden layers are ReLU")
            _call("ngp_x_mlp_rf_backward_act", enc, *head, _ptr(sample_index, "i", "sample_index", True), *tail,
                  _scaler_ptr(scaler), int(act[0]), int(act[1]), float(act[2]), probe_as="ngp_x_mlp_rf_backward")
        elif sample_index is not None or scaler is not None:
            _call("ngp_x_mlp_rf_backward_list", enc, *head, _ptr(sample_index, "i", "sample_index", True), *tail,
                  _scaler_ptr(scaler), probe_as="ngp_x_mlp_rf_backward")
        else:
            _call("ngp_x_mlp_rf_backward", enc, *head, *tail)


class _EngineBackend:
    """Kernels of the fused training step (extensions)."""

    @staticmethod
    def grid_encode_forward_slab(xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, L, max_level, S,
                                 H, gridtype=0, align_corners=False, interp=0, binned_workspace=None, dydx=None,
                                 level_cost=None):
        """dydx: optional [L, stride, 3, 2] slab receiving d out / d x01 (pose refinement: ray_gradients).
        level_cost: optional sequence of max_level positive floats, the relative cost of a tile of each level for these
        points -- the level -> XCD placement is balanced on it (placement only)."""
        args = (_ptr(xyzs, "f", "xyzs"), float(bound),
                _ptr(embeddings, "f", "embeddings"), _ptr(offsets, "i", "offsets"), _ptr(out, "f", "out"),
                _ptr(inputs01, "f", "inputs01", True), _ptr(B_dev, "i", "B_dev", True), B_cap, stride, L, max_level,
                float(S), H, gridtype, int(bool(align_corners)), interp,
                binned_workspace.data_ptr() if binned_workspace is not None else None, embeddings.shape[0])
        if level_cost is not None:
            if len(level_cost) != max_level:
                raise RuntimeError("grid_encode_forward_slab: level_cost must hold max_level floats")
            if dydx is not None and dydx.numel() < L * stride * 6:
                raise RuntimeError("grid_encode_forward_slab: dydx must hold L * stride * 3 * 2 floats")
            cost = (ctypes.c_float * max_level)(*[float(c) for c in level_cost])
            _call("ngp_x_grid_encode_forward_slab_placed", xyzs, *args, _ptr(dydx, "f", "dydx", True), cost,
                  probe_as="ngp_x_grid_encode_forward_slab_jac" if dydx is not None else "ngp_x_grid_encode_forward_slab")
        elif dydx is None:
            _call("ngp_x_grid_encode_forward_slab", xyzs, *args)
        else:
            if dydx.numel() < L * stride * 6:
                raise RuntimeError("grid_encode_forward_slab: dydx must hold L * stride * 3 * 2 floats")
            _call("ngp_x_grid_encode_forward_slab_jac", xyzs, *args, _ptr(dydx, "f", "dydx"))

    @staticmethod
    def grid_encode_forward_slab_levels(xyzs, bound, embeddings, offsets, out, inputs01, B_dev, B_cap, stride, L, level_lo,
                                        level_hi, S, H, gridtype=0, align_corners=False, interp=0, dydx=None):
        """grid_encode_forward_slab for the levels [level_lo, level_hi) only; the other levels' slab rows stay as they are
        (inputs01 is written by the call that covers level 0)."""
        if dydx is not None and dydx.numel() < L * stride * 6:
            raise RuntimeError("grid_encode_forward_slab_levels: dydx must hold L * stride * 3 * 2 floats")
        _call("ngp_x_grid_encode_forward_slab_levels", xyzs, _ptr(xyzs, "f", "xyzs"), float(bound),
              _ptr(embeddings, "f", "embeddings"), _ptr(offsets, "i", "offsets"), _ptr(out, "f", "out"),
              _ptr(inputs01, "f", "inputs01", True), _ptr(B_dev, "i", "B_dev", True), B_cap, stride, L, int(level_lo),
              int(level_hi), float(S), H, gridtype, int(bool(align_corners)), interp, _ptr(dydx, "f", "dydx", True))

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, ts, rays, M, N, T_thresh, weights, weights_sum, depth, image):
        _call("ngp_x_composite_rays_train_forward", rays, _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"),
              _ptr(ts, "f", "ts"), _ptr(rays, "i", "rays"), M, N, float(T_thresh), _ptr(weights, "f", "weights"),
              _ptr(weights_sum, "f", "weights_sum"), _ptr(depth, "f", "depth"), _ptr(image, "f", "image"))

    @staticmethod
    def composite_rays_train_backward(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                      weights_sum, depth, image, M, N, T_thresh, grad_sigmas, grad_rgbs):
        _call("ngp_x_composite_rays_train_backward", rays, _ptr(grad_weights, "f", "grad_weights"),
              _ptr(grad_weights_sum, "f", "grad_weights_sum"), _ptr(grad_depth, "f", "grad_depth"),
              _ptr(grad_image, "f", "grad_image"), _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"),
              _ptr(ts, "f", "ts"), _ptr(rays, "i", "rays"), _ptr(weights_sum, "f", "weights_sum"),
              _ptr(depth, "f", "depth"), _ptr(image, "f", "image"), M, N, float(T_thresh),
              _ptr(grad_sigmas, "f", "grad_sigmas"), _ptr(grad_rgbs, "f", "grad_rgbs"))

    @staticmethod
    def composite_mse_backward(gt_rgba, bg_rgb, bg_const, sigmas, rgbs, ts, rays, weights_sum, depth, image, M, N,
                               T_thresh, grad_sigmas, grad_rgbs, loss_out):
        _call("ngp_x_composite_mse_backward", rays, _ptr(gt_rgba, "f", "gt_rgba"), _ptr(bg_rgb, "f", "bg_rgb", True),
              float(bg_const), _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"), _ptr(ts, "f", "ts"),
              _ptr(rays, "i", "rays"), _ptr(weights_sum, "f", "weights_sum"), _ptr(depth, "f", "depth"),
              _ptr(image, "f", "image"), M, N, float(T_thresh), _ptr(grad_sigmas, "f", "grad_sigmas"),
              _ptr(grad_rgbs, "f", "grad_rgbs"), _ptr(loss_out, "f", "loss_out"))

    @staticmethod
    def composite_mse_train(gt_rgba, bg_rgb, bg_const, sigmas, rgbs, ts, rays, M, N, T_thresh, weights_sum, depth, image,
                            grad_sigmas, grad_rgbs, loss_out, live=None):
        """live = (live_n [N], live_idx [M], live_count [1]) int32: also list the samples in front of the early stop."""
        args = [_ptr(gt_rgba, "f", "gt_rgba"), _ptr(bg_rgb, "f", "bg_rgb", True),
                float(bg_const), _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"), _ptr(ts, "f", "ts"),
                _ptr(rays, "i", "rays"), M, N, float(T_thresh), _ptr(weights_sum, "f", "weights_sum"),
                _ptr(depth, "f", "depth"), _ptr(image, "f", "image"), _ptr(grad_sigmas, "f", "grad_sigmas"),
                _ptr(grad_rgbs, "f", "grad_rgbs"), _ptr(loss_out, "f", "loss_out")]
        if live is None:
            _call("ngp_x_composite_mse_train", rays, *args)
        else:
            _call("ngp_x_composite_mse_train_idx", rays, *args, _ptr(live[0], "i", "live_n"), _ptr(live[1], "i", "live_idx"),
                  _ptr(live[2], "i", "live_count"))

    @staticmethod
    def composite_hdr_train(gt_rgba, bg_rgb, bg_const, exposure, weight, inv_norm, sigmas, rgbs, ts, rays, M, N, T_thresh,
                            weights_sum, depth, image, grad_sigmas, grad_rgbs, loss_out):
        """composite_mse_train with the HDR loss of train_utils.py:512-536 (exposure [N], weight [N,3] or None)."""
        _call("ngp_x_composite_hdr_train", rays, _ptr(gt_rgba, "f", "gt_rgba"), _ptr(bg_rgb, "f", "bg_rgb", True),
              float(bg_const), _ptr(exposure, "f", "exposure"), _ptr(weight, "f", "weight", True), float(inv_norm),
              _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"), _ptr(ts, "f", "ts"),
              _ptr(rays, "i", "rays"), M, N, float(T_thresh), _ptr(weights_sum, "f", "weights_sum"),
              _ptr(depth, "f", "depth"), _ptr(image, "f", "image"), _ptr(grad_sigmas, "f", "grad_sigmas"),
              _ptr(grad_rgbs, "f", "grad_rgbs"), _ptr(loss_out, "f", "loss_out"))

    @staticmethod
    def composite_train_live(gt_rgba, bg_rgb, bg_const, exposure, weight, inv_norm, n_live, sigmas, rgbs, ts, rays, M, N,
                             T_thresh, weights_sum, depth, image, grad_sigmas, grad_rgbs, loss_out, lambda_entropy=0.0,
                             live=None, sample_term=None, lambda_sample=0.0, term_weight=None):
        """composite_mse_train (exposure None) / composite_hdr_train over the first n_live[0] ray slots (None: all), plus
        lambda_entropy * mean entropy of the rays' accumulated opacity (train_utils.py:554-557).
        live = (live_n [N], live_idx [M], live_count [1], live_off [N] or None) int32: also list the samples in front of the
        early stop (and where each ray's entries start).
        sample_term [M]: loss += lambda_sample * sum_i weights[i] * sample_term[i] (the orientation term); term_weight [M]
        (optional) <- lambda_sample * weights."""
        if sample_term is not None:
            lv = live if live is not None else (None, None, None, None)
            _call("ngp_x_composite_train_terms", rays, _ptr(gt_rgba, "f", "gt_rgba"), _ptr(bg_rgb, "f", "bg_rgb", True),
                  float(bg_const), _ptr(exposure, "f", "exposure", True), _ptr(weight, "f", "weight", True), float(inv_norm),
                  _ptr(n_live, "i", "n_live", True), float(lambda_entropy), _ptr(sample_term, "f", "sample_term"),
                  float(lambda_sample), _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"), _ptr(ts, "f", "ts"),
                  _ptr(rays, "i", "rays"), M, N, float(T_thresh), _ptr(weights_sum, "f", "weights_sum"),
                  _ptr(depth, "f", "depth"), _ptr(image, "f", "image"), _ptr(grad_sigmas, "f", "grad_sigmas"),
                  _ptr(grad_rgbs, "f", "grad_rgbs"), _ptr(loss_out, "f", "loss_out"), _ptr(lv[0], "i", "live_n", True),
                  _ptr(lv[1], "i", "live_idx", True), _ptr(lv[2], "i", "live_count", True),
                  _ptr(lv[3], "i", "live_off", True), _ptr(term_weight, "f", "term_weight", True),
                  probe_as="ngp_x_composite_train_live")
            return
        args = [_ptr(gt_rgba, "f", "gt_rgba"), _ptr(bg_rgb, "f", "bg_rgb", True),
                float(bg_const), _ptr(exposure, "f", "exposure", True), _ptr(weight, "f", "weight", True), float(inv_norm),
                _ptr(n_live, "i", "n_live", True), float(lambda_entropy), _ptr(sigmas, "f", "sigmas"), _ptr(rgbs, "f", "rgbs"),
                _ptr(ts, "f", "ts"),
                _ptr(rays, "i", "rays"), M, N, float(T_thresh), _ptr(weights_sum, "f", "weights_sum"),
                _ptr(depth, "f", "depth"), _ptr(image, "f", "image"), _ptr(grad_sigmas, "f", "grad_sigmas"),
                _ptr(grad_rgbs, "f", "grad_rgbs"), _ptr(loss_out, "f", "loss_out")]
        if live is None:
            _call("ngp_x_composite_train_live", rays, *args)
        else:
            _call("ngp_x_composite_train_live_idx", rays, *args, _ptr(live[0], "i", "live_n"), _ptr(live[1], "i", "live_idx"),
                  _ptr(live[2], "i", "live_count"), _ptr(live[3], "i", "live_off", True))

    @staticmethod
    def step_window(step_counter, step_offset, iters, start_annealing, end_annealing, L, level_w, flags=None, baa=False):
        """baa: the BAA-NGP weights (network.py:77-97) instead of BARF's (:99-109)."""
        _call("ngp_x_step_window_baa" if baa else "ngp_x_step_window", level_w, _ptr(step_counter, "u", "step_counter"),
              int(step_offset), float(iters), float(start_annealing), float(end_annealing), int(L),
              _ptr(level_w, "f", "level_w"), _ptr(flags, "i", "flags", True))

    @staticmethod
    def slab_window(slab, stride, L, level_w, M_dev, M, backward=False, scale_only=False):
        """BAA-NGP blend on the level-major encoder slab, in place (backward: its adjoint on the gradient slab);
        scale_only: the BARF window f'_l = w_l f_l (self-adjoint)."""
        _call("ngp_x_slab_window", slab, _ptr(slab, "f", "slab"), int(stride), int(L), _ptr(level_w, "f", "level_w"),
              _ptr(M_dev, "i", "M_dev", True), int(M), 2 if scale_only else int(bool(backward)))

    @staticmethod
    def orientation_term(dh_denc, dydx, stride, L, bound, sigmas, dirs, M_dev, M, term, dterm_ddirs=None, act=None):
        """term [M] <- min(0, n . -v)^2 per sample with n = (-normalize(d sigma / d xyz) + 1) / 2 (nerf/renderer.py:558-571);
        dh_denc: mlp_backend.density_gradient's slab, dydx: the Jacobian slab of grid_encode_forward_slab.
        dterm_ddirs [M,3] (optional) <- d term / d dirs.  act: the field's activations (a softplus density changes d sigma / d h0)."""
        args = [_ptr(dh_denc, "f", "dh_denc"), _ptr(dydx, "f", "dydx"), stride, L, float(bound),
                _ptr(sigmas, "f", "sigmas"), _ptr(dirs, "f", "dirs"), _ptr(M_dev, "i", "M_dev", True), M, _ptr(term, "f", "term"),
                _ptr(dterm_ddirs, "f", "dterm_ddirs", True)]
        if act is not None and int(act[1]) != 0:
            _call("ngp_x_orientation_term_act", term, *args, int(act[1]), float(act[2]), probe_as="ngp_x_orientation_term")
        else:
            _call("ngp_x_orientation_term", term, *args)

    @staticmethod
    def ray_gradients(denc, dydx, stride, L, bound, ddirs, ts, rays, N, M, grad_rays_o, grad_rays_d, live=None, terms=None):
        """live = (live_n, live_off): the backward ran over the list of live samples -- denc in list order.
        terms = (term_weight [M], dterm_ddirs [M,3]): per sample their product joins the direction gradient."""
        if terms is not None:
            lv = live if live is not None else (None, None)
            _call("ngp_x_ray_gradients_terms", rays, _ptr(denc, "f", "denc"), _ptr(dydx, "f", "dydx"), stride, L, float(bound),
                  _ptr(ddirs, "f", "ddirs", True), _ptr(ts, "f", "ts"), _ptr(rays, "i", "rays"), _ptr(lv[0], "i", "live_n", True),
                  _ptr(lv[1], "i", "live_off", True), _ptr(terms[0], "f", "term_weight"), _ptr(terms[1], "f", "dterm_ddirs"),
                  N, M, _ptr(grad_rays_o, "f", "grad_rays_o"), _ptr(grad_rays_d, "f", "grad_rays_d"),
                  probe_as="ngp_x_ray_gradients")
            return
        if live is not None:
            _call("ngp_x_ray_gradients_list", rays, _ptr(denc, "f", "denc"), _ptr(dydx, "f", "dydx"), stride, L, float(bound),
                  _ptr(ddirs, "f", "ddirs", True), _ptr(ts, "f", "ts"), _ptr(rays, "i", "rays"), _ptr(live[0], "i", "live_n"),
                  _ptr(live[1], "i", "live_off"), N, M, _ptr(grad_rays_o, "f", "grad_rays_o"),
                  _ptr(grad_rays_d, "f", "grad_rays_d"), probe_as="ngp_x_ray_gradients")
            return
        _call("ngp_x_ray_gradients", rays, _ptr(denc, "f", "denc"), _ptr(dydx, "f", "dydx"), stride, L, float(bound),
              _ptr(ddirs, "f", "ddirs", True), _ptr(ts, "f", "ts"), _ptr(rays, "i", "rays"), N, M,
              _ptr(grad_rays_o, "f", "grad_rays_o"), _ptr(grad_rays_d, "f", "grad_rays_d"))

    @staticmethod
    def pose_gradient(index, grad_rays_o, grad_rays_d, N, V, W, intrinsics, grad_pose):
        fx, fy, cx, cy = [float(v) for v in intrinsics]
        _call("ngp_x_pose_gradient", index, _ptr(index, "i", "index"), _ptr(grad_rays_o, "f", "grad_rays_o"),
              _ptr(grad_rays_d, "f", "grad_rays_d"), N, V, W, fx, fy, cx, cy, _ptr(grad_pose, "f", "grad_pose"))

    @staticmethod
    def pose_update(xi, base, grad_pose, flags, exp_avg, exp_avg_sq, lr0, gamma, beta1, beta2, eps, refined, grad_xi=None,
                    scaler=None):
        """grad_pose None: only refined = compose(exp(xi), base); else one Adam step on xi first (when flags[0] != 0)."""
        _call("ngp_x_pose_update", xi, _ptr(xi, "f", "xi"), _ptr(base, "f", "base"), _ptr(grad_pose, "f", "grad_pose", True),
              xi.shape[0], _ptr(flags, "i", "flags", True), _ptr(exp_avg, "f", "exp_avg", True),
              _ptr(exp_avg_sq, "f", "exp_avg_sq", True), float(lr0), float(gamma), float(beta1), float(beta2), float(eps),
              _ptr(refined, "f", "refined"), _ptr(grad_xi, "f", "grad_xi", True), _scaler_ptr(scaler))

    @staticmethod
    def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, zero_grad=False):
        _call("ngp_x_adam_step", param, _ptr(param, "f", "param"), _ptr(grad, "f", "grad"),
              _ptr(exp_avg, "f", "exp_avg"), _ptr(exp_avg_sq, "f", "exp_avg_sq"), param.numel(), float(lr),
              float(beta1), float(beta2), float(eps), int(step), int(bool(zero_grad)))

    @staticmethod
    def adam_step_dev(param, grad, exp_avg, exp_avg_sq, hyper, beta1, beta2, eps, zero_grad=False, skip=None):
        """skip: a one-element int32 device tensor (LossScaler.found); non-zero at run time = the launch does nothing."""
        _call("ngp_x_adam_step_dev", param, _ptr(param, "f", "param"), _ptr(grad, "f", "grad"),
              _ptr(exp_avg, "f", "exp_avg"), _ptr(exp_avg_sq, "f", "exp_avg_sq"), param.numel(),
              _ptr(hyper, "f", "hyper"), float(beta1), float(beta2), float(eps), int(bool(zero_grad)),
              _ptr(skip, "i", "skip", True))

    @staticmethod
    def schedule_step(step_counter, hyper, lr0, decay_steps, beta1, beta2):
        _call("ngp_x_schedule_step", hyper, _ptr(step_counter, "u", "step_counter"), _ptr(hyper, "f", "hyper"),
              float(lr0), float(decay_steps), float(beta1), float(beta2))

    @staticmethod
    def step_begin(step_counter, hyper, lr0, decay_steps, beta1, beta2, loss_out=None, samples_seen=None,
                   sample_counter=None, binned_workspace=None, L=0, n_rows_total=0, single_segment=False, scaling=None):
        """scaling: a LossScaler -- the previous step is settled (GradScaler.update) and Adam's t follows the steps taken."""
        if samples_seen is not None and (samples_seen.dtype != torch.int64 or not samples_seen.is_cuda):
            raise RuntimeError("samples_seen must be an int64 CUDA tensor")
        _call("ngp_x_step_begin", hyper, _ptr(step_counter, "u", "step_counter"), _ptr(hyper, "f", "hyper"), float(lr0),
              float(decay_steps), float(beta1), float(beta2), _ptr(loss_out, "f", "loss_out", True),
              samples_seen.data_ptr() if samples_seen is not None else None,
              _ptr(sample_counter, "i", "sample_counter", True),
              binned_workspace.data_ptr() if binned_workspace is not None else None, int(L), int(n_rows_total),
              int(bool(single_segment)), *_scaler_args(scaling))

    @staticmethod
    def adam_step_dev2(a, b, hyper, beta1, beta2, eps, skip=None):
        """a, b = (param, grad, exp_avg, exp_avg_sq, zero_grad) of two tensors updated by one launch; a's gradient may
        be bfloat16 (the data-parallel wire format)."""
        args = []
        a16 = a[1].dtype == torch.bfloat16
        for name, (p_, g_, m_, v_, z_) in (("a", a), ("b", b)):
            if g_.numel() != p_.numel():
                raise RuntimeError(f"grad_{name} and param_{name} differ in size")
            args += [_ptr(p_, "f", f"param_{name}"), _ptr(g_, "h" if a16 and name == "a" else "f", f"grad_{name}"),
                     _ptr(m_, "f", f"exp_avg_{name}"), _ptr(v_, "f", f"exp_avg_sq_{name}"), p_.numel(), int(bool(z_))]
        _call("ngp_x_adam_step_dev2", hyper, *args, _ptr(hyper, "f", "hyper"), float(beta1), float(beta2), float(eps),
              int(a16), _ptr(skip, "i", "skip", True))

    @staticmethod
    def counter_add(counter, delta=1):
        _call("ngp_x_counter_add", counter, _ptr(counter, "u", "counter"), int(delta))

    @staticmethod
    def sample_rays(images, poses, intrinsics, N, seed, draw, rays_o, rays_d, gt_rgba, noises=None, bg_rgb=None,
                    index=None, view_ldirs=None, rays_ldir=None, adaptive=None, exposure=None):
        """`draw`: int32 device tensor (read at run time) or a Python int.  view_ldirs [V,3] + rays_ldir [N,3]: per-ray
        light directions of the light-conditioned configuration.  exposure = (view_exposure [V], out [N])."""
        V, H, W, C = images.shape
        fx, fy, cx, cy = [float(v) for v in intrinsics]
        on_dev = torch.is_tensor(draw)
        args = (_ptr(images, "b", "images"), V, H, W, C, _ptr(poses, "f", "poses"), fx, fy,
                cx, cy, N, int(seed) & (2 ** 64 - 1), _ptr(draw, "u", "draw") if on_dev else None,
                0 if on_dev else int(draw) & 0xffffffff, _ptr(rays_o, "f", "rays_o"), _ptr(rays_d, "f", "rays_d"),
                _ptr(gt_rgba, "f", "gt_rgba"), _ptr(noises, "f", "noises", True), _ptr(bg_rgb, "f", "bg_rgb", True),
                _ptr(index, "i", "index", True))
        if adaptive is not None or exposure is not None:
            # (prev_samples, prev_live, live, num_points): see ngp_x_sample_rays_adaptive
            prev_samples, prev_live, live, num_points = adaptive if adaptive is not None else (None, None, None, 0)
            vexp, out_exp = exposure if exposure is not None else (None, None)
            _call("ngp_x_sample_rays_adaptive", images, *args, _ptr(view_ldirs, "f", "view_ldirs", True),
                  _ptr(rays_ldir, "f", "rays_ldir", True), _ptr(prev_samples, "i", "prev_samples", True),
                  _ptr(prev_live, "i", "prev_live", True), _ptr(live, "i", "live", True), int(num_points),
                  _ptr(vexp, "f", "view_exposure", True), _ptr(out_exp, "f", "exposure", True))
        elif view_ldirs is None and rays_ldir is None:
            _call("ngp_x_sample_rays", images, *args)
        else:
            _call("ngp_x_sample_rays_lit", images, *args, _ptr(view_ldirs, "f", "view_ldirs"),
                  _ptr(rays_ldir, "f", "rays_ldir"))

    @staticmethod
    def density_grid_workspace_bytes(H):
        return int(load().ngp_x_density_grid_workspace_bytes(H))

    @staticmethod
    def density_grid_sample(grid_cas, H, span, half, n_uniform, n_occupied, full, seed, draw, workspace, indices, xyzs):
        """`draw`: int32 device tensor or a Python int (as in sample_rays)."""
        on_dev = torch.is_tensor(draw)
        _call("ngp_x_density_grid_sample", grid_cas, _ptr(grid_cas, "f", "grid_cas"), H, float(span), float(half),
              n_uniform, n_occupied, int(bool(full)), int(seed) & (2 ** 64 - 1),
              _ptr(draw, "u", "draw") if on_dev else None, 0 if on_dev else int(draw) & 0xffffffff,
              _ptr(workspace, "b", "workspace"), workspace.numel(), _ptr(indices, "i", "indices"),
              _ptr(xyzs, "f", "xyzs"))

    @staticmethod
    def density_grid_scatter(indices, sigmas, n, tmp_cas):
        _call("ngp_x_density_grid_scatter", sigmas, _ptr(indices, "i", "indices"), _ptr(sigmas, "f", "sigmas"), n,
              _ptr(tmp_cas, "f", "tmp_cas"))

    @staticmethod
    def density_grid_update(grid, tmp, decay, stats):
        _call("ngp_x_density_grid_update", grid, _ptr(grid, "f", "grid"), _ptr(tmp, "f", "tmp"), grid.numel(),
              float(decay), _ptr(stats, "f", "stats"))

    @staticmethod
    def packbits_mean(grid, stats, density_thresh, bitfield):
        _call("ngp_x_packbits_mean", grid, _ptr(grid, "f", "grid"), bitfield.numel(), _ptr(stats, "f", "stats"),
              float(density_thresh), _ptr(bitfield, "b", "bitfield"))

    @staticmethod
    def near_far_from_aabb_v2(rays_o, rays_d, aabb, N, min_near, nears, fars):
        _call("ngp_x_near_far_from_aabb_v2", rays_o, _ptr(rays_o, "f", "rays_o"), _ptr(rays_d, "f", "rays_d"),
              _ptr(aabb, "f", "aabb"), N, float(min_near), _ptr(nears, "f", "nears"), _ptr(fars, "f", "fars"))


gridencoder_backend = _GridBackend()
engine_backend = _EngineBackend()
mlp_backend = _MlpBackend()
mlp_rf_backend = _MlpRfBackend()
shencoder_backend = _SHBackend()
freqencoder_backend = _FreqBackend()
raymarching_backend = _RayBackend()
