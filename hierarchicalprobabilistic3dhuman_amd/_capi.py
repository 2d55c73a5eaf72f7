"""ctypes binding of libhps.so (C ABI: include/hps.h).

PyTorch is only plumbing here: tensors own device memory, ``data_ptr()`` and the current HIP stream
are handed to the library.  There is no CPU implementation behind these calls -- if the library or a
HIP device is missing the call fails loudly.

``dev_library()`` is a context manager for tests / tests/dev only: inside it every call goes to libhps_dev.so
(include/hps_dev.h: the product code plus earlier kernel generations, alternate variants, tuning switches).
"""
import contextlib
import ctypes
import os

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libhps.so")
DEV_LIB_PATH = os.path.join(_PKG_DIR, "libhps_dev.so")

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int

# name -> argtypes (restype is always int unless listed in _RESTYPES)
_PROTOTYPES = {
    "hps_version": [],
    "hps_last_error": [],
    "hps_stream_create_cu_partition": [_I, _I, _c.POINTER(_P)],
    "hps_stream_destroy": [_P],
    "hps_smpl_pose_prep": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _I, _P],
    "hps_smpl_blend": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hps_smpl_lbs": [_P, _I, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P],
    "hps_smpl_mesh_fused": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "hps_smpl_mesh_fused_picks": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "hps_smpl_mesh_fused_shared_shape": [_P] * 8 + [_I, _I, _P] + [_I] * 5 + [_P, _P, _I, _P],
    "hps_smpl_split_bf16x3_bytes": [_I, _I],
    "hps_smpl_split_bf16x3_mesh_tile": [],
    "hps_smpl_split_bf16x3": [_P, _I, _I, _I, _I, _P, _P],
    "hps_smpl_mesh_fused_shared_shape_bf16x3": [_P] * 8 + [_I, _I, _P] + [_I] * 4 + [_P, _P, _I, _P],
    "hps_smpl_v_shaped": [_P, _I, _P, _I, _P, _P, _I, _I, _P],
    "hps_smpl_mesh_fused_np": [_I],
    "hps_smpl_joints": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P],
    "hps_vertex_uncertainty": [_P, _P, _I, _I, _I, _P],
    "hps_joints_and_uncertainty": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _I, _P],
    "hps_query_workspace": [_I, _c.c_int64, _c.c_int64, _c.c_int64],
    "hps_mf_sample": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _c.c_float, _c.c_float, _P, _P, _P, _c.c_uint64,
                      _c.c_int64, _P, _I, _P, _P, _P, _P],
    "hps_infer_assemble": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hps_quat_to_rotmat": [_P, _P, _I, _P],
    "hps_rot6d_to_rotmat": [_P, _P, _I, _P],
    "hps_batch_rodrigues": [_P, _P, _I, _P],
    "hps_linear": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "hps_head_trunk": [_P] + [_I] + [_P] * 14 + [_I] * 7 + [_P],
    "hps_head_joint_level": [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_float, _P, _P, _I, _I, _P],
    "hps_head_joint_level_svd": [_P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_float, _P, _P, _P, _P, _I, _I, _I, _P],
    "hps_svd3_packed": [_P, _P, _I, _I, _P],
    "hps_host_svd3_emulated": [_P, _P, _I, _I],
    "hps_host_svd_flavor": [],
    "hps_host_svd3_packed": [_P, _P, _I, _I],
    "hps_host_bind_lapack": [_c.c_char_p],
    "hps_head_svd_finish": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "hps_canny_edges": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _c.c_float, _I, _P],
    "hps_canny_edge_map": [_P, _P, _I, _P, _c.c_int64, _I, _I, _I, _I, _c.c_float, _I, _P],
    "hps_proxy_rep": [_P, _P, _P, _P, _I, _I, _I, _I, _c.c_float, _P],
    "hps_pointset_errors": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "hps_heatmaps_to_joints2d": [_P, _P, _P, _I, _I, _I, _c.c_float, _P],
    "hps_sample_joints2d_error": [_P, _P, _I, _P, _P, _P, _c.c_float, _P, _I, _I, _P],
    "hps_conv2d_bn_act_pad": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "hps_conv2d_bn_act_pad_down": [_P] * 9 + [_I] * 14 + [_P, _P],
    "hps_conv3x3_winograd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "hps_conv3x3_winograd_workspace": [_I, _I, _I, _I, _I],
    "hps_stem_phase_frames_bytes": [_I, _I, _I],
    "hps_stem_phase_split": [_P, _P, _I, _I, _I, _I, _P],
    "hps_proxy_rep_phase_frames": [_P, _P, _P, _P, _I, _I, _I, _I, _c.c_float, _P],
    "hps_stem_winograd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "hps_stem_pool_side_bytes": [_I, _I, _I],
    "hps_stem_winograd_pooled": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "hps_stem_winograd_pooled_nchw": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "hps_sums_f64": [_P, _P, _P, _I, _c.c_double, _P, _P, _P, _P],
    "hps_sizeof_enc_op": [],
    "hps_encoder_run": [_P, _I, _P],
    "hps_head_pose_levels": [_P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_float, _P, _P, _P, _P, _P, _P,
                             _P, _P, _I, _I, _I, _I, _P],
    "hps_nchw_to_padded_nhwc": [_P, _P, _I, _I, _I, _I, _I, _P],
    "hps_nchw_to_padded_nhwc_generic": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "hps_maxpool3x3s2_pad": [_P, _P, _I, _I, _I, _I, _I, _P],
    "hps_global_avgpool_pad": [_P, _P, _I, _I, _I, _I, _I, _P],
}
_RESTYPES = {"hps_last_error": _c.c_char_p, "hps_smpl_split_bf16x3_bytes": _c.c_size_t, "hps_query_workspace": _c.c_int64, "hps_conv3x3_winograd_workspace": _c.c_size_t,
             "hps_stem_phase_frames_bytes": _c.c_size_t, "hps_stem_pool_side_bytes": _c.c_size_t}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None
_dev_lib = None
_use_dev = False


class EncOp(_c.Structure):
    """include/hps.h: hps_enc_op."""
    _fields_ = [("kind", _I), ("x", _P), ("w", _P), ("scale", _P), ("shift", _P), ("residual", _P), ("y", _P),
                ("splitk_ws", _P)] + [(n, _I) for n in ("B", "H", "W", "ipad", "Cin", "Cout", "KH", "KW", "stride", "pad", "opad",
                                                       "relu", "row_mode", "variant", "ksplit")] + [
                    ("w_down", _P), ("scale_down", _P), ("shift_down", _P), ("y_down", _P)]


ENC_RELAYOUT, ENC_CONV, ENC_MAXPOOL, ENC_AVGPOOL, ENC_CONV_WINOGRAD, ENC_STEM_SPLIT, ENC_STEM_WINOGRAD, ENC_RELAYOUT_GENERIC = 0, 1, 2, 3, 4, 5, 6, 7
ENC_STEM_WINOGRAD_POOLED, ENC_STEM_WINOGRAD_POOLED_NCHW, ENC_CONV_DOWN = 8, 9, 10
SVD_HOST, SVD_DEVICE, SVD_DEVICE_FMA = 0, 1, 2
SVD_ROUNDING_REFERENCE, SVD_ROUNDING_FMA = 0, 1
HEAD_WIDE_WORKGROUPS = 0x100


class HpsError(RuntimeError):
    pass


def _open(path, prototypes, what):
    if not os.path.exists(path):
        raise HpsError(
            "%s is missing (%s). Build it with `python -m hierarchicalprobabilistic3dhuman_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path." % (what, path))
    lib = ctypes.CDLL(path)
    for name, argtypes in prototypes.items():
        fn = getattr(lib, name)          # AttributeError if the build is stale: fail loudly
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _I)
    # the head's host SVD uses the MKL sgesdd_ PyTorch itself links (same routine as the reference's torch.svd)
    torch_cpu = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so")
    if os.path.exists(torch_cpu):
        lib.hps_host_bind_lapack(torch_cpu.encode())
    return lib


def load(dev=False):
    """dlopen libhps.so (dev=True: libhps_dev.so) and attach prototypes; raises if the library has not been built."""
    global _lib, _dev_lib
    if dev:
        if _dev_lib is None:
            _dev_lib = _open(DEV_LIB_PATH, dict(_PROTOTYPES, **_dev_prototypes()), "libhps_dev.so")
        return _dev_lib
    if _lib is None:
        _lib = _open(LIB_PATH, _PROTOTYPES, "libhps.so")
    return _lib


def _dev_prototypes():
    """ctypes prototypes of the dev library's extra entry points (include/hps_dev.h).  They are test infrastructure and live on the
    tests' side of the tree -- tests/devlib.py, loaded here by path -- together with the helpers that call them; the package itself
    names no hps_dev_* symbol."""
    import importlib.util
    path = os.path.join(os.path.dirname(_PKG_DIR), "tests", "devlib.py")
    if not os.path.exists(path):
        raise HpsError("libhps_dev.so is test infrastructure: its prototypes live in tests/devlib.py, which is missing (%s)" % path)
    spec = importlib.util.spec_from_file_location("hps_tests_devlib_prototypes", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.DEV_PROTOTYPES


@contextlib.contextmanager
def dev_library():
    """Route every call inside the block to libhps_dev.so (tests and tests/dev only)."""
    global _use_dev
    prev, _use_dev = _use_dev, True
    try:
        yield load(dev=True)
    finally:
        _use_dev = prev


def require_device(t, what="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HpsError(
            "%s must live on a HIP (MI355X) device; this package has no CPU path "
            "(the CPU oracle under oracle/ is test infrastructure only)" % what)
    if t.device.index != torch.cuda.current_device():
        # kernels are launched on the CURRENT device's current stream (stream()): a tensor of another device would be
        # addressed from the wrong GPU.  The reference's `device` arguments become torch.cuda.set_device(device).
        raise HpsError("%s lives on cuda:%d but the current device is cuda:%d; call torch.cuda.set_device(%d) "
                       "(or use `with torch.cuda.device(...)`) before calling into libhps"
                       % (what, t.device.index, torch.cuda.current_device(), t.device.index))


def ptr(t, dtype=torch.float32, what="tensor"):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    require_device(t, what)
    if t.dtype != dtype:
        raise HpsError("%s: expected %s, got %s" % (what, dtype, t.dtype))
    if not t.is_contiguous():
        raise HpsError("%s must be contiguous" % what)
    return _P(t.data_ptr())


def iptr(t, what="index tensor"):
    return ptr(t, torch.int32, what)


def stream():
    return _P(torch.cuda.current_stream().cuda_stream)


def cu_partition_stream(first_cu, num_cus):
    """torch stream restricted to CUs [first_cu, first_cu + num_cus) of every XCD (include/hps.h: hps_stream_create_cu_partition).
    The stream lives as long as the process (a handful are ever created: one pair per InferencePipeline)."""
    s = _P()
    call("hps_stream_create_cu_partition", int(first_cu), int(num_cus), _c.byref(s))
    if not _partition_streams:
        import atexit
        atexit.register(_destroy_partition_streams)
    _partition_streams.append(s.value)
    return torch.cuda.ExternalStream(s.value)


_partition_streams = []


def _destroy_partition_streams():
    """At interpreter exit: drain and destroy the CU-partition streams this process created (a tool that hooks the runtime's teardown --
    rocprofv3 -- otherwise finds live streams it never saw created by hipStreamCreate and crashes in its own finaliser)."""
    try:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        lib = load()
        while _partition_streams:
            lib.hps_stream_destroy(_P(_partition_streams.pop()))
    except Exception:
        pass


_svd_flavor = None
_svd_flavor_exact = None


def svd_flavor():
    """Rounding flavour (include/hps.h: HPS_SVD_ROUNDING_*) of the in-kernel SVD that reproduces THIS host's LAPACK (the MKL
    sgesdd_ behind torch.svd) bit for bit -- calibrated once per process by hps_host_svd_flavor.  If neither flavour matches
    (another LAPACK build), the reference-BLAS rounding is used and a warning says so."""
    global _svd_flavor, _svd_flavor_exact
    if _svd_flavor is None:
        fl = load(dev=_use_dev).hps_host_svd_flavor()
        _svd_flavor_exact = fl in (SVD_ROUNDING_REFERENCE, SVD_ROUNDING_FMA)
        if not _svd_flavor_exact:
            import warnings
            warnings.warn("libhps: neither rounding flavour of the in-kernel 3x3 SVD reproduces this host's LAPACK sgesdd bit for "
                          "bit; using reference rounding (expect about one differently signed singular-vector pair in 10^4 "
                          "matrices relative to torch.svd on this host; svd_mode='host' is exact)")
            fl = SVD_ROUNDING_REFERENCE
        _svd_flavor = fl
    return _svd_flavor


def svd_flavor_is_exact():
    """True when the calibrated flavour reproduces this host's LAPACK bit for bit (the normal case with PyTorch's MKL)."""
    svd_flavor()
    return bool(_svd_flavor_exact)


def call(name, *args):
    lib = load(dev=_use_dev)
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.hps_last_error()
        raise HpsError("%s failed (code %d): %s" % (name, rc, msg.decode() if msg else ""))


WS_CONV_SPLITK, WS_SMPL_MP, WS_SMPL_XT, WS_SMPL_A, WS_SMPL_VPOSED, WS_HEAD_F, WS_HEAD_USV = range(7)


def query_workspace(what, d0=0, d1=0, d2=0):
    """include/hps.h: hps_query_workspace -- bytes (HPS_WS_SMPL_MP: a count) of a caller-provided scratch buffer."""
    n = load(dev=_use_dev).hps_query_workspace(int(what), int(d0), int(d1), int(d2))
    if n < 0:
        raise HpsError("hps_query_workspace(%d, %d, %d, %d) failed" % (what, d0, d1, d2))
    return int(n)


def f32c(t):
    """fp32 contiguous view/copy on the same device."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
