"""CPU: libhps.so builds for gfx950, loads, and exports every entry point include/hps.h declares
(no compute calls without a GPU); the product path refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from hierarchicalprobabilistic3dhuman_amd import _capi, build as hps_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="hps.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hps_[a-z0-9_]+)\s*\(", text)))


def _exported(path):
    """EVERY defined dynamic symbol of the shared object (not only the hps_ prefix): the libraries are built with
    -fvisibility=hidden and a linker version script, so kernel stubs, kernel handles and C++ helpers must not appear."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_library_builds_and_exports_all_declared_symbols():
    path = hps_build.build(force=False, verbose=False)
    lib = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), "libhps.so does not export %s" % name
    assert set(declared) == set(_capi.EXPORTED_SYMBOLS), "ctypes prototypes out of sync with include/hps.h"
    # the product library is the product path only: exactly the declared ABI, no tuning switches, no legacy kernels
    assert _exported(path) == set(declared)
    assert not [n for n in declared if n.startswith("hps_dev_")]


def test_dev_library_is_separate_and_exports_the_dev_header():
    path = hps_build.build(force=False, verbose=False, dev=True)
    dev_declared = set(_declared_symbols("hps_dev.h")) - set(_declared_symbols())
    import devlib
    assert dev_declared == set(devlib.DEV_PROTOTYPES), "ctypes prototypes (tests/devlib.py) out of sync with include/hps_dev.h"
    assert _exported(path) == dev_declared | set(_declared_symbols())
    assert os.path.basename(path) == "libhps_dev.so" and path != _capi.LIB_PATH


def test_version_and_error_string():
    lib = _capi.load()
    assert lib.hps_version() == 502
    assert isinstance(lib.hps_last_error(), bytes)


def test_query_workspace_is_the_single_source_of_scratch_sizes():
    """hps_query_workspace(op, dims...) -- every caller-provided scratch size comes from the library."""
    q = _capi.query_workspace
    assert q(_capi.WS_CONV_SPLITK, 4, 64 * 8 * 8, 512) == 4 * 64 * 8 * 8 * 512 * 4 and q(_capi.WS_CONV_SPLITK, 1, 10, 10) == 0
    assert q(_capi.WS_SMPL_MP, 6528) == 6528 and q(_capi.WS_SMPL_MP, 6529) == 6656 and q(_capi.WS_SMPL_MP, 1) == 128
    assert q(_capi.WS_SMPL_XT, 6528, 224) == 224 * 6528 * 4
    assert q(_capi.WS_SMPL_A, 3, 24) == 3 * 24 * 12 * 4
    assert q(_capi.WS_SMPL_VPOSED, 2, 6890) == 2 * 20736 * 4
    assert q(_capi.WS_HEAD_F, 64, 5) == 64 * 5 * 9 * 4 and q(_capi.WS_HEAD_USV, 64, 5) == 64 * 5 * 21 * 4
    lib = _capi.load()
    # the K-slice buffer of the Winograd layer4 geometry: 4 slices x B x 8 x 8 x Cout floats; none for the 16x16-block geometry,
    # one slice when the 8 x 8 geometry has fewer than 32 chunks
    w = lib.hps_conv3x3_winograd_workspace
    assert w(64, 8, 8, 512, 512) == 4 * 64 * 64 * 512 * 4 and w(5, 8, 8, 64, 128) == 1 * 5 * 64 * 128 * 4
    assert w(64, 16, 16, 256, 256) == 0 and w(64, 64, 64, 64, 64) == 0 and w(0, 8, 8, 512, 512) == 0
    assert lib.hps_query_workspace(99, 1, 1, 1) == -1 and b"unknown item" in lib.hps_last_error()
    assert lib.hps_query_workspace(_capi.WS_SMPL_MP, -1, 0, 0) == -1
    with pytest.raises(_capi.HpsError):
        q(99)


def test_library_has_no_tuning_switches_and_the_header_says_what_is_global():
    """include/hps.h promised 'no synchronisation / no mutable global state' while hps_dev_* switches were
    exported.  The switches now exist only in libhps_dev.so and the header names the remaining exceptions."""
    text = open(os.path.join(ROOT, "include", "hps.h")).read()
    assert "hps_dev_" not in text and "HPS_DEV" not in text            # not even in a comment (no workspace kind, no constant
    assert "documented exceptions" in text and "hps_host_bind_lapack" in text       # whose only consumer is a dev symbol)
    assert not [n for n in _exported(_capi.LIB_PATH) if n.startswith("hps_dev_")]
    # ... and the product package names no dev symbol either: the dev prototypes and the helpers that call them live in tests/devlib.py;
    # what stays in the package is the loader (_capi.dev_library)
    pkg = os.path.join(ROOT, "hierarchicalprobabilistic3dhuman_amd")
    for name in sorted(f for f in os.listdir(pkg) if f.endswith(".py")):
        src = open(os.path.join(pkg, name)).read()
        assert "hps_dev_" not in src.replace("hps_dev_*", ""), name
        if name not in ("_capi.py", "build.py"):
            assert "dev_library" not in src and "libhps_dev" not in src, name
    lib = _capi.load()
    assert lib.hps_query_workspace(7, 64, 0, 0) == -1                  # the dev build's counter workspace is not a product item


def test_modules_are_copyable_and_reload_resets_caches(net_cpu):
    """ADVICE r1 (no GPU needed): deepcopy / pickle drop the device-bound caches; load_state_dict through the parent resets
    the encoder's prepared weights as well as the head's."""
    import copy
    import pickle
    net = copy.deepcopy(net_cpu[0])
    net._prepared, net.image_encoder._prepared, net._pinned_bufs = {"stale": 1}, {"stale": 2}, {"f": torch.zeros(4)}
    for clone in (copy.deepcopy(net), pickle.loads(pickle.dumps(net))):
        assert clone._prepared is None and clone.image_encoder._prepared is None and clone._pinned_bufs == {}
        assert len(clone.image_encoder._frames) == 0
    assert net._prepared == {"stale": 1}                          # the original keeps its own
    net.load_state_dict(net_cpu[1])
    assert net._prepared is None and net.image_encoder._prepared is None


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host before any launch."""
    lib = _capi.load()
    rc = lib.hps_smpl_lbs(None, 20670, None, None, None, 4, 24, None, None, 1, 6890, None)
    assert rc == -1 and b"null pointer" in lib.hps_last_error()
    rc = lib.hps_smpl_blend(None, None, None, None, 1, 1, 16, 128, 128, 128, None)
    assert rc == -1


def test_struct_mirror_and_composite_argument_checks():
    """The ctypes mirror of hps_enc_op has the layout the library was compiled with; the composites validate on the host."""
    lib = _capi.load()
    assert lib.hps_sizeof_enc_op() == ctypes.sizeof(_capi.EncOp)
    assert lib.hps_encoder_run(None, 0, None) == 0                     # empty list: nothing to do
    ops = (_capi.EncOp * 1)(_capi.EncOp(kind=99))
    assert lib.hps_encoder_run(ops, 1, None) == -1 and b"unknown op kind" in lib.hps_last_error()
    ops = (_capi.EncOp * 1)(_capi.EncOp(kind=_capi.ENC_CONV))          # null tensors are caught before any launch
    assert lib.hps_encoder_run(ops, 1, None) == -1 and b"null pointer" in lib.hps_last_error()
    rc = lib.hps_conv2d_bn_act_pad(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None,
                                   ctypes.c_void_p(16), 1, 8, 8, 0, 64, 64, 3, 3, 1, 1, 1, 1, 0, 0, 1, None, None)
    assert rc == -1 and b"halo" in lib.hps_last_error()                # input halo smaller than the padding


def test_effective_cpus_is_sane():
    from hierarchicalprobabilistic3dhuman_amd import sharding
    n = sharding.effective_cpus()
    assert isinstance(n, int) and 1 <= n <= (os.cpu_count() or 1)


def test_product_refuses_cpu_tensors(smpl_assets, net_cpu):
    from hierarchicalprobabilistic3dhuman_amd.smpl_official import SMPL
    from hierarchicalprobabilistic3dhuman_amd import rigid_transform_utils as rtu, sampling_utils as su
    smpl = SMPL(smpl_assets[0])
    with pytest.raises(_capi.HpsError):
        smpl(betas=torch.zeros(1, 10))
    with pytest.raises(_capi.HpsError):
        rtu.rot6d_to_rotmat(torch.zeros(2, 6))
    with pytest.raises(_capi.HpsError):
        su.pose_matrix_fisher_sampling_torch(torch.zeros(1, 23, 3, 3), torch.zeros(1, 23, 3), torch.zeros(1, 23, 3, 3), 2)
    with pytest.raises(_capi.HpsError):
        net_cpu[0](torch.zeros(1, 18, 256, 256))
