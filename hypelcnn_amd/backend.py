"""ctypes binding of libhypel_hip.so (include/hypel.h) -- the only compute backend of the product.

There is deliberately NO CPU fallback here: if the HIP library is missing or there is no GPU,
`HipBackend()` raises.  (tests/ inject a numpy emulation of this same interface to exercise the
host-side planner without a GPU; that emulation lives under tests/, not in the package.)

A backend exposes one method per C-ABI entry point.  Pointer arguments are `Ref`s
(flat torch tensor + element offset); `bind(name, args)` resolves them once and returns a
zero-argument callable, so a planned step is replayed as a flat list of pre-bound C calls.
"""
import ctypes
import os
import time
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HYPEL_LIB_PATH: an alternative build of the same ABI (A/B experiments on one GPU box)
LIB_PATH = os.environ.get("HYPEL_LIB_PATH") or os.path.join(_HERE, "csrc", "libhypel_hip.so")

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
GEMM_BM = 128
ABI_VERSION = 6  # include/hypel.h HYPEL_ABI_VERSION: a library built from other headers is refused at load time

SEG_DTYPE = np.dtype([("a_off", "<i8"), ("b_off", "<i8"), ("k", "<i4"), ("reserved", "<i4")])
GROUP_DTYPE = np.dtype([("c_off", "<i8"), ("seg_begin", "<i4"), ("seg_count", "<i4"), ("rows", "<i4"),
                        ("reserved", "<i4")])
LOSS_TERM_DTYPE = np.dtype([("a_off", "<i8"), ("b_off", "<i8"), ("da_off", "<i8"), ("db_off", "<i8"), ("lda", "<i8"),
                            ("ldb", "<i8"), ("ldda", "<i8"), ("lddb", "<i8"), ("rows", "<i8"), ("mode", "<i4"), ("c", "<i4"),
                            ("acc_da", "<i4"), ("acc_db", "<i4"), ("target", "<f4"), ("gcoef", "<f4"), ("pscale", "<f4"),
                            ("slot", "<i4")])
LOSS_NONE = -(1 << 63)
MTILE_DTYPE = np.dtype([("c_off", "<i8"), ("a_off0", "<i8"), ("b_off0", "<i8"), ("m0", "<i4"), ("rows", "<i4"),
                        ("n0", "<i4"), ("n", "<i4"), ("seg_begin", "<i4"), ("seg_count", "<i4"), ("k0", "<i4"),
                        ("flags", "<i4"), ("lda", "<i4"), ("ldb", "<i4"), ("ldc", "<i4"), ("reserved", "<i4")])
REDUCE_ENTRY_DTYPE = np.dtype([("partial_off", "<i8"), ("out_off", "<i8"), ("stride", "<i8"), ("count", "<i8"),
                               ("n_splits", "<i4"), ("flags", "<i4")])
TILE_DTYPE = np.dtype([("group", "<i4"), ("m0", "<i4"), ("rows", "<i4"), ("seg_begin", "<i4"), ("seg_count", "<i4"),
                       ("k0", "<i4"), ("c_off", "<i8"), ("a_off0", "<i8"), ("b_off0", "<i8"),
                       ("flags", "<i4"), ("n", "<i4")])
TILE_PLAIN = 1  # include/hypel.h HYPEL_TILE_PLAIN: K-slice partial (no bias / accumulate / shortcut gather)
COPY_BLOCK_DTYPE = np.dtype([("src_off", "<i8"), ("dst_off", "<i8"), ("rows", "<i4"), ("cols", "<i4"), ("src_ld", "<i4"),
                             ("dst_ld", "<i4"), ("flags", "<i4"), ("reserved", "<i4")])


class Ref:
    """A device (or, for the test emulation, host) address: flat tensor + element offset."""
    __slots__ = ("t", "off")

    def __init__(self, t, off=0):
        self.t = t
        self.off = int(off)

    def ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()

    def __add__(self, elems):
        return Ref(self.t, self.off + int(elems))


def build_library(verbose=False):
    """Compile libhypel_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "libhypel_hip.so"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_P, _I64, _I32, _F, _U64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_uint64
_F64 = ctypes.c_double

# name -> argument ctypes (stream appended automatically); mirrors include/hypel.h one to one
SIGNATURES = {
    "nhwc_to_pnc": [_P, _P, _I64, _I32, _I32, _I64],
    "pnc_to_nhwc": [_P, _I64, _P, _I64, _I32, _I32],
    "fill_f32": [_P, _I64, _F],
    "seg_gemm_f32": [_P, _I64, _I32, _P, _I64, _I32, _P, _I64, _I32, _P, _P, _P, _I32, _P, _I32],
    "seg_gemm_stats_f32": [_P, _I64, _I32, _P, _I64, _I32, _P, _I64, _I32, _P, _P, _P, _I32, _P, _I32, _P],
    "seg_gemm_res_f32": [_P, _I64, _I32, _P, _I64, _I32, _P, _I64, _I32, _P, _P, _P, _I32, _P, _I32, _P, _I64, _P],
    "reduce_splits_f32": [_P, _I64, _I32, _P, _I64, _I32, _P, _I32, _I64],
    "reduce_splits_pair_f32": [_P, _I64, _I64, _P, _P, _I64, _I64, _P, _I32, _I32],
    "seg_gemm_multi_f32": [_P, _I32, _I32, _I32, _P, _P, _I32],
    "copy_blocks_f32": [_P, _P, _I32, _I64],
    "reduce_splits_wave_multi_f32": [_P, _P, _I32, _I64],
    "reduce_splits_multi_f32": [_P, _P, _I32],
    "reduce_splits_multi_sized_f32": [_P, _P, _I32, _I64],
    "col_stats_partial": [_P, _I64, _I64, _I32, _I32, _P],
    "bn_act_small_fwd": [_P, _I64, _I64, _I32, _F, _P, _I32, _F, _P, _I64, _P, _P, _P, _P, _F, _P, _I64],
    "bn_act_small_bwd": [_P, _I64, _P, _I64, _I64, _I32, _P, _P, _P, _I32, _F, _P, _I64, _P, _I64, _P, _I32],
    "bn_finalize": [_P, _I32, _I32, _I64, _I32, _F, _P, _P, _P, _P, _F],
    "bn_merge_partials": [_P, _I32, _I32, _I64, _I32, _P],
    "bn_finalize_ranks": [_P, _I32, _I32, _F, _P, _P, _P, _P, _F],
    "rstd_from_var": [_P, _I32, _F, _P],
    "bn_act_fwd": [_P, _I64, _I64, _I32, _P, _P, _P, _I32, _F, _P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _I64],
    "bn_act_bwd_reduce": [_P, _I64, _P, _I64, _I64, _I32, _P, _P, _P, _I32, _F, _P, _I64, _I32, _P],
    "act_bias_bwd_reduce": [_P, _I64, _P, _I64, _I64, _I32, _I32, _F, _P, _I64, _I32, _P, _P, _I64],
    "bwd_reduce_finalize": [_P, _I32, _I32, _P, _P, _I32],
    "bn_act_bwd_apply": [_P, _I64, _P, _I64, _I64, _I32, _P, _P, _P, _I32, _F, _P, _I64, _P, _P, _I64],
    "bn_act_bwd_apply_global": [_P, _I64, _P, _I64, _I64, _I32, _P, _P, _P, _I32, _F, _P, _I64, _P, _I64, _P, _I64],
    "chanmap_bwd": [_P, _I64, _I64, _I32, _P, _I64, _I32, _P, _I32],
    "softmax_xent": [_P, _I64, _I64, _I32, _P, _I64, _P, _P, _I64, _F],
    "mse": [_P, _I64, _P, _I64, _I64, _I32, _P, _P, _I64, _F, _P],
    "sum_f32": [_P, _I64, _F, _P, _P],
    "adam_tf1": [_P, _P, _P, _P, _I64, _F, _F, _F, _F],
    "momentum_tf1": [_P, _P, _P, _I64, _F, _F],
    "adam_tf1_guarded": [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _P],
    "momentum_tf1_guarded": [_P, _P, _P, _I64, _F, _F, _P],
    "loss_guard_f32": [_P, _P, _P],
    "mse_partial_f32": [_P, _I64, _P, _I64, _I64, _I32, _P, _I64, _F, _P],
    "loss_finalize_f32": [_P, _I32, _P, _F64, _P, _P, _P, _P],
    "dropout_mask": [_P, _I64, _F, _U64, _P],
    "step_inc": [_P],
    "argmax_confusion": [_P, _I64, _I64, _I32, _P, _P, _P],
    "gather_patches_f32": [_P, _P, _I64, _I64, _I32, _I32, _P, _I64, _I32, _P],
    "gather_patches_2x_f32": [_P, _P, _I64, _I64, _I32, _I32, _I32, _P, _I64, _I32, _P],
    "augment_patches_f32": [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    "gather_pairs_f32": [_P, _P, _P, _I64, _I32, _P, _P, _P, _F, _P, _P],
    "argmax_scatter": [_P, _I64, _I64, _I32, _P, _P, _I64],
    "lrn_fwd": [_P, _I64, _I64, _I32, _I32, _F, _F, _F, _P, _I64],
    "lrn_bwd": [_P, _I64, _P, _I64, _I64, _I32, _I32, _F, _F, _F, _P, _I64, _I32],
    "gan_generator_fwd": [_P, _I64, _I64, _I32, _P, _P, _I32, _P, _I64],
    "gan_generator_bwd": [_P, _I64, _P, _I64, _I64, _I32, _P, _P, _I32, _P, _I64, _I32, _P, _P],
    "dense_stack_fwd": [_P, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P, _P, _P, _I64],
    "dense_stack_bwd": [_P, _I64, _P, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P, _P, _P, _I64, _I32,
                        _P, _P],
    "gan_generator_fwd_keep": [_P, _I64, _I64, _I32, _P, _P, _I32, _P, _I64, _P],
    "gan_generator_bwd_kept": [_P, _I64, _P, _I64, _I64, _I32, _P, _P, _I32, _P, _I64, _I32, _P, _P, _P],
    "gan_generator_fwd_tap": [_P, _I64, _I64, _I32, _P, _P, _P, _I64, _P, _I64, _P],
    "gan_generator_bwd_tap": [_P, _I64, _P, _I64, _P, _I64, _I64, _I32, _P, _P, _P, _I64, _I32, _P, _P, _P],
    "copy_pair_f32": [_P, _P, _I64, _P, _P, _I64],
    "gan_generator_fwd_apps": [_P, _I64, _I64, _I32, _I64, _I64, _I32, _P, _P, _I32, _P, _I64, _P],
    "gan_generator_bwd_apps": [_P, _I64, _P, _I64, _I64, _I32, _I64, _I64, _I64, _I64, _I32, _P, _P, _I32, _P, _I64, _I32,
                               _P, _P, _P],
    "dense_stack_fwd_apps": [_P, _I64, _I64, _I32, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P, _P, _P,
                             _I64],
    "dense_stack_bwd_apps": [_P, _I64, _P, _I64, _I64, _I32, _I64, _I64, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _I32,
                             _I32, _F, _P, _P, _P, _I64, _I32, _P, _P],
    "gan_loss": [_I32, _P, _I64, _P, _I64, _I64, _I32, _F, _F, _P, _I32, _P, _I64, _I32, _P, _I64, _I32, _P],
    "l2_reg": [_P, _I64, _F, _P, _I32, _P, _P],
    "loss_finalize_slots": [_P, _I32, _P, _I32],
    "loss_terms_slots": [_P, _P, _I32, _P],
    "l2norm_fwd": [_P, _I64, _I64, _I32, _P, _I64, _P],
    "l2norm_parts_fwd": [_P, _I64, _I64, _I32, _I32, _P, _I64, _P],
    "l2norm_parts_bwd": [_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _P, _I64, _I32],
    "l2norm_segs_fwd": [_P, _I64, _I64, _I32, _I32, _I32, _P, _I64, _P],
    "l2norm_segs_bwd": [_P, _I64, _P, _I64, _I64, _I32, _I32, _I32, _P, _P, _I64, _I32],
    "l2norm_bwd": [_P, _I64, _P, _I64, _I64, _I32, _P, _P, _I64, _I32],
    "nce_loss": [_P, _I64, _P, _I64, _I64, _I32, _I32, _F, _F, _P, _I32, _P, _I64, _I32, _P, _I64, _I32, _P],
}
NO_STREAM = {"version", "last_error", "device_info"}


class HypelError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise HypelError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    lib.hypel_last_error.restype = ctypes.c_char_p
    lib.hypel_version.restype = ctypes.c_int
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, "hypel_" + name, None)
        if fn is None:
            continue  # optional entry points (checked by tests/test_abi.py against the header)
        fn.argtypes = list(sig) + [_P]
        fn.restype = ctypes.c_int
    lib.hypel_graph_begin_capture.argtypes = [_P]
    lib.hypel_graph_end_capture.argtypes = [_P, ctypes.POINTER(_P)]
    lib.hypel_graph_launch.argtypes = [_P, _P]
    lib.hypel_graph_destroy.argtypes = [_P]
    lib.hypel_device_info.argtypes = [ctypes.POINTER(_I32), ctypes.POINTER(_I32)]
    lib.hypel_gan_generator_blocks.argtypes = [_I64]
    lib.hypel_gan_generator_blocks.restype = ctypes.c_int
    lib.hypel_dense_stack_blocks.argtypes = [_I64]
    lib.hypel_dense_stack_blocks.restype = ctypes.c_int
    lib.hypel_gan_generator_blocks_apps.argtypes = [_I64, _I32]
    lib.hypel_gan_generator_blocks_apps.restype = ctypes.c_int
    lib.hypel_dense_stack_blocks_apps.argtypes = [_I64, _I32]
    lib.hypel_dense_stack_blocks_apps.restype = ctypes.c_int
    lib.hypel_dense_stack_supported.argtypes = [_I32] * 6
    lib.hypel_dense_stack_supported.restype = ctypes.c_int
    lib.hypel_gan_generator_tap_supported.argtypes = [_I32]
    lib.hypel_gan_generator_tap_supported.restype = ctypes.c_int
    lib.hypel_gan_generator_keep_floats.argtypes = [_I64, _I32, _I32]
    lib.hypel_gan_generator_keep_floats.restype = ctypes.c_int64
    return lib


COLLECTIVES = ("_allreduce", "_allgather")


def bind_collective(name, args):
    """Host-side pseudo-launches of a plan (synchronised batch norm): `_allreduce (ref, count)` sums `count` floats in
    place over the data-parallel ranks, `_allgather (src, count, dst)` writes every rank's `count` floats to
    dst[rank * count ...].  torch.distributed orders them with the current stream (RCCL on the GPU box, gloo in the CPU
    tests); they cannot be captured, so a compiled tower cuts its HIP graphs around them (`host` attribute)."""
    import torch.distributed as dist
    if name == "_allreduce":
        ref, count = args

        def call():
            note_collective()
            dist.all_reduce(ref.t[ref.off:ref.off + count], op=dist.ReduceOp.SUM)
    else:
        src, count, dst = args

        def call():
            # into ONE long-lived tensor: the list form of all_gather stages through a temporary whose release (by the
            # process group's watchdog thread, whenever it reaps the finished work) makes the caching allocator record
            # and poll an event -- which invalidates a HIP-graph capture that happens to be under way on the compute
            # stream ("operation not permitted on an event last recorded in a capturing stream", 1 run in 8)
            world = dist.get_world_size()
            note_collective()
            dist.all_gather_into_tensor(dst.t[dst.off:dst.off + world * count], src.t[src.off:src.off + count])

    call.host = True
    return call


_collective_epoch = [0]  # bumped by every collective the package issues (note_collective)


def _pg_sequence_number():
    """Collective sequence number of the default process group (None when the build does not expose it): moves with
    every collective issued on it, whoever issued it."""
    try:
        import torch.distributed as dist
        return int(dist.distributed_c10d._get_default_group()._get_sequence_number_for_group())
    except Exception:  # noqa: BLE001 -- private API: absent / renamed means "unknown", the package epoch still counts
        return None


def note_collective():
    """Called right before any torch.distributed collective of the package: a HIP-graph capture must not start while
    the RCCL watchdog still polls that collective (HipBackend.settle_before_capture waits it out -- once per burst of
    captures, not once per capture, as long as no collective was issued in between)."""
    _collective_epoch[0] += 1


class HipBackend:
    """Launches hand-written gfx950 kernels through the C-ABI on a torch CUDA(HIP) stream."""
    name = "hip"
    _settled_epoch = -1  # value of the collective epoch at the last settle_before_capture() pause
    _streams = {}  # device index -> (main stream, side stream), shared by every backend object of the process

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise HypelError("no HIP device visible: the hypelcnn_amd compute path needs an MI355X "
                             "(there is no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.lib = load_library()
        if self.lib.hypel_version() != ABI_VERSION:
            raise HypelError(f"libhypel_hip.so ABI version {self.lib.hypel_version()} != {ABI_VERSION} (stale build or HYPEL_LIB_PATH)")
        # All hypel launches (and the torch plumbing ops around them) run on ONE dedicated non-default
        # stream: HIP cannot capture the legacy null stream into a graph, and a private stream keeps the
        # step ordered without device-wide syncs.  The stream is per DEVICE, not per backend object: torch's
        # "current stream" is process-global state, so a second HipBackend with streams of its own would silently
        # move the first one's copies (set_input, .cpu()) onto a stream its kernels are not ordered with.
        key = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if key not in HipBackend._streams:
            HipBackend._streams[key] = torch.cuda.Stream(self.device)
        self.stream = HipBackend._streams[key]
        torch.cuda.set_stream(self.stream)

    # -- memory (PyTorch is the allocator: plumbing only) --
    def empty(self, n, dtype=torch.float32):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    def zeros(self, n, dtype=torch.float32):
        t = torch.empty(int(n), dtype=dtype, device=self.device)
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return t
        if nbytes % 4:  # (byte tables of odd length: not on any step path)
            return t.zero_()
        self.call("fill_f32", Ref(t.view(torch.uint8).view(torch.float32)), nbytes // 4, 0.0)  # hypel_fill, not a torch kernel
        return t

    def upload(self, array):
        """numpy array (any dtype, incl. structured tables) -> flat device tensor."""
        a = np.ascontiguousarray(array)
        if a.dtype.fields is not None:
            t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy())
        else:
            t = torch.from_numpy(a.reshape(-1).copy())
        return t.to(self.device)

    def stream_handle(self):
        return self.stream.cuda_stream

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    def gan_generator_blocks(self, n):
        return int(self.lib.hypel_gan_generator_blocks(int(n)))

    def dense_stack_blocks(self, n):
        return int(self.lib.hypel_dense_stack_blocks(int(n)))

    def gan_generator_blocks_apps(self, n, n_apps):
        return int(self.lib.hypel_gan_generator_blocks_apps(int(n), int(n_apps)))

    def dense_stack_blocks_apps(self, n, n_apps):
        return int(self.lib.hypel_dense_stack_blocks_apps(int(n), int(n_apps)))

    def dense_stack_supported(self, widths):
        w = list(widths) + [0] * (5 - len(widths))
        return len(widths) <= 5 and bool(self.lib.hypel_dense_stack_supported(len(widths) - 1, *[int(v) for v in w]))

    def gan_generator_tap_supported(self, bands):
        return bool(self.lib.hypel_gan_generator_tap_supported(int(bands)))

    def gan_generator_keep_floats(self, n, bands, only_encoder):
        return int(self.lib.hypel_gan_generator_keep_floats(int(n), int(bands), int(only_encoder)))

    # -- launches --
    def bind(self, name, args, stream=None):
        if name in COLLECTIVES:
            return bind_collective(name, args)
        fn = getattr(self.lib, "hypel_" + name)
        st = self.stream_handle() if stream is None or stream == 0 else stream
        cargs = []
        for a in args:
            if isinstance(a, Ref):
                cargs.append(a.ptr())
            elif a is None:
                cargs.append(None)
            else:
                cargs.append(a)
        cargs.append(st)
        cargs = tuple(cargs)
        lib = self.lib

        def call():
            rc = fn(*cargs)
            if rc != 0:
                raise HypelError(f"hypel_{name} failed ({rc}): {lib.hypel_last_error().decode()}")

        return call

    def call(self, name, *args):
        self.bind(name, args)()

    # -- HIP graph capture --
    def settle_before_capture(self):
        """A HIP-graph capture on the compute stream must not overlap the RCCL process group's watchdog thread while
        that thread still polls / reaps finished collectives: HIP then fails the capture AND the watchdog's event query
        ("operation not permitted on an event last recorded in a capturing stream", which terminates the process).
        torch.cuda.graph() coordinates with the watchdog internally; a capture started through the C-ABI cannot, so it
        waits until the device is idle and the watchdog (100 ms period) has nothing left to look at.  Reproduced and
        cured in tools/exp/capture_vs_watchdog.py: 8 of 9 runs of 60 capture bursts die without the pause, 0 of 6 with
        it.  No-op without an initialised NCCL process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
            return
        # "anything issued since the last pause?" = the package's own epoch (note_collective) AND the process group's
        # collective sequence number, which also moves for dist.* calls made outside the package (bench.py, user code);
        # HYPEL_CAPTURE_SETTLE_ALWAYS=1 pauses before every capture regardless
        epoch = (_collective_epoch[0], _pg_sequence_number())
        if os.environ.get("HYPEL_CAPTURE_SETTLE_ALWAYS") != "1" and HipBackend._settled_epoch == epoch:
            return  # no collective since the last pause: the watchdog has nothing new to poll (a burst of captures --
            #         every batch size of a run, TrainStep.precapture -- pays the pause once)
        torch.cuda.synchronize(self.device)
        time.sleep(float(os.environ.get("HYPEL_CAPTURE_SETTLE_S", "0.5")))
        HipBackend._settled_epoch = (_collective_epoch[0], _pg_sequence_number())

    def capture(self, launches, settled=False):
        """Capture a list of bound launches into a hipGraphExec; returns a replay callable.
        settled: the caller already called settle_before_capture() and issued no collective since."""
        if not settled:
            self.settle_before_capture()
        st = self.stream_handle()
        lib = self.lib
        if lib.hypel_graph_begin_capture(st) != 0:
            raise HypelError(lib.hypel_last_error().decode())
        try:
            for f in launches:
                f()
        finally:
            exec_ = _P()
            rc = lib.hypel_graph_end_capture(st, ctypes.byref(exec_))
        if rc != 0:
            raise HypelError(lib.hypel_last_error().decode())
        handle = exec_.value

        def replay():
            if lib.hypel_graph_launch(handle, st) != 0:
                raise HypelError(lib.hypel_last_error().decode())

        replay.handle = handle
        return replay
