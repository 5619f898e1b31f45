"""ctypes binding of libnerfmatch_amd.so (the C ABI declared in include/nerfmatch_amd.h).

The product path has NO fallback: if the HIP library is missing or a kernel call fails, these
helpers raise.  Tensors are passed as raw device pointers (`tensor.data_ptr()`) and work is
enqueued on torch's current HIP stream.
"""
import ctypes as C
from pathlib import Path

import torch

PKG = Path(__file__).resolve().parent
import os

LIB_PATH = Path(os.environ.get("NERFMATCH_AMD_LIB", PKG / "lib" / "libnerfmatch_amd.so"))  # env override: kernel A/B builds

_lib = None

vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class NerfWeights(C.Structure):
    _fields_ = [
        ("pts_w", vp * 8), ("pts_b", vp * 8), ("alpha_w", vp), ("alpha_b", vp), ("feat_w", vp), ("feat_b", vp),
        ("views_w", vp), ("views_b", vp), ("rgb_w", vp), ("rgb_b", vp), ("app_dim", i32),
    ]


# name -> (restype, argtypes); must list every symbol of include/nerfmatch_amd.h (tests check this)
SIGNATURES = {
    "nm_abi_version": (i32, []),
    "nm_error_string": (C.c_char_p, [i32]),
    "nm_probe_mfma_f16": (i32, [vp, i32, i32, vp]),
    "nm_stream_create_cu_mask": (i32, [C.POINTER(C.c_uint32), i32, C.POINTER(vp)]),
    "nm_stream_destroy": (i32, [vp]),
    "nm_stream_cus": (i32, [vp]),
    "nm_params_fingerprint": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp]),
    "nm_raygen_count": (i32, [i32, i32, i32]),
    "nm_raygen": (i32, [vp, vp, i32, i32, i32, f32, vp, vp, vp]),
    "nm_raygen_batch": (i32, [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "nm_sample_coarse": (i32, [vp, vp, i32, i32, vp, vp]),
    "nm_resample": (i32, [vp, vp, vp, i32, i32, f32, i32, vp, vp]),
    "nm_resample_ex": (i32, [vp, vp, vp, i32, i32, f32, i32, vp, vp, vp]),
    "nm_resample_scaled": (i32, [vp, vp, vp, f32, i32, i32, f32, i32, vp, vp, vp]),
    "nm_nerf_blob_floats": (sz, []),
    "nm_nerf_pack": (i32, [C.POINTER(NerfWeights), vp]),
    "nm_nerf_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nm_nerf_blob_bytes_bf16x3": (sz, []),
    "nm_nerf_pack_bf16x3": (i32, [C.POINTER(NerfWeights), vp]),
    "nm_nerf_workspace_bytes_bf16x3": (sz, []),
    "nm_nerf_fwd_bf16x3": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nm_nerf_fwd_bf16x3_ex": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nm_nerf_pack_fp16x3": (i32, [C.POINTER(NerfWeights), vp]),
    "nm_nerf_fwd_fp16x3": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nm_nerf_pack_fp16x3_scaled": (i32, [C.POINTER(NerfWeights), vp, vp]),
    "nm_nerf_fwd_fp16x3_ex": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nm_nerf_fwd_guarded": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nm_nerf_blob_bytes_bwd_bf16x3": (sz, []),
    "nm_nerf_points_gate_bytes": (sz, [i32]),
    "nm_nerf_pack_bwd_bf16x3": (i32, [C.POINTER(NerfWeights), vp]),
    "nm_nerf_points_fwd_bf16x3": (i32, [vp, vp, vp, i32, vp, vp, vp]),
    "nm_nerf_points_fwd_rays_bf16x3": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "nm_nerf_points_bwd_bf16x3": (i32, [vp, vp, vp, i32, vp, vp, vp, vp]),
    "nm_nerf_points_fwd_rays_tap_bf16x3": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp]),
    "nm_nerf_points_bwd_tap_bf16x3": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "nm_nerf_blob_bytes_fp16x1": (sz, []),
    "nm_nerf_pack_fp16x1": (i32, [C.POINTER(NerfWeights), vp]),
    "nm_nerf_fwd_fp16x1": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nm_unnormalize_points": (i32, [vp, vp, i32, vp, vp]),
    "nm_inerf_encode": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "nm_inerf_composite4": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "nm_inerf_composite4_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "nm_inerf_encode_bwd": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "nm_inerf_encode_bwd2": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "nm_inerf_pose_grad": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp]),
    "nm_inerf_composite": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, vp, vp]),
    "nm_inerf_composite_bwd": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "nm_inerf_composite_ex": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp]),
    "nm_inerf_composite_bwd_ex": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "nm_inerf_ray_sums": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp]),
    "nm_inerf_ray_sums_bwd": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "nm_linear": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "nm_linear_ex": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "nm_linear_ex_bf16x3": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "nm_linear_blob_bytes_bf16x3": (sz, [i32, i32]),
    "nm_linear_qkv_bf16x3": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "nm_attention_presplit": (i32, [vp, i32, vp, i32, i32, i32, i32, f32, vp, vp]),
    "nm_linear_pack_bf16x3": (i32, [vp, i32, i32, vp, vp]),
    "nm_linear_pack_t_bf16x3": (i32, [vp, i32, i32, vp, vp]),
    "nm_linear_bf16x3": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "nm_layernorm": (i32, [vp, vp, vp, i32, i32, f32, vp, vp]),
    "nm_layernorm2": (i32, [vp, vp, vp, i32, f32, vp, vp, vp, vp, i32, f32, vp, i32, vp]),
    "nm_attention": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp]),
    "nm_attention_ld": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp]),
    "nm_attention_ex": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp]),
    "nm_linear_pack_perm_bf16x3": (i32, [vp, i32, i32, vp, vp]),
    "nm_encoder_tail_bf16x3": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp]),
    "nm_encoder_tail_bwd_bf16x3": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp]),
    "nm_encoder_tail_save_bf16x3": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]),
    "nm_attention_workspace_bytes": (sz, [i32, i32, i32]),
    "nm_attention_fp8_workspace_bytes": (sz, [i32, i32, i32]),
    "nm_attention_fp8": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp]),
    "nm_attention_ws": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp]),
    "nm_add_sine_pe": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "nm_mip_encode": (i32, [vp, vp, sz, i32, i32, i32, i32, vp, vp, vp]),
    "nm_fourier_embed": (i32, [vp, sz, i32, i32, vp, vp]),
    "nm_feature_normalize": (i32, [vp, i32, i32, i32, vp, vp]),
    "nm_cat_fourier": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "nm_cat_fourier_bwd": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "nm_match_workspace_bytes": (sz, [i32, i32, i32]),
    "nm_dual_softmax_match": (i32, [vp, vp, i32, i32, i32, f32, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "nm_match_fused_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "nm_dual_softmax_match_fused": (i32, [vp, vp, i32, i32, i32, i32, f32, vp, vp, f32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "nm_dual_softmax_match_ex": (i32, [vp, vp, i32, i32, i32, f32, vp, vp, f32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "nm_fine_windows": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, i32, vp, vp]),
    "nm_fine_windows_batch": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, vp, vp]),
    "nm_assemble_matches": (i32, [vp, vp, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, vp, vp]),
    "nm_gather_rows": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "nm_fine_pt_proj": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "nm_fine_stage": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, f32,
                            vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "nm_fine_window_layer": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, f32,
                                   vp, vp, vp, vp]),
    "nm_fine_expectation": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    # training side (train.hip, attention_bwd.hip, match.hip)
    "nm_linear_wgrad_workspace_bytes": (sz, [i32, i32, i32]),
    "nm_linear_wgrad": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "nm_linear_wgrad_bf16x3": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "nm_linear_wgrad_bias_bf16x3": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "nm_col_sum": (i32, [vp, i32, i32, i32, vp, vp]),
    "nm_gelu": (i32, [vp, sz, vp, vp]),
    "nm_gelu_bwd": (i32, [vp, vp, sz, vp, vp]),
    "nm_relu_bwd": (i32, [vp, vp, sz, vp, vp]),
    "nm_layernorm_bwd": (i32, [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]),
    "nm_l2norm_bwd": (i32, [vp, vp, i32, i32, vp, vp]),
    "nm_attention_bwd_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "nm_attention_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "nm_attention_bwd_lse": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "nm_attention_ws_lse": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp]),
    "nm_fine_windows_bwd": (i32, [vp, i32, i32, i32, vp, vp, i32, i32, i32, vp, vp]),
    "nm_fine_expectation_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "nm_focal_count": (i32, [vp, sz, vp, vp]),
    "nm_match_focal_loss": (i32, [vp, i32, i32, i32, f32, f32, i32, vp, sz, vp, vp, vp, vp]),
    "nm_match_focal_loss_bwd": (i32, [vp, vp, vp, i32, i32, i32, f32, f32, i32, f32, vp, vp, sz, vp, vp, vp, vp, vp, vp]),
}

NM_NERF_SKIP_RGB = 1
NM_NERF_FEAT_MAX = 2
NM_ACT_NONE, NM_ACT_RELU, NM_ACT_GELU = 0, 1, 2
NM_ERR_UNSUPPORTED = 2
NM_ATTN_BF16X3 = 1
NM_NERF_ZERO_TAIL = 4
NM_MATCH_BF16X3 = 1
NM_MATCH_STATS_ONLY = 2


class NerfmatchAmdError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NerfmatchAmdError(
                f"{LIB_PATH} not found: build the HIP extension first (python -m nerfmatch_amd.build). "
                "nerfmatch_amd has no CPU / eager fallback."
            )
        h = C.CDLL(str(LIB_PATH))
        missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(h, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype, fn.argtypes = res, args
        h._nm_missing = missing  # tests/test_abi.py requires this to be empty
        _lib = h
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().nm_error_string(code).decode()
        raise NerfmatchAmdError(f"{what} failed: {msg} (code {code})")


def stream():
    """torch's current stream of the CURRENT device; dptr() checks that every tensor handed to a kernel lives there (the C
    side launches on the stream it is given and never calls hipSetDevice)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))  # (= torch.cuda.current_stream().cuda_stream, a tenth of the host time)


_PART_STREAMS = {}


def destroy_partition_streams():
    """Destroys the CU-masked streams of this process (explicit call only; every tensor that was used on one of them must be gone: the caching
    allocator records an event on each stream a block was used on when the block is freed).  NOT registered at interpreter exit -- a round-6
    version did that and crashed untraced processes in Py_FinalizeEx: module globals are cleared AFTER the exit handlers, and freeing a tensor
    that had been on a partition stream then recorded an event on a destroyed stream.  Left alone, the streams go with the process."""
    for st in list(_PART_STREAMS.values()):
        st.synchronize()
        check(lib().nm_stream_destroy(vp(st.cuda_stream)), "nm_stream_destroy")
    _PART_STREAMS.clear()


def cu_mask_words(ncu, first_cu=0, n_cus=0, xcds=None, n_xcd=8):
    """The 32-bit words of a compute-unit mask for hipExtStreamCreateWithCUMask on a chip of `ncu` units in `n_xcd` XCDs, where bit i is
    unit i // n_xcd of XCD i % n_xcd (profiles/r6_cumask_probe.log): bits [first_cu, first_cu + n_cus), or -- xcds = (first, count) -- every
    unit of `count` whole XCDs."""
    if xcds is None:
        if n_cus <= 0 or first_cu < 0 or first_cu + n_cus > ncu:
            raise NerfmatchAmdError(f"compute-unit range [{first_cu}, {first_cu + n_cus}) outside the device's {ncu} units")
        bits = range(first_cu, first_cu + n_cus)
    else:
        if xcds[1] <= 0 or xcds[0] < 0 or xcds[0] + xcds[1] > n_xcd:
            raise NerfmatchAmdError(f"XCD range {tuple(xcds)} outside the device's {n_xcd} XCDs")
        bits = [i for i in range(ncu) if xcds[0] <= i % n_xcd < xcds[0] + xcds[1]]
    mask = [0] * ((ncu + 31) // 32)
    for i in bits:
        mask[i >> 5] |= 1 << (i & 31)
    return mask


def partition_stream(n_cus, first_cu=0, device=None, xcds=None):
    """torch stream (ExternalStream over nm_stream_create_cu_mask) whose kernels run on a subset of the current device's compute units:
    mask bits [first_cu, first_cu + n_cus) -- n_cus / 8 units of every XCD -- or, with xcds = (first, count), all 32 units of `count` whole
    XCDs.  One stream per (device, subset), kept for the life of the process.  Persistent kernels launched on it size their grids to the
    subset (nm_stream_cus)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (dev, int(first_cu), int(n_cus), None if xcds is None else tuple(xcds))
    st = _PART_STREAMS.get(key)
    if st is None:
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        mask = cu_mask_words(ncu, first_cu, n_cus, xcds)
        words = len(mask)
        with torch.cuda.device(dev):
            h = vp()
            check(lib().nm_stream_create_cu_mask((C.c_uint32 * words)(*mask), words, C.byref(h)), "nm_stream_create_cu_mask")
            st = _PART_STREAMS[key] = torch.cuda.ExternalStream(h.value, device=dev)
    return st


def dptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA(HIP) tensor of the expected dtype, or NULL for None.  The tensor must live on
    the current device: one process per GPU calls torch.cuda.set_device(LOCAL_RANK) once (GenericModelEvaluator and bench.py
    do), and a pointer of another device on this device's stream would launch the kernel on the wrong GPU."""
    if t is None:
        return C.c_void_p(0)
    if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise NerfmatchAmdError(f"expected contiguous {dtype} device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    if t.device.index != torch._C._cuda_getDevice():
        raise NerfmatchAmdError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                                "call torch.cuda.set_device(LOCAL_RANK) (one process per GPU)")
    return C.c_void_p(t.data_ptr())


def hptr(t):
    """Host pointer of a contiguous CPU fp32 tensor."""
    if t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
        raise NerfmatchAmdError("expected contiguous fp32 host tensor")
    return C.c_void_p(t.data_ptr())


PRECISIONS = ("fp32", "bf16x3", "fp16x3", "fp16x1", "bwd_bf16x3")


def pack_nerf_weights(sd, prefix, precision="fp32", act_log2=None):
    """state-dict (reference key names) -> packed host blob for nm_nerf_fwd (1-D fp32 tensor) or, with
    precision="bf16x3", for nm_nerf_fwd_bf16x3 (1-D uint8 tensor).  act_log2 (fp16x3 only): the 12 input-scale exponents of
    nm_nerf_pack_fp16x3_scaled (None = weights scaled, activations as they are)."""
    if precision not in PRECISIONS:
        raise NerfmatchAmdError(f"precision must be one of {PRECISIONS}")
    L = lib()
    keep = []

    def host(name):
        t = sd[f"{prefix}.{name}"].detach().to("cpu", torch.float32).contiguous()
        keep.append(t)
        return t.data_ptr()

    w = NerfWeights()
    for i in range(8):
        w.pts_w[i] = host(f"pts_linears.{i}.weight")
        w.pts_b[i] = host(f"pts_linears.{i}.bias")
    w.alpha_w, w.alpha_b = host("alpha_linear.weight"), host("alpha_linear.bias")
    w.feat_w, w.feat_b = host("feature_linear.weight"), host("feature_linear.bias")
    w.views_w, w.views_b = host("views_linears.0.weight"), host("views_linears.0.bias")
    w.rgb_w, w.rgb_b = host("rgb_linear.weight"), host("rgb_linear.bias")
    in_dim = sd[f"{prefix}.views_linears.0.weight"].shape[1]
    w.app_dim = in_dim - 283
    shapes = {0: (256, 90), 5: (256, 346)}
    for i in range(8):
        exp = shapes.get(i, (256, 256))
        if tuple(sd[f"{prefix}.pts_linears.{i}.weight"].shape) != exp:
            raise NerfmatchAmdError(f"{prefix}.pts_linears.{i}.weight has shape {tuple(sd[f'{prefix}.pts_linears.{i}.weight'].shape)}, kernel is built for {exp}")
    if precision == "fp16x1":  # (the dtype of the blob tensor tells ops.nerf_fwd which kernel family it belongs to)
        blob = torch.empty(L.nm_nerf_blob_bytes_fp16x1() // 2, dtype=torch.float16)
        check(L.nm_nerf_pack_fp16x1(C.byref(w), C.c_void_p(blob.data_ptr())), "nm_nerf_pack_fp16x1")
        return blob
    if precision == "bwd_bf16x3":  # transposed weights for nm_nerf_points_bwd_bf16x3
        blob = torch.empty(L.nm_nerf_blob_bytes_bwd_bf16x3(), dtype=torch.uint8)
        check(L.nm_nerf_pack_bwd_bf16x3(C.byref(w), C.c_void_p(blob.data_ptr())), "nm_nerf_pack_bwd_bf16x3")
        return blob
    if precision == "bf16x3":
        blob = torch.empty(L.nm_nerf_blob_bytes_bf16x3(), dtype=torch.uint8)
        check(L.nm_nerf_pack_bf16x3(C.byref(w), C.c_void_p(blob.data_ptr())), "nm_nerf_pack_bf16x3")
        return blob
    if precision == "fp16x3":  # same size and slot structure as the bf16x3 blob; int16 marks the kernel family
        blob = torch.empty(L.nm_nerf_blob_bytes_bf16x3() // 2, dtype=torch.int16)
        al = None if act_log2 is None else (C.c_int * 12)(*[int(v) for v in act_log2])
        check(L.nm_nerf_pack_fp16x3_scaled(C.byref(w), al, C.c_void_p(blob.data_ptr())), "nm_nerf_pack_fp16x3_scaled")
        return blob
    blob = torch.empty(L.nm_nerf_blob_floats(), dtype=torch.float32)
    check(L.nm_nerf_pack(C.byref(w), C.c_void_p(blob.data_ptr())), "nm_nerf_pack")
    return blob


_GC_DEPTH = [0]
_GC_LOCK = __import__("threading").RLock()  #  the depth counter and the freeze / unfreeze calls of two threads must not interleave


@__import__("contextlib").contextmanager
def steady_gc():
    """Localisation / refinement loops run with the objects that exist at their start exempt from Python's cyclic collector
    (gc.freeze(); undone at exit).  A process that has imported torch and built two networks holds ~2e5 container objects; every
    full (generation-2) collection walks them all -- measured at 80-90 ms on the GPU box's host, landing on whichever step happens
    to allocate the triggering object: one iNeRF step in seventeen took 100 ms instead of 10.5, a 30 ms localisation batch now and
    then 110 ms (round 5, scripts/probe_step_spikes.py, profiles/r5_step_spikes_gc.log).  Frozen objects are not scanned; what the loop itself allocates is
    collected as before.  Nothing is collected up front (a forced collection would cost the same 80 ms on every call).  Re-entrant and
    thread-safe (one process-wide depth counter under a lock: gc.freeze() is process-wide too, so the heap stays frozen until the LAST
    loop of any thread has left); NERFMATCH_AMD_NO_GC_FREEZE=1 switches it off.  Side effect to know about: gc.unfreeze() moves the
    frozen objects into the oldest generation, so the next full collection after a loop sees all of them at once."""
    import gc

    if os.environ.get("NERFMATCH_AMD_NO_GC_FREEZE") == "1":  # opt-out for applications that manage the collector themselves
        yield
        return
    with _GC_LOCK:
        if _GC_DEPTH[0] == 0:
            # an application that froze its heap itself (serving frameworks do at start-up) keeps its own arrangement: nothing is touched then
            _GC_DEPTH.append(gc.get_freeze_count() == 0)
            if _GC_DEPTH[1]:
                gc.freeze()
        _GC_DEPTH[0] += 1
    try:
        yield
    finally:
        with _GC_LOCK:
            _GC_DEPTH[0] -= 1
            if _GC_DEPTH[0] == 0 and _GC_DEPTH.pop():
                gc.unfreeze()  # (the last loop to leave: a thread still inside its loop keeps the heap frozen)
