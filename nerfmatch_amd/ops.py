"""Thin torch-tensor wrappers over the C ABI (one function per entry point of include/nerfmatch_amd.h).

Everything here enqueues hand-written gfx950 kernels on torch's current stream; there is no eager
fallback.  Tensors must live on the GPU, be contiguous and fp32 (int64 / uint8 where stated)."""
import contextlib
import ctypes as C

import torch

from . import _lib
from ._lib import check, dptr, hptr, lib, stream

NEAR_PLANE = 0.01  # reference: nerfmatch/nerf/render_utils.py:72

# Measurement hook (bench.py sets it, nothing else does): callable(tag, flop) -> context manager that brackets the launches of one
# native call with HIP events on the launch stream.  None in production: the wrappers below then cost a null context.
KERNEL_PROBE = None


def _probe(tag, flop):
    return contextlib.nullcontext() if KERNEL_PROBE is None else KERNEL_PROBE(tag, flop)


def _f32(t):
    return t.to(torch.float32).contiguous()


def _mask_u8(m):
    """Mask -> contiguous uint8 device tensor (or None).  A bool mask is re-viewed, not converted: same bytes, no launch."""
    if m is None:
        return None
    m = m.contiguous()
    if m.dtype != torch.bool:  # the reference's `mask.bool()` (c2f_trainer.py:296-298): non-zero = valid
        m = m.ne(0)
    return m.view(torch.uint8)


_SIDE_STREAMS = {}


def side_stream(dev):
    """One extra (non-blocking) HIP stream per device for small transfers and checks that must not queue behind -- or lengthen -- the main
    stream's chain of dependent launches: the match counts' read-back, the parameter fingerprints."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


# ----------------------------------------------------------------------------- parameter fingerprints
class ParamGuard:
    """Notices parameters whose VALUES changed behind the derived copies this package keeps of them (packed / split / transposed blobs,
    host values), i.e. writes through `.data`, which change neither data_ptr nor _version -- the keys those caches use.
    check() is ONE small launch: it sums every parameter's words on the device and compares with the sums taken at the last baseline;
    `flag` (device int32[2], element 1) goes up on a mismatch.  The first check() for a given (data_ptr, _version) state of the parameters
    takes the baseline -- taken at the start of a pass, before that pass packs anything, so copies and baseline describe the same values.
    No synchronisation here: callers read `flag` where they synchronise anyway."""

    CHUNK = 4096  # words per workgroup: FP_CHUNK of csrc/capi.hip

    def __init__(self, module):
        """module: an nn.Module (every parameter of it and its children is watched; a Parameter OBJECT that is replaced later --
        load_state_dict(assign=True), `layer.weight = nn.Parameter(...)` -- is picked up: the list is validated by identity on every check)."""
        self.module = module
        self.key = None
        self.tables = None
        self.flag = None
        self._collect()

    def _collect(self):
        seen, self.slots = set(), []
        for m in self.module.modules():
            for n, p in m._parameters.items():
                if p is not None and id(p) not in seen and p.is_cuda and p.numel() > 0 and p.element_size() in (4, 8):
                    seen.add(id(p))
                    self.slots.append((m, n, p))
        self.params = [p for _, _, p in self.slots]

    def _state(self):
        for m, n, p in self.slots:
            if m._parameters.get(n) is not p:  # (a replaced Parameter object: walk the module again)
                self._collect()
                break
        return tuple([(p.data_ptr(), p._version) for p in self.params])

    def _build(self, dev):
        ptrs, words, bt, bo = [], [], [], []
        for i, p in enumerate(self.params):
            if not p.is_contiguous():
                raise _lib.NerfmatchAmdError("ParamGuard: parameters must be contiguous")
            n = p.numel() * p.element_size() // 4
            ptrs.append(p.data_ptr()); words.append(n)
            for off in range(0, n, self.CHUNK):
                bt.append(i); bo.append(off)
        i64 = lambda v: torch.tensor(v, dtype=torch.int64).to(dev)
        self.tables = dict(ptrs=i64(ptrs), words=i64(words), bt=torch.tensor(bt, dtype=torch.int32).to(dev), bo=i64(bo), nblk=len(bt),
                           cur=torch.zeros(len(ptrs), dtype=torch.int64, device=dev), ref=torch.zeros(len(ptrs), dtype=torch.int64, device=dev),
                           ptr_key=tuple(ptrs))
        self.flag = torch.zeros(2, dtype=torch.int32, device=dev)

    def check(self):
        """Enqueue the fingerprint launch (baseline when the parameters' (data_ptr, _version) state is new) on the device's SIDE stream,
        ordered behind everything the current stream holds at this moment -- a write to a parameter enqueued there is seen -- and beside
        whatever the current stream does next: the pass's chain of dependent launches is not lengthened.  Everything of this object runs
        on that one side stream (tables, launches, reset).  Returns True when this call took a baseline."""
        state = self._state()
        if not self.params:
            return True
        dev = self.params[0].device
        side = side_stream(dev)
        here = torch.cuda.Event()
        here.record(torch.cuda.current_stream(dev))
        side.wait_event(here)
        baseline = state != self.key
        with torch.cuda.stream(side):
            if self.tables is None or self.tables["ptr_key"] != tuple(s[0] for s in state):
                self._build(dev)
                baseline = True
            self.key = state
            t = self.tables
            u64 = torch.int64
            check(lib().nm_params_fingerprint(dptr(t["ptrs"], u64), dptr(t["words"], u64), dptr(t["bt"], torch.int32), dptr(t["bo"], u64),
                                              len(self.params), t["nblk"], dptr(t["cur"], u64), dptr(t["ref"], u64), dptr(self.flag, torch.int32),
                                              int(baseline), stream()), "nm_params_fingerprint")
        return baseline

    def reset(self):
        """Forget the baseline: the next check() takes a new one (call together with dropping the derived caches)."""
        self.key = None
        if getattr(self, "flag", None) is not None:
            with torch.cuda.stream(side_stream(self.flag.device)):
                self.flag.zero_()  # (sticky on the device until the next baseline: cleared so that a reader in between sees "clean")
                self.tables["cur"].zero_()


# ----------------------------------------------------------------------------- NeRF half
def raygen(K, c2w_norm, H, W, device, ds=8, near=NEAR_PLANE, out=None, flag=None):
    """rays (R,12) on `device` for the sub-sampled pixel grid; also returns the far-fallback flag tensor.
    `out` / `flag` let a caller fill a slice of a larger (batched) ray bundle."""
    L = lib()
    kinv = torch.linalg.inv(K.detach().to("cpu", torch.float32)).contiguous()
    pose = c2w_norm.detach().to("cpu", torch.float32).contiguous()
    R = L.nm_raygen_count(int(H), int(W), int(ds))
    rays = torch.empty(R, 12, device=device, dtype=torch.float32) if out is None else out
    assert rays.shape == (R, 12)
    flag = torch.empty(1, device=device, dtype=torch.int32) if flag is None else flag
    check(L.nm_raygen(hptr(kinv), hptr(pose), int(H), int(W), int(ds), float(near), dptr(rays), dptr(flag, torch.int32), stream()), "nm_raygen")
    return rays, flag


def raygen_batch(K, c2ws_norm, H, W, device, ds=8, near=NEAR_PLANE):
    """Q normalised poses (Q,4,4) -> rays (Q*R,12) and far-fallback flags (Q,) with one launch per kernel."""
    L = lib()
    kinv = torch.linalg.inv(K.detach().to("cpu", torch.float32)).contiguous()
    poses = c2ws_norm.detach().to("cpu", torch.float32).reshape(-1, 16).contiguous()
    Q = poses.shape[0]
    R = L.nm_raygen_count(int(H), int(W), int(ds))
    rays = torch.empty(Q * R, 12, device=device, dtype=torch.float32)
    flags = torch.empty(Q, device=device, dtype=torch.int32)
    check(L.nm_raygen_batch(hptr(kinv), hptr(poses), Q, int(H), int(W), int(ds), float(near), dptr(rays), dptr(flags, torch.int32), stream()),
          "nm_raygen_batch")
    return rays, flags


def sample_coarse(rays, t_rand, S):
    R = rays.shape[0]
    assert t_rand.shape == (R, S + 1)
    t = torch.empty(R, S + 1, device=rays.device, dtype=torch.float32)
    check(lib().nm_sample_coarse(dptr(rays), dptr(t_rand), R, int(S), dptr(t), stream()), "nm_sample_coarse")
    return t


def resample(t, weights, jitter, padding=0.01, randomized=True, want_tail_flag=False, jitter_scale=1.0):
    """want_tail_flag: also return the device int32[1] flag nm_resample_ex raises when the fence posts j > S/2 do NOT
    coincide (the premise of nerf_fwd(zero_tail=True)); pass it on as `tail_flag`.  jitter_scale: the kernel uses jitter * jitter_scale
    (one fp32 product where it reads the value: the same bits as scaling the tensor first, one launch less)."""
    R, n = t.shape
    S = n - 1
    assert weights.shape == (R, S)
    out = torch.empty_like(t)
    flag = torch.empty(1, device=t.device, dtype=torch.int32) if want_tail_flag else None
    check(lib().nm_resample_scaled(dptr(t), dptr(weights), dptr(jitter), float(jitter_scale), R, S, float(padding), int(bool(randomized)), dptr(out),
                                   dptr(flag, torch.int32), stream()), "nm_resample_scaled")
    return (out, flag) if want_tail_flag else out


class Fp16Guard:
    """Safety net of one fp16x3 blob (round 4): the device status block nm_nerf_fwd_fp16x3_ex writes (saturation flag + range
    telemetry, int32[16]) and the fp32 blob of the same parameters for the device-side fall-back nm_nerf_fwd_guarded.  NeRF.packed
    attaches one to every fp16x3 blob (`blob.nm_guard`); nerf_fwd refuses an fp16x3 blob without one.
    ONE STREAM PER STATUS BLOCK: the guarded pass consumes the flag (completion counter status[12], event count status[11]); two guarded
    launches on different streams against the same block could clear the flag before the other has read it.  A blob -- and with it its
    guard -- is used by one stream at a time (the evaluator's render stream; give a second concurrent renderer its own NeRF.packed blob)."""

    def __init__(self, device, blob32, act_log2=None):
        self.status = torch.zeros(16, dtype=torch.int32, device=device)
        self.blob32 = blob32
        self.act_log2 = None if act_log2 is None else [int(v) for v in act_log2]

    def read(self):
        """(saturated, ranges[10]) -- SYNCHRONISES; ranges are in the scaled units of the blob (divide by 2^act_log2).
        saturated = some launch on this block hit the fp16 limit since the last reset(): bit 0 of status[0] (raised by the fp16x3
        kernel, cleared again by the guarded fp32 pass that rewrote that launch) or the sticky event count status[11]."""
        h = self.status.cpu()
        return bool(int(h[0]) & 1) or int(h[11]) > 0, h[1:11].view(torch.float32).tolist()

    def reset(self):
        self.status.zero_()


def nerf_fwd(blob, rays, t, app_row=None, tap_layer=-1, white_bg=False, var_scale=-1.0, need_rgb=True, need_feat=True,
             feat_max=False, want_raw=False, want_sample_feat=False, zero_tail=False, tail_flag=None, guard=None):
    """One fused pass.  Returns dict(weights, feat, pts, rgb, depth, acc[, raw, sample_feat]).
    zero_tail: the intervals s > S/2 have zero width (t from `resample(..., randomized=True)`), see NM_NERF_ZERO_TAIL in the
    header; same outputs, about half the work on the bf16x3 path.  `tail_flag` = the device flag of
    `resample(..., want_tail_flag=True)` for this very `t`: the kernel then checks the premise itself and evaluates every
    sample when it does not hold; without it the caller vouches for the premise.
    fp16x3 blobs run GUARDED: the kernel raises a device flag when an operand reached the fp16 limit, and a second launch
    (the fp32 kernel, which exits at once unless that flag is set) rewrites the outputs -- no host synchronisation, never a
    silently clamped result.  guard: an Fp16Guard (default: the one NeRF.packed attached to the blob)."""
    R, n = t.shape
    S_req = n - 1
    # Row lengths the kernels tile natively: 32, 64 and multiples of 128 (a tile = 128 samples = 4 wavefronts of 32).  Any other
    # `num_pts` (the reference takes any, renderer.py:86,100,196-213) runs on the next native length with the fence posts padded by
    # copies of the last one: a zero-width interval has alpha = 0, i.e. weight EXACTLY 0 and transmittance unchanged, so every output
    # is the sum the unpadded row defines plus exact zeros (cost: the padded samples still go through the MLP -- 96 -> 128: +33 %).
    S = S_req if (S_req in (32, 64) or S_req % 128 == 0) else (32 if S_req < 32 else 64 if S_req < 64 else (S_req + 127) // 128 * 128)
    if S != S_req:
        if S_req < 1:
            raise _lib.NerfmatchAmdError("nerf_fwd needs at least one interval per ray")
        t = torch.cat([t, t[:, -1:].expand(R, S - S_req)], 1).contiguous()
        zero_tail, tail_flag = False, None  # (the premise is about the second half of the REQUESTED row)
    dev = rays.device
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    out = dict(weights=new(R, S), pts=new(R, 3), depth=new(R), acc=new(R))
    out["feat"] = new(R, 256) if need_feat else None
    out["rgb"] = new(R, 3) if need_rgb else None
    out["raw"] = new(R, S, 4) if want_raw else None
    out["sample_feat"] = new(R, S, 256) if want_sample_feat else None
    flags = (0 if need_rgb else _lib.NM_NERF_SKIP_RGB) | (_lib.NM_NERF_FEAT_MAX if feat_max else 0) | (_lib.NM_NERF_ZERO_TAIL if zero_tail else 0)
    # the blob's dtype tells the kernel family: fp32 blob -> fp32 MFMA kernel, uint8 -> bf16x3 split, int16 -> fp16x3 split, float16 -> fp16x1
    common = (dptr(rays), dptr(t), dptr(app_row), R, S, int(tap_layer), int(bool(white_bg)),
              float(var_scale), flags, dptr(out["weights"]), dptr(out["feat"]), dptr(out["pts"]),
              dptr(out["rgb"]), dptr(out["depth"]), dptr(out["acc"]), dptr(out["raw"]), dptr(out["sample_feat"]))
    if blob.dtype == torch.uint8:
        ws = _nerf_workspace(dev) if (need_feat or want_sample_feat) else None
        check(lib().nm_nerf_fwd_bf16x3_ex(dptr(blob, torch.uint8), *common, dptr(ws, torch.uint8), dptr(tail_flag, torch.int32), stream()),
              "nm_nerf_fwd_bf16x3_ex")
    elif blob.dtype == torch.int16:  # fp16 hi/lo-split blob (NeRF.packed(device, "fp16x3"))
        g = guard if guard is not None else getattr(blob, "nm_guard", None)
        if g is None:
            raise _lib.NerfmatchAmdError("fp16x3 blob without an Fp16Guard: take it from NeRF.packed(device, 'fp16x3') (operands beyond "
                                         "+-65504 would be clamped silently)")
        ws = _nerf_workspace(dev) if (need_feat or want_sample_feat) else None
        check(lib().nm_nerf_fwd_fp16x3_ex(dptr(blob, torch.int16), *common, dptr(ws, torch.uint8), dptr(tail_flag, torch.int32),
                                          dptr(g.status, torch.int32), stream()), "nm_nerf_fwd_fp16x3_ex")
        check(lib().nm_nerf_fwd_guarded(dptr(g.blob32), *common, dptr(g.status, torch.int32), stream()), "nm_nerf_fwd_guarded")
    elif blob.dtype == torch.float16:  # single-product fp16 blob (NeRF.packed(device, "fp16x1"))
        ws = _nerf_workspace(dev) if (need_feat or want_sample_feat) else None
        check(lib().nm_nerf_fwd_fp16x1(dptr(blob, torch.float16), *common, dptr(ws, torch.uint8), dptr(tail_flag, torch.int32), stream()),
              "nm_nerf_fwd_fp16x1")
    else:
        check(lib().nm_nerf_fwd(dptr(blob), *common, stream()), "nm_nerf_fwd")
    if S != S_req:
        for k in ("weights", "raw", "sample_feat"):
            if out[k] is not None:
                out[k] = out[k][:, :S_req].contiguous()
    return out


_NERF_WS = {}


def _nerf_workspace(dev):
    """Scratch of the persistent bf16x3 kernel, one per (device, stream): calls on one stream are serialised."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _NERF_WS.get(key)
    if ws is None:
        ws = _NERF_WS[key] = torch.empty(lib().nm_nerf_workspace_bytes_bf16x3(), dtype=torch.uint8, device=dev)
    return ws


def unnormalize_points(pts, unnorm):
    m = unnorm.detach().to("cpu", torch.float32).contiguous()
    out = torch.empty_like(pts)
    check(lib().nm_unnormalize_points(dptr(pts), hptr(m), pts.shape[0], dptr(out), stream()), "nm_unnormalize_points")
    return out


# ----------------------------------------------------------------------------- matcher half
# Arithmetic of the nn.Linear layers: "fp32" (v_mfma_f32_32x32x2_f32, nm_linear) or "bf16x3" (bf16 MFMA on hi/lo-split
# operands, fp32-accurate, nm_linear_bf16x3).  Module-level switch like ATTENTION_PRECISION.
LINEAR_PRECISION = "fp32"
_LINEAR_BLOBS = {}
_LINEAR_LIMIT = 256   # entries per generation
_LINEAR_GRAVE = []    # the previous generation (see _linear_evict)
_LINEAR_RECENT = __import__("collections").deque(maxlen=16)  # the last blobs handed out stay allocated whatever the limit is


def _linear_evict():
    """Start a new generation of the blob cache.  The entries of the old one stay ALLOCATED until the next eviction: an op that takes
    several blobs (the encoder tail takes three) fetches their device pointers one after the other and launches afterwards -- had the
    second fetch freed the first blob (round 1-4: `_LINEAR_BLOBS.clear()`), the third one's allocation could land on its memory and the pack
    kernel would overwrite weights the launch had not consumed yet.  (Seen in round 5 as an order-dependent test failure once the suite
    had created more than 256 weight tensors; a training loop, whose parameter versions change every step, evicts every few steps.)"""
    global _LINEAR_GRAVE
    _LINEAR_GRAVE = list(_LINEAR_BLOBS.values())
    _LINEAR_BLOBS.clear()


def _linear_blob(weight):
    """Split / re-ordered copy of a weight matrix for nm_linear_bf16x3, cached until the tensor changes."""
    w = weight.detach()
    key = (w.data_ptr(), w._version, tuple(w.shape), w.device.index)
    hit = _LINEAR_BLOBS.get(key)
    if hit is None:
        N, K = w.shape
        blob = torch.empty(lib().nm_linear_blob_bytes_bf16x3(N, K), dtype=torch.uint8, device=w.device)
        wc = w.contiguous()
        check(lib().nm_linear_pack_bf16x3(dptr(wc), N, K, dptr(blob, torch.uint8), stream()), "nm_linear_pack_bf16x3")
        if len(_LINEAR_BLOBS) >= _LINEAR_LIMIT:
            _linear_evict()
        hit = _LINEAR_BLOBS[key] = (blob, w)  # keeps the source tensor alive so that its data_ptr is not reused
    _LINEAR_RECENT.append(hit)
    return hit[0]


def _linear_blob_perm(weight):
    """The same blob with the K dimension in accumulator order (nm_linear_pack_perm_bf16x3): the weights of a product whose
    input is the previous product's output still sitting in accumulator registers (nm_encoder_tail_bf16x3)."""
    w = weight.detach()
    key = ("perm", w.data_ptr(), w._version, tuple(w.shape), w.device.index)
    hit = _LINEAR_BLOBS.get(key)
    if hit is None:
        N, K = w.shape
        blob = torch.empty(lib().nm_linear_blob_bytes_bf16x3(N, K), dtype=torch.uint8, device=w.device)
        wc = w.contiguous()
        check(lib().nm_linear_pack_perm_bf16x3(dptr(wc), N, K, dptr(blob, torch.uint8), stream()), "nm_linear_pack_perm_bf16x3")
        if len(_LINEAR_BLOBS) >= _LINEAR_LIMIT:
            _linear_evict()
        hit = _LINEAR_BLOBS[key] = (blob, w)
    _LINEAR_RECENT.append(hit)
    return hit[0]


def _linear_blob_t(weight):
    """The packed blob of weight.T made straight from the stored (N_out, K_in) tensor (nm_linear_pack_t_bf16x3): a training step, whose
    parameter versions change every step, then needs neither the transposed copy nor a second trip through the cache for it."""
    w = weight.detach()
    key = ("packT", w.data_ptr(), w._version, tuple(w.shape), w.device.index)
    hit = _LINEAR_BLOBS.get(key)
    if hit is None:
        K, N = w.shape  # stored (out = K of the product, in = N of the product): the packed matrix is (N, K)
        blob = torch.empty(lib().nm_linear_blob_bytes_bf16x3(N, K), dtype=torch.uint8, device=w.device)
        wc = w.contiguous()
        check(lib().nm_linear_pack_t_bf16x3(dptr(wc), N, K, dptr(blob, torch.uint8), stream()), "nm_linear_pack_t_bf16x3")
        if len(_LINEAR_BLOBS) >= _LINEAR_LIMIT:
            _linear_evict()
        hit = _LINEAR_BLOBS[key] = (blob, w)
    _LINEAR_RECENT.append(hit)
    return hit[0]


def linear_t(dy, weight):
    """dy (..., N_out) @ weight (N_out, K_in) -> (..., K_in): a linear layer's input gradient.  Split-bf16 arithmetic: nm_linear_bf16x3 on the
    transposed-pack blob; otherwise nm_linear on the cached transposed copy (ops.transposed)."""
    No, Ki = weight.shape
    if LINEAR_PRECISION == "bf16x3" and No % 8 == 0 and Ki % 8 == 0:
        d2 = dy.reshape(-1, No).contiguous()
        y = torch.empty(d2.shape[0], Ki, device=dy.device, dtype=torch.float32)
        if d2.shape[0]:
            check(lib().nm_linear_ex_bf16x3(dptr(d2), dptr(_linear_blob_t(weight), torch.uint8), None, None, None, None, d2.shape[0], Ki, No,
                                            _lib.NM_ACT_NONE, dptr(y), stream()), "nm_linear_ex_bf16x3")
        return y.reshape(*dy.shape[:-1], Ki)
    return linear(dy, transposed(weight))


def transposed(weight):
    """weight.T as a contiguous tensor, cached until the tensor changes (same generations as the blobs): the dX product of a linear
    layer's backward pass, dy @ W, is nm_linear with W^T as the weight.  A fresh `.t().contiguous()` per call is a copy kernel AND a
    miss of the blob cache (a new tensor every time: one pack per layer and call); with frozen parameters -- the matching term of the
    iNeRF refinement differentiates through the matcher five times per query -- both happen once."""
    w = weight.detach()
    key = ("T", w.data_ptr(), w._version, tuple(w.shape), w.device.index)
    hit = _LINEAR_BLOBS.get(key)
    if hit is None:
        if len(_LINEAR_BLOBS) >= _LINEAR_LIMIT:
            _linear_evict()
        hit = _LINEAR_BLOBS[key] = (w.t().contiguous(), w)
    _LINEAR_RECENT.append(hit)
    return hit[0]


ENCODER_TAIL_FUSED = True  # False: the four separate launches (A/B runs, tests)


def encoder_tail_supported(dim, inner, hidden, act):
    """nm_encoder_tail_bf16x3 takes model dim = attention inner dim = FFN hidden dim = 256, GELU, split-bf16 arithmetic."""
    return ENCODER_TAIL_FUSED and LINEAR_PRECISION == "bf16x3" and dim == inner == hidden == 256 and act == _lib.NM_ACT_GELU


def encoder_tail(att, xh, w_out, norm2, ffn0, ffn2):
    """y = xh + FFN(LN2(xh + att @ w_out.T)) as ONE kernel (csrc/encoder_tail.hip).  att, xh (..., 256); norm2: nn.LayerNorm;
    ffn0 / ffn2: the two nn.Linear layers of the feed-forward network."""
    dim = xh.shape[-1]
    a2, x2 = att.reshape(-1, dim).contiguous(), xh.reshape(-1, dim).contiguous()
    y = torch.empty_like(x2)
    if x2.shape[0]:
        check(lib().nm_encoder_tail_bf16x3(dptr(a2), dptr(x2), dptr(_linear_blob(w_out), torch.uint8), dptr(_linear_blob_perm(ffn0.weight), torch.uint8),
                                           dptr(_linear_blob_perm(ffn2.weight), torch.uint8), dptr(norm2.weight), dptr(norm2.bias), dptr(ffn0.bias),
                                           dptr(ffn2.bias), x2.shape[0], dim, float(norm2.eps), dptr(y), stream()), "nm_encoder_tail_bf16x3")
    return y.reshape(xh.shape)


ENCODER_TAIL_BWD_FUSED = True  # False: the separate backward launches (A/B runs, tests)


def encoder_tail_save(att, xh, w_out, norm2, ffn0, ffn2):
    """(y, a, u): encoder_tail's y (bit-identical) plus the two intermediates encoder_tail_bwd needs, in ONE launch (nm_encoder_tail_save_bf16x3)."""
    dim = xh.shape[-1]
    a2, x2 = att.reshape(-1, dim).contiguous(), xh.reshape(-1, dim).contiguous()
    y, a, u = torch.empty_like(x2), torch.empty_like(x2), torch.empty_like(x2)
    if x2.shape[0]:
        check(lib().nm_encoder_tail_save_bf16x3(dptr(a2), dptr(x2), dptr(_linear_blob(w_out), torch.uint8), dptr(_linear_blob_perm(ffn0.weight), torch.uint8),
                                                dptr(_linear_blob_perm(ffn2.weight), torch.uint8), dptr(norm2.weight), dptr(norm2.bias), dptr(ffn0.bias),
                                                dptr(ffn2.bias), x2.shape[0], dim, float(norm2.eps), dptr(y), dptr(a), dptr(u), stream()),
              "nm_encoder_tail_save_bf16x3")
    return y.reshape(xh.shape), a, u


def encoder_tail_bwd(dy, a_pre, u_pre, w_out, norm2, ffn0, ffn2):
    """(d_att, d_xh) of y = xh + FFN(LN2(xh + att @ w_out.T)) for FROZEN parameters in ONE launch (csrc/encoder_tail_bwd.hip): a_pre / u_pre are
    the forward pass's LayerNorm-2 input and GELU input.  The blobs are those of the TRANSPOSED matrices (cached: the parameters are frozen)."""
    dim = dy.shape[-1]
    d2, a2, u2 = dy.reshape(-1, dim).contiguous(), a_pre.reshape(-1, dim).contiguous(), u_pre.reshape(-1, dim).contiguous()
    d_att, d_xh = torch.empty_like(d2), torch.empty_like(d2)
    if d2.shape[0]:
        check(lib().nm_encoder_tail_bwd_bf16x3(dptr(d2), dptr(a2), dptr(u2), dptr(_linear_blob(transposed(ffn2.weight)), torch.uint8),
                                               dptr(_linear_blob_perm(transposed(ffn0.weight)), torch.uint8),
                                               dptr(_linear_blob_perm(transposed(w_out)), torch.uint8), dptr(norm2.weight), d2.shape[0], dim,
                                               float(norm2.eps), dptr(d_att), dptr(d_xh), stream()), "nm_encoder_tail_bwd_bf16x3")
    return d_att.reshape(dy.shape), d_xh.reshape(dy.shape)


def linear(x, weight, bias=None, residual=None, act=_lib.NM_ACT_NONE, pre=None, gate=None):
    """y = (act(x @ weight.T + bias + pre) + residual) * [gate > 0] for x (..., K); weight (N, K) as stored by nn.Linear."""
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K).contiguous()
    M = x2.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    if M == 0:
        return y.reshape(*x.shape[:-1], N)
    r2 = None if residual is None else residual.reshape(-1, N).contiguous()
    p2 = None if pre is None else pre.reshape(-1, N).contiguous()
    g2 = None if gate is None else gate.reshape(-1, N).contiguous()
    if LINEAR_PRECISION == "bf16x3" and K % 8 == 0 and N % 8 == 0:
        check(lib().nm_linear_ex_bf16x3(dptr(x2), dptr(_linear_blob(weight), torch.uint8), dptr(bias), dptr(p2), dptr(r2), dptr(g2), M, N, K,
                                        int(act), dptr(y), stream()), "nm_linear_ex_bf16x3")
    elif LINEAR_PRECISION in ("fp32", "bf16x3"):
        check(lib().nm_linear_ex(dptr(x2), dptr(weight), dptr(bias), dptr(p2), dptr(r2), dptr(g2), M, N, K, int(act), dptr(y), stream()),
              "nm_linear_ex")
    else:
        raise _lib.NerfmatchAmdError(f"LINEAR_PRECISION must be 'fp32' or 'bf16x3', got {LINEAR_PRECISION!r}")
    return y.reshape(*x.shape[:-1], N)


def layernorm(x, gamma, beta, eps=1e-5):
    dim = x.shape[-1]
    x2 = x.reshape(-1, dim).contiguous()
    y = torch.empty_like(x2)
    if x2.shape[0]:
        check(lib().nm_layernorm(dptr(x2), dptr(gamma), dptr(beta), x2.shape[0], dim, float(eps), dptr(y), stream()), "nm_layernorm")
    return y.reshape(x.shape)


def layernorm_pair(x0, ln0, x1, ln1):
    """(LN0(x0), LN1(x1)) in ONE launch (nm_layernorm2) when both have the same width -- the two pre-norms of a cross-attention layer;
    a row's arithmetic is nm_layernorm's, so the results are the same bits as two calls of `layernorm`."""
    dim = x0.shape[-1]
    if x1.shape[-1] != dim or x0.numel() == 0 or x1.numel() == 0:
        return layernorm(x0, ln0.weight, ln0.bias, ln0.eps), layernorm(x1, ln1.weight, ln1.bias, ln1.eps)
    a, b = x0.reshape(-1, dim).contiguous(), x1.reshape(-1, dim).contiguous()
    ya, yb = torch.empty_like(a), torch.empty_like(b)
    check(lib().nm_layernorm2(dptr(a), dptr(ln0.weight), dptr(ln0.bias), a.shape[0], float(ln0.eps), dptr(ya), dptr(b), dptr(ln1.weight), dptr(ln1.bias),
                              b.shape[0], float(ln1.eps), dptr(yb), dim, stream()), "nm_layernorm2")
    return ya.reshape(x0.shape), yb.reshape(x1.shape)


# Arithmetic of the two attention contractions: "fp32" (v_mfma_f32_32x32x2_f32), "bf16x3" (bf16 MFMA on hi/lo-split
# operands, fp32-accurate) or "fp8" (ONE e4m3 MFMA per product block: the THROUGHPUT configuration of BASELINE config 5, not a
# parity arithmetic; head_dim 32 sequences only, everything else falls to bf16x3).  Module-level switch so that the encoder
# modules need no extra plumbing.
ATTENTION_PRECISION = "fp32"


def _attn_flags():
    if ATTENTION_PRECISION not in ("fp32", "bf16x3", "fp8"):
        raise _lib.NerfmatchAmdError(f"ATTENTION_PRECISION must be 'fp32', 'bf16x3' or 'fp8', got {ATTENTION_PRECISION}")
    return _lib.NM_ATTN_BF16X3 if ATTENTION_PRECISION in ("bf16x3", "fp8") else 0


_ATTN_FP8_WS = {}


def _attention_fp8(qp, kp, vp_, ldq, ldk, ldv, B, L, S, heads, scale, out, dev):
    need = lib().nm_attention_fp8_workspace_bytes(int(B), int(S), int(heads))
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _ATTN_FP8_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _ATTN_FP8_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    check(lib().nm_attention_fp8(qp, kp, vp_, ldq, ldk, ldv, int(B), int(L), int(S), int(heads), float(scale), dptr(ws, torch.uint8), dptr(out),
                                 stream()), "nm_attention_fp8")


def _use_fp8(L, S, head_dim):
    return ATTENTION_PRECISION == "fp8" and head_dim == 32 and not (L <= 64 and S <= 64)


def attention(q, k, v, heads, scale):
    """q (B,L,C), k/v (B,S,C) -> (B,L,C); softmax((q*scale).k) v per head."""
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    B, L, Cc = q.shape
    S = k.shape[1]
    out = torch.empty_like(q)
    if B * L and _use_fp8(L, S, Cc // heads):
        _attention_fp8(dptr(q), dptr(k), dptr(v), Cc, Cc, Cc, B, L, S, heads, scale, out, q.device)
    elif B * L:
        flags = _attn_flags()
        with _probe("nm_attention_ws", 4.0 * B * L * S * Cc):
            check(lib().nm_attention_ws(dptr(q), dptr(k), dptr(v), Cc, Cc, Cc, B, L, S, int(heads), Cc // heads, float(scale), flags,
                                        _attn_workspace(q.device, B, S, heads, flags, L, Cc // heads), dptr(out), stream()), "nm_attention_ws")
    return out


_ATTN_WS = {}


def _attn_workspace(dev, B, S, heads, flags, L=None, head_dim=32):
    """Scratch for the pre-split K / V operands of the bf16x3 kernel, one (growing) buffer per (device, stream).  NULL for the
    shapes nm_attention_ws does not route to that kernel (head dim != 32, or the <= 64-token windows of the fine stage: with
    thousands of windows the request would be ~1 GB and re-grow -- a device allocation, milliseconds -- whenever a batch had
    more matches than any before)."""
    if not (flags & _lib.NM_ATTN_BF16X3) or head_dim != 32 or (L is not None and L <= 64 and S <= 64):
        return C.c_void_p(0)
    need = lib().nm_attention_workspace_bytes(int(B), int(S), int(heads))
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _ATTN_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _ATTN_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    return dptr(ws, torch.uint8)


def lse_supported(L, S, head_dim):
    """True when the attention forward can keep the log-sum-exp for the backward pass (nm_attention_ws_lse: the split-bf16 kernel)."""
    return ATTENTION_PRECISION == "bf16x3" and head_dim == 32 and not (L <= 64 and S <= 64)


def attention_fused(qkv, q_cols, k_cols, v_cols, B, L, S, heads, scale, kv=None, want_lse=False):
    """Attention reading q / k / v as column slices of fused projection buffers.
    qkv: (B*L, ld) holding q at column offset q_cols (and k, v too when kv is None); kv: (B*S, ldkv) holding k and v.
    want_lse (training forward): -> (out, nlse) with nlse (B, heads, L) = -(log-sum-exp), log2 domain, for attention_bwd_fused -- or
    (out, None) where the kernel at hand does not keep it."""
    src_kv = qkv if kv is None else kv
    ldq, ldkv = qkv.shape[1], src_kv.shape[1]
    dim = (k_cols[1] - k_cols[0])
    out = torch.empty(B * L, dim, device=qkv.device, dtype=torch.float32)
    esz = 4
    qp = C.c_void_p(qkv.data_ptr() + q_cols[0] * esz)
    kp = C.c_void_p(src_kv.data_ptr() + k_cols[0] * esz)
    vp_ = C.c_void_p(src_kv.data_ptr() + v_cols[0] * esz)
    assert qkv.is_contiguous() and src_kv.is_contiguous() and qkv.dtype == torch.float32
    if _use_fp8(L, S, dim // heads):
        _attention_fp8(qp, kp, vp_, ldq, ldkv, ldkv, B, L, S, heads, scale, out, qkv.device)
        return (out.reshape(B, L, dim), None) if want_lse else out.reshape(B, L, dim)
    flags = _attn_flags()
    if want_lse and lse_supported(L, S, dim // heads):
        nlse = torch.empty(B, heads, L, device=qkv.device, dtype=torch.float32)
        check(lib().nm_attention_ws_lse(qp, kp, vp_, ldq, ldkv, ldkv, B, L, S, int(heads), dim // heads, float(scale), flags,
                                        _attn_workspace(qkv.device, B, S, heads, flags, L, dim // heads), dptr(out), dptr(nlse), stream()),
              "nm_attention_ws_lse")
        return out.reshape(B, L, dim), nlse
    with _probe("nm_attention_ws", 4.0 * B * L * S * dim):
        check(lib().nm_attention_ws(qp, kp, vp_, ldq, ldkv, ldkv, B, L, S, int(heads), dim // heads, float(scale), flags,
                                    _attn_workspace(qkv.device, B, S, heads, flags, L, dim // heads), dptr(out), stream()), "nm_attention_ws")
    return (out.reshape(B, L, dim), None) if want_lse else out.reshape(B, L, dim)


def projected_attention_supported(K, heads, head_dim, L, S):
    """True when attention_projected can take the projection + attention of a layer (split-bf16 arithmetic on both, head dim
    32, whole 128-column chunks per role, whole 32-key tiles)."""
    return (LINEAR_PRECISION == "bf16x3" and ATTENTION_PRECISION == "bf16x3" and head_dim == 32 and (32 * heads) % 128 == 0 and
            S % 32 == 0 and K % 8 == 0 and not (L <= 64 and S <= 64))


def attention_projected(x_q, w_q, x_kv, w_stack, B, L, S, heads, scale):
    """softmax(q k^T scale) v with the projections fused in front: the keys and values never exist as fp32 rows -- the
    projection GEMM writes them split into bf16 hi / lo parts, laid out as the MFMA operands of the attention kernel.
    Self attention: x_kv is None, x_q (B*L, K), w_stack = [Wq; Wk; Wv].  Cross attention: x_q (B*L, K) is projected with w_q
    by the plain GEMM, x_kv (B*S, K) with w_stack = [Wk; Wv]."""
    dev = x_q.device
    inner = 32 * heads
    flags = _attn_flags()
    ws = _attn_workspace(dev, B, S, heads, flags, L, 32)
    if x_kv is None:
        x2 = x_q.contiguous()
        q = torch.empty(B * L, inner, device=dev, dtype=torch.float32)
        check(lib().nm_linear_qkv_bf16x3(dptr(x2), dptr(_linear_blob(w_stack), torch.uint8), B * L, x2.shape[1], inner, int(heads), int(S),
                                         dptr(q), ws, stream()), "nm_linear_qkv_bf16x3")
    else:
        q = linear(x_q, w_q)
        x2 = x_kv.contiguous()
        check(lib().nm_linear_qkv_bf16x3(dptr(x2), dptr(_linear_blob(w_stack), torch.uint8), B * S, x2.shape[1], 0, int(heads), int(S), None, ws,
                                         stream()), "nm_linear_qkv_bf16x3")
    out = torch.empty(B * L, inner, device=dev, dtype=torch.float32)
    with _probe("attn32_v3_kernel", 4.0 * B * L * S * inner):  # exactly one launch of the attention kernel (operands pre-split by the GEMM)
        check(lib().nm_attention_presplit(dptr(q), inner, ws, int(B), int(L), int(S), int(heads), float(scale), dptr(out), stream()),
              "nm_attention_presplit")
    return out.reshape(B, L, inner)


def nchw_to_tokens(x, pe_table=None):
    """(B,C,h,w) -> (B,h*w,C), optionally adding the sine PE table (C,Hmax,Wmax)."""
    x = x.contiguous()
    B, Cc, h, w = x.shape
    y = torch.empty(B, h * w, Cc, device=x.device, dtype=torch.float32)
    th, tw = (pe_table.shape[1], pe_table.shape[2]) if pe_table is not None else (0, 0)
    check(lib().nm_add_sine_pe(dptr(x), dptr(pe_table), B, h, w, Cc, th, tw, dptr(y), stream()), "nm_add_sine_pe")
    return y


def feature_normalize(x):
    """x (B,N,D) is centred IN PLACE (per set b: minus the mean over its N rows); returns the centred set divided by its largest row norm
    (nm_feature_normalize; the coarse model's `pt_feat_norm` option)."""
    B, N, D = x.shape
    y = torch.empty_like(x)
    check(lib().nm_feature_normalize(dptr(x), B, N, D, dptr(y), stream()), "nm_feature_normalize")
    return y


def cat_fourier(feat, pt3d, num_freqs=15):
    """(n,C),(n,3) -> (n, ld) = [feat | x | sin/cos(2^f x)...] zero padded to a multiple of 8 columns."""
    n, Cc = feat.shape
    ld = ((Cc + 3 + 6 * num_freqs + 7) // 8) * 8
    out = torch.empty(n, ld, device=feat.device, dtype=torch.float32)
    check(lib().nm_cat_fourier(dptr(feat), dptr(pt3d), n, Cc, int(num_freqs), dptr(out), stream()), "nm_cat_fourier")
    return out


def cat_fourier_bwd(dy, pt3d, C, num_freqs=15):
    """d loss / d pt3d (n,3) from the gradient dy (n, ld) of cat_fourier's output."""
    dy, pt3d = dy.contiguous(), pt3d.contiguous()
    g = torch.empty_like(pt3d)
    if pt3d.shape[0]:
        check(lib().nm_cat_fourier_bwd(dptr(dy), dptr(pt3d), pt3d.shape[0], int(C), int(num_freqs), dptr(g), stream()), "nm_cat_fourier_bwd")
    return g


_ws_cache = {}


def _match_workspace(dev, need):
    """Scratch of the matching kernels (similarity matrix + statistics), one growing buffer per (device, stream): calls on one
    stream are serialised, calls on different streams must not share it."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = _ws_cache[key] = torch.empty(need, device=dev, dtype=torch.uint8)
    return ws


def invalidate_caches():
    """Drop every derived-weight cache of this module (packed bf16x3 weight blobs).  The caches are keyed on (data_ptr,
    _version) of the source tensor; writes through `.data` (EMA updates, manual weight surgery) do not bump `_version`, so
    call this -- and `module.invalidate()` on NeRF / matcher modules -- after such writes."""
    _linear_evict()


# Arithmetic of the similarity GEMM of the dual-softmax matcher: "fp32" or "bf16x3" (cf. LINEAR_PRECISION)
MATCH_PRECISION = "fp32"


def dual_softmax_match(im, pt, scale, im_mask=None, pt_mask=None, threshold=0.0, mutual=True, want_conf=True, want_norm=False,
                       conf_out=None, defer_count=False):
    """im (M,C), pt (N,C) -> dict(i_ids, j_ids, mconf [K], conf (M,N) | None, im_norm, pt_norm).
    K is read back from the device (one 4-byte D2H copy), as the reference's torch.where does implicitly."""
    M, Cc = im.shape
    N = pt.shape[0]
    dev = im.device
    L = lib()
    need = L.nm_match_workspace_bytes(M, N, Cc)
    ws = _match_workspace(dev, need)
    conf = (conf_out if conf_out is not None else torch.empty(M, N, device=dev, dtype=torch.float32)) if want_conf else None
    assert conf is None or (conf.shape == (M, N) and conf.is_contiguous())
    imn = torch.empty(M, Cc, device=dev, dtype=torch.float32) if want_norm else None
    ptn = torch.empty(N, Cc, device=dev, dtype=torch.float32) if want_norm else None
    oi = torch.empty(M, device=dev, dtype=torch.int64)
    oj = torch.empty(M, device=dev, dtype=torch.int64)
    oc = torch.empty(M, device=dev, dtype=torch.float32)
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    im_m, pt_m = _mask_u8(im_mask), _mask_u8(pt_mask)
    if MATCH_PRECISION not in ("fp32", "bf16x3"):
        raise _lib.NerfmatchAmdError(f"MATCH_PRECISION must be 'fp32' or 'bf16x3', got {MATCH_PRECISION!r}")
    flags = _lib.NM_MATCH_BF16X3 if MATCH_PRECISION == "bf16x3" else 0
    check(L.nm_dual_softmax_match_ex(dptr(im), dptr(pt), M, N, Cc, float(scale), dptr(im_m, torch.uint8), dptr(pt_m, torch.uint8),
                                     float(threshold), int(bool(mutual)), flags, dptr(conf), dptr(imn), dptr(ptn), dptr(oi, torch.int64),
                                     dptr(oj, torch.int64), dptr(oc), dptr(cnt, torch.int32), dptr(ws, torch.uint8), C.c_size_t(need),
                                     stream()), "nm_dual_softmax_match_ex")
    if defer_count:  # the caller reads `count` back later (one synchronisation for a whole batch) and slices itself
        return dict(i_ids=oi, j_ids=oj, mconf=oc, conf=conf, im_norm=imn, pt_norm=ptn, count=cnt)
    k = int(cnt.item())
    return dict(i_ids=oi[:k], j_ids=oj[:k], mconf=oc[:k], conf=conf, im_norm=imn, pt_norm=ptn, count=cnt)


MATCH_FUSED = True  # False: always the per-pair path that materialises the similarity matrix (A/B runs, tests)
_fused_ws = {}


def _dual_softmax_match_fused(im, pt, scale, im_mask, pt_mask, threshold, mutual):
    B, M, Cc = im.shape
    N = pt.shape[1]
    dev = im.device
    L = lib()
    need = L.nm_match_fused_workspace_bytes(B, M, N, Cc)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _fused_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _fused_ws[key] = torch.empty(need, device=dev, dtype=torch.uint8)
    # (one allocation for the three lists; the compaction kernel writes every slot: matches first, zeros behind the count -- the
    # single-pair path's speculative fine stage reads the first `cap` slots as indices before the count is known)
    buf = torch.empty(B * M * 20, device=dev, dtype=torch.uint8)
    oi = buf[: B * M * 8].view(torch.int64).view(B, M)
    oj = buf[B * M * 8: B * M * 16].view(torch.int64).view(B, M)
    oc = buf[B * M * 16:].view(torch.float32).view(B, M)
    cnt = torch.empty(B, device=dev, dtype=torch.int32)
    im_m, pt_m = _mask_u8(im_mask), _mask_u8(pt_mask)
    with _probe("nm_dual_softmax_match_fused", 2.0 * B * M * N * Cc):
        rc = L.nm_dual_softmax_match_fused(dptr(im), dptr(pt), B, M, N, Cc, float(scale), dptr(im_m, torch.uint8), dptr(pt_m, torch.uint8),
                                           float(threshold), int(bool(mutual)), dptr(oi, torch.int64), dptr(oj, torch.int64), dptr(oc),
                                           dptr(cnt, torch.int32), dptr(ws, torch.uint8), C.c_size_t(need), stream())
    if rc == _lib.NM_ERR_UNSUPPORTED:
        return None
    check(rc, "nm_dual_softmax_match_fused")
    return dict(i_ids=oi, j_ids=oj, mconf=oc, count=cnt, conf=None, im_norm=None, pt_norm=None)


def dual_softmax_match_batch(im, pt, scale, im_mask=None, pt_mask=None, threshold=0.0, mutual=True, want_conf=True, want_norm=False):
    """Batched form: im (B,M,C), pt (B,N,C) -> dict(i_ids, j_ids (B,M) int64, mconf (B,M), count (B,) int32 [device],
    conf (B,M,N) | None, im_norm, pt_norm).  One allocation per output for the whole batch and no per-element torch calls
    (the host issues the B kernel sequences back to back); the valid prefix of row b has count[b] entries."""
    B, M, Cc = im.shape
    N = pt.shape[1]
    dev = im.device
    L = lib()
    if MATCH_PRECISION not in ("fp32", "bf16x3"):
        raise _lib.NerfmatchAmdError(f"MATCH_PRECISION must be 'fp32' or 'bf16x3', got {MATCH_PRECISION!r}")
    flags = _lib.NM_MATCH_BF16X3 if MATCH_PRECISION == "bf16x3" else 0
    im, pt = im.contiguous(), pt.contiguous()
    if not want_conf and not want_norm and MATCH_PRECISION == "bf16x3" and MATCH_FUSED:
        # inference without the confidence matrix: the similarity never leaves the registers and the whole batch is ONE launch
        # sequence (csrc/match_fused.hip); shapes / temperatures it does not take fall through to the per-pair path below
        r = _dual_softmax_match_fused(im, pt, scale, im_mask, pt_mask, threshold, mutual)
        if r is not None:
            return r
    # (only the per-pair path needs the M x N similarity workspace -- ~92 MB at 4800^2; the fused path above never allocates it)
    need = L.nm_match_workspace_bytes(M, N, Cc)
    ws = _match_workspace(dev, need)
    conf = torch.empty(B, M, N, device=dev, dtype=torch.float32) if want_conf else None
    imn = torch.empty(B, M, Cc, device=dev, dtype=torch.float32) if want_norm else None
    ptn = torch.empty(B, N, Cc, device=dev, dtype=torch.float32) if want_norm else None
    oi = torch.zeros(B, M, device=dev, dtype=torch.int64)
    oj = torch.zeros(B, M, device=dev, dtype=torch.int64)
    oc = torch.zeros(B, M, device=dev, dtype=torch.float32)
    cnt = torch.zeros(B, device=dev, dtype=torch.int32)
    im_m, pt_m = _mask_u8(im_mask), _mask_u8(pt_mask)
    off = lambda t, b, stride: C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr() + b * stride)
    st, wsp = stream(), dptr(ws, torch.uint8)
    for b in range(B):
        check(L.nm_dual_softmax_match_ex(off(im, b, M * Cc * 4), off(pt, b, N * Cc * 4), M, N, Cc, float(scale), off(im_m, b, M), off(pt_m, b, N),
                                         float(threshold), int(bool(mutual)), flags, off(conf, b, M * N * 4), off(imn, b, M * Cc * 4),
                                         off(ptn, b, N * Cc * 4), off(oi, b, M * 8), off(oj, b, M * 8), off(oc, b, M * 4), off(cnt, b, 4),
                                         wsp, C.c_size_t(need), st), "nm_dual_softmax_match_ex")
    return dict(i_ids=oi, j_ids=oj, mconf=oc, count=cnt, conf=conf, im_norm=imn, pt_norm=ptn)


def fine_windows(ffeat_chw, i_ids, count, win=5, stride=4):
    """ffeat (C,Hf,Wf), i_ids (K,) int64 -> (K, win*win, C)."""
    Cc, Hf, Wf = ffeat_chw.shape
    K = i_ids.shape[0]
    out = torch.empty(K, win * win, Cc, device=ffeat_chw.device, dtype=torch.float32)
    if K:
        check(lib().nm_fine_windows(dptr(ffeat_chw), Cc, Hf, Wf, dptr(i_ids, torch.int64), dptr(count, torch.int32), K, int(win), int(stride),
                                    dptr(out), stream()), "nm_fine_windows")
    return out


def fine_windows_batch(ffeat, map_ids, i_ids, count, win=5, stride=4):
    """ffeat (B,C,Hf,Wf), map_ids / i_ids (K,) int64 -> (K, win*win, C): the window of match k comes from map map_ids[k]."""
    B, Cc, Hf, Wf = ffeat.shape
    K = i_ids.shape[0]
    out = torch.empty(K, win * win, Cc, device=ffeat.device, dtype=torch.float32)
    if K:
        check(lib().nm_fine_windows_batch(dptr(ffeat), B, Cc, Hf, Wf, dptr(map_ids, torch.int64), dptr(i_ids, torch.int64),
                                          dptr(count, torch.int32), K, int(win), int(stride), dptr(out), stream()), "nm_fine_windows_batch")
    return out


def assemble_matches(pt2d, pt3d, i_ids, j_ids, expec_f, mconf, win, fine_ds):
    """One pair's match assembly in one launch (nm_assemble_matches): -> mpt2d_c (K,2), mpt2d_f (K,2), mpt3d (K,3), pred_mask (K,) bool."""
    K, dev = i_ids.shape[0], pt2d.device
    c2, f2, p3 = (torch.empty(K, 2, device=dev), torch.empty(K, 2, device=dev), torch.empty(K, 3, device=dev))
    mask = torch.empty(K, device=dev, dtype=torch.bool)
    if K:
        check(lib().nm_assemble_matches(dptr(pt2d), dptr(pt3d), dptr(i_ids, torch.int64), dptr(j_ids, torch.int64), dptr(expec_f), dptr(mconf), K,
                                        float(win), float(fine_ds), dptr(c2), dptr(f2), dptr(p3), dptr(mask, torch.bool), stream()), "nm_assemble_matches")
    return c2, f2, p3, mask


FINE_LAYER_FUSED = True  # False: window gather + the generic layer kernels (A/B runs, tests)
FINE_STAGE_ONE_LAUNCH = True  # False: point side (nm_fine_pt_proj) and image side (nm_fine_window_layer) as two launches (A/B runs)


def fine_window_layer_supported(block, win_sz, C):
    """nm_fine_window_layer takes ONE pre-norm self-attention layer of width 128 with 8 heads of 16, bias-free attention projections, a
    GELU feed-forward 128 -> 128 -> 128 with biases, 5 x 5 windows (the shipped c2f configuration), split-bf16 arithmetic."""
    if not FINE_LAYER_FUSED or LINEAR_PRECISION != "bf16x3" or win_sz != 5 or C != 128:
        return False
    ok = block.__dict__.get("_nm_fwl_shape_ok")  # (the module's structure does not change after construction: examined once)
    if ok is None:
        ok = block.__dict__["_nm_fwl_shape_ok"] = _fine_window_layer_shape_ok(block)
    return ok


def _fine_window_layer_shape_ok(block):
    from .modules.attention import GenericEncoderLayer

    if len(block.layers) != 1:
        return False
    l = block.layers[0]
    if not isinstance(l, GenericEncoderLayer) or l.norm_type != "pre" or l.att_mode != "self":
        return False
    at, ff = l.attention, l.feedforward
    lin = (at.proj_q, at.proj_k, at.proj_v, at.proj_out[0])
    return (at.att_type == "full" and at.head_num == 8 and at.head_dim == 16 and all(m.bias is None and tuple(m.weight.shape) == (128, 128) for m in lin)
            and ff.act == _lib.NM_ACT_GELU and all(tuple(ff.layers[i].weight.shape) == (128, 128) and ff.layers[i].bias is not None for i in (0, 2))
            and tuple(l.norm1[0].weight.shape) == (128,) and tuple(l.norm2.weight.shape) == (128,))


def fine_window_layer(ffeat, map_ids, i_ids, count, block, stride=4, pt_f=None, pt_proj=None):
    """(K, 25, 128): the matches' 5 x 5 windows of `ffeat` through the block's one encoder layer, in one launch (nm_fine_window_layer).
    With pt_f (K, 128), the point-side fine features: returns FineMatching's expectation (K, 3) instead -- the layer's output is consumed
    inside the kernel and never stored.  With pt_proj = (src (rows, C0), ids (K,), lin0, lin1) the point side is computed inside too
    (nm_fine_stage: the whole fine stage in one launch)."""
    B, C, Hf, Wf = ffeat.shape
    K = i_ids.shape[0]
    want_expec = pt_f is not None or pt_proj is not None
    out = None if want_expec else torch.empty(K, 25, C, device=ffeat.device, dtype=torch.float32)
    expec = torch.empty(K, 3, device=ffeat.device, dtype=torch.float32) if want_expec else None
    if pt_f is not None:
        pt_f = pt_f.contiguous()
    p_src = p_ids = p_w0 = p_b0 = p_w1 = p_b1 = None
    p_c0 = 0
    if pt_proj is not None:
        p_src, p_ids, lin0, lin1 = pt_proj
        p_src, p_ids = p_src.contiguous(), p_ids.contiguous()
        p_c0 = p_src.shape[1]
        p_w0, p_w1 = transposed(lin0.weight), transposed(lin1.weight)
        p_b0 = None if lin0.bias is None else lin0.bias.detach()
        p_b1 = None if lin1.bias is None else lin1.bias.detach()
    if K:
        l = block.layers[0]
        at, ff, n1, n2 = l.attention, l.feedforward, l.norm1[0], l.norm2
        ffeat, map_ids, i_ids = ffeat.contiguous(), map_ids.contiguous(), i_ids.contiguous()
        ws = (at.proj_q.weight, at.proj_k.weight, at.proj_v.weight, at.proj_out[0].weight, ff.layers[0].weight, ff.layers[2].weight)
        # the six packed matrices, looked up once per parameter state (this sits between two launches of the one-query step's tail)
        vkey = (ws[0]._version, ws[1]._version, ws[2]._version, ws[3]._version, ws[4]._version, ws[5]._version, ws[0].data_ptr(), ws[5].data_ptr())
        hit = block.__dict__.get("_nm_fwl_blobs")
        # (an eviction -- invalidate_caches() after a write through .data, or the generation limit -- drops the entries: packed again then)
        if hit is None or hit[0] != vkey or any(_LINEAR_BLOBS.get(k) is None for k in hit[2]):
            blobs = [_linear_blob_perm(w) for w in ws]
            keys = [("perm", w.data_ptr(), w._version, tuple(w.shape), w.device.index) for w in ws]
            hit = block.__dict__["_nm_fwl_blobs"] = (vkey, blobs, keys)
        blobs = hit[1]
        b1, b2 = ff.layers[0].bias.detach(), ff.layers[2].bias.detach()
        u8 = torch.uint8
        check(lib().nm_fine_stage(dptr(ffeat), B, C, Hf, Wf, dptr(map_ids, torch.int64), dptr(i_ids, torch.int64), dptr(count, torch.int32), K, 5,
                                  int(stride), 8, dptr(n1.weight), dptr(n1.bias), float(n1.eps), dptr(blobs[0], u8), dptr(blobs[1], u8),
                                  dptr(blobs[2], u8), dptr(blobs[3], u8), dptr(n2.weight), dptr(n2.bias), float(n2.eps), dptr(blobs[4], u8), dptr(b1),
                                  dptr(blobs[5], u8), dptr(b2), float(at.attend.scale()), dptr(out), dptr(pt_f), dptr(p_src), dptr(p_ids, torch.int64),
                                  int(p_c0), dptr(p_w0), dptr(p_b0), dptr(p_w1), dptr(p_b1), dptr(expec), stream()), "nm_fine_stage")
    return expec if want_expec else out


FINE_PT_PROJ_FUSED = True  # False: gather + two nm_linear launches (A/B runs, tests)


def fine_pt_proj_supported(lin0, lin1):
    return (FINE_PT_PROJ_FUSED and lin1.out_features == 128 and lin1.in_features == 128 and lin0.out_features == 128
            and lin0.in_features % 4 == 0 and lin0.in_features <= 512)


def fine_pt_proj(src, ids, count, lin0, lin1):
    """out (K, 128) = lin1(lin0(src[ids])) for the first `count` slots (zeros behind them) in one launch (nm_fine_pt_proj): the point side
    of the fine stage, `pt_ffeat_proj`.  lin0 / lin1: the two nn.Linear modules; their transposed weights are cached (ops.transposed)."""
    K, dev = ids.shape[0], src.device
    out = torch.empty(K, 128, device=dev, dtype=torch.float32)
    if K:
        src, ids = src.contiguous(), ids.contiguous()
        w0t, w1t = transposed(lin0.weight), transposed(lin1.weight)
        b0 = None if lin0.bias is None else lin0.bias.detach()
        b1 = None if lin1.bias is None else lin1.bias.detach()
        check(lib().nm_fine_pt_proj(dptr(src), dptr(ids, torch.int64), dptr(count, torch.int32), K, src.shape[1], 128, dptr(w0t), dptr(b0), dptr(w1t),
                                    dptr(b1), dptr(out), stream()), "nm_fine_pt_proj")
    return out


def gather_rows(src, ids, count):
    K, dim = ids.shape[0], src.shape[1]
    out = torch.empty(K, dim, device=src.device, dtype=torch.float32)
    if K:
        check(lib().nm_gather_rows(dptr(src), dptr(ids, torch.int64), dptr(count, torch.int32), K, dim, dptr(out), stream()), "nm_gather_rows")
    return out


def fine_expectation(pt_f, win_f, count, win=5):
    K, ww, Cc = win_f.shape
    out = torch.empty(K, 3, device=pt_f.device, dtype=torch.float32)
    if K:
        check(lib().nm_fine_expectation(dptr(pt_f), dptr(win_f), dptr(count, torch.int32), K, int(win), Cc, dptr(out), stream()), "nm_fine_expectation")
    return out


# ----------------------------------------------------------------------------- training side
_WGRAD_WS = {}


def _scratch(cache, dev, need):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = cache.get(key)
    if ws is None or ws.numel() < need:
        ws = cache[key] = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    return ws


def _aligned16(t):
    """t, or a copy of it when its first element does not sit on a 16-byte boundary (a contiguous view at an odd offset into its storage)."""
    return t if t.data_ptr() % 16 == 0 else t.clone()


def linear_wgrad(dy, x, out=None):
    """dw (N,K) = dy(M,N)^T @ x(M,K)  (added onto `out` when given)."""
    dy, x = _aligned16(dy.contiguous()), _aligned16(x.contiguous())
    M, N = dy.shape
    K = x.shape[1]
    dw = out if out is not None else torch.empty(N, K, device=dy.device, dtype=torch.float32)
    if M == 0:
        return dw if out is not None else dw.zero_()
    need = lib().nm_linear_wgrad_workspace_bytes(M, N, K)
    ws = _scratch(_WGRAD_WS, dy.device, need)
    if LINEAR_PRECISION == "bf16x3" and N % 2 == 0 and K % 4 == 0:  # (the arithmetic of the dX GEMMs of the same backward pass; round 6: 3 x faster than the fp32-MFMA kernel)
        check(lib().nm_linear_wgrad_bf16x3(dptr(dy), dptr(x), M, N, K, int(out is not None), dptr(dw), dptr(ws, torch.uint8), ws.numel(), stream()),
              "nm_linear_wgrad_bf16x3")
        return dw
    if N % 4 or K % 4:
        raise _lib.NerfmatchAmdError("linear_wgrad (fp32 kernel): N and K must be multiples of 4")
    check(lib().nm_linear_wgrad(dptr(dy), dptr(x), M, N, K, int(out is not None), dptr(dw), dptr(ws, torch.uint8), ws.numel(), stream()),
          "nm_linear_wgrad")
    return dw


def linear_wgrad_bias(dy, x):
    """(dw, db) = (dy^T @ x, column sums of dy): one launch on the split-bf16 path (dy read once), two otherwise."""
    dy, x = _aligned16(dy.contiguous()), _aligned16(x.contiguous())
    M, N = dy.shape
    K = x.shape[1]
    if not (LINEAR_PRECISION == "bf16x3" and N % 2 == 0 and K % 4 == 0 and M > 0):
        return linear_wgrad(dy, x), col_sum(dy)
    dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)
    db = torch.empty(N, device=dy.device, dtype=torch.float32)
    ws = _scratch(_WGRAD_WS, dy.device, lib().nm_linear_wgrad_workspace_bytes(M, N, K))
    check(lib().nm_linear_wgrad_bias_bf16x3(dptr(dy), dptr(x), M, N, K, 0, dptr(dw), dptr(db), dptr(ws, torch.uint8), ws.numel(), stream()),
          "nm_linear_wgrad_bias_bf16x3")
    return dw, db


def col_sum(dy):
    dy = dy.contiguous()
    M, N = dy.shape
    out = torch.zeros(N, device=dy.device, dtype=torch.float32)
    if M:
        check(lib().nm_col_sum(dptr(dy), M, N, 1, dptr(out), stream()), "nm_col_sum")
    return out


def gelu(u):
    u = u.contiguous()
    h = torch.empty_like(u)
    if u.numel():
        check(lib().nm_gelu(dptr(u), u.numel(), dptr(h), stream()), "nm_gelu")
    return h


def gelu_bwd(u, dh):
    u, dh = u.contiguous(), dh.contiguous()
    du = torch.empty_like(u)
    if u.numel():
        check(lib().nm_gelu_bwd(dptr(u), dptr(dh), u.numel(), dptr(du), stream()), "nm_gelu_bwd")
    return du


def relu_bwd(h, dh):
    """dh where the ReLU's output h is positive, else 0."""
    h, dh = h.contiguous(), dh.contiguous()
    du = torch.empty_like(h)
    if h.numel():
        check(lib().nm_relu_bwd(dptr(h), dptr(dh), h.numel(), dptr(du), stream()), "nm_relu_bwd")
    return du


def layernorm_bwd(x, gamma, dy, eps=1e-5, param_grads=True):
    """-> dx (like x), dgamma (dim), dbeta (dim); param_grads=False: (dx, None, None) -- no parameter-gradient reduction, no zero fills."""
    dim = x.shape[-1]
    x2, dy2 = x.reshape(-1, dim).contiguous(), dy.reshape(-1, dim).contiguous()
    dx = torch.empty_like(x2)
    dg = db = None
    if param_grads:
        dg_db = torch.zeros(2, dim, device=x.device, dtype=torch.float32)  # (one fill launch for both)
        dg, db = dg_db[0], dg_db[1]
    if x2.shape[0]:
        check(lib().nm_layernorm_bwd(dptr(x2), dptr(gamma), dptr(dy2), x2.shape[0], dim, float(eps), dptr(dx), dptr(dg), dptr(db), stream()),
              "nm_layernorm_bwd")
    return dx.reshape(x.shape), dg, db


def l2norm_bwd(f, dy):
    f, dy = f.contiguous(), dy.contiguous()
    df = torch.empty_like(f)
    if f.shape[0]:
        check(lib().nm_l2norm_bwd(dptr(f), dptr(dy), f.shape[0], f.shape[1], dptr(df), stream()), "nm_l2norm_bwd")
    return df


_ATTN_BWD_WS = {}


def attention_bwd_fused(q_src, q_col, kv_src, k_col, v_col, o, d_o, B, L, S, heads, scale, nlse=None):
    """Backward of attention_fused: q / k / v are column slices (offsets in floats) of the fused projection buffers q_src
    (B*L, ldq) and kv_src (B*S, ldkv).  Returns (d_q_src, d_kv_src) of the same shapes with the three slices filled (other
    columns zero); d_kv_src is d_q_src when both are the same buffer (self attention)."""
    dim = o.shape[-1]
    o2, d2 = o.reshape(B * L, dim).contiguous(), d_o.reshape(B * L, dim).contiguous()
    same = kv_src is q_src
    # (every column is written when the buffers hold exactly the slices: [q | k | v], or q alone beside [k | v]: no zero fill then)
    dq_src = torch.empty_like(q_src) if q_src.shape[1] == (3 * dim if same else dim) else torch.zeros_like(q_src)
    dkv_src = dq_src if same else (torch.empty_like(kv_src) if kv_src.shape[1] == 2 * dim else torch.zeros_like(kv_src))
    if B * L == 0 or S == 0:
        return dq_src.zero_(), dkv_src.zero_()
    flags = _attn_flags() if dim // heads == 32 else 0
    need = lib().nm_attention_bwd_workspace_bytes(B, L, S, int(heads), flags)
    ws = _scratch(_ATTN_BWD_WS, q_src.device, need)
    ldq, ldkv = q_src.shape[1], kv_src.shape[1]
    off = lambda t, c: C.c_void_p(t.data_ptr() + 4 * c)
    if nlse is not None and not (flags & _lib.NM_ATTN_BF16X3):
        nlse = None  # (the precision switch moved between forward and backward: the fp32 kernels rebuild the log-sum-exp themselves)
    check(lib().nm_attention_bwd_lse(off(q_src, q_col), off(kv_src, k_col), off(kv_src, v_col), dptr(o2), dptr(d2), ldq, ldkv, ldkv, dim, dim,
                                     B, L, S, int(heads), dim // heads, float(scale), off(dq_src, q_col), off(dkv_src, k_col),
                                     off(dkv_src, v_col), ldq, ldkv, ldkv, flags, dptr(nlse), dptr(ws, torch.uint8), ws.numel(), stream()),
          "nm_attention_bwd_lse")
    return dq_src, dkv_src


def attention_bwd(q, k, v, o, d_o, heads, scale):
    """Gradients of softmax attention: q, o, d_o (B,L,C); k, v (B,S,C) -> dq, dk, dv (arithmetic per ATTENTION_PRECISION)."""
    q, k, v, o, d_o = (t.contiguous() for t in (q, k, v, o, d_o))
    B, L, Cc = q.shape
    S = k.shape[1]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    if B * L == 0 or S == 0:
        return dq.zero_(), dk.zero_(), dv.zero_()
    flags = _attn_flags() if Cc // heads == 32 else 0
    need = lib().nm_attention_bwd_workspace_bytes(B, L, S, int(heads), flags)
    ws = _scratch(_ATTN_BWD_WS, q.device, need)
    check(lib().nm_attention_bwd(dptr(q), dptr(k), dptr(v), dptr(o), dptr(d_o), Cc, Cc, Cc, Cc, Cc, B, L, S, int(heads), Cc // heads,
                                 float(scale), dptr(dq), dptr(dk), dptr(dv), Cc, Cc, Cc, flags, dptr(ws, torch.uint8), ws.numel(), stream()),
          "nm_attention_bwd")
    return dq, dk, dv


def fine_windows_bwd(dwin, shape_chw, i_ids, count, win=5, stride=4, out=None):
    """Scatter-add of window gradients (K, win*win, C) into a (C,Hf,Wf) map (`out` accumulated onto when given)."""
    Cc, Hf, Wf = shape_chw
    K = i_ids.shape[0]
    d = out if out is not None else torch.zeros(Cc, Hf, Wf, device=dwin.device, dtype=torch.float32)
    if K:
        dwin = dwin.contiguous()
        check(lib().nm_fine_windows_bwd(dptr(dwin), Cc, Hf, Wf, dptr(i_ids, torch.int64), dptr(count, torch.int32), K, int(win),
                                        int(stride), dptr(d), stream()), "nm_fine_windows_bwd")
    return d


def fine_expectation_bwd(pt_f, win_f, d_expec, count, win=5):
    K, ww, Cc = win_f.shape
    d_pt, d_win = torch.empty_like(pt_f), torch.empty_like(win_f)
    if K:
        d_expec = d_expec.contiguous()
        check(lib().nm_fine_expectation_bwd(dptr(pt_f), dptr(win_f), dptr(d_expec), dptr(count, torch.int32), K, int(win), Cc,
                                            dptr(d_pt), dptr(d_win), stream()), "nm_fine_expectation_bwd")
    return d_pt, d_win
