"""numpy model of the wavefront-level dataflow of nerf_fwd.hip (CPU-only test helper).

Emulates v_mfma_f32_32x32x2_f32 operand/result layouts (cdna_hip_programming.md section 3):
  A: lane l holds A[i = l & 31][k = l >> 5];  B: lane l holds B[k = l >> 5][j = l & 31]
  D: lane l, register r holds D[(r & 3) + 8*(r >> 2) + 4*(l >> 5)][l & 31]
and replays the kernel's layer chain on the packed weight blob, so that nm_nerf_pack and the
"result layout == next layer's B layout" argument are checked without a GPU.
"""
import numpy as np

XK, HK, VK = 45, 128, 150
OFF_BIAS, OFF_BVIEWS, OFF_WALPHA, OFF_WRGB, OFF_MISC, SMALL = 0, 2304, 2432, 2688, 3072, 3088
OFF_WX0 = SMALL
OFF_WX5 = OFF_WX0 + XK * 512
OFF_WH = OFF_WX5 + XK * 512
OFF_WV = OFF_WH + 8 * HK * 512

LANE = np.arange(64)
ROW = LANE & 31
HI = LANE >> 5


def nrow(r, hi):
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def mfma(a, b, d):
    """d: (32,32) += A(32x2) . B(2x32) with per-lane scalars a, b (64,)"""
    return d + np.outer(a[:32], b[:32]).astype(np.float32) + np.outer(a[32:], b[32:]).astype(np.float32)


def d_to_regs(d):
    """(32,32) matrix -> [lane][16] register view"""
    out = np.empty((64, 16), np.float32)
    for r in range(16):
        out[:, r] = d[nrow(r, HI), ROW]
    return out


def gemm_part(blob, base, nks, nobg, xs, acc):
    w = blob[base : base + nks * nobg * 256].reshape(nks, nobg, 64, 4)
    for ks in range(nks):
        b = xs(ks)
        for o in range(nobg):
            for c in range(4):
                acc[4 * o + c] = mfma(w[ks, o, :, c], b, acc[4 * o + c])
    return acc


def bias_blocks(vec, nblocks):
    # accumulator init: every column (sample) of block ob starts at bias[32*ob + row]
    return [np.repeat(vec[32 * ob : 32 * ob + 32, None], 32, axis=1).astype(np.float32) for ob in range(nblocks)]


def run_wave(blob, ipe90, dirpe27, app16, tap, need_rgb=True):
    """ipe90 (32,90), dirpe27 (32,27) [sin12 | cos12 | raw3], app16 (16,) or None for the 32 samples of one wave.
    Returns sigma_raw (32,), tap features (32,256), rgb (32,3)."""
    small = blob[:SMALL]

    def ipe_at(ks):
        return np.where(HI == 0, ipe90[ROW, ks], ipe90[ROW, 45 + ks]).astype(np.float32)

    x = None  # [lane][128]
    tapped = None
    sigma = None
    for l in range(9):
        if l == 8 and not need_rgb:
            break
        acc = bias_blocks(small[OFF_BIAS + l * 256 : OFF_BIAS + (l + 1) * 256], 8)
        if l in (0, 5):
            acc = gemm_part(blob, OFF_WX0 if l == 0 else OFF_WX5, XK, 2, ipe_at, acc)
        if l != 0:
            xx = x
            acc = gemm_part(blob, OFF_WH + (l - 1) * HK * 512, HK, 2, lambda ks: xx[:, ks], acc)
        regs = np.concatenate([d_to_regs(a) for a in acc], axis=1)  # [lane][ob*16 + r]
        x = np.maximum(regs, 0.0) if l < 8 else regs
        if l == tap:
            tapped = x.copy()
        if l == 7:
            wa = small[OFF_WALPHA : OFF_WALPHA + 256]
            part = np.zeros(64, np.float32)
            for ob in range(8):
                for r in range(16):
                    part += x[:, ob * 16 + r] * wa[32 * ob + nrow(r, HI)]
            sigma = part[:32] + part[32:] + small[OFF_MISC]

    def regs_to_samples(regs, nblk):
        out = np.zeros((32, 32 * nblk), np.float32)
        for ob in range(nblk):
            for r in range(16):
                out[ROW, 32 * ob + nrow(r, HI)] = regs[:, ob * 16 + r]
        return out

    feat = regs_to_samples(tapped, 8)
    rgb = None
    if need_rgb:
        vx = np.zeros((64, VK - HK), np.float32)
        for k in range(12):
            vx[:, k] = np.where(HI == 0, dirpe27[ROW, k], dirpe27[ROW, 12 + k])
        vx[:, 12] = np.where(HI == 0, dirpe27[ROW, 24], dirpe27[ROW, 25])
        vx[:, 13] = np.where(HI == 0, dirpe27[ROW, 26], 0.0)
        if app16 is not None:
            for j in range(8):
                vx[:, 14 + j] = np.where(HI == 0, app16[2 * j], app16[2 * j + 1])
        av = bias_blocks(small[OFF_BVIEWS : OFF_BVIEWS + 128], 4)
        xx = x
        av = gemm_part(blob, OFF_WV, HK, 1, lambda ks: xx[:, ks], av)
        av = gemm_part(blob, OFF_WV + HK * 256, VK - HK, 1, lambda ks: vx[:, ks], av)
        hv = np.maximum(np.concatenate([d_to_regs(a) for a in av], axis=1), 0.0)
        wr = small[OFF_WRGB : OFF_WRGB + 384].reshape(3, 128)
        p = np.zeros((3, 64), np.float32)
        for ob in range(4):
            for r in range(16):
                for c in range(3):
                    p[c] += hv[:, ob * 16 + r] * wr[c, 32 * ob + nrow(r, HI)]
        pre = p[:, :32] + p[:, 32:] + small[OFF_MISC + 1 : OFF_MISC + 4, None]
        rgb = (1.0 / (1.0 + np.exp(-pre))).T
    return sigma, feat, rgb


# ----------------------------------------------------------------------------------------------------------------------
# fp16x3 blob of nerf_fwd_bf16.hip (round 4: power-of-two operand scaling).  CPU model of the SCALE BOOK-KEEPING only: the
# weight slots are read back into plain matrices (hi + lo, still carrying their pack-time factor), and the layer chain is
# replayed with the kernel's re-packing rule  v = relu(fma(acc, s_l, bias'_l))  and head / tap descaling, in float64.
F16 = dict(OFF_BIAS=0, OFF_BVIEWS=2304, OFF_WALPHA=2432, OFF_WRGB=2688, OFF_MISC=3072, OFF_SCALE=3088, OFF_DESCALE=3104, OFF_INSCALE=3112,
           SMALL_PAD=4096, SLOT_BYTES=16384, XS=6, HS=16, VS=3)


def unpack_fp16x3(blob_i16, app_dim):
    """int16 blob -> (small fp32 block, dict name -> effective weight matrix [out, in] as float64 = (hi + lo), scaled as packed)."""
    raw = blob_i16.view(np.uint8) if blob_i16.dtype != np.uint8 else blob_i16
    small = raw[: F16["SMALL_PAD"] * 4].view(np.float32).copy()
    slots = raw[F16["SMALL_PAD"] * 4:].view(np.float16)
    per = F16["SLOT_BYTES"] // 2
    g = [0]

    def read(nob, ncols, colfn, W):
        s = slots[g[0] * per: (g[0] + 1) * per].astype(np.float64)
        g[0] += 1
        for obo in range(nob):
            for ln in range(64):
                for i in range(8):
                    c = colfn(ln >> 5, i)
                    if c < 0:
                        continue
                    hi = s[((obo * 2 + 0) * 64 + ln) * 8 + i]
                    lo = s[((obo * 2 + 1) * 64 + ln) * 8 + i]
                    W[32 * obo + (ln & 31), c] = hi + lo

    def ipe(W):
        for m in range(F16["XS"]):
            read(8, 90, lambda h, i, m=m: (45 * h + 8 * m + i) if (8 * m + i) < 45 else -1, W)  # K-slot (m, half, i) <-> encoding 45 half + 8 m + i

    def hid(W, col0, nob):
        for ks in range(F16["HS"]):
            read(nob, 256, lambda h, i, ks=ks: col0 + 32 * (ks >> 1) + nrow(8 * (ks & 1) + i, h), W)

    mats = {}
    for l in range(8):
        W = np.zeros((256, 90 if l == 0 else (346 if l == 5 else 256)))
        if l == 0:
            ipe(W)
        else:
            hid(W, 90 if l == 5 else 0, 8)
        if l == 5:
            ipe(W)
        mats[f"pts{l}"] = W
    # (no feature_linear slots: nerf_pack_split folds it into the views layer's hidden columns, round 4)
    ldv = 283 + app_dim
    W = np.zeros((128, ldv))

    def ext_col(e, h, i):
        f = 16 * e + 8 * h + i
        if f < 27:
            return 256 + f
        if f < 43 and app_dim:
            return 283 + (f - 27)
        return -1

    def hid_col(ks, h, i):
        return 32 * (ks >> 1) + nrow(8 * (ks & 1) + i, h)

    def read2(colA, colB):
        """paired slot of the views layer (slot_step4x2): blocks 0..3 = the four output blocks for the first K-step, 4..7 for the second"""
        sl = slots[g[0] * per: (g[0] + 1) * per].astype(np.float64)
        g[0] += 1
        for obo in range(8):
            for ln in range(64):
                for i in range(8):
                    c = (colA if obo < 4 else colB)(ln >> 5, i)
                    if c < 0:
                        continue
                    W[32 * (obo & 3) + (ln & 31), c] = sl[((obo * 2 + 0) * 64 + ln) * 8 + i] + sl[((obo * 2 + 1) * 64 + ln) * 8 + i]

    for s2 in range(F16["HS"] // 2):
        read2(lambda h, i, k=2 * s2: hid_col(k, h, i), lambda h, i, k=2 * s2 + 1: hid_col(k, h, i))
    read2(lambda h, i: ext_col(0, h, i), lambda h, i: ext_col(1, h, i))
    read(4, ldv, lambda h, i: ext_col(2, h, i), W)
    mats["views"] = W
    assert g[0] == 134, g[0]
    return small, mats


def run_chain_fp16x3(small, mats, ipe90, dirpe27, app16, tap):
    """float64 replay of the kernel's scaled layer chain for n samples: returns (sigma_raw, tapped activations in TRUE units, rgb)."""
    sm = small.astype(np.float64)
    s_ipe, s_dir, s_app = sm[F16["OFF_INSCALE"]], sm[F16["OFF_INSCALE"] + 1], sm[F16["OFF_INSCALE"] + 2]
    x0 = ipe90.astype(np.float64) * s_ipe
    bias = lambda l: sm[F16["OFF_BIAS"] + l * 256: F16["OFF_BIAS"] + (l + 1) * 256]
    h = None
    tapped = None
    for l in range(8):
        W = mats[f"pts{l}"]
        xin = x0 if l == 0 else (np.concatenate([x0, h], 1) if l == 5 else h)
        acc = xin @ W.T
        h = np.maximum(acc * sm[F16["OFF_SCALE"] + l] + bias(l), 0.0)  # re-packed: carries the next layer's input scale
        if l == tap:
            tapped = h * sm[F16["OFF_DESCALE"] + l]
    sigma = h @ sm[F16["OFF_WALPHA"]: F16["OFF_WALPHA"] + 256] + sm[F16["OFF_MISC"]]
    # views layer on layer 7's (scaled) activations: its hidden columns hold views_w[:, :256] @ feature_w, its bias the folded one
    ex = [dirpe27.astype(np.float64) * s_dir]
    if app16 is not None:
        ex.append(np.repeat(app16.astype(np.float64)[None] * s_app, h.shape[0], 0))
    v = np.maximum(np.concatenate([h] + ex, 1) @ mats["views"].T + sm[F16["OFF_BVIEWS"]: F16["OFF_BVIEWS"] + 128], 0.0)
    pre = v @ sm[F16["OFF_WRGB"]: F16["OFF_WRGB"] + 384].reshape(3, 128).T + sm[F16["OFF_MISC"] + 1: F16["OFF_MISC"] + 4]
    return sigma, tapped, 1.0 / (1.0 + np.exp(-pre))
