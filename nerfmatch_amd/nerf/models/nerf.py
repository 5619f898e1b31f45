"""Parameter container for one NeRF MLP with the reference's module / state-dict layout.

Mirrors the constructor of nerfmatch/nerf/models/nerf.py:29-65 (`pts_linears`, `views_linears`,
`feature_linear`, `alpha_linear`, `rgb_linear`), so reference checkpoints load unchanged.  The arithmetic
itself lives in the fused HIP kernel (csrc/nerf_fwd.hip): this module only owns the parameters and
packs them into the MFMA operand order the kernel streams."""
import torch
import torch.nn as nn

from ...utils import update_configs
from ... import _lib


class NeRF(nn.Module):
    default_config = {
        "layer_num": 8, "hid_dim": 256, "xyz_dim": 3, "dirs_dim": 3, "app_dim": 0, "output_dim": 4,
        "skips": [4], "use_viewdirs": False, "out_3d_pnt": False, "out_add_ch": 0, "stop_layer": -1,
    }

    def __init__(self, config):
        super().__init__()
        c = update_configs(self.default_config, config)
        if not (c.layer_num == 8 and c.hid_dim == 256 and list(c.skips) == [4] and c.use_viewdirs and c.output_dim == 4
                and c.xyz_dim == 90 and c.dirs_dim == 27 and c.app_dim in (0, 16) and not c.out_3d_pnt):
            raise NotImplementedError(
                "nerfmatch_amd's fused kernel is built for the shipped NeRFMatch architecture: 8x256, skip at 4, "
                f"IPE 90 + view PE 27 (+16 appearance), view-dependent rgb; got {vars(c)}")
        self.layer_num, self.hid_dim, self.skips = c.layer_num, c.hid_dim, list(c.skips)
        self.xyz_dim, self.dirs_dim, self.app_dim, self.output_dim = c.xyz_dim, c.dirs_dim, c.app_dim, c.output_dim
        self.use_viewdirs, self.out_3d_pnt, self.stop_layer = c.use_viewdirs, c.out_3d_pnt, c.stop_layer
        hid = c.hid_dim
        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.xyz_dim, hid)] +
            [nn.Linear(hid + (self.xyz_dim if i in self.skips else 0), hid) for i in range(self.layer_num - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.dirs_dim + hid + self.app_dim, hid // 2)])
        self.feature_linear = nn.Linear(hid, hid)
        self.alpha_linear = nn.Linear(hid, 1)
        self.rgb_linear = nn.Linear(hid // 2, self.output_dim - 1)
        self._blob = None
        self._blob_key = None

    def _param_key(self):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters())

    def packed(self, device, precision="fp32"):
        """Device blob in MFMA operand order (fp32 kernel) or as pre-split bf16 hi/lo K-step slots (bf16x3 kernel);
        re-packed whenever a parameter was replaced or modified."""
        pkey = self._param_key()
        if self._blob is None or self._blob_key != pkey:
            self._blob, self._blob_key = {}, pkey  # one blob per (device, precision) for the current parameters
        hit = self._blob.get((str(device), precision))
        if hit is None:
            sd = {f"m.{k}": v for k, v in self.state_dict().items()}
            hit = self._blob[(str(device), precision)] = _lib.pack_nerf_weights(sd, "m", precision).to(device)
        return hit

    def invalidate(self):
        """Forget the packed blobs (call after writing parameters through `.data`, which does not bump `_version`)."""
        self._blob = self._blob_key = None
        self.__dict__.pop("_field_key", None)

    def forward(self, x, ret_pfeat=0, pfeat_mask=None, val=False):
        """Per-sample evaluation with the reference's signature (nerf/models/nerf.py:94-144): x (..., 90 + 27 [+ 16]) ->
        outputs (..., 4) = [sigmoid rgb, raw sigma] and, with ret_pfeat > 0, the features of layer `stop_layer` (last layer
        when negative).  The render path never comes through here (it uses the fused kernel); callers that reach into the
        network directly (the reference's iNeRF loop, nerfmatch_evaluator.py:402-406) get the same numbers from a chain of
        the HIP GEMM kernels (nm_linear_ex; the skip and view inputs enter as `pre` addends instead of concatenations)."""
        from ...inerf import XD, XI, FineField  # GEMM-chain packing of one MLP (shared with the iNeRF refinement)

        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).to(torch.float32)
        n, dev = x2.shape[0], x2.device
        key = (self._param_key(), str(dev))
        if self.__dict__.get("_field_key") != key:
            self.__dict__["_field"], self.__dict__["_field_key"] = FineField(self, dev), key
        field = self.__dict__["_field"]
        xi = torch.zeros(n, XI, device=dev)
        xi[:, : self.xyz_dim] = x2[:, : self.xyz_dim]
        xd = torch.zeros(n, XD, device=dev)
        xd[:, : self.dirs_dim + self.app_dim] = x2[:, self.xyz_dim:]
        logit, sig, (h, _) = field.forward(xi, xd)
        outputs = torch.cat([torch.sigmoid(logit[:, :3]), sig[:, :1]], -1).reshape(*lead, 4)
        if ret_pfeat > 0:
            feats = h[self.stop_layer if self.stop_layer >= 0 else self.layer_num - 1]
            feats = feats.reshape(*lead, feats.shape[-1])
            if pfeat_mask is not None and self.stop_layer < 0:
                feats = feats[..., pfeat_mask, :]
            return outputs, feats
        return outputs
