"""Worker of tests/test_dist_gpu.py::test_two_processes_on_one_gpu: one rank of a gloo process group (or the single-process reference run)
that localises a stream of one-query batches through the real HIP path on cuda:0 and writes the gathered records to a file.

    python tests/two_proc_worker.py <out.pt>          (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment; WORLD_SIZE 1 = no group)
"""
import os
import sys
from argparse import Namespace
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch  # noqa: E402

H, W, S, NQ = 64, 96, 64, 6


def run(out_path):
    import torch.distributed as dist

    import nerfmatch_amd
    from nerfmatch_amd import dist as nmdist
    from nerfmatch_amd import synth
    from nerfmatch_amd.nerf.renderer import NerfRenderer
    from nerfmatch_amd.nerfmatch_evaluator import NeRFMatchEvaluator
    from test_evaluator_gpu import make_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)  # BOTH ranks on the one GPU: what is under test is process isolation, not a second device
    gpu = torch.device("cuda:0")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    R = (H // 8) * (W // 8)
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H)), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=3, style="surface"))
    ren.to(gpu).eval()
    raw = ren.render_novel_views

    def seeded(img_hw, K, c2ws, unnorm, device, **kw):  # the samplers' random tensors belong to the query, not to the process that draws them
        qs = [int(round(float(torch.as_tensor(c)[0, 3]) * 1e6)) % (2**31 - 1) for c in torch.as_tensor(c2ws).reshape(-1, 4, 4)]
        tr = torch.cat([torch.rand(R, S + 1, generator=torch.Generator().manual_seed(q)) for q in qs])
        jt = torch.cat([synth.resample_jitter((R, S + 1), q + 1) for q in qs])
        return raw(img_hw, K, c2ws, unnorm, device, t_rand=tr.to(gpu), jitter=jt.to(gpu), **kw)

    ren.render_novel_views = seeded
    ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=0), data=Namespace()))
    ev.model.load_state_dict(synth.matcher_state_dict("c2f", seed=0), strict=False)
    ev.model.to(gpu).eval()
    assert ev.device == gpu
    nerfmatch_amd.set_precision("bf16x3")
    agreed = nmdist.agree_calibration(ren, gpu)
    batches = [make_batch(H, W, q) for q in range(NQ)]
    out = ev.eval_data_loader(renderer=ren, data_loader=batches, solver="none", query2query=True, mutual=True)
    torch.cuda.synchronize()
    mine = {q: {k: batches[q][k].cpu() for k in ("pt3d", "pt_feat", "mpt2d_f", "mpt3d", "mconf")} for q in range(NQ) if "mpt3d" in batches[q]}
    torch.save(dict(rank=rank, world=world, agreed=bool(agreed), query_idx=out["query_idx"], num_matches=out["num_matches"], c2w_est=out["c2w_est"],
                    R_err=out["R_err"], t_err=out["t_err"], mine=mine,
                    scales=(tuple(ren.nerf_coarse._act_log2[str(gpu)]), tuple(ren.nerf_fine._act_log2[str(gpu)]))), out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1])
