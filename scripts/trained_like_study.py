import sys, torch, numpy as np, math
sys.path.insert(0, '/root/repo')
from nerfmatch_amd import synth
from oracle import nerf_oracle as no
torch.set_grad_enabled(False)
H,W,S=128,256,128
K = torch.tensor([[240.0, 0, W / 2], [0, 240.0, H / 2], [0, 0, 1]])
unnorm = synth.unnorm_scene(); c2w = unnorm @ synth.camera_pose(11)
R=(H//8)*(W//8)
t_rand, jit = synth.uniform01((R,S+1),1), synth.resample_jitter((R,S+1),2)
def base(gain, fcut):
    sd = synth.nerf_state_dict(seed=0)
    damp = torch.tensor([min(1.0, 2.0**(fcut-i)) for i in range(15)]).repeat_interleave(3).repeat(2)  # 90 columns: scale-major xyz-minor, two halves
    for net in ("nerf_coarse","nerf_fine"):
        for i in range(8):
            sd[f"{net}.pts_linears.{i}.weight"] *= gain[i]
        sd[f"{net}.pts_linears.0.weight"] *= damp[None]
        sd[f"{net}.pts_linears.5.weight"][:, :90] *= damp[None]
        sd[f"{net}.alpha_linear.bias"] *= 0
    for k in list(sd):
        if k.startswith("nerf_coarse.pts_linears") or k.startswith("nerf_coarse.alpha_linear"):
            sd[k.replace("nerf_coarse","nerf_fine")] = sd[k].clone()
    return sd
def stats(p):
    for key in ('coarse','fine'):
        w = p[f'weights_{key}']; raw = p[f'raw_{key}']
        print(' ', key, 'sigma range %.1f %.1f'%(raw[...,3].min().item(), raw[...,3].max().item()), 'frac>0 %.2f'%(raw[...,3]>0).float().mean().item(),
              'acc mean %.3f'%p[f'acc_{key}'].mean().item(), 'maxw median %.3f'%w.max(-1)[0].median().item(), 'n(w>0.01) median', (w>0.01).sum(-1).float().median().item(),
              'feat absmax %.2f'%p[f'feat_{key}'].abs().max().item(), 'sfeat absmax %.2f'%p[f'sfeat_{key}'].abs().max().item(), 'rgb range %.3f %.3f'%(p[f'rgb_{key}'].min(), p[f'rgb_{key}'].max()))
for gain, fcut, q, dstd in [([3.2]*8, 4, 0.75, 500.), ([3.2]*8, 4, 0.75, 1000.),([3.2]*8, 4, 0.75, 2500.)]:
    sd = base(gain, fcut)
    p = no.render_novel_view(sd,(H,W),K,c2w,unnorm,t_rand,jit,S,S,stop_layer=3, keep_raw=True)['preds']
    s = p['raw_coarse'][...,3]
    print('gain', gain[0], 'fcut', fcut, 'pre-bias coarse: mean %.3f std %.3f along-ray std %.3f step std %.4f'%(s.mean(), s.std(), s.std(-1).mean(), (s[:,1:]-s[:,:-1]).std()))
    # calibrate both nets on the coarse sample positions: fine net evaluated there too
    for net, key in (("nerf_coarse","coarse"),("nerf_fine","fine")):
        s = p[f'raw_coarse'][...,3]
        g = dstd / s.std().item()
        thr = torch.quantile(s.flatten()[:200000], q).item()
        sd[f"{net}.alpha_linear.weight"] *= g
        sd[f"{net}.alpha_linear.bias"] += -thr*g
        print('  ', net, 'gain %.1f bias %.1f'%(g, -thr*g))
    p = no.render_novel_view(sd,(H,W),K,c2w,unnorm,t_rand,jit,S,S,stop_layer=3, keep_raw=True)['preds']
    stats(p)
