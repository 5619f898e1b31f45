import sys, subprocess, os
# A/B: ops.transposed cached vs fresh transposes, same box, alternating processes
for rep in range(2):
    for mode in ("cached", "fresh"):
        env = dict(os.environ, NM_AB_T=mode)
        out = subprocess.run([sys.executable, "-c", """
import os, sys, runpy
sys.path.insert(0, '/root/repo')
from nerfmatch_amd import ops
if os.environ['NM_AB_T'] == 'fresh':
    ops.transposed = lambda w: w.detach().t().contiguous()
sys.argv = ['perf_train.py', '480', '480', '2', 'bf16x3']
src = open('/root/repo/scripts/perf_train.py').read().replace('n = 5', 'n = 20')
exec(compile(src, 'perf_train.py', 'exec'), {'__file__': '/root/repo/scripts/perf_train.py', '__name__': '__main__'})
"""], env=env, capture_output=True, text=True)
        print(mode, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
