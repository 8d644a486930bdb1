import sys, time, torch
sys.path.insert(0,'.')
from omfs_4d_video_gen_amd.engine import synthetic
from omfs_4d_video_gen_amd.engine.flame_rig import FlameRig
from omfs_4d_video_gen_amd.engine.trainer import Renderer, Trainer, View
W,H,N=1920,1080,300000
srig=synthetic.make_rig(0); rig=FlameRig.from_synthetic(srig); seq=synthetic.make_flame_sequence(16,0)
cams=synthetic.make_camera_arc(W,H,16); g0=synthetic.make_gaussians(N,rig.n_faces,0); g1=synthetic.make_gaussians(N,rig.n_faces,1)
tr=Renderer(rig,seq,g1,W,H); views=[]
for i,c in enumerate(cams):
    v=View(c,i); v.target=tr.render(v).clone(); views.append(v)
t=Trainer(rig,seq,g0,views,W,H,start_sh_degree=3)
for _ in range(50): t.step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(200): t.step()
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host enqueue ms/step", (t1-t0)/200*1e3, "total ms/step", (t2-t0)/200*1e3)
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(100): t.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
