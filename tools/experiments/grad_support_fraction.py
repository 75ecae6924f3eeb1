import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from multiview_inpaint_amd import raster as R, synthetic as syn
W,H,N=1920,1080,1500000
cam=syn.make_camera(W,H,50.0); sc=syn.make_scene(N,cam,3,seed=0)
t={k: torch.tensor(v,device="cuda") for k,v in sc.items() if k!="sh_degree"}
d="cuda"
rs=R.GaussianRasterizationSettings(image_height=H,image_width=W,tanfovx=cam["tanfovx"],tanfovy=cam["tanfovy"],bg=torch.zeros(3,device=d),scale_modifier=1.0,viewmatrix=torch.tensor(cam["viewmatrix"],device=d),projmatrix=torch.tensor(cam["projmatrix"],device=d),sh_degree=3,campos=torch.tensor(cam["campos"],device=d),prefiltered=False)
kw=dict(shs=t["shs"],scales=t["scales"],rotations=t["rotations"])
c,radii,dep,st=R.rasterize_forward(rs,t["means3D"],t["opacities"],**kw)
g=R.rasterize_backward(rs,st,torch.randn(3,H,W,device=d),t["means3D"],**kw)
vis=(radii>0)
nz=(g["shs"].abs().amax(dim=(1,2))>0)
nz2=(g["opacities"].abs().reshape(-1)>0)
print("visible",float(vis.float().mean()),"nonzero dL/dshs",float(nz.float().mean()),"nonzero dL/dopacity",float(nz2.float().mean()))
