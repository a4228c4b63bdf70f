import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import gan_class_transfer2_amd as g
from oracle import denoiser_oracle as O
L=g._lib; dev=torch.device('cuda',0)
B,H,W,Cin,Cout=2,16,16,64,128
rng=np.random.default_rng(15)
x=np.maximum(rng.standard_normal((B,H,W,Cin)),0).astype(np.float32)
w=(rng.standard_normal((4,4,Cin,Cout))*0.1).astype(np.float32)
dz=rng.standard_normal((B,H//2,W//2,Cout)).astype(np.float32)
prev=rng.standard_normal((B,H,W,Cin)).astype(np.float32)
contrib=O.conv4s2_bwd(x.astype(np.float64),w.astype(np.float64),dz.astype(np.float64))[0]*(x>0)
t=lambda a: torch.tensor(a,device=dev)
dxd=t(prev); db=torch.full((32,),3.0,device=dev); db2=torch.full((32,),-1.0,device=dev)
s=torch.cuda.current_stream().cuda_stream
L.call("gct2_conv4s2_dgrad",0,t(dz).data_ptr(),Cout,t(w).data_ptr(),t(x).data_ptr(),Cin,dxd.data_ptr(),Cin,B,H,W,Cin,Cout,1,db.data_ptr(),32,db2.data_ptr(),s)
torch.cuda.synchronize()
print(db[:6].cpu().numpy()-3, contrib.reshape(-1,Cin).sum(0)[:6], prev.reshape(-1,Cin).sum(0)[:6], np.abs(prev).reshape(-1,Cin).sum(0)[:3])
for acc in (0, 1):
    dxd=t(prev); db=torch.zeros(32,device=dev); db2=torch.zeros(32,device=dev)
    L.call("gct2_conv4s2_dgrad",0,t(dz).data_ptr(),Cout,t(w).data_ptr(),t(x).data_ptr(),Cin,dxd.data_ptr(),Cin,B,H,W,Cin,Cout,acc,db.data_ptr(),32,db2.data_ptr(),s)
    torch.cuda.synchronize()
    out=dxd.cpu().numpy()
    print("acc",acc,"db",db[:4].cpu().numpy(),"colsum(out)",out.reshape(-1,Cin).sum(0)[:4],"db2",db2[:3].cpu().numpy(), out.reshape(-1,Cin).sum(0)[32:35])
