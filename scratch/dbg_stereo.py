import sys, torch
sys.path.insert(0, "/root/repo")
from oracle import ref_loss, ref_synthesize as rs, ref_pose
from xpt_mde_2021_amd.utils import synthetic_data as sd
from xpt_mde_2021_amd.utils import convert_pose as cp
from xpt_mde_2021_amd.model.synthesize.synthesize_base import SynthesizeMultiScale
from xpt_mde_2021_amd.hip import ops
dev = torch.device("cuda")
B,H,W=2,64,208
f = sd.make_features(B,H,W,5,99,True)
g = torch.Generator().manual_seed(5)
depth_ms=[sd.smooth_depth(B,H//s,W//s,g,lo=1.0,hi=60.0) for s in (1,2,4,8)]
T=f["stereo_T_LR"]
p_ref = ref_pose.pose_matr2rvec_batch(torch.linalg.inv(T.double()).unsqueeze(1))
p_gpu = cp.pose_matr2rvec_batch(torch.linalg.inv(T.to(dev)).unsqueeze(1))
print("pose ref", p_ref[0], "gpu", p_gpu[0].cpu())
src = f["image5d_R"][:, -1].unsqueeze(1)
ref = rs.synthesize_multi_scale(src.double(), f["intrinsic"].double(), [d.double() for d in depth_ms], p_ref)
out = SynthesizeMultiScale()(src.to(dev), f["intrinsic"].to(dev), [d.to(dev) for d in depth_ms], p_gpu)
for s,(a,b) in enumerate(zip(out,ref)):
    diff=(a.cpu().double()-b).abs()
    print("scale",s,"maxdiff",diff.max().item(),"nbad",(diff>1e-4).sum().item(),"of",diff.numel(), "valid ref",(b.abs().sum(-1)>0).double().mean().item(), "valid gpu", (a.abs().sum(-1)>0).double().mean().item())
    tg = rs.tf_resize_bilinear(f["image5d"][:,-1], a.shape[2:4])
    print("   L1 ref", ref_loss.photometric_loss_l1(b, tg.double()), "gpu", ops.photometric("L1", a, tg.to(dev)).cpu())
